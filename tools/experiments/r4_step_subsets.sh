#!/bin/bash
# round 4 (verdict item 7): what the small-matrix class costs INSIDE the step -- the step of the set without the eight 256-thread
# matrices, without the four small tile streams, without both, and those classes alone
out=gpurun_out/r4q; mkdir -p $out
BIG=PFlow_742,soc-Pokec,mouse_gene,TSOPF_RS_b2383,Si41Ge41H72,crankseg_2,nd6k,thread
TTS4=nxp1,analytics,boyd2,language
S256=ASIC_680k,crystk03,trans5,ford2,lowThrust_7,c-52,hangGlider_3,poli_large
run() { tag=$1; shift; python3 bench.py --no-cpu-baseline --no-extras --no-verify --steps 300 --warmup 100 --per-matrix-reps 0 "$@" > $out/$tag.log 2>&1
  echo "$tag: $(grep -o '"ms_per_step": [0-9.]*' $out/$tag.log | tr '\n' ' ')"; }
run all
run big8 --matrices $BIG
run big8_tts4 --matrices $BIG,$TTS4
run big8_s256 --matrices $BIG,$S256
run small12 --matrices $TTS4,$S256
run tts4 --matrices $TTS4
run s256 --matrices $S256
run big_slices --matrices PFlow_742,mouse_gene,TSOPF_RS_b2383,Si41Ge41H72,crankseg_2,nd6k,thread
run pokec --matrices soc-Pokec
run pokec_tts4 --matrices soc-Pokec,$TTS4
