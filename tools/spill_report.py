#!/usr/bin/env python3
"""Where do the SGPR spills of the kernels sit?    Reads hispmv_amd/csrc/build/kernels.s (`make -C hispmv_amd/csrc asm`)
and reports per kernel: v_writelane_b32 (spill stores) and the v_readlane_b32 that read the spill VGPRs back (reloads), in total and
INSIDE LOOPS (a loop = a label that a later branch jumps back to).  Usage: tools/spill_report.py [kernels.s] [out.json]"""
import json
import re
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
src = Path(sys.argv[1]) if len(sys.argv) > 1 else ROOT / "hispmv_amd/csrc/build/kernels.s"
txt = src.read_text().split("\n")
names = [l.split()[-1] for l in txt if l.strip().startswith(".amdhsa_kernel ")]
start_of = {}
for i, l in enumerate(txt):
    m = re.match(r"^(\S+):\s*;\s*@", l)
    if m and m.group(1) in names:
        start_of[m.group(1)] = i
rows = []
for n, start in start_of.items():
    end = start
    while end < len(txt) and "s_endpgm" not in txt[end]:
        end += 1
    body = txt[start:end + 1]
    labels = {m.group(1): i for i, l in enumerate(body) if (m := re.match(r"^(\.LBB\d+_\d+):", l))}
    loops = []
    for i, l in enumerate(body):
        m = re.search(r"s_c?branch\S*\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            loops.append((labels[m.group(1)], i))
    inloop = lambda i: any(a <= i <= b for a, b in loops)
    wl = [i for i, l in enumerate(body) if "v_writelane_b32" in l]
    spill_vgprs = {re.search(r"v_writelane_b32 (v\d+)", body[i]).group(1) for i in wl}
    rl = [i for i, l in enumerate(body) if "v_readlane_b32" in l and any(re.search(rf", {v}, ", l) for v in spill_vgprs)]
    rows.append(dict(kernel=n, lines=len(body), spill_stores=len(wl), spill_stores_in_loops=sum(map(inloop, wl)),
                     spill_reloads=len(rl), spill_reloads_in_loops=sum(map(inloop, rl)), loops=len(loops)))
rows.sort(key=lambda q: -q["spill_stores"])
for q in rows:
    if q["spill_stores"]:
        print(f'{q["kernel"][:78]:80s} stores {q["spill_stores"]:4d} (in loops {q["spill_stores_in_loops"]:4d})  reloads {q["spill_reloads"]:4d} (in loops {q["spill_reloads_in_loops"]:4d})')
if len(sys.argv) > 2:
    Path(sys.argv[2]).write_text(json.dumps(rows, indent=1) + "\n")
