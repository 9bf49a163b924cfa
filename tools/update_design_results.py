#!/usr/bin/env python3
"""Regenerates the '### Current results' block of DESIGN.md from a bench.py --details file (and the traffic
summary in profiles/):   tools/update_design_results.py gpurun_out/bench_final.json [profiles/r1_traffic.json]"""
import json
import re
import sys
from pathlib import Path

root = Path(__file__).resolve().parents[1]
d = json.loads(Path(sys.argv[1]).read_text())
s, rows = d["summary"], d["per_matrix"]
traffic = json.loads(Path(sys.argv[2]).read_text()) if len(sys.argv) > 2 else None
fam = {}
sys.path.insert(0, str(root))
from hispmv_amd import matrices as M  # noqa: E402
for name, _r, _n, f, _p in M.SUITESPARSE_SET:
    fam[name] = f
cb = s.get("cpu_baseline") or {}
lines = ["### Current results (MI355X, round 1)",
         f"`python bench.py` (defaults: {s['steps']} steps, {s['warmup']} warm-up, {s['config']['streams']} streams; the `BENCH` line of "
         f"this round): **{s['value']} GFLOP/s** over the set, {s['ms_per_step']} ms per step, {s['hbm_gbs_algorithmic']} GB/s algorithmic = "
         f"**{s['hbm_pct_of_peak']} % of the 8 TB/s peak** ({100 * s['hbm_gbs_algorithmic'] / 6270:.0f} % of the 6.27 TB/s the bare "
         f"stream reaches); `roofline.achieved` {s['roofline']['achieved']} GB/s (HIP events)."]
if cb:
    lines.append(f"CPU baseline on the same box ({cb.get('cores')} host threads, {cb.get('kind')}): {cb.get('value')} {cb.get('unit')} "
                 f"({cb.get('sample')}).")
if traffic:
    alg = sum(M.algorithmic_bytes(r["rows"], r["rows"], r["nnz"]) for r in rows)
    lines.append(f"rocprofv3 (`profiles/r1_summary.md`): HBM traffic {traffic['hbm_bytes_per_step'] / 1e6:.0f} MB per step vs {alg / 1e6:.0f} MB "
                 f"algorithmic ({traffic['hbm_bytes_per_step'] / alg:.2f}x; the surplus is x lines fetched for the L2 gathers of the "
                 f"scattered matrices, fillers and the y read-modify-write of column tiles).")
lines += ["", "Per matrix, each timed alone between two HIP events with the largest matrix streamed in between (cold Infinity Cache);",
          "time = slice kernel (+ fix-up launch, + one launch per extra column tile):", "",
          "| matrix (stand-in family) | rows | nnz | µs | GFLOP/s | alg. GB/s | % of 8 TB/s | launch plan (threads/slices per WG/LDS window/column tiles) |",
          "|---|---:|---:|---:|---:|---:|---:|---|"]
for r in rows:
    src = fam.get(r["name"], "?") if str(r.get("source", "synthetic")).startswith("synthetic") else "real file"
    lines.append(f"| {r['name']} ({src}) | {r['rows']} | {r['nnz']} | {r['us']} | {r['gflops']} | {r['alg_gbs']} | {r['pct_hbm_peak']} | {r['plan']} |")
block = "\n".join(lines) + "\n\n"
p = root / "DESIGN.md"
t = p.read_text()
m = re.search(r"### Current results \(MI355X, round 1\)\n.*?\n(?=Reading: )", t, re.S)
assert m, "results block not found"
p.write_text(t[:m.start()] + block + t[m.end():])
print("updated", p)
