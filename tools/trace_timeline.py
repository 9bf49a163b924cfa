#!/usr/bin/env python3
"""Timeline of the last steps of a kernel trace (rocprofv3 --kernel-trace --output-format csv): for every hispmv kernel of a
step its workgroup size, grid, LDS, start and end relative to the step's first kernel.  tools/trace_timeline.py <dir> [steps]"""
import csv, glob, sys
d = sys.argv[1]; want = int(sys.argv[2]) if len(sys.argv) > 2 else 2
f = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))[-1]
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("hispmv::", ""), int(r["Workgroup_Size_X"]),
              int(r["Grid_Size_X"]), int(r["LDS_Block_Size"]), r["Queue_Id"]) for r in csv.DictReader(open(f)) if "hispmv::" in r["Kernel_Name"]))
steps, cur = [], []
for e in ev:
    cur.append(e)
    if "merge_multi" in e[2] or "tail_multi" in e[2]:
        steps.append(cur); cur = []
for st in steps[-want:]:
    t0 = st[0][0]
    print(f"step: span {(max(e[1] for e in st) - t0) / 1e3:.1f} us")
    for s0, e0, name, wg, grid, lds, q in st:
        print(f"  {name:28s} wg {wg:5d} grid {grid // max(wg,1):6d} lds {lds:7d} q {q:>3s}  {(s0 - t0) / 1e3:8.1f} -> {(e0 - t0) / 1e3:8.1f}  ({(e0 - s0) / 1e3:6.1f} us)")
