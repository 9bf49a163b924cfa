#!/usr/bin/env python3
"""Prints the last N kernel launches of a rocprofv3 kernel trace with durations and gaps:
   tools/trace_seq.py <dir-or-csv> [N]"""
import csv, glob, sys
src = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
files = [src] if src.endswith(".csv") else glob.glob(src + "/**/*kernel_trace.csv", recursive=True)
for f in files:
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    prev = None
    for r in rows[-n:]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        gap = (s - prev) / 1e3 if prev else 0.0
        name = r["Kernel_Name"].replace("hispmv::", "").replace("void ", "")[:44]
        print(f"{name:44s} q{r['Queue_Id']:>2s} grid {r['Grid_Size_X']:>8s} wg {r['Workgroup_Size_X']:>5s} lds {r['LDS_Block_Size']:>7s} vgpr {r['VGPR_Count']:>4s} dur {(e - s) / 1e3:7.1f} us gap {gap:6.1f}")
        prev = e
