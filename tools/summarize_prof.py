#!/usr/bin/env python3
"""Condenses gpurun_out/prof_<tag>/ (written by tools/profile_round.sh on the GPU box) into the small
tracked files profiles/<tag>_kernel_stats.csv and profiles/<tag>_summary.md.

HBM traffic follows MI355X_MICROARCH.md "HBM": FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE
reports exactly half of the bytes of a wide (16 B/lane) coalesced streaming read, so the read side is
doubled; WRITE_SIZE is exact for dword-per-lane stores.  Collected in separate --pmc passes."""
import csv
import glob
import json
import re
import sys
from collections import defaultdict
from pathlib import Path

tag = sys.argv[1] if len(sys.argv) > 1 else "r1"
src = Path("gpurun_out") / f"prof_{tag}"
dst = Path("profiles")
dst.mkdir(exist_ok=True)


def first(pattern):
    import os
    g = sorted(glob.glob(str(src / pattern), recursive=True), key=os.path.getmtime)   # gpurun merges runs: newest wins
    return g[-1] if g else None


out = [f"# rocprofv3 summary, round tag `{tag}`", "", "Command: `tools/profile_round.sh " + tag + "` (trace pass: bench.py --steps 30 --warmup 200; counter passes: --steps 5 --warmup 1; 20-matrix set)", ""]
stats = first("trace/**/*kernel_stats.csv")
if stats:
    rows = list(csv.DictReader(open(stats)))
    with open(dst / f"{tag}_kernel_stats.csv", "w") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for r in rows:
            w.writerow([r["Name"][:110], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])
    out += ["## Kernel time (`--kernel-trace --stats`)", "", "| kernel | calls | avg us | total ms | % |", "|---|---:|---:|---:|---:|"]
    for r in rows[:8]:
        out.append(f"| `{r['Name'][:70]}` | {r['Calls']} | {float(r['AverageNs'])/1e3:.2f} | {float(r['TotalDurationNs'])/1e6:.3f} | {float(r['Percentage']):.1f} |")
    out.append("")


# Step span from the kernel trace: the independent main launches of a step run on two HIP streams (their durations
# overlap, so the sum of the kernel averages exceeds the step time); a step ends with its merge launch.
trace = first("trace/**/*kernel_trace.csv")
if trace:
    ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(trace)) if "hispmv::" in r["Kernel_Name"]))
    spans, start, busy = [], None, 0
    for s0, e0, name in ev:
        if start is None:
            start = s0
        if "merge_multi" in name or "tail_multi" in name:
            spans.append((e0 - start) / 1e3)
            start = None
    if spans:
        tail = spans[-30:] if len(spans) > 60 else spans[len(spans) // 2:]          # the timed steps (past warm-up)
        out += [f"Step span in the trace (first launch of a step to the end of its merge launch, last {len(tail)} steps): "
                f"{sum(tail) / len(tail):.1f} us on average (min {min(tail):.1f}); under the profiler the chip clocks lower than in an "
                "un-profiled run (MI355X_MICROARCH.md, DVFS), bench.py's HIP-event time is the un-profiled figure.", ""]


# The step kernel (the default path of a call that shares the chip between its matrices): ONE launch holds the slice groups and tiles of
# every class, so its algorithmic bytes are the whole step's (sum over bench.py's launch_classes) and its average duration in the main
# trace gives the dominant kernel's roofline fraction directly.
step_kernel_row = None
if trace:
    classes_main = {}
    try:
        for line in (src / "trace.log").read_text().splitlines():
            if line.startswith("{") and "launch_classes" in line:
                classes_main = json.loads(line)["launch_classes"]
    except Exception:
        pass
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(trace)) if "spmv_step_kernel" in r["Kernel_Name"]]
    if d and classes_main:
        tail = d[-30:]
        avg = sum(tail) / len(tail)
        ab = sum(v["algorithmic_bytes_per_launch"] for k, v in classes_main.items() if k.startswith("spmv_"))
        step_kernel_row = {"kernel": "spmv_step_kernel", "workgroup": 1024, "launches_averaged": len(tail), "avg_us": round(avg, 2), "min_us": round(min(tail), 2),
                           "matrices": sum((v["matrices"] for k, v in classes_main.items() if k.startswith("spmv_")), []), "algorithmic_bytes_per_launch": ab,
                           "achieved_gbs": round(ab / avg / 1e3, 1), "frac_of_8TBs": round(ab / avg / 1e3 / 8000.0, 4)}
        (dst / f"{tag}_step_kernel_roofline.json").write_text(json.dumps({"method": "main kernel trace (default path), last 30 launches of spmv_step_kernel; algorithmic bytes = sum over "
                                                                                     "bench.py launch_classes (8*nnz+16*rows+4 per matrix); peak 8 TB/s", "kernel": step_kernel_row}, indent=1) + "\n")
        out += ["## The step kernel (default path, main trace, last 30 steps)", "",
                f"`spmv_step_kernel`: {avg:.2f} us per launch on average (min {min(tail):.2f}) for {ab / 1e6:.1f} MB algorithmic = {ab / avg / 1e3:.1f} GB/s = "
                f"**{ab / avg / 1e3 / 8000.0:.4f} of 8 TB/s** (one launch per step holds the groups and tiles of all {len(step_kernel_row['matrices'])} matrices; "
                "the tail launch follows it).", ""]

# Per-kernel roofline fraction from the SINGLE-STREAM trace (HISPMV_BATCH_STREAMS=1 HISPMV_BATCH_GRAPH=0: the launches of a
# step run one after the other, so a kernel's duration is its own): algorithmic bytes of the matrices in the grid
# (bench.py's "launch_classes", from the JSON line in the pass's log) / average duration / 8 TB/s.
single = first("trace_single/**/*kernel_trace.csv")
if single:
    classes = {}
    try:
        for line in (src / "trace_single.log").read_text().splitlines():
            if line.startswith("{") and "launch_classes" in line:
                classes = json.loads(line)["launch_classes"]
    except Exception:
        pass
    rows_k = [r for r in csv.DictReader(open(single)) if "hispmv::" in r["Kernel_Name"]]
    # the timed steps are the last 30 of 230: keep the last 30 launches of every (kernel, workgroup size)
    per = defaultdict(list)
    for r in rows_k:
        per[(re.sub(r"<.*", "", r["Kernel_Name"].split("(")[0].replace("void ", "").replace("hispmv::", "")).strip(), int(r["Workgroup_Size_X"]))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    table = []
    for (name, wg), d in sorted(per.items(), key=lambda kv: -sum(kv[1][-30:])):
        tail = d[-30:]
        avg = sum(tail) / len(tail)
        key = f"{name}/{wg}t" if ("slices_multi" in name or "tts_multi" in name) else name
        cl = classes.get(key, {})
        ab = cl.get("algorithmic_bytes_per_launch")
        table.append({"kernel": name, "workgroup": wg, "launches_averaged": len(tail), "avg_us": round(avg, 2), "min_us": round(min(tail), 2),
                      "matrices": cl.get("matrices"), "algorithmic_bytes_per_launch": ab,
                      "achieved_gbs": round(ab / avg / 1e3, 1) if ab else None, "frac_of_8TBs": round(ab / avg / 1e3 / 8000.0, 4) if ab else None})
    with open(dst / f"{tag}_kernel_stats_single_stream.csv", "w") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "workgroup_threads", "launches_averaged", "avg_us", "min_us", "algorithmic_bytes_per_launch", "achieved_GBs", "frac_of_8TBs", "matrices"])
        for t in table:
            w.writerow([t["kernel"], t["workgroup"], t["launches_averaged"], t["avg_us"], t["min_us"], t["algorithmic_bytes_per_launch"], t["achieved_gbs"],
                        t["frac_of_8TBs"], " ".join(t["matrices"] or [])])
    (dst / f"{tag}_per_kernel_roofline.json").write_text(json.dumps({"method": "single-stream kernel trace (HISPMV_BATCH_STREAMS=1 HISPMV_BATCH_GRAPH=0), last 30 launches of every kernel; "
                                                                      "algorithmic bytes per grid from bench.py launch_classes (8*nnz+16*rows+4 per matrix); peak 8 TB/s",
                                                                      "kernels": table, "step_us_sum_of_kernels": round(sum(t["avg_us"] for t in table), 1)}, indent=1) + "\n")
    out += ["## Per kernel, one stream (`HISPMV_BATCH_STREAMS=1 HISPMV_BATCH_GRAPH=0`: the grids of the classes one after the other, not the step kernel; last 30 steps)", "",
            "| kernel | workgroup | avg us | algorithmic MB per launch | GB/s | frac of 8 TB/s | matrices |", "|---|---:|---:|---:|---:|---:|---|"]
    for t in table:
        ab = t["algorithmic_bytes_per_launch"]
        out.append(f"| `{t['kernel']}` | {t['workgroup']} | {t['avg_us']:.2f} | {ab / 1e6:.1f} | {t['achieved_gbs']} | {t['frac_of_8TBs']} | {' '.join(t['matrices'] or [])} |" if ab else
                   f"| `{t['kernel']}` | {t['workgroup']} | {t['avg_us']:.2f} | – | – | – | (not counted as algorithmic) |")
    out += ["", f"Sum of the kernel averages: {sum(t['avg_us'] for t in table):.1f} us per step on one stream.", ""]


def pmc(kind, counter):
    f = first(f"{kind}/**/*counter_collection.csv")
    if not f:
        return None
    tot = defaultdict(float)
    calls = defaultdict(int)
    for r in csv.DictReader(open(f)):
        if r.get("Counter_Name") == counter:
            name = r["Kernel_Name"]
            tot[name] += float(r["Counter_Value"])
            calls[name] += 1
    return tot, calls


fetch, write = pmc("fetch", "FETCH_SIZE"), pmc("write", "WRITE_SIZE")
traffic = {}
if fetch or write:
    out += ["## HBM traffic (`--pmc FETCH_SIZE` and `--pmc WRITE_SIZE`, separate passes)", "",
            "| kernel | launches | FETCH_SIZE KiB/launch (raw) | read MB/launch (x2 gfx950 correction) | WRITE MB/launch |", "|---|---:|---:|---:|---:|"]
    names = set((fetch[0] if fetch else {}).keys()) | set((write[0] if write else {}).keys())
    for n in sorted(names, key=lambda k: -(fetch[0].get(k, 0) if fetch else 0)):
        if "hispmv" not in n:
            continue
        fk = fetch[0].get(n, 0) / max(1, fetch[1].get(n, 1)) if fetch else 0
        wk = write[0].get(n, 0) / max(1, write[1].get(n, 1)) if write else 0
        traffic[n[:60]] = {"launches": fetch[1].get(n, 0) if fetch else 0, "read_bytes_per_launch": fk * 1024 * 2, "write_bytes_per_launch": wk * 1024}
        out.append(f"| `{n[:60]}` | {fetch[1].get(n, 0) if fetch else 0} | {fk:.0f} | {fk*1024*2/1e6:.2f} | {wk*1024/1e6:.2f} |")
    out.append("")
PASSES = 8    # tools/profile_round.sh: --warmup 1 --steps 5 + bench.py's two untimed passes (stream assignment), no per-matrix pass
for log in ("fetch.log", "write.log", "trace.log"):     # bench.py states how often it went over the set
    try:
        for line in (src / log).read_text().splitlines():
            if line.startswith("{") and "passes_over_set" in line:
                PASSES = int(json.loads(line)["passes_over_set"])
        break
    except Exception:
        pass
tot_r = sum(v["read_bytes_per_launch"] * v["launches"] for v in traffic.values())
tot_w = sum(v["write_bytes_per_launch"] * v["launches"] for v in traffic.values())
summary = {"passes": PASSES, "hbm_read_bytes_per_step": tot_r / PASSES, "hbm_write_bytes_per_step": tot_w / PASSES,
           "hbm_bytes_per_step": (tot_r + tot_w) / PASSES, "kernels": traffic,
           "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), KiB units, FETCH_SIZE x2 (gfx950)"}
out += [f"HBM traffic per step (one pass over the 20 matrices): read {tot_r / PASSES / 1e6:.1f} MB + write {tot_w / PASSES / 1e6:.1f} MB "
        f"= {(tot_r + tot_w) / PASSES / 1e6:.1f} MB; algorithmic bytes per step: 1421.1 MB.", ""]
(dst / f"{tag}_summary.md").write_text("\n".join(out) + "\n")
(dst / f"{tag}_traffic.json").write_text(json.dumps(summary, indent=1) + "\n")
print("\n".join(out))
