#!/usr/bin/env python3
"""Condenses gpurun_out/prof_<tag>/ (written by tools/profile_round.sh on the GPU box) into the small
tracked files profiles/<tag>_kernel_stats.csv and profiles/<tag>_summary.md.

HBM traffic follows MI355X_MICROARCH.md "HBM": FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE
reports exactly half of the bytes of a wide (16 B/lane) coalesced streaming read, so the read side is
doubled; WRITE_SIZE is exact for dword-per-lane stores.  Collected in separate --pmc passes."""
import csv
import glob
import json
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
        if "merge_multi" in name:
            spans.append((e0 - start) / 1e3)
            start = None
    if spans:
        tail = spans[-30:] if len(spans) > 60 else spans[len(spans) // 2:]          # the timed steps (past warm-up)
        out += [f"Step span in the trace (first launch of a step to the end of its merge launch, last {len(tail)} steps): "
                f"{sum(tail) / len(tail):.1f} us on average (min {min(tail):.1f}); under the profiler the chip clocks lower than in an "
                "un-profiled run (MI355X_MICROARCH.md, DVFS), bench.py's HIP-event time is the un-profiled figure.", ""]


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
