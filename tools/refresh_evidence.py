#!/usr/bin/env python3
"""After tools/evidence_pass.sh has run on the GPU box (gpurun merges its output into gpurun_out/): copies the figures the
documents cite into profiles/ and regenerates the generated block of DESIGN.md ('results:begin' .. 'results:end').

    tools/refresh_evidence.py r2        # round tag: profiles/r2_*"""
import json
import re
import shutil
import subprocess
import sys
from pathlib import Path

root = Path(__file__).resolve().parents[1]
tag = sys.argv[1] if len(sys.argv) > 1 else "r2"
out = root / "gpurun_out"
prof = root / "profiles"
sys.path.insert(0, str(root))
from hispmv_amd import matrices as M  # noqa: E402

subprocess.run([sys.executable, str(root / "tools" / "summarize_prof.py"), tag], check=True, stdout=subprocess.DEVNULL)
for w in ("dense", "model", "powerlaw"):
    files = sorted((out / f"prof_{tag}_{w}" / "trace").glob("*/*_kernel_stats.csv"))
    if files:
        shutil.copy(files[-1], prof / f"{tag}_{w}_kernel_stats.csv")
if (out / "parity_report.json").exists():
    shutil.copy(out / "parity_report.json", prof / f"{tag}_parity_report.json")
drv = out / f"{tag}m" / "bench_driver_style.log"
if drv.exists():           # the line of `python bench.py --gpus 1 --steps 20 --warmup 5` (what the driver runs at round end)
    dl = [q for q in drv.read_text().splitlines() if q.startswith("{")]
    if dl:
        json.loads(dl[-1])
        (prof / f"{tag}_bench_line_driver_style.json").write_text(dl[-1] + "\n")
details = out / f"{tag}m" / "details_default.json"
line = (out / f"{tag}m" / "bench_default.log").read_text().strip().splitlines()[-1]
json.loads(line)
(prof / f"{tag}_bench_line.json").write_text(line + "\n")
shutil.copy(details, prof / f"{tag}_bench_details.json")

d = json.loads(details.read_text())
s, rows = d["summary"], d["per_matrix"]
fam = {name: f for name, _r, _n, f, _p in M.SUITESPARSE_SET}
traffic = json.loads((prof / f"{tag}_traffic.json").read_text())
alg = sum(M.algorithmic_bytes(r["rows"], r["rows"], r["nnz"]) for r in rows)
cb, su, ss, rf = s["cpu_baseline"], s["standin_uniform"], s["strong_scaling"], s["roofline"]
hbm = traffic.get("hbm_bytes_per_step")
drv_note = ""
if (prof / f"{tag}_bench_line_driver_style.json").exists():
    dd = json.loads((prof / f"{tag}_bench_line_driver_style.json").read_text())
    drv_note = (f" Driver-style (`--gpus 1 --steps 20 --warmup 5`, `profiles/{tag}_bench_line_driver_style.json`): {dd['value']} GFLOP/s, "
                f"{dd['ms_per_step']} ms per step, frac {dd['roofline']['frac']} (pessimistic family {dd['standin_uniform']['roofline_frac']}); "
                f"both lines carry `y_checked: {str(dd['y_checked']).lower()}` (max backward error {dd['y_check']['max_backward_error']:.2e} over {dd['y_check']['rows_checked']} rows).")
lines = [
    f"`python bench.py` (defaults: {s['steps']} steps, {s['warmup']} warm-up; `profiles/{tag}_bench_line.json`, "
    f"`profiles/{tag}_bench_details.json`): **{s['value']} GFLOP/s** over the set, {s['ms_per_step']} ms per step, "
    f"{s['hbm_gbs_algorithmic']} GB/s algorithmic = **{s['hbm_pct_of_peak']} % of the 8 TB/s peak**; `roofline.achieved` "
    f"{rf['achieved']} GB/s (HIP events), `frac` {rf['frac']}." + drv_note,
    f"The same step with the pessimistic stand-in family (`standin_uniform`): {su['value']} GFLOP/s, {su['ms_per_step']} ms, "
    f"frac {su['roofline_frac']}. Six largest matrices alone (`strong_scaling` at n_gpus = 1): {ss['value']} GFLOP/s, "
    f"{ss['ms_per_step']} ms per step.",
    f"CPU baseline on the same box ({cb['host']['logical']} logical / {cb['host']['physical']} physical CPUs visible, cgroup quota "
    f"{cb['host']['cgroup_cpu_max']:.0f} CPUs → {cb['cores']} threads): MKL `mkl_sparse_s_mv` "
    f"{max([v for k, v in cb.items() if k.startswith('mkl_') and k.endswith('_gflops')] or [None])} GFLOP/s, OpenMP restatement "
    f"{max([v for k, v in cb.items() if k.startswith('omp_') and k.endswith('_gflops')] or [None])} GFLOP/s (`cpu_baseline.value` = the better of the two: "
    f"{cb['value']}), the reference's single-thread loop {cb.get('cpu_spmv_1_thread_gflops')} GFLOP/s.",
]
if hbm:
    lines.append(f"rocprofv3 (`profiles/{tag}_summary.md`): HBM traffic {hbm / 1e6:.0f} MB per step against {alg / 1e6:.0f} MB algorithmic "
                 f"({hbm / alg:.2f}×: the 6-byte elements read less than the definition counts).")
lines += ["", "Per matrix, each timed alone between two HIP events with the largest matrix streamed in between (cold Infinity Cache; GFLOP/s and",
          "GB/s from that time), and as the average of 20 launches back to back (the reference's `rp_time` loop, `spmv-host.cpp:120-154`);",
          "plan = threads / slices per workgroup / LDS window / column tiles / share of compact (6-byte) slices; `1024t/28s/0KiB` = tile "
          "stream (28-slice blocks):", "",
          "| matrix (stand-in family) | rows | nnz | µs | µs back to back | GFLOP/s | alg. GB/s | % of 8 TB/s | plan |", "|---|---:|---:|---:|---:|---:|---:|---:|---|"]
for r in rows:
    lines.append(f"| {r['name']} ({fam.get(r['name'], '?')}) | {r['rows']} | {r['nnz']} | {r['us']} | {r.get('us_back_to_back', '–')} | {r['gflops']} | {r['alg_gbs']} | "
                 f"{r['pct_hbm_peak']} | {r['plan']} |")
block = "<!-- results:begin (tools/refresh_evidence.py) -->\n" + "\n".join(lines) + "\n<!-- results:end -->"
p = root / "DESIGN.md"
t = p.read_text()
m = re.search(r"<!-- results:begin.*?<!-- results:end -->", t, re.S)
assert m, "DESIGN.md has no results block"
p.write_text(t[:m.start()] + block + t[m.end():])
for w in ("dense", "model", "powerlaw"):
    f = out / f"{tag}m" / f"details_{w}.json"
    if f.exists():
        sw = json.loads(f.read_text())["summary"]
        shutil.copy(f, prof / f"{tag}_bench_details_{w}.json")
        print(f"{w}: {sw['ms_per_step']} ms per step, {sw['value']} {sw['unit']}, frac {sw['roofline']['frac']}  (DESIGN.md 'Other workloads' is prose: update by hand)")
print("profiles/ and DESIGN.md refreshed from", details)
