#!/usr/bin/env python3
"""Per-CU occupancy of one benchmark step, from the workgroup trace of the diagnostic library.

    make -C hispmv_amd/csrc wgtrace                          # libhispmv_wgtrace.so (kernels built with -DHISPMV_WG_TRACE=1)
    HISPMV_LIB=$PWD/hispmv_amd/lib/libhispmv_wgtrace.so python3 tools/wg_timeline.py [--uniform] [--out gpurun_out/wg.json]

Every workgroup of the multi-matrix kernels records {start, end} on the 100 MHz constant clock, the CU it ran on
(XCC_ID + HW_ID) and what it was.  For ONE step in steady state (the step after 200 warm-up steps) this prints and writes:
  * per kind (1024-thread slice groups, 256-thread slice groups, tiles): workgroups, CU-time, mean / max duration, first
    start and last end relative to the step;
  * per CU: busy time = union of its workgroups' intervals, idle time inside the step, number of workgroups;
  * the gaps between consecutive workgroups on a CU (what a workgroup change costs) and the idle time at the step's end;
  * per table entry (matrix part): workgroups, mean duration, span.
The trace costs a barrier and ~6 scalar instructions per workgroup; the step with it is within 1 % of the product library's.
"""
import argparse
import ctypes
import json
import os
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--uniform", action="store_true", help="the pessimistic stand-in family")
    ap.add_argument("--matrices", default="")
    ap.add_argument("--powerlaw", action="store_true", help="bench.py's powerlaw workload (R-MAT scale 20 + Zipf at soc-Pokec's shape) instead of the set")
    ap.add_argument("--out", default="gpurun_out/wg_timeline.json")
    ap.add_argument("--warm", type=int, default=200)
    ap.add_argument("--steps", type=int, default=3, help="traced steps (the LAST one is analysed, all are summarised)")
    args = ap.parse_args()
    lib_path = os.environ.get("HISPMV_LIB", "")
    assert "wgtrace" in lib_path, "run with HISPMV_LIB=.../libhispmv_wgtrace.so"
    import torch
    import bench
    from hispmv_amd import _lib
    setter = ctypes.CDLL(lib_path).hispmv_wg_trace_set
    setter.argtypes = [ctypes.c_void_p, ctypes.c_uint]
    setter.restype = ctypes.c_int

    names = [n for n in args.matrices.split(",") if n]
    r = bench.Runner(0, 1)
    mats = bench.load_powerlaw() if args.powerlaw else bench.load_set(names, 0, 1, args.uniform)
    r.add(mats)
    step = r.batch_step(mats)
    for _ in range(args.warm):
        step()
    torch.cuda.synchronize()
    cap = 1 << 16
    out_steps = []
    for s in range(args.steps):
        buf = torch.zeros(4 + 4 * cap, dtype=torch.int64, device=r.dev)
        torch.cuda.synchronize()
        assert setter(buf.data_ptr(), cap) == 0
        for _ in range(5):              # the traced step is the LAST of five back-to-back steps: steady state, no host gap before it
            step()
        torch.cuda.synchronize()
        assert setter(None, 0) == 0
        raw = buf.cpu().numpy().view(np.uint64)
        n = int(raw[0])
        assert n <= cap, "trace buffer too small"
        rec = raw[4:4 + 4 * n].reshape(n, 4)
        per_step = n // 5
        assert per_step * 5 == n
        # records arrive in completion order; the steps do not overlap (the tail launch separates them): sort by start, cut in five
        rec = rec[np.argsort(rec[:, 0], kind="stable")]
        out_steps.append(analyse(rec[-per_step:], mats))
    res = dict(standin="uniform" if args.uniform else "structured", lib=os.path.basename(lib_path), steps=out_steps)
    os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(res, f, indent=1)
    last = out_steps[-1]
    print(json.dumps({k: last[k] for k in ("span_us", "cus_seen", "cu_busy_frac", "per_kind", "gaps", "end_of_step")}, indent=1))


def analyse(rec, mats):
    t0 = rec[:, 0].astype(np.int64)
    t1 = rec[:, 1].astype(np.int64)
    base = t0.min()
    a = (t0 - base) / 100.0          # us
    b = (t1 - base) / 100.0
    hw = (rec[:, 2] & np.uint64(0xffffffff)).astype(np.int64)
    xcc = (rec[:, 2] >> np.uint64(32)).astype(np.int64)
    cu = (xcc << 8) | ((hw >> 8) & 0xff)             # XCC_ID | SE_ID, SH_ID, CU_ID of HW_ID
    kind = (rec[:, 3] >> np.uint64(56)).astype(np.int64)
    entry = ((rec[:, 3] >> np.uint64(40)) & np.uint64(0xffff)).astype(np.int64)
    span = float(b.max())
    names = {1: "slices_1024t", 2: "slices_256t", 3: "tiles_1024t"}
    per_kind = {}
    for k, nm in names.items():
        m = kind == k
        if not m.any():
            continue
        d = b[m] - a[m]
        per_kind[nm] = dict(workgroups=int(m.sum()), cu_time_us=round(float(d.sum()), 1), mean_us=round(float(d.mean()), 2),
                            max_us=round(float(d.max()), 2), first_start_us=round(float(a[m].min()), 2), last_end_us=round(float(b[m].max()), 2))
    cus = np.unique(cu)
    busy, gaps_all, n_wg, last_end = [], [], [], []
    # a CU hosts up to four 256-thread workgroups at once: busy = union of intervals
    for c in cus:
        m = cu == c
        o = np.argsort(a[m])
        aa, bb = a[m][o], b[m][o]
        cur_s, cur_e, tot = aa[0], bb[0], 0.0
        for s, e in zip(aa[1:], bb[1:]):
            if s > cur_e:
                tot += cur_e - cur_s
                gaps_all.append(s - cur_e)
                cur_s, cur_e = s, e
            else:
                cur_e = max(cur_e, e)
        tot += cur_e - cur_s
        busy.append(tot)
        n_wg.append(int(m.sum()))
        last_end.append(cur_e)
    busy = np.array(busy)
    gaps_all = np.array(gaps_all) if gaps_all else np.zeros(1)
    last_end = np.array(last_end)
    per_entry = []
    for k in names:
        for e in np.unique(entry[kind == k]):
            m = (kind == k) & (entry == e)
            d = b[m] - a[m]
            per_entry.append(dict(kind=names[k], entry=int(e), workgroups=int(m.sum()), mean_us=round(float(d.mean()), 2),
                                  max_us=round(float(d.max()), 2), first_start_us=round(float(a[m].min()), 1),
                                  last_end_us=round(float(b[m].max()), 1), cu_time_us=round(float(d.sum()), 1)))
    return dict(
        workgroups=int(len(rec)), span_us=round(span, 2), cus_seen=int(len(cus)),
        cu_busy_frac=round(float(busy.sum() / (len(cus) * span)), 4),
        cu_busy_us=dict(mean=round(float(busy.mean()), 1), min=round(float(busy.min()), 1), max=round(float(busy.max()), 1)),
        workgroups_per_cu=dict(mean=round(float(np.mean(n_wg)), 1), min=int(np.min(n_wg)), max=int(np.max(n_wg))),
        per_kind=per_kind,
        gaps=dict(count=int(len(gaps_all)), mean_us=round(float(gaps_all.mean()), 2), median_us=round(float(np.median(gaps_all)), 2),
                  p90_us=round(float(np.percentile(gaps_all, 90)), 2), sum_per_cu_us=round(float(gaps_all.sum() / len(cus)), 1)),
        end_of_step=dict(mean_idle_before_end_us=round(float((span - last_end).mean()), 1),
                         cus_done_20us_early=int((span - last_end > 20).sum()), cus_done_10us_early=int((span - last_end > 10).sum())),
        start_of_step=dict(note="first workgroup start per CU, us", mean=round(float(np.mean([a[cu == c].min() for c in cus])), 2)),
        per_entry=per_entry,
        matrices=[m["name"] for m in mats],
    )


if __name__ == "__main__":
    main()
