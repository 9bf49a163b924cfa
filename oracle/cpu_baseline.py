#!/usr/bin/env python3
"""oracle/cpu_baseline.py -- TEST INFRASTRUCTURE ONLY: the `cpu_baseline` leg of bench.py, run as a CHILD process.

The reference times its CPU path with MKL on 24 pinned threads (cpu/src/main.cpp:136, cpu/env.sh:2-4:
OMP_NUM_THREADS=24 OMP_PLACES=cores OMP_PROC_BIND=close, 200 repetitions per matrix, cpu/run_spmv.sh:6).  This
script does the same on the GPU box's host cores for a bounded sample of the SAME workload bench.py times on the
GPU, in a process that imports neither torch nor HIP and holds ONE OpenMP runtime:

  --impl mkl   mkl_sparse_s_mv / cblas_sgemv through libmklbench.so (only MKL's own libiomp5 in the process)
  --impl omp   the OpenMP restatement of cpu_spmv / naive_gemv in liboracle.so (libgomp)
               (--one-thread adds one pass of the reference's single-thread loops, cpu/src/main.cpp:11-23, :53-71)

The caller sets the environment (places, binding) and the thread counts to sweep; this script prints one JSON object:
per thread count the flops, seconds, repetitions and GFLOP/s of the sample.  Matrices are regenerated here from their
seeds (hispmv_amd/matrices.py is loaded as a plain file: importing the package would load the HIP library).
"""
from __future__ import annotations

import argparse
import ctypes as C
import importlib.util
import json
import sys
import time
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent


def load_matrices_module():
    spec = importlib.util.spec_from_file_location("hispmv_matrices", ROOT / "hispmv_amd" / "matrices.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def workload(M, name: str, names):
    """-> list of ("sparse", label, rows, cols, rp, ci, va) / ("dense", label, rows, cols)."""
    out = []
    if name == "set":
        for m in M.benchmark_set(names or None, False):
            if "rp" in m:
                out.append(("sparse", m["name"], m["rows"], m["cols"], m["rp"], m["ci"], m["va"]))
    elif name == "powerlaw":
        n, _, r, c, v = M.rmat_coo(20)
        rp, ci, va = M.coo_to_csr_sorted(r, c, v, n)
        out.append(("sparse", "rmat20", n, n, rp, ci, va))
        rp, ci, va = M.zipf_csr(1632803, 1632803, 30622600, 1.2, 7)
        out.append(("sparse", "zipf1.2_pokec_shape", 1632803, 1632803, rp, ci, va))
    elif name == "model":
        for i, (kind, W, rows, cols, _b) in enumerate(M.model_test_layers(0)):
            if kind == "dense":
                out.append(("dense", f"layer{i}", rows, cols))
            else:
                rp, ci, va = M.coo_to_csr_sorted(W[0], W[1], W[2], rows)
                out.append(("sparse", f"layer{i}", rows, cols, rp, ci, va))
    elif name == "dense":
        for n in (512, 1024, 2048, 4096, 8192):            # cpu/run_gemv.sh:9-13
            out.append(("dense", f"gemv_{n}x{n}", n, n))
    else:
        raise SystemExit(f"unknown workload {name}")
    return out


def run_items(args, items, threads, budget, lib, oracle, flops_of, total_flops):
    res = dict(threads=threads, items=[], flops=0.0, seconds=0.0)
    for it in items:
        share = budget * flops_of(it) / total_flops          # time in proportion to the work, like equal repetitions
        reps = C.c_int()
        if it[0] == "sparse":
            _, label, rows, cols, rp, ci, va = it
            if args.impl == "mkl":
                thr = C.c_int()
                t = lib.mb_spmv(rows, cols, rp.ctypes.data, ci.ctypes.data, va.ctypes.data, threads, share, 200, C.byref(reps), C.byref(thr))
                n = reps.value
            else:
                t, n = oracle.cpu_bench_spmv(rp, ci, va, cols, 0, threads, share, 200)
        else:
            _, label, rows, cols = it
            if args.impl == "mkl":
                t = lib.mb_gemv(rows, cols, threads, share, 10000, C.byref(reps))     # cpu/run_gemv.sh:6 (10000 there)
                n = reps.value
            else:
                t, n = oracle.cpu_bench_gemv(rows, cols, 0, threads, share, 10000)
        if t is None or t < 0:
            return dict(res, error=f"{label}: failed ({t})")
        res["items"].append(dict(name=label, seconds_per_rep=t, reps=n, gflops=flops_of(it) / t / 1e9))
        res["flops"] += flops_of(it) * n
        res["seconds"] += t * n
    res["gflops"] = res["flops"] / res["seconds"] / 1e9 if res["seconds"] > 0 else None
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--impl", choices=["mkl", "omp"], required=True)
    ap.add_argument("--threads", default="1", help="comma-separated thread counts to sweep (one process, one OpenMP runtime)")
    ap.add_argument("--workload", default="set")
    ap.add_argument("--names", default="")
    ap.add_argument("--budget", type=float, default=10.0, help="seconds of timed CPU work per thread count")
    ap.add_argument("--one-thread", action="store_true",
                    help="omp only: also one pass of the reference's single-thread loops over the three largest items")
    args = ap.parse_args()
    M = load_matrices_module()
    items = workload(M, args.workload, [n for n in args.names.split(",") if n])
    items = [(it[0], it[1], it[2], it[3]) + tuple(np.ascontiguousarray(a, dt) for a, dt in zip(it[4:], (np.int32, np.int32, np.float32)))
             for it in items]
    flops_of = lambda it: (2.0 * (int(it[4][-1]) + it[2]) if it[0] == "sparse" else 2.0 * it[2] * it[3] + it[2])
    total_flops = sum(flops_of(it) for it in items)
    out = dict(impl=args.impl, workload=args.workload, n_items=len(items), runs=[])
    lib = oracle = None
    if args.impl == "mkl":
        lib = C.CDLL(str(HERE / "libmklbench.so"))
        lib.mb_spmv.restype = C.c_double
        lib.mb_spmv.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        lib.mb_gemv.restype = C.c_double
        lib.mb_gemv.argtypes = [C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, C.POINTER(C.c_int)]
        if not lib.mb_mkl_available():
            print(json.dumps(dict(out, error="libmkl_rt not found")))
            return
    else:
        sys.path.insert(0, str(ROOT))
        import oracle                              # liboracle.so (libgomp)
    for t in [int(q) for q in args.threads.split(",") if q]:
        out["runs"].append(run_items(args, items, t, args.budget, lib, oracle, flops_of, total_flops))
    if args.one_thread and args.impl == "omp":
        fl = sec = 0.0
        for it in sorted(items, key=lambda q: -flops_of(q))[:3]:
            if it[0] == "sparse":
                _, label, rows, cols, rp, ci, va = it
                x = ((np.arange(cols, dtype=np.float32) + 1) / (np.arange(cols, dtype=np.float32) + 2)).astype(np.float32)
                y = (np.float32(-2.0) * (np.arange(rows, dtype=np.float32) + 1) / (np.arange(rows, dtype=np.float32) + 2)).astype(np.float32)
                t0 = time.perf_counter()
                oracle.cpu_spmv(rp, ci, va, x, y, 0.85, -2.06, 1)          # cpu/src/main.cpp:11-23
                sec += time.perf_counter() - t0
            else:
                t, _n = oracle.cpu_bench_gemv(it[2], it[3], 0, 1, 0.0, 2)     # cpu/src/main.cpp:53-71
                sec += t
            fl += flops_of(it)
        out["one_thread_gflops"] = fl / sec / 1e9 if sec > 0 else None
    print(json.dumps(out))


if __name__ == "__main__":
    main()
