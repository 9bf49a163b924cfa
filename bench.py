#!/usr/bin/env python3
"""bench.py -- the reference's headline benchmark on MI355X: fp32 y = alpha*A*x + beta*y over the
20-matrix SuiteSparse set of get_tb_matrices.py:57-78 (BASELINE.json configs[1]); metric = the
reference's own GFLOP/s convention 2*(nnz+rows)/t (spmv-host.cpp:100,185) plus achieved HBM GB/s.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass over the set: one SpMV launch per matrix, back to back on one HIP stream,
every operand already resident in HBM.  The working set of a step (~1.45 GB of packed stream) is
far larger than the 256 MiB Infinity Cache, so every launch streams its matrix from HBM.
SuiteSparse files cannot be downloaded here; matrices/<name>/<name>.mtx is used when present,
otherwise the seeded stand-in with the real matrix's rows and nnz (hispmv_amd/matrices.py).

N > 1 (weak scaling): rank k holds the k-th row block of the set scaled N-fold (same per-GPU nnz),
split on the nnz prefix; x is replicated; the partial sums of rows cut by a rank boundary are
exchanged with one RCCL all_gather per step (hispmv_amd/dist.py) -- never an all-reduce of y.

Prints ONE JSON line on rank 0 (contract in the task statement) with "roofline" and "cpu_baseline".
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

ALPHA, BETA = 0.55, -2.05            # common/src/spmv-host.cpp:43-44 (the FPGA/GPU drivers' scalars)
HBM_PEAK_GBS = 8000.0                # MI355X spec (MI355X_MICROARCH.md, HBM)
HW = ("bench.xclbin", 24, 1, 1, 2, 5, True, False, True)   # apps tuple; only sizes the default arena


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--matrices", type=str, default="", help="comma-separated subset of the set (debug)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--per-matrix-reps", type=int, default=10)
    ap.add_argument("--details", type=str, default="", help="write the per-matrix table to this JSON file")
    ap.add_argument("--streams", type=int, default=4,
                    help="HIP streams the SpMVs of a step are spread over (independent matrices may overlap)")
    ap.add_argument("--launch", choices=["batch", "streams"], default="batch",
                    help="batch: one hispmv_spmv_device_batch call per step (matrices with the same workgroup size share a "
                         "grid); streams: one launch per matrix, spread over --streams HIP streams")
    ap.add_argument("--standin", choices=["structured", "uniform"], default="structured",
                    help="stand-in family of the mesh-origin matrices: structured FEM-like (default) or unstructured band")
    return ap.parse_args()


def load_set(names, rank, world, uniform=False):
    """-> list of dicts with the rank's CSR shard of every matrix (host arrays)."""
    from hispmv_amd import matrices as M
    out = []
    import zlib
    for name, rows, nnz, fam, par in M.SUITESPARSE_SET:
        if names and name not in names:
            continue
        real = M.real_matrix_path(name)
        if real is not None and world == 1:
            out.append(dict(name=name, source="file:" + str(real), path=str(real)))
            continue
        seed = zlib.crc32(name.encode())
        rp, ci, va, used = M.make_standin(name, rows, nnz, fam, par, seed + rank, uniform)
        if world == 1:
            out.append(dict(name=name, source=f"synthetic:{used}", rows=rows, cols=rows, nnz=int(rp[-1]), rp=rp, ci=ci, va=va))
            continue
        # N-fold scaled matrix = `world` stacked blocks; this rank's shard is cut inside rows on both sides
        from hispmv_amd.dist import shard_of_stacked_blocks
        nxt = M.make_standin(name, rows, nnz, fam, par, seed + rank + 1, uniform)[:3] if rank + 1 < world else None
        sh = shard_of_stacked_blocks((rp, ci, va), nxt, rows, rows, rank, world)
        out.append(dict(name=name, source=f"synthetic:{used}", rows=sh.n_rows, cols=rows * world, nnz=int(sh.row_ptr[-1]),
                        rp=sh.row_ptr, ci=sh.col_idx, va=sh.values, shard=sh))
    return out


def cpu_baseline(mats, budget_s=20.0):
    """The oracle (kind "port") timed on the host cores over the same matrices: the OpenMP CSR
    restatement of cpu/src/main.cpp:11-23 and, when the image has libmkl_rt, mkl_sparse_s_mv called
    exactly as cpu/src/main.cpp:26-49 does (the reference's timed CPU path, 200 reps in
    cpu/run_spmv.sh:6; here reps are bounded so the whole leg stays within ~budget_s)."""
    import oracle   # checker/baseline only -- never on the product path
    cores = len(os.sched_getaffinity(0))
    os.environ.setdefault("OMP_NUM_THREADS", str(cores))
    res = {"omp": [0.0, 0.0], "mkl": [0.0, 0.0]}
    used = {"omp": cores, "mkl": 0}
    per_matrix_budget = budget_s / max(1, len(mats)) / 2
    sample = []
    for m in mats:
        if "rp" not in m:
            continue
        rows, cols, nnz = m["rows"], m["cols"], m["nnz"]
        x = ((np.arange(cols, dtype=np.float32) + 1) / (np.arange(cols, dtype=np.float32) + 2)).astype(np.float32)
        y0 = (np.float32(-2.0) * (np.arange(rows, dtype=np.float32) + 1) / (np.arange(rows, dtype=np.float32) + 2)).astype(np.float32)
        fl = 2.0 * (nnz + rows)
        t1, nt, _ = oracle.omp_spmv_timed(m["rp"], m["ci"], m["va"], x, y0, 0.85, -2.06, 1)
        reps = int(min(200, max(2, per_matrix_budget / max(t1, 1e-6))))
        t, nt, _ = oracle.omp_spmv_timed(m["rp"], m["ci"], m["va"], x, y0, 0.85, -2.06, reps)
        res["omp"][0] += fl * reps; res["omp"][1] += t * reps; used["omp"] = nt
        if oracle.mkl_available():
            r1 = oracle.mkl_spmv(m["rp"], m["ci"], m["va"], cols, x, y0, 0.85, -2.06, 1, cores)
            if r1 is not None:
                reps_m = int(min(200, max(2, per_matrix_budget / max(r1[0], 1e-6))))
                r = oracle.mkl_spmv(m["rp"], m["ci"], m["va"], cols, x, y0, 0.85, 0.0, reps_m, cores)
                res["mkl"][0] += fl * reps_m; res["mkl"][1] += r[0] * reps_m; used["mkl"] = r[1]
        sample.append(m["name"])
    # the reference's own single-thread loop (cpu_spmv, cpu/src/main.cpp:11-23) on the three largest matrices, one pass each
    one_fl = one_t = 0.0
    for m in sorted((q for q in mats if "rp" in q), key=lambda q: -q["nnz"])[:3]:
        x = ((np.arange(m["cols"], dtype=np.float32) + 1) / (np.arange(m["cols"], dtype=np.float32) + 2)).astype(np.float32)
        y0 = np.zeros(m["rows"], np.float32)
        t0 = time.perf_counter()
        oracle.cpu_spmv(m["rp"], m["ci"], m["va"], x, y0, 0.85, -2.06, 1)
        one_t += time.perf_counter() - t0
        one_fl += 2.0 * (m["nnz"] + m["rows"])
    omp = res["omp"][0] / res["omp"][1] / 1e9 if res["omp"][1] > 0 else None
    mkl = res["mkl"][0] / res["mkl"][1] / 1e9 if res["mkl"][1] > 0 else None
    primary = "mkl_sparse_s_mv" if mkl is not None else "openmp_csr"
    return {
        "value": round(mkl if mkl is not None else omp, 3), "unit": "GFLOP/s",
        "cores": used["mkl"] if mkl is not None else used["omp"], "kind": "port", "impl": primary,
        "openmp_csr_gflops": None if omp is None else round(omp, 3),
        "mkl_gflops": None if mkl is None else round(mkl, 3),
        "cpu_spmv_1_thread_gflops": round(one_fl / one_t / 1e9, 3) if one_t > 0 else None,
        "sample": f"{len(sample)} matrices of the same set ({', '.join(sample[:3])}...), reps bounded to ~{budget_s:.0f} s total "
                  f"(reference: 200 reps each, cpu/run_spmv.sh:6), alpha=0.85 beta=-2.06 as cpu/src/main.cpp:147-148",
    }


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("launch with `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`")
        args.gpus = world

    import torch
    import torch.distributed as dist
    import pyhispmv
    from hispmv_amd import matrices as M

    # HISPMV_BENCH_REHEARSAL=1: every rank on GPU 0 with the gloo backend -- rehearses the N > 1 code path on a
    # one-GPU box (numbers from such a run mean nothing)
    rehearsal = os.environ.get("HISPMV_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    names = [n for n in args.matrices.split(",") if n]
    t0 = time.time()
    mats = load_set(names, rank, world, args.standin == "uniform")
    t_gen = time.time() - t0

    fpga = pyhispmv.FpgaHandle(HW[0], local_rank, *HW[1:])
    fpga.set_arena_bytes(64 << 30)
    t0 = time.time()
    for m in mats:
        if "path" in m:
            m["idx"] = fpga.create_sparse_handle_from_mtx(m["path"], 0)
        else:
            m["idx"] = fpga.create_sparse_handle_from_csr(m["rp"], m["ci"], m["va"], m["rows"], m["cols"])
        assert m["idx"] >= 0, f"{m['name']}: arena full"
    fpga.load_matrices()
    t_prep = time.time() - t0
    for m in mats:
        info = fpga.matrix_info(m["idx"])
        m.update(rows=info["rows"], cols=info["cols"], nnz=info["nnz"], n_slices=info["n_slices"],
                 device_bytes=info["device_bytes"], prep_seconds=info["prep_seconds"], n_split=info["n_split_rows"],
                 plan=f'{info["block_threads"]}t/{info["group_slices"]}s/{info["lds_bytes"] // 1024}KiB/{info["col_tiles"]}ct')
        g = torch.Generator(device="cpu").manual_seed(1234 + m["idx"])
        m["x"] = torch.rand(m["cols"], generator=g, dtype=torch.float32).to(dev)
        m["b"] = torch.rand(m["rows"], generator=g, dtype=torch.float32).to(dev)
        m["y"] = torch.zeros(m["rows"], dtype=torch.float32, device=dev)
        if m.get("shard") is not None and m["shard"].tail_open:
            m["b"][-1] = 0.0        # the cut row belongs to the next rank: this rank only contributes alpha*partial

    # a dedicated (non-default) HIP stream: every launch, event and collective of the timed region is on it
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    sptr = stream.cuda_stream
    assert sptr != 0
    if world > 1:
        from hispmv_amd.dist import BoundaryExchange
        exch = BoundaryExchange(len(mats), dev)

    # optional extra streams: the matrices of a step are independent, so their launches may overlap; every
    # side stream is fenced against the main stream's start and end events
    n_streams = max(1, args.streams)
    side = [torch.cuda.Stream(device=dev) for _ in range(n_streams - 1)]
    lanes = [stream] + side
    # longest-processing-time-first: one untimed pass measures each launch alone, then every matrix goes to the
    # stream with the least work so far
    cost = {}
    for i, m in enumerate(mats):
        fpga.spmv_device(m["idx"], m["x"].data_ptr(), m["b"].data_ptr(), m["y"].data_ptr(), ALPHA, BETA, sptr)
        a_ev, b_ev = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a_ev.record(stream)
        fpga.spmv_device(m["idx"], m["x"].data_ptr(), m["b"].data_ptr(), m["y"].data_ptr(), ALPHA, BETA, sptr)
        b_ev.record(stream)
        torch.cuda.synchronize()
        cost[i] = a_ev.elapsed_time(b_ev)
    order = sorted(range(len(mats)), key=lambda i: -cost[i])
    load = [0.0] * n_streams
    for i in order:
        k = min(range(n_streams), key=lambda q: load[q])
        mats[i]["lane"] = k
        load[k] += cost[i]

    batch = fpga.prepare_batch([m["idx"] for m in mats], [m["x"].data_ptr() for m in mats], [m["b"].data_ptr() for m in mats],
                               [m["y"].data_ptr() for m in mats])

    def step_batch():
        fpga.spmv_device_batch(batch, ALPHA, BETA, sptr)
        if world > 1:
            exch.run(mats, ALPHA)

    def step_streams():
        if n_streams > 1:
            fork = torch.cuda.Event()
            fork.record(stream)
            for s2 in side:
                s2.wait_event(fork)
        for i in order:
            m = mats[i]
            fpga.spmv_device(m["idx"], m["x"].data_ptr(), m["b"].data_ptr(), m["y"].data_ptr(), ALPHA, BETA,
                             lanes[m["lane"]].cuda_stream)
        if n_streams > 1:
            for s2 in side:
                join = torch.cuda.Event()
                join.record(s2)
                stream.wait_event(join)
        if world > 1:
            exch.run(mats, ALPHA)

    step = step_batch if args.launch == "batch" else step_streams
    if args.launch == "batch":
        n_streams = 1

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t_start = time.perf_counter()
    ev0.record(stream)
    for _ in range(args.steps):
        step()
    ev1.record(stream)
    fence()
    t_wall = time.perf_counter() - t_start
    t_dev = ev0.elapsed_time(ev1) * 1e-3          # HIP events on the launch stream
    fpga.synchronize()                            # raises if a bounded in-kernel wait (carry look-back) expired
    if world > 1:
        tt = torch.tensor([t_wall, t_dev], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        t_wall, t_dev = float(tt[0]), float(tt[1])

    # per-matrix table (outside the timed region): events around `reps` launches of one matrix; the
    # other matrices are touched in between so that each measurement starts from a cold Infinity Cache
    table = []
    if rank == 0 and args.per_matrix_reps > 0:
        big = max(mats, key=lambda q: q["device_bytes"])
        for m in mats:
            ts = []
            for _ in range(max(1, args.per_matrix_reps)):
                if m is not big or len(mats) == 1:
                    fpga.spmv_device(big["idx"], big["x"].data_ptr(), big["b"].data_ptr(), big["y"].data_ptr(), ALPHA, BETA, sptr)
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(stream)
                fpga.spmv_device(m["idx"], m["x"].data_ptr(), m["b"].data_ptr(), m["y"].data_ptr(), ALPHA, BETA, sptr)
                b.record(stream)
                torch.cuda.synchronize()
                ts.append(a.elapsed_time(b) * 1e-3)
            t = float(np.median(ts))
            ab = M.algorithmic_bytes(m["rows"], m["cols"] // world, m["nnz"])
            table.append(dict(name=m["name"], source=m["source"], rows=m["rows"], nnz=m["nnz"], us=round(t * 1e6, 2),
                              gflops=round(M.flops(m["rows"], m["nnz"]) / t / 1e9, 2), alg_gbs=round(ab / t / 1e9, 1),
                              pct_hbm_peak=round(100 * ab / t / 1e9 / HBM_PEAK_GBS, 2), slices=m["n_slices"],
                              split_rows=m["n_split"], prep_s=round(m["prep_seconds"], 3), plan=m["plan"]))

    flops_step = sum(M.flops(m["rows"], m["nnz"]) for m in mats)
    # (x is replicated at full length on every rank, but a rank's block touches 1/world of it: count that part)
    bytes_step = sum(M.algorithmic_bytes(m["rows"], m["cols"] // world, m["nnz"]) for m in mats)
    if rank == 0:
        total_flops = flops_step * world * args.steps
        total_bytes = bytes_step * world * args.steps
        value = total_flops / t_wall / 1e9
        achieved = bytes_step * args.steps / t_dev / 1e9          # per GPU, device time
        launches = len(mats) * args.steps
        geo = math.exp(sum(math.log(r["gflops"]) for r in table) / len(table)) if table else None
        # HBM traffic per launch from the PMC passes of tools/profile_round.sh (same command under rocprofv3;
        # FETCH_SIZE x2 + WRITE_SIZE, MI355X_MICROARCH.md "HBM"), if a summary has been committed
        traffic, traffic_src = None, None
        cands = sorted((ROOT / "profiles").glob("*_traffic.json"), key=lambda q: q.stat().st_mtime)
        if cands and world == 1 and not names:
            try:
                tj = json.loads(cands[-1].read_text())
                traffic = int(tj["hbm_bytes_per_step"] / max(1, len(mats)))
                traffic_src = f"profiles/{cands[-1].name}"
            except Exception:
                traffic = None
        out = {
            "metric": "SpMV GFLOP/s, SuiteSparse set (20 matrices), fp32 y=alpha*A*x+beta*y, flops=2*(nnz+rows)",
            "value": round(value, 2), "unit": "GFLOP/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(t_wall / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE.json configs[1]: full get_tb_matrices.py SuiteSparse set, one SpMV per matrix per step"
                                   + (f", scaled {world}x in rows and nnz-split over {world} GPUs" if world > 1 else ""),
                       "matrices": len(mats), "nnz_per_step_per_gpu": int(sum(m["nnz"] for m in mats)),
                       "sources": sorted(set(m["source"].split(":")[0] for m in mats)),
                       "alpha": ALPHA, "beta": BETA, "launch": args.launch, "streams": n_streams, "parallelism": f"nnz-split x{world}" if world > 1 else "single GPU"},
            "passes_over_set": args.warmup + args.steps + 2,   # + the two untimed passes that size the stream assignment
            "hbm_gbs_algorithmic": round(total_bytes / t_wall / 1e9, 1),
            "hbm_pct_of_peak": round(100 * total_bytes / t_wall / 1e9 / (HBM_PEAK_GBS * world), 2),
            "geomean_gflops_per_matrix": None if geo is None else round(geo, 2),
            "roofline": {"bound": "hbm", "kernel": ("spmv_slices_multi_kernel (matrices of one workgroup size share a grid; + one fix-up launch per round)"
                                                    if args.launch == "batch" else "spmv_slices_kernel (+ carry fix-up launches)"),
                         "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch_avg": int(bytes_step / len(mats)),
                         "avg_launch_us": round(t_dev / launches * 1e6, 3),
                         "note": "achieved = sum over the set of (8*nnz+16*rows+4) B / HIP-event time of the timed region on the launch stream"},
            "host": {"gen_s": round(t_gen, 1), "prep_upload_s": round(t_prep, 1)},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(mats)
        else:
            out["cpu_baseline"] = None
        if args.details:
            Path(args.details).parent.mkdir(parents=True, exist_ok=True)
            Path(args.details).write_text(json.dumps({"summary": out, "per_matrix": table}, indent=1) + "\n")
        print(json.dumps(out), flush=True)
    fpga.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
