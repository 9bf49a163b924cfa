#!/usr/bin/env python3
"""bench.py -- the reference's headline benchmark on MI355X: fp32 y = alpha*A*x + beta*y over the
20-matrix SuiteSparse set of get_tb_matrices.py:57-78 (BASELINE.json configs[1]); metric = the
reference's own GFLOP/s convention 2*(nnz+rows)/t (spmv-host.cpp:100,185) plus achieved HBM GB/s.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload set|powerlaw|dense|model] [--scaling weak|strong]

--gpus N > 1 from a plain `python bench.py`: this process starts `python -m torch.distributed.run` with N ranks as a
CHILD (before touching any GPU), relays rank 0's JSON line and exits with the children's status.  Started by
torch.distributed.run itself (RANK/WORLD_SIZE in the environment) it is one of the ranks.

A "step" is one pass over the workload, every operand already resident in HBM:
  set       (default) one SpMV per matrix of the set, ONE hispmv_spmv_device_batch call -- which the library issues as the step
            kernel (one persistent workgroup per CU drawing the slice groups and tiles of all matrices from a queue) + one tail
            launch; "batch_call" / "rank_step" of the JSON line say how the step just timed was issued.  ~1.2 GB of packed stream per
            step, far more than the 256 MiB Infinity Cache: every launch streams from HBM.  SuiteSparse files cannot be
            downloaded here; matrices/<name>/<name>.mtx is used when present, otherwise the seeded stand-in with the
            real matrix's rows and nnz (hispmv_amd/matrices.py).  The JSON line also carries the pessimistic stand-in
            family ("standin_uniform") and the strong-scaling figure of the six largest matrices ("strong_scaling").
  powerlaw  BASELINE.json configs[2]: R-MAT scale 20 + Zipf(1.2) at soc-Pokec's shape
  dense     the GeMV sizes of cpu/run_gemv.sh:9-13 (512 .. 8192 square) through the dense overlay
  model     BASELINE.json configs[3]: the three layers of apps/model_test.py (dense 8192x4096, sparse 8192x8192 d=0.1,
            sparse 1024x8192 d=0.25), device-resident vectors, one launch per layer

N > 1: --scaling weak (default): rank k holds the k-th row block of the set scaled N-fold (same per-GPU nnz), split on
the nnz prefix; --scaling strong: the six largest matrices (+ synthetic banded ones of --strong-gb GB)
nnz-split over the ranks (hispmv_amd/dist.py: shard_csr).  x is replicated; the partial sums of rows cut by a rank
boundary are exchanged with one RCCL all_gather per step -- never an all-reduce of y.

Prints ONE JSON line on rank 0 (contract in the task statement) with "roofline" and "cpu_baseline".
"""
from __future__ import annotations

import argparse
import json
import math
import os
import socket
import subprocess
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

ALPHA, BETA = 0.55, -2.05            # common/src/spmv-host.cpp:43-44 (the FPGA/GPU drivers' scalars)
HBM_PEAK_GBS = 8000.0                # MI355X spec (MI355X_MICROARCH.md, HBM)
HW = ("bench.xclbin", 24, 1, 1, 2, 5, True, False, True)   # apps tuple; only sizes the default arena
STRONG_SET = ["PFlow_742", "soc-Pokec", "mouse_gene", "TSOPF_RS_b2383", "Si41Ge41H72", "crankseg_2"]


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: the chip reaches its steady clocks after ~15 ms of load -- the set's step measures 0.315 ms in 20 steps behind
    # 3 warm-up steps, 0.301 behind 50, 0.295-0.299 in 500 steps behind 100-200 (tools/experiments/run_r2_au.sh); 700 steps
    # take 0.2 s
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--workload", choices=["set", "powerlaw", "dense", "model"], default="set")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak", help="what the main line measures when --gpus > 1")
    ap.add_argument("--strong-gb", type=str, default="2", help="comma-separated sizes (GB of stream) of synthetic row-shardable "
                    "banded matrices added to the strong-scaling set, e.g. 2,4,8 (SURVEY.md 8d C5); generated rank-locally as 64 stacked "
                    "blocks.  Default 2: with the six SuiteSparse-size matrices alone a rank of 8 has ~30 us of kernels per step, less "
                    "than the launch and exchange latency around them; empty = only the six")
    ap.add_argument("--matrices", type=str, default="", help="comma-separated subset of the set (debug)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--preheat", type=float, default=0.3, help="seconds of untimed steps before the warm-up (clock ramp; 0 = none)")
    ap.add_argument("--no-verify", action="store_true", help="skip the fp64 self-check of one step's y (outside the timed region)")
    ap.add_argument("--no-extras", action="store_true", help="skip the standin_uniform and strong_scaling sub-measurements")
    ap.add_argument("--cpu-budget", type=float, default=5.0, help="seconds of timed CPU work per implementation and thread count")
    ap.add_argument("--per-matrix-reps", type=int, default=10)
    ap.add_argument("--details", type=str, default="", help="write the per-matrix table to this JSON file")
    ap.add_argument("--streams", type=int, default=4,
                    help="HIP streams the SpMVs of a step are spread over with --launch streams")
    ap.add_argument("--launch", choices=["batch", "streams"], default="batch",
                    help="batch: one hispmv_spmv_device_batch call per step (matrices with the same workgroup size share a "
                         "grid); streams: one launch per matrix, spread over --streams HIP streams")
    ap.add_argument("--standin", choices=["structured", "uniform"], default="structured",
                    help="stand-in family of the mesh-origin matrices: structured FEM-like (default) or unstructured band")
    return ap.parse_args()


# ------------------------------------------------------------------------------------------------------------------
# launcher: N ranks as children of a process that never touches the GPU
# ------------------------------------------------------------------------------------------------------------------
def launch_ranks(args) -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve())] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=None, text=True, env=env)
    line = None
    for out in proc.stdout:                        # rank 0's JSON line is relayed; anything else goes to stderr
        if out.startswith("{") and '"metric"' in out:
            line = out.strip()
        else:
            sys.stderr.write(out)
    rc = proc.wait()
    if line is not None:
        print(line, flush=True)
    elif rc == 0:
        rc = 1
    return rc


# ------------------------------------------------------------------------------------------------------------------
# CPU baseline: child processes without torch / HIP (oracle/cpu_baseline.py)
# ------------------------------------------------------------------------------------------------------------------
def host_cpu_info():
    aff = sorted(os.sched_getaffinity(0))
    cores = set()
    for c in aff:
        try:
            base = Path(f"/sys/devices/system/cpu/cpu{c}/topology")
            cores.add((int((base / "physical_package_id").read_text()), int((base / "core_id").read_text())))
        except Exception:
            cores.add((0, c))
    quota = None
    try:
        q, per = Path("/sys/fs/cgroup/cpu.max").read_text().split()
        if q != "max":
            quota = float(q) / float(per)
    except Exception:
        pass
    return dict(logical=len(aff), physical=len(cores), cgroup_cpu_max=quota)


def host_threads_per_rank(local_world: int) -> int:
    """CPUs the job may use (physical cores in the affinity mask, capped by the cgroup quota) divided by the ranks on this host."""
    info = host_cpu_info()
    limit = info["logical"]
    if info["cgroup_cpu_max"]:
        limit = max(1, min(limit, int(info["cgroup_cpu_max"])))
    return max(1, limit // max(1, local_world))


def summarize_ranks(per_rank):
    """rank_breakdown of the JSON line: per_rank = one dict per rank (rank, batch_us, exchange_us, nnz, rows, formats, ...);
    adds min / max / mean over the ranks of every numeric field and the rank holding the maximum of the two times -- what one
    needs to tell a slow rank (imbalanced shard, another format) from a slow exchange in the first N > 1 run."""
    per_rank = sorted(per_rank, key=lambda q: q["rank"])
    out = {"ranks": len(per_rank), "per_rank": per_rank}
    for key in ("batch_us", "exchange_us", "step_us_eager", "nnz", "rows", "host_threads", "prep_upload_s"):
        vals = [(q[key], q["rank"]) for q in per_rank if isinstance(q.get(key), (int, float))]
        if not vals:
            continue
        xs = [v for v, _ in vals]
        out[key] = {"min": min(xs), "max": max(xs), "mean": round(sum(xs) / len(xs), 3), "argmax_rank": max(vals)[1]}
    if "batch_us" in out and out["batch_us"]["mean"] > 0:
        out["batch_imbalance_max_over_mean"] = round(out["batch_us"]["max"] / out["batch_us"]["mean"], 4)
    return out


def cpu_baseline(workload: str, names, budget_s: float):
    """kind "port": mkl_sparse_s_mv / cblas_sgemv called as cpu/src/main.cpp:26-49,74-96 does (when the box has
    libmkl_rt) and the OpenMP restatement of cpu_spmv, each in a fresh child process that loads neither torch nor HIP,
    pinned like cpu/env.sh:2-4, first-touched in parallel, on 24 threads (cpu/src/main.cpp:136) and on every physical
    core the cgroup lets this process use; the best figure is reported with its thread count."""
    info = host_cpu_info()
    limit = info["physical"]
    if info["cgroup_cpu_max"]:
        limit = max(1, min(limit, int(info["cgroup_cpu_max"])))
    counts = sorted({min(24, limit), limit})
    script = str(ROOT / "oracle" / "cpu_baseline.py")
    results, errors = {}, []
    for impl in ("mkl", "omp"):
        env = dict(os.environ)
        env.update(OMP_PLACES="cores", OMP_PROC_BIND="close", OMP_NUM_THREADS=str(max(counts)), MKL_NUM_THREADS=str(max(counts)),
                   MKL_DYNAMIC="FALSE", OMP_DYNAMIC="FALSE")
        env.pop("MKL_THREADING_LAYER", None)
        cmd = [sys.executable, script, "--impl", impl, "--threads", ",".join(map(str, counts)), "--workload", workload,
               "--budget", str(budget_s)] + (["--names", ",".join(names)] if names else []) + (["--one-thread"] if impl == "omp" else [])
        try:
            p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
            line = [q for q in p.stdout.splitlines() if q.startswith("{")]
            results[impl] = json.loads(line[-1]) if line else {"error": (p.stderr or "no output")[-300:]}
        except Exception as ex:       # the baseline must never take the GPU number down with it
            results[impl] = {"error": repr(ex)[:300]}
        if "error" in results[impl]:
            errors.append(f"{impl}: {results[impl]['error']}")
    best = None
    table = {}
    for impl, r in results.items():
        for run in r.get("runs", []):
            if run.get("gflops"):
                table[f"{impl}_{run['threads']}t_gflops"] = round(run["gflops"], 3)
                if best is None or run["gflops"] > best[0]:
                    best = (run["gflops"], run["threads"], impl)
    one = results.get("omp", {}).get("one_thread_gflops")
    out = {"value": round(best[0], 3) if best else None, "unit": "GFLOP/s", "cores": best[1] if best else 0, "kind": "port",
           "impl": {"mkl": "cblas_sgemv (libmkl_rt via dlopen)" if workload == "dense" else "mkl_sparse_s_mv (+ cblas_sgemv for dense layers), libmkl_rt via dlopen",
                    "omp": "OpenMP restatement of naive_gemv" if workload == "dense" else "OpenMP restatement of cpu_spmv (+ naive_gemv for dense layers)"}[best[2]] if best else None,
           "threads_swept": counts, **table,
           "cpu_spmv_1_thread_gflops": round(one, 3) if one else None,
           "host": info,
           "sample": f"the whole '{workload}' workload" + (f" ({','.join(names)})" if names else "") +
                     f", per implementation and thread count about {budget_s:.0f} s of timed repetitions (<= 200 per matrix; reference: 200 each, "
                     "cpu/run_spmv.sh:6), split over the matrices in proportion to their flops; child processes without torch/HIP, "
                     "OMP_PLACES=cores OMP_PROC_BIND=close (cpu/env.sh:2-4), parallel first touch; vectors and alpha as cpu/src/main.cpp:147,173-178 "
                     "(beta = 0 in the repeated calls: the reference's in-place loop overflows, SURVEY.md App. B.4)"}
    if errors:
        out["errors"] = errors
    return out


# ------------------------------------------------------------------------------------------------------------------
# workloads
# ------------------------------------------------------------------------------------------------------------------
def load_set(names, rank, world, uniform=False):
    """-> list of dicts with the rank's CSR shard of every matrix (host arrays): weak-scaling layout."""
    from hispmv_amd import matrices as M
    if world == 1:
        return M.benchmark_set(names or None, uniform)
    out = []
    import zlib
    for name, rows, nnz, fam, par in M.SUITESPARSE_SET:
        if names and name not in names:
            continue
        seed = zlib.crc32(name.encode())
        rp, ci, va, used = M.make_standin(name, rows, nnz, fam, par, seed + rank, uniform)
        # N-fold scaled matrix = `world` stacked blocks; this rank's shard is cut inside rows on both sides
        from hispmv_amd.dist import shard_of_stacked_blocks
        nxt = M.make_standin(name, rows, nnz, fam, par, seed + rank + 1, uniform)[:3] if rank + 1 < world else None
        sh = shard_of_stacked_blocks((rp, ci, va), nxt, rows, rows, rank, world)
        out.append(dict(name=name, source=f"synthetic:{used}", rows=sh.n_rows, cols=rows * world, nnz=int(sh.row_ptr[-1]),
                        rp=sh.row_ptr, ci=sh.col_idx, va=sh.values, shard=sh, x_touch=rows))
    return out


def load_strong(names, rank, world, gb_sizes):
    """Strong scaling: every rank generates the whole stand-in and keeps its nnz-equal element range (shard_csr); the
    synthetic GB-size banded matrices are 64 stacked row blocks of which a rank generates only its own."""
    from hispmv_amd import matrices as M
    from hispmv_amd.dist import shard_csr
    out = []
    for m in M.benchmark_set(names, False):
        if "rp" not in m:
            continue
        if world == 1:
            out.append(dict(m, x_touch=m["cols"], full_rows=m["rows"], full_nnz=m["nnz"]))
            continue
        sh = shard_csr(m["rp"], m["ci"], m["va"], world, rank)
        out.append(dict(name=m["name"], source=m["source"], rows=sh.n_rows, cols=m["cols"], nnz=int(sh.row_ptr[-1]), rp=sh.row_ptr,
                        ci=sh.col_idx, va=sh.values, shard=sh, x_touch=m["cols"] // world, full_rows=m["rows"], full_nnz=m["nnz"]))
    return out + load_strong_gb(rank, world, gb_sizes)


def load_strong_gb(rank, world, gb_sizes):
    """The synthetic row-shardable banded matrices of SURVEY.md 8d C5: `gb` GB of stream as 64 stacked row blocks; a rank
    generates (and holds) only its own blocks -- row-aligned shards, nothing is cut."""
    from hispmv_amd import matrices as M
    out = []
    for gb in gb_sizes:
        blocks, per_row, band = 64, 50, 20000
        nnz_b = int(gb * 1e9 / 8 / blocks)
        rows_b = nnz_b // per_row
        mine = range(blocks * rank // world, blocks * (rank + 1) // world)
        rps, cis, vas = [np.zeros(1, np.int64)], [], []
        for b in mine:
            rp, ci, va = M.synth_csr(rows_b, rows_b, nnz_b, "banded", band, 900 + b)
            rps.append(rp[1:].astype(np.int64) + rps[-1][-1])
            cis.append(ci.astype(np.int64) + b * rows_b)
            vas.append(va)
        rp = np.concatenate(rps)
        assert rp[-1] < 2 ** 31
        out.append(dict(name=f"banded_{gb:g}GB", source="synthetic:banded", rows=rows_b * len(mine), cols=rows_b * blocks, nnz=int(rp[-1]),
                        rp=rp.astype(np.int32), ci=np.concatenate(cis).astype(np.int32), va=np.concatenate(vas), x_touch=rows_b * len(mine),
                        full_rows=rows_b * blocks, full_nnz=nnz_b * blocks, shard=None))
    return out


def load_powerlaw():
    from hispmv_amd import matrices as M
    n, _, r, c, v = M.rmat_coo(20)
    rp, ci, va = M.coo_to_csr_sorted(r, c, v, n)
    out = [dict(name="rmat20", source="synthetic:rmat", rows=n, cols=n, nnz=int(rp[-1]), rp=rp, ci=ci, va=va)]
    rp, ci, va = M.zipf_csr(1632803, 1632803, 30622600, 1.2, 7)
    out.append(dict(name="zipf1.2_pokec_shape", source="synthetic:zipf", rows=1632803, cols=1632803, nnz=int(rp[-1]), rp=rp, ci=ci, va=va))
    return out


class Runner:
    """One FpgaHandle, device vectors and a timing helper shared by the main measurement and the sub-measurements."""

    def __init__(self, local_rank, world, dist_on=False):
        import torch
        import pyhispmv
        self.torch, self.world = torch, world
        self.dist_on = dist_on or world > 1
        self.dev = torch.device("cuda", local_rank)
        self.fpga = pyhispmv.FpgaHandle(HW[0], local_rank, *HW[1:])
        self.fpga.set_arena_bytes(200 << 30)
        # a dedicated (non-default) HIP stream: every launch, event and collective of a timed region is on it
        self.stream = torch.cuda.Stream(device=self.dev)
        torch.cuda.set_stream(self.stream)
        self.sptr = self.stream.cuda_stream
        assert self.sptr != 0
        self.passes = 0            # passes over the workload issued so far (every step of every phase: what a profiler sees)

    def add(self, mats):
        """Creates the handles (sparse from CSR / file, dense from an array), loads them, attaches device vectors."""
        torch, fpga = self.torch, self.fpga
        t0 = time.time()
        for m in mats:
            if m.get("dense") is not None:
                m["idx"] = fpga.create_dense_handle(m["dense"].reshape(-1), m["rows"], m["cols"])
            elif "path" in m:
                # flavour 1 = the cpu/ driver's reader (readMatrixCSC + convertCSCtoCSR, cpu/src/helper_functions.cpp:91-241): the
                # index-parity target of north_star and the loader whose restatement is pinned bit-exact by the reference
                # itself (oracle/_ref); the common/ flavour (0) is parity-unpinned
                m["idx"] = fpga.create_sparse_handle_from_mtx(m["path"], 1)
            else:
                m["idx"] = fpga.create_sparse_handle_from_csr(m["rp"], m["ci"], m["va"], m["rows"], m["cols"])
            assert m["idx"] >= 0, f"{m['name']}: arena full"
        fpga.load_matrices()
        t_prep = time.time() - t0
        for m in mats:
            info = fpga.matrix_info(m["idx"])
            m.update(rows=info["rows"], cols=info["cols"], nnz=info["nnz"], n_slices=info["n_slices"], device_bytes=info["device_bytes"],
                     prep_seconds=info["prep_seconds"], n_split=info["n_split_rows"], compact_slices=info["compact_slices"],
                     plan=f'{info["block_threads"]}t/{info["group_slices"]}s/{info["lds_bytes"] // 1024}KiB/{info["col_tiles"]}ct'
                          + (f'/{100 * info["compact_slices"] // max(1, info["n_slices"])}%c' if not info["is_dense"] else ""))
            g = torch.Generator(device="cpu").manual_seed(1234 + m["idx"])
            m["x"] = torch.rand(m["cols"], generator=g, dtype=torch.float32).to(self.dev)
            m["b"] = torch.rand(m["rows"], generator=g, dtype=torch.float32).to(self.dev)
            m["y"] = torch.zeros(m["rows"], dtype=torch.float32, device=self.dev)
            if m.get("shard") is not None and m["shard"].tail_open:
                m["b"][-1] = 0.0        # the cut row belongs to the next rank: this rank only contributes alpha*partial
        return t_prep

    def batch_step(self, mats, exch=None):
        fpga, sptr = self.fpga, self.sptr
        batch = fpga.prepare_batch([m["idx"] for m in mats], [m["x"].data_ptr() for m in mats], [m["b"].data_ptr() for m in mats],
                                   [m["y"].data_ptr() for m in mats])
        if exch is not None:
            exch.prepare(mats)

            def step():
                self.passes += 1
                fpga.spmv_device_batch(batch, ALPHA, BETA, sptr)
                exch.run(mats, ALPHA, prepared=True)
            # The rank's whole step -- batch launches, tail pack, the RCCL all_gather, tail apply -- as ONE graph launch: the
            # collective runs on the process group's own stream, and eagerly every step pays two stream hand-offs around it
            # (one rank, RCCL: 0.325 ms per step against 0.300 without the exchange).  Captured with torch.cuda.graph (RCCL
            # kernels are capturable; the library sees the capture and issues plain launches); any failure keeps the eager step.
            if os.environ.get("HISPMV_BENCH_STEP_GRAPH", "1") == "1" and exch.send.is_cuda and exch.dist.get_backend() == "nccl":
                torch = self.torch
                try:
                    for _ in range(3):
                        step()
                    self.fence()
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, stream=self.stream):
                        step()
                    self.fence()
                    g.replay()
                    self.fence()
                    self.step_graph = g                 # (keeps the graph alive)
                    self.step_mode = "graph"

                    def replay():
                        self.passes += 1
                        g.replay()
                    return replay
                except Exception as ex:                 # capture refused: eager step
                    sys.stderr.write(f"bench.py: step graph not used ({type(ex).__name__}: {str(ex)[:200]})\n")
                    try:
                        torch.cuda.synchronize()
                    except Exception:
                        pass
                    self.step_mode = "eager"
        else:
            def step():
                self.passes += 1
                fpga.spmv_device_batch(batch, ALPHA, BETA, sptr)
        return step

    def fence(self):
        if self.dist_on:
            import torch.distributed as dist
            dist.barrier()
        self.torch.cuda.synchronize()

    def preheat(self, step, seconds=0.3, max_steps=4000):
        """Untimed clock ramp: the chip reaches its steady clocks only after ~15 ms of load (the set's step measured 0.315
        ms in 20 steps behind 3 warm-up steps, 0.296-0.300 behind 100-200), so the measurement must not depend on the
        caller's --warmup.  Steps are issued for `seconds` of wall time (bounded by max_steps), then the device drains;
        reported as preheat_ms, outside every timed region."""
        t0 = time.perf_counter()
        if seconds <= 0:
            return 0.0
        if self.dist_on:
            # a step contains a collective: every rank must issue the SAME number of steps -- a count computed from `seconds`
            # alone (800 steps per 0.3 s: about that long for the 20-matrix step), never from a rank's own clock
            for _ in range(max(20, int(seconds / 0.3 * 800))):
                step()
            self.fence()
            return (time.perf_counter() - t0) * 1e3
        n = 0
        while time.perf_counter() - t0 < seconds and n < max_steps:
            for _ in range(20):
                step()
            n += 20
            self.torch.cuda.synchronize()
        self.fence()
        return (time.perf_counter() - t0) * 1e3

    def verify(self, mats, step):
        """One step outside the timed region, every rank's y checked against a host fp64 product of ITS shard (scipy CSR
        in fp64 -- not the oracle: bench.py touches oracle/ only in the cpu_baseline leg): backward error
        max_i |y_i - y64_i| / (|alpha| sum_j |a_ij x_j| + |beta b_i|) over the rank's rows.  The first row of a shard whose
        head is open also needs the tails of the ranks before it: they are gathered (fp64, host) over the process group
        with the chain rule of hispmv_amd.dist.chain_weights.  -> (worst error over ranks, rows checked over ranks)."""
        import scipy.sparse as sp
        torch = self.torch
        for m in mats:                               # the step must WRITE every y: results left over from earlier steps do not count
            m["y"].fill_(float("nan"))
        self.fence()
        step()
        self.fence()
        worst, n_rows, skipped = 0.0, 0, 0
        tails = []                                   # per matrix: (head_open, tail_open, single_row, tail value, tail magnitude)
        exp_heads = []
        for m in mats:
            y = m["y"].cpu().numpy().astype(np.float64)
            x = m["x"].cpu().numpy().astype(np.float64)
            b = m["b"].cpu().numpy().astype(np.float64)
            if m.get("dense") is not None:
                W = m["dense"].astype(np.float64)
                p, a = W @ x, np.abs(W) @ np.abs(x)
            elif "rp" in m:
                A = sp.csr_matrix((np.asarray(m["va"], np.float64), np.asarray(m["ci"]), np.asarray(m["rp"])), shape=(m["rows"], m["cols"]))
                p = A @ x
                A.data = np.abs(A.data)
                a = A @ np.abs(x)
            elif "path" in m:
                # a real .mtx file: its host CSR comes from the library's host-only reader, flavour 1 = the cpu/ driver's loader
                # (hispmv_prep_from_mtx; bit-exact against the reference's own helper_functions.cpp, tests/test_prep_host.py) --
                # the same reader the handle was created with, so `matrices_without_host_csr` stays 0 when real files are present
                try:
                    from hispmv_amd.prep import prep_from_mtx
                    pr = prep_from_mtx(m["path"], 1)
                    A = sp.csr_matrix((pr.values.astype(np.float64), pr.col_idx, pr.row_ptr), shape=(pr.rows, pr.cols))
                    del pr
                    p = A @ x
                    A.data = np.abs(A.data)
                    a = A @ np.abs(x)
                except Exception as ex:
                    sys.stderr.write(f"bench.py: no host CSR for {m['name']} ({type(ex).__name__}: {str(ex)[:200]})\n")
                    skipped += 1
                    tails.append((0, 0, 0, 0.0, 0.0)); exp_heads.append(None)
                    continue
            else:
                skipped += 1
                tails.append((0, 0, 0, 0.0, 0.0)); exp_heads.append(None)
                continue
            y64 = ALPHA * p + BETA * b
            mag = np.abs(ALPHA) * a + np.abs(BETA * b)
            sh = m.get("shard")
            lo, hi = 0, m["rows"]
            if sh is not None and sh.head_open:
                lo = 1
            tails.append((int(bool(sh and sh.head_open)), int(bool(sh and sh.tail_open)), int(bool(sh and sh.n_rows == 1)),
                          float(y64[-1]) if m["rows"] else 0.0, float(mag[-1]) if m["rows"] else 0.0))
            exp_heads.append((float(y64[0]), float(mag[0]), float(y[0])) if (sh is not None and sh.head_open) else None)
            if hi > lo:
                err = np.abs(y[lo:hi] - y64[lo:hi]) / np.maximum(mag[lo:hi], 1e-300)
                worst = max(worst, float(err.max()))
                n_rows += hi - lo
        if self.dist_on:
            import torch.distributed as dist
            from hispmv_amd.dist import chain_weights
            allt = [None] * dist.get_world_size()
            dist.all_gather_object(allt, tails)
            rank = dist.get_rank()
            for i, eh in enumerate(exp_heads):
                if eh is None:
                    continue
                flags = np.array([[allt[r][i][0], allt[r][i][1], allt[r][i][2]] for r in range(len(allt))], np.float32)
                w = chain_weights(flags, rank)
                y0 = eh[0] + sum(w[r] * allt[r][i][3] for r in range(len(allt)))
                m0 = eh[1] + sum(w[r] * allt[r][i][4] for r in range(len(allt)))
                worst = max(worst, abs(eh[2] - y0) / max(m0, 1e-300))
                n_rows += 1
            tt = torch.tensor([worst, float(n_rows), float(skipped)], dtype=torch.float64, device=self.dev)
            mx = tt.clone(); dist.all_reduce(mx, op=dist.ReduceOp.MAX)
            sm = tt.clone(); dist.all_reduce(sm, op=dist.ReduceOp.SUM)
            worst, n_rows, skipped = float(mx[0]), int(sm[1]), int(sm[2])
        return worst, n_rows, skipped

    def time_steps(self, step, steps, warmup):
        """EXACTLY `steps` steps between two fences (barrier + device synchronise), HIP events on the launch stream;
        -> (wall seconds, device seconds), max over ranks."""
        torch = self.torch
        for _ in range(warmup):
            step()
        self.fence()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev0.record(self.stream)
        for _ in range(steps):
            step()
        ev1.record(self.stream)
        self.fence()
        t_wall = time.perf_counter() - t0
        t_dev = ev0.elapsed_time(ev1) * 1e-3
        self.fpga.synchronize()                       # raises if a bounded in-kernel wait (carry look-back) expired
        if self.dist_on:
            import torch.distributed as dist
            tt = torch.tensor([t_wall, t_dev], dtype=torch.float64, device=self.dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            t_wall, t_dev = float(tt[0]), float(tt[1])
        return t_wall, t_dev


def latest_traffic(n_launches):
    """HBM bytes per launch from the committed PMC summary of the latest ROUND (tools/profile_round.sh: FETCH_SIZE x2 +
    WRITE_SIZE in separate passes, MI355X_MICROARCH.md "HBM"): profiles/r<N>_traffic.json with the largest N -- chosen by
    the round tag in the file name, never by modification time (arbitrary after a fresh checkout or push)."""
    import re
    def round_of(q):
        m = re.match(r"r(\d+)_traffic\.json$", q.name)
        return int(m.group(1)) if m else -1
    cands = sorted((q for q in (ROOT / "profiles").glob("r*_traffic.json") if round_of(q) >= 0), key=round_of)
    for c in reversed(cands):
        try:
            tj = json.loads(c.read_text())
            return int(tj["hbm_bytes_per_step"] / max(1, n_launches)), f"profiles/{c.name}"
        except Exception:
            continue
    return None, None


def recorded_one_gpu_strong(names):
    """The n_gpus = 1 figure of the strong-scaling set from the committed bench line of the latest round
    (profiles/r<N>_bench_line.json, chosen by the round tag), when it was measured on the same matrices."""
    import re
    def round_of(q):
        m = re.match(r"r(\d+)_bench_line\.json$", q.name)
        return int(m.group(1)) if m else -1
    for c in sorted((q for q in (ROOT / "profiles").glob("r*_bench_line.json") if round_of(q) >= 0), key=round_of, reverse=True):
        try:
            ss = json.loads(c.read_text()).get("strong_scaling") or {}
            if ss.get("n_gpus") == 1 and ss.get("matrices") == list(names) and ss.get("value"):
                return float(ss["value"]), f"profiles/{c.name}"
        except Exception:
            continue
    return None, None


# ------------------------------------------------------------------------------------------------------------------
def main():
    args = parse_args()
    under_torchrun = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if args.gpus > 1 and not under_torchrun:
        sys.exit(launch_ranks(args))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world

    # N ranks share one host: every rank's preprocessor gets its share of the CPUs this job may use (cgroup quota / affinity),
    # not all of them (8 ranks x 16 threads on a 16-CPU quota oversubscribed the host 8-fold in round 3).  Set before the
    # library is loaded: host_threads() reads HISPMV_HOST_THREADS once.
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    if local_world > 1 and "HISPMV_HOST_THREADS" not in os.environ:
        os.environ["HISPMV_HOST_THREADS"] = str(host_threads_per_rank(local_world))

    import torch
    import torch.distributed as dist
    from hispmv_amd import matrices as M

    # HISPMV_BENCH_REHEARSAL=1: every rank on GPU 0 with the gloo backend -- rehearses the N > 1 code path on a
    # one-GPU box (numbers from such a run mean nothing)
    rehearsal = os.environ.get("HISPMV_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    backend = None
    # HISPMV_BENCH_FORCE_DIST=1: initialise the process group, fences and the boundary exchange even with one rank (under
    # torch.distributed.run with --nproc-per-node 1): the RCCL calls of the N > 1 path on a one-GPU box
    dist_on = world > 1 or (under_torchrun and os.environ.get("HISPMV_BENCH_FORCE_DIST") == "1")
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        backend = dist.get_backend()
    ranks_seen = dist.get_world_size() if dist_on else 1

    names = [n for n in args.matrices.split(",") if n]
    gb_sizes = [float(q) for q in args.strong_gb.split(",") if q]
    R = Runner(local_rank, world, dist_on)
    fpga, sptr, stream = R.fpga, R.sptr, R.stream
    strong_main = world > 1 and args.scaling == "strong" and args.workload == "set"

    t0 = time.time()
    if args.workload == "set":
        mats = load_strong(names or STRONG_SET, rank, world, gb_sizes) if strong_main else load_set(names, rank, world, args.standin == "uniform")
    elif args.workload == "powerlaw":
        mats = load_powerlaw()
    elif args.workload == "dense":
        rng = np.random.default_rng(0)
        mats = [dict(name=f"gemv_{n}x{n}", source="synthetic:dense", rows=n, cols=n, dense=rng.random((n, n), dtype=np.float32) - np.float32(0.5))
                for n in (512, 1024, 2048, 4096, 8192)]                                   # cpu/run_gemv.sh:9-13
    else:
        mats = []
        for i, (kind, W, rows, cols, _bias) in enumerate(M.model_test_layers(0)):
            if kind == "dense":
                mats.append(dict(name=f"layer{i}_dense_{rows}x{cols}", source="synthetic:model_test", rows=rows, cols=cols, dense=W))
            else:
                rp, ci, va = M.coo_to_csr_sorted(W[0], W[1], W[2], rows)
                mats.append(dict(name=f"layer{i}_sparse_{rows}x{cols}", source="synthetic:model_test", rows=rows, cols=cols, rp=rp, ci=ci, va=va))
    assert world == 1 or args.workload == "set", "--gpus > 1 is defined for the SuiteSparse set"
    t_gen = time.time() - t0
    t_prep = R.add(mats)

    exch = None
    if dist_on:
        from hispmv_amd.dist import BoundaryExchange
        exch = BoundaryExchange(len(mats), R.dev, fpga=fpga)

    # ---- the main timed region --------------------------------------------------------------------------------------
    n_streams = 1
    if args.launch == "batch":
        step = R.batch_step(mats, exch)
    else:
        # one launch per matrix over --streams HIP streams: longest-processing-time-first assignment from one untimed pass
        n_streams = max(1, args.streams)
        side = [torch.cuda.Stream(device=R.dev) for _ in range(n_streams - 1)]
        lanes = [stream] + side
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
        if exch is not None:
            exch.prepare(mats)

        def step():
            if n_streams > 1:
                fork = torch.cuda.Event()
                fork.record(stream)
                for s2 in side:
                    s2.wait_event(fork)
            for i in order:
                m = mats[i]
                fpga.spmv_device(m["idx"], m["x"].data_ptr(), m["b"].data_ptr(), m["y"].data_ptr(), ALPHA, BETA, lanes[m["lane"]].cuda_stream)
            if n_streams > 1:
                for s2 in side:
                    join = torch.cuda.Event()
                    join.record(s2)
                    stream.wait_event(join)
            if exch is not None:
                exch.run(mats, ALPHA, prepared=True)

    preheat_ms = R.preheat(step, args.preheat)
    t_wall, t_dev = R.time_steps(step, args.steps, args.warmup)
    main_call_info = None            # how the library issued the step just timed (before the sub-measurements issue theirs)
    if args.launch == "batch":
        try:
            main_call_info = fpga.batch_call_info()
        except Exception:
            main_call_info = None
    # self-check (outside the timed region): one more step, every rank's y against a host fp64 product of its shard
    y_err, y_rows, y_skipped = (R.verify(mats, step) if not args.no_verify else (None, 0, len(mats)))
    # True / False when rows were checked; None when nothing was checkable (--no-verify, or no host CSR for any matrix): skipped,
    # not failed -- the line says so and the exit status stays 0
    y_checked = None if (y_err is None or y_rows == 0) else bool(y_err < 1e-5)

    # ---- rank breakdown (outside the timed region): the EAGER twin of the step with HIP events between its parts, on every rank
    def rank_breakdown(mats_, exch_, reps=20):
        batch = fpga.prepare_batch([m["idx"] for m in mats_], [m["x"].data_ptr() for m in mats_], [m["b"].data_ptr() for m in mats_],
                                   [m["y"].data_ptr() for m in mats_])
        if exch_ is not None:
            exch_.prepare(mats_)
        evs = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(reps)]
        R.fence()
        for k in range(reps + 3):
            e = evs[max(0, k - 3)]
            e[0].record(stream)
            R.passes += 1
            fpga.spmv_device_batch(batch, ALPHA, BETA, sptr)
            e[1].record(stream)
            if exch_ is not None:
                exch_.run(mats_, ALPHA, prepared=True)
            e[2].record(stream)
        R.fence()
        b_us = float(np.median([e[0].elapsed_time(e[1]) for e in evs])) * 1e3
        x_us = float(np.median([e[1].elapsed_time(e[2]) for e in evs])) * 1e3 if exch_ is not None else None
        fmts = {}
        for m in mats_:
            info = fpga.matrix_info(m["idx"])
            key = ("dense" if info["is_dense"] else ("tile_stream" if info["format"] == 1 else "slices") + f'/{info["block_threads"]}t'
                   + (f'/{info["col_tiles"]}{"band" if info["tile_kind"] == 2 else "col"}tiles' if info["col_tiles"] > 1 else ""))
            fmts[key] = fmts.get(key, 0) + 1
        mine = dict(rank=rank, batch_us=round(b_us, 2), exchange_us=None if x_us is None else round(x_us, 2),
                    step_us_eager=round(b_us + (x_us or 0.0), 2), nnz=int(sum(m["nnz"] for m in mats_)), rows=int(sum(m["rows"] for m in mats_)),
                    cut_heads=int(sum(1 for m in mats_ if m.get("shard") is not None and m["shard"].head_open)),
                    cut_tails=int(sum(1 for m in mats_ if m.get("shard") is not None and m["shard"].tail_open)),
                    formats=fmts, host_threads=int(os.environ.get("HISPMV_HOST_THREADS", "0")) or None, prep_upload_s=round(t_prep, 2))
        if dist_on:
            allr = [None] * dist.get_world_size()
            dist.all_gather_object(allr, mine)
        else:
            allr = [mine]
        return summarize_ranks(allr)

    breakdown = rank_breakdown(mats, exch) if args.launch == "batch" else None
    passes_main = R.passes + (args.warmup + args.steps + 2 if args.launch == "streams" else 0)      # before the sub-measurements add theirs

    def alg_bytes(m):
        if m.get("dense") is not None:
            return 4 * m["rows"] * m["cols"] + 4 * m["cols"] + 8 * m["rows"]               # B_gemv
        # (x is replicated at full length on every rank, but a rank's block touches about 1/world of it: count that part)
        return M.algorithmic_bytes(m["rows"], m.get("x_touch", m["cols"]), m["nnz"])

    def flops_of(m):
        return 2 * m["rows"] * m["cols"] + m["rows"] if m.get("dense") is not None else M.flops(m["rows"], m["nnz"])     # cpu/src/main.cpp:233 / :187

    # ---- per-matrix table (outside the timed region): events around one launch of one matrix; the largest matrix
    # is streamed in between so that each measurement starts from a cold Infinity Cache
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
            # ... and the same launch repeated back to back (what the reference's rp_time loop times, spmv-host.cpp:120-154:
            # operands of a small matrix stay in the caches)
            reps = 20
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            fpga.spmv_device(m["idx"], m["x"].data_ptr(), m["b"].data_ptr(), m["y"].data_ptr(), ALPHA, BETA, sptr)
            a.record(stream)
            for _ in range(reps):
                fpga.spmv_device(m["idx"], m["x"].data_ptr(), m["b"].data_ptr(), m["y"].data_ptr(), ALPHA, BETA, sptr)
            b.record(stream)
            torch.cuda.synchronize()
            t_warm = a.elapsed_time(b) * 1e-3 / reps
            ab = alg_bytes(m)
            table.append(dict(name=m["name"], source=m["source"], rows=m["rows"], nnz=m["nnz"], us=round(t * 1e6, 2), us_back_to_back=round(t_warm * 1e6, 2),
                              gflops=round(flops_of(m) / t / 1e9, 2), alg_gbs=round(ab / t / 1e9, 1),
                              pct_hbm_peak=round(100 * ab / t / 1e9 / HBM_PEAK_GBS, 2), slices=m["n_slices"],
                              split_rows=m["n_split"], prep_s=round(m["prep_seconds"], 3), plan=m["plan"]))

    flops_step = sum(flops_of(m) for m in mats)
    bytes_step = sum(alg_bytes(m) for m in mats)

    # ---- sub-measurements (same JSON line, so that neither stand-in family nor scaling mode is cherry-picked) ----------
    extras = {}
    if args.workload == "set" and not args.no_extras and not names and args.launch == "batch":
        if world == 1 and args.standin == "structured":
            # the pessimistic family: the 8 mesh-origin matrices as unstructured bands (no column reuse between rows)
            uni = [m for m in M.benchmark_set(None, True) if m.get("family") == "fem"]
            R.add(uni)
            swapped = [next((u for u in uni if u["name"] == m["name"]), m) for m in mats]
            sstep = R.batch_step(swapped)
            R.preheat(sstep, min(args.preheat, 0.15))         # (its own call signature: plan, graph capture and clocks settle outside the timed region)
            tw, td = R.time_steps(sstep, args.steps, args.warmup)
            fl, by = sum(flops_of(m) for m in swapped), sum(alg_bytes(m) for m in swapped)
            extras["standin_uniform"] = {"value": round(fl * args.steps / tw / 1e9, 2), "unit": "GFLOP/s", "ms_per_step": round(tw / args.steps * 1e3, 4),
                                         "roofline_frac": round(by * args.steps / td / 1e9 / HBM_PEAK_GBS, 4),
                                         "note": "same step with the 8 mesh-origin matrices generated as unstructured bands (--standin uniform)"}
        if not strong_main:
            # strong scaling of the six largest matrices: the SAME matrices at every N, nnz-split over the ranks
            smats = [m for m in mats if m["name"] in STRONG_SET] if world == 1 else load_strong(STRONG_SET, rank, world, gb_sizes)
            if world == 1 and gb_sizes:
                extra_gb = load_strong_gb(rank, world, gb_sizes)
                R.add(extra_gb)
                smats = smats + extra_gb
            if world > 1:
                R.add(smats)
            sexch = None
            if world > 1:
                from hispmv_amd.dist import BoundaryExchange
                sexch = BoundaryExchange(len(smats), R.dev, fpga=fpga)
            sstep = R.batch_step(smats, sexch)
            R.preheat(sstep, min(args.preheat, 0.15))
            tw, td = R.time_steps(sstep, args.steps, args.warmup)
            s_err, s_rows, _ = (R.verify(smats, sstep) if not args.no_verify else (None, 0, 0))
            fl = sum(M.flops(m.get("full_rows", m["rows"]), m.get("full_nnz", m["nnz"])) for m in smats)
            by = sum(M.algorithmic_bytes(m.get("full_rows", m["rows"]), m["cols"], m.get("full_nnz", m["nnz"])) for m in smats)
            ss_val = fl * args.steps / tw / 1e9
            rec1, rec1_src = recorded_one_gpu_strong([m["name"] for m in smats])
            extras["strong_scaling"] = {"matrices": [m["name"] for m in smats], "n_gpus": world, "value": round(ss_val, 2),
                                        "rank_breakdown": rank_breakdown(smats, sexch) if world > 1 else None,
                                        "efficiency": {"value": None if not rec1 else round(ss_val / (world * rec1), 4), "one_gpu_value": rec1, "one_gpu_source": rec1_src,
                                                       "definition": "value / (n_gpus x the recorded n_gpus = 1 value of the same matrices)"},
                                        "unit": "GFLOP/s", "ms_per_step": round(tw / args.steps * 1e3, 4),
                                        "hbm_gbs_algorithmic": round(by * args.steps / tw / 1e9, 1), "scaling": "strong",
                                        "y_checked": bool(s_err is not None and s_err < 1e-5 and s_rows > 0), "y_max_backward_error": s_err,
                                        "note": "whole-job rate of the same matrices nnz-split over n_gpus ranks (shard_csr); speed-up = value / the n_gpus = 1 value"}

    if args.workload == "model" and rank == 0 and not args.no_extras:
        # FpgaHandle.linear with 8 input vectors per call (apps/model_test.py --batch_size; fpga_handle.cpp:323-388): host
        # buffers in and out, so the call is PCIe-inclusive; kernel_us = events around the launches alone (8 / 4 / 2
        # vectors per pass over the matrix: gemv_rows_kernel<4,.,8>, spmv_slices_batched_kernel<.,.,4>)
        rng = np.random.default_rng(99)
        lin = []
        for m in mats:
            xs = rng.random(8 * m["cols"], dtype=np.float32)
            bias = rng.random(m["rows"], dtype=np.float32)
            fpga.linear(m["idx"], xs, bias)
            ks, ws = [], []
            for _ in range(7):
                t1 = time.perf_counter()
                fpga.linear(m["idx"], xs, bias)
                ws.append(time.perf_counter() - t1)
                ks.append(fpga.last_kernel_ms() * 1e-3)
            k, w8 = float(np.median(ks)), float(np.median(ws))
            lin.append({"name": m["name"], "vectors": 8, "kernel_us": round(k * 1e6, 2), "call_us_with_pcie": round(w8 * 1e6, 1),
                        "gflops_kernel": round(8 * flops_of(m) / k / 1e9, 1),
                        "matrix_passes_per_s_x_bytes_GBs": round(alg_bytes(m) * 8 / k / 1e9, 1)})
        # one vector per call from host buffers (FpgaHandle.run_kernel through linear's alpha = beta = 1): what apps/model_test.py pays per
        # layer with --batch_size 1 -- the x / bias copy in, the launches, y written straight into pinned memory (no copy back)
        one = []
        for m in mats:
            xs = rng.random(m["cols"], dtype=np.float32)
            bias = rng.random(m["rows"], dtype=np.float32)
            for _ in range(3):
                fpga.linear(m["idx"], xs, bias)
            ks, ws = [], []
            for _ in range(15):
                t1 = time.perf_counter()
                fpga.linear(m["idx"], xs, bias)
                ws.append(time.perf_counter() - t1)
                ks.append(fpga.last_kernel_ms() * 1e-3)
            one.append({"name": m["name"], "kernel_us": round(float(np.median(ks)) * 1e6, 2), "call_us_with_pcie": round(float(np.median(ws)) * 1e6, 1)})
        extras["host_vector_call"] = {"layers": one, "note": "one vector per FpgaHandle.linear call from host buffers, median of 15; y lands in pinned "
                                      "host memory without a copy back (HISPMV_HOST_Y=copy: through a device buffer, as until round 4)"}
        extras["linear_batch8"] = {"layers": lin, "note": "8 vectors per FpgaHandle.linear call; kernel_us is the device time of the launches, "
                                   "call_us_with_pcie the whole call from host buffers; the last column counts the matrix bytes once per "
                                   "vector, i.e. the rate a one-vector-per-pass kernel would need"}

    if rank == 0:
        if strong_main:
            total_flops = sum(M.flops(m["full_rows"], m["full_nnz"]) for m in mats) * args.steps
            total_bytes = sum(M.algorithmic_bytes(m["full_rows"], m["cols"], m["full_nnz"]) for m in mats) * args.steps
        else:
            total_flops = flops_step * world * args.steps
            total_bytes = bytes_step * world * args.steps
        value = total_flops / t_wall / 1e9
        achieved = bytes_step * args.steps / t_dev / 1e9          # per GPU, device time
        launches = len(mats) * args.steps
        geo = math.exp(sum(math.log(r["gflops"]) for r in table) / len(table)) if table else None
        traffic, traffic_src = latest_traffic(len(mats)) if (world == 1 and not names and args.workload == "set") else (None, None)
        workloads = {"set": "BASELINE.json configs[1]: full get_tb_matrices.py SuiteSparse set, one SpMV per matrix per step",
                     "powerlaw": "BASELINE.json configs[2]: R-MAT scale 20 (ef 16, duplicates kept) + Zipf(1.2) row lengths at soc-Pokec's shape",
                     "dense": "dense overlay GeMV, sizes of cpu/run_gemv.sh:9-13 (512..8192 square)",
                     "model": "BASELINE.json configs[3]: apps/model_test.py layers 4096->8192 dense, 8192->8192 d=0.1, 8192->1024 d=0.25, one launch per layer"}
        classes = {}
        for m in mats:
            info = fpga.matrix_info(m["idx"])
            key = ("gemv_rows_multi_kernel" if info["is_dense"] else
                   f"spmv_tts_multi_kernel/{info['block_threads']}t" if info["format"] == 1 else f"spmv_slices_multi_kernel/{info['block_threads']}t")
            cl = classes.setdefault(key, {"matrices": [], "algorithmic_bytes_per_launch": 0, "flops_per_launch": 0})
            cl["matrices"].append(m["name"]); cl["algorithmic_bytes_per_launch"] += int(alg_bytes(m)); cl["flops_per_launch"] += int(flops_of(m))
        # the kernels of a step, named from what the handles actually are (launch_classes below): one grid per class of a batch call --
        # or, when the call shares the chip between its matrices, ONE spmv_step_kernel launch whose queue holds the groups and tiles of
        # all classes (the library says which: hispmv_batch_call_info)
        call_info = main_call_info
        if call_info and call_info["step_kernel"]:
            dominant = (f"spmv_step_kernel (ONE launch per step: {call_info['items']} items -- slice groups and tiles of all {len(mats)} matrices -- drawn from a queue "
                        "by one persistent 1024-thread workgroup per CU; the classes it replaces: " + ", ".join(f"{k} ({len(v['matrices'])})" for k, v in classes.items())
                        + "); + one spmv_tail_multi_kernel launch per step (cut rows, merge of column-tile partial vectors)")
        elif args.launch == "batch":
            dominant = " + ".join(f"{k} ({len(v['matrices'])} matri{'x' if len(v['matrices']) == 1 else 'ces'}: {', '.join(v['matrices'][:8])}{', ...' if len(v['matrices']) > 8 else ''})"
                                  for k, v in classes.items())
            if any(k.startswith("spmv_") for k in classes):
                dominant += "; + one spmv_tail_multi_kernel launch per step when rows are cut by slice boundaries or a matrix has column parts"
        else:
            dominant = "spmv_slices_kernel / spmv_tts_kernel / gemv_rows_kernel, one launch sequence per matrix (+ carry fix-up launches)"
        out = {
            "metric": {"set": "SpMV GFLOP/s, SuiteSparse set (20 matrices), fp32 y=alpha*A*x+beta*y, flops=2*(nnz+rows)",
                       "powerlaw": "SpMV GFLOP/s, power-law matrices, fp32 y=alpha*A*x+beta*y, flops=2*(nnz+rows)",
                       "dense": "GeMV GFLOP/s, dense overlay, fp32 y=alpha*W*x+beta*y, flops=2*rows*cols+rows",
                       "model": "GFLOP/s over the three model_test layers (dense + sparse), fp32"}[args.workload],
            "value": round(value, 2), "unit": "GFLOP/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(t_wall / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "strong" if strong_main else "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": workloads[args.workload]
                                   + (f", the six largest matrices nnz-split over {world} GPUs" if strong_main else
                                      f", scaled {world}x in rows and nnz-split over {world} GPUs" if world > 1 else ""),
                       "matrices": len(mats), "nnz_per_step_per_gpu": int(sum(m["nnz"] for m in mats)),
                       "sources": sorted(set(m["source"].split(":")[0] for m in mats)), "standin": args.standin,
                       "alpha": ALPHA, "beta": BETA, "launch": args.launch, "streams": n_streams,
                       "parallelism": f"nnz-split x{world}" if world > 1 else "single GPU"},
            "ranks_seen": ranks_seen, "backend": backend, "rank_step": getattr(R, "step_mode", "eager" if dist_on else ("library graph" if os.environ.get("HISPMV_BATCH_GRAPH", "0") not in ("", "0") else
                                                     "library launches (step kernel + tail, one stream)" if (call_info and call_info["step_kernel"]) else "library launches (two streams, plain)")),
            "batch_call": call_info,
            # every pass over the workload this process issued for the MAIN measurement -- preheat, warm-up, timed steps, the
            # self-check's step, the rank breakdown's eager steps: what a profiler divides its per-kernel totals by
            "passes_over_set": passes_main,
            "hbm_gbs_algorithmic": round(total_bytes / t_wall / 1e9, 1),
            "hbm_pct_of_peak": round(100 * total_bytes / t_wall / 1e9 / (HBM_PEAK_GBS * world), 2),
            "geomean_gflops_per_matrix": None if geo is None else round(geo, 2),
            "roofline": {"bound": "hbm", "kernel": dominant,
                         "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch_avg": int(bytes_step / len(mats)),
                         "avg_launch_us": round(t_dev / launches * 1e6, 3),
                         "note": "achieved = sum over the workload of the algorithmic bytes (SpMV: 8*nnz+16*rows+4, GeMV: 4*rows*cols+4*cols+8*rows) "
                                 "/ HIP-event time of the timed region on the launch stream"},
            "y_checked": y_checked,
            "y_check": {"max_backward_error": y_err, "rows_checked": y_rows, "matrices_without_host_csr": y_skipped, "gate": 1e-5,
                        "how": "one step after the timed region; every rank: host fp64 product (scipy CSR) of its shard, cut rows with the "
                               "gathered fp64 tails of the ranks before it"},
            "preheat_ms": round(preheat_ms, 1),
            "rank_breakdown": breakdown,
            "launch_classes": classes,
            "host": {"gen_s": round(t_gen, 1), "prep_upload_s": round(t_prep, 1)},
        }
        out.update(extras)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.workload, names, args.cpu_budget)
        else:
            out["cpu_baseline"] = None
        if args.details:
            Path(args.details).parent.mkdir(parents=True, exist_ok=True)
            Path(args.details).write_text(json.dumps({"summary": out, "per_matrix": table}, indent=1) + "\n")
        print(json.dumps(out), flush=True)
    fpga.close()
    if dist_on:
        dist.destroy_process_group()
    if not args.no_verify and y_checked is False:
        sys.exit(f"bench.py: the y self-check failed (max backward error {y_err}, rows checked {y_rows})")


if __name__ == "__main__":
    main()
