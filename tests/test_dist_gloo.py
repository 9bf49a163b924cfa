"""Multi-GPU path on CPU: world_size 2 and 3 `gloo` process groups exercise the nnz-balanced sharding and
the boundary exchange of hispmv_amd/dist.py.  The local SpMV of each rank is played by the oracle
(checker), exactly where a GPU rank would call hispmv_spmv_device; ownership, tails and the chain of a
row that passes through a whole rank are what is under test."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle
from hispmv_amd.dist import BoundaryExchange, chain_weights, shard_csr, split_points
from util import bwd_err


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def make_matrices():
    rng = np.random.default_rng(21)
    mats = []
    for k, (rows, cols, nnz, heavy) in enumerate([(300, 200, 6000, 0), (50, 400, 9000, 7000), (4000, 50, 9000, 0), (64, 64, 0, 0)]):
        r = rng.integers(0, rows, nnz)
        if heavy:
            r[:heavy] = 20                   # one row holding most of the matrix: spans several ranks
        if k == 2:
            r = r - (r % 3 == 1)             # many empty rows
        c = rng.integers(0, cols, nnz)
        v = rng.random(nnz).astype(np.float32) - 0.5
        order = np.lexsort((c, r))
        rp = np.zeros(rows + 1, np.int64)
        np.add.at(rp, r + 1, 1)
        mats.append(dict(rows=rows, cols=cols, rp=np.cumsum(rp).astype(np.int32), ci=c[order].astype(np.int32), va=v[order],
                         x=rng.random(cols).astype(np.float32), b=rng.random(rows).astype(np.float32)))
    return mats


def _worker(rank, world, port, alpha, beta, out_q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mats = make_matrices()
    local = []
    for m in mats:
        sh = shard_csr(m["rp"], m["ci"], m["va"], world, rank)
        y0 = sh.local_bias(m["b"])
        y = oracle.cpu_spmv(sh.row_ptr, sh.col_idx, sh.values, m["x"], y0, alpha, beta, 1) if sh.n_rows else np.zeros(0, np.float32)
        local.append(dict(y=torch.from_numpy(y.copy()), shard=sh))
    ex = BoundaryExchange(len(mats), torch.device("cpu"))
    for _ in range(2):                        # the second run must not double count (tails are re-read, heads re-added)
        for ent, m in zip(local, mats):
            sh = ent["shard"]
            if sh.n_rows:
                ent["y"].copy_(torch.from_numpy(oracle.cpu_spmv(sh.row_ptr, sh.col_idx, sh.values, m["x"], sh.local_bias(m["b"]), alpha, beta, 1)))
        ex.run(local, alpha)
    res = []
    for ent in local:
        sh = ent["shard"]
        n_own = sh.n_rows - (1 if sh.tail_open else 0)
        res.append((sh.row_begin, n_own, ent["y"][:n_own].numpy().copy(), sh.head_open, sh.tail_open))
    out_q.put((rank, res))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_nnz_sharded_spmv_with_boundary_exchange(world):
    alpha, beta = 0.85, -2.06
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, alpha, beta, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    mats = make_matrices()
    for i, m in enumerate(mats):
        y = np.full(m["rows"], np.nan, np.float32)
        cover = np.zeros(m["rows"], int)
        for r in range(world):
            row_begin, n_own, yl, _, _ = results[r][i]
            y[row_begin:row_begin + n_own] = yl
            cover[row_begin:row_begin + n_own] += 1
        assert (cover == 1).all(), "every row has exactly one owner"
        y64, mag = oracle.spmv_f64(m["rp"], m["ci"], m["va"], m["x"], m["b"], alpha, beta)
        assert bwd_err(y, y64, mag) < 1e-5
    if world == 3:   # the heavy row of matrix 1 passes through the middle rank
        assert results[1][1][3] and results[1][1][4]


def test_split_points_and_chain_weights():
    assert split_points(10, 3).tolist() == [0, 4, 7, 10]
    assert split_points(0, 2).tolist() == [0, 0, 0]
    flags = np.array([[0, 1, 0], [1, 1, 1], [1, 1, 1], [1, 0, 0]], np.float32)     # one row over 4 ranks
    assert chain_weights(flags, 3).tolist() == [1, 1, 1, 0]
    assert chain_weights(flags, 0).tolist() == [0, 0, 0, 0]
    flags = np.array([[0, 1, 0], [1, 1, 0], [1, 0, 0]], np.float32)                 # rank 1 holds more than that row
    assert chain_weights(flags, 2).tolist() == [0, 1, 0]
    assert chain_weights(flags, 1).tolist() == [1, 0, 0]


def test_shards_cover_matrix_exactly():
    m = make_matrices()[1]
    for world in (1, 2, 5, 8):
        tot = 0
        for rank in range(world):
            sh = shard_csr(m["rp"], m["ci"], m["va"], world, rank)
            tot += sh.values.size
            assert sh.row_ptr[-1] == sh.values.size == sh.col_idx.size
        assert tot == m["va"].size


def test_stacked_block_shards_cut_rows_and_reassemble():
    """bench.py's weak-scaling workload: `world` stacked blocks, every rank boundary inside a row."""
    import zlib
    from hispmv_amd import matrices as M
    from hispmv_amd.dist import shard_of_stacked_blocks
    name, rows, nnz, fam, par = [t for t in M.SUITESPARSE_SET if t[0] == "ford2"][0]
    world, seed = 3, zlib.crc32(name.encode())
    blocks = [M.make_standin(name, rows, nnz, fam, par, seed + k)[:3] for k in range(world)]
    offs = np.cumsum([0] + [int(b[0][-1]) for b in blocks])
    rp = np.concatenate([[0]] + [b[0][1:].astype(np.int64) + offs[k] for k, b in enumerate(blocks)])
    ci = np.concatenate([b[1].astype(np.int64) + k * rows for k, b in enumerate(blocks)])
    va = np.concatenate([b[2] for b in blocks])
    rng = np.random.default_rng(0)
    x = rng.random(world * rows).astype(np.float32)
    bias = rng.random(world * rows).astype(np.float32)
    y64, mag = oracle.spmv_f64(rp.astype(np.int32), ci.astype(np.int32), va, x, bias, 0.55, -2.05)
    sh = [shard_of_stacked_blocks(blocks[k], blocks[k + 1] if k + 1 < world else None, rows, rows, k, world) for k in range(world)]
    assert [s.head_open for s in sh] == [False, True, True] and [s.tail_open for s in sh] == [True, True, False]
    assert sum(s.values.size for s in sh) == va.size
    flags = np.array([[s.head_open, s.tail_open, s.n_rows == 1] for s in sh], np.float32)
    ys = [oracle.cpu_spmv(s.row_ptr, s.col_idx, s.values, x, s.local_bias(bias), 0.55, -2.05, 1) for s in sh]
    tails = np.array([y[-1] if s.tail_open else 0.0 for y, s in zip(ys, sh)])
    Y = np.full(world * rows, np.nan)
    for k, s in enumerate(sh):
        y = ys[k].copy()
        if s.head_open:
            y[0] += float((chain_weights(flags, k) * tails).sum())
        n = s.n_rows - (1 if s.tail_open else 0)
        assert np.isnan(Y[s.row_begin:s.row_begin + n]).all()
        Y[s.row_begin:s.row_begin + n] = y[:n]
    assert not np.isnan(Y).any() and bwd_err(Y, y64, mag) < 1e-5


@pytest.mark.parametrize("world", [2, 5, 8])
def test_virtual_ranks_loopback_world(world):
    """hispmv_amd.dist.LoopbackWorld: `world` virtual ranks in ONE process (what tests/test_gpu_dist_full.py runs at world 8 on
    the one-GPU box: the card takes at most 6 processes) -- the same shards, pack / apply halves and chain weights as the
    process-per-rank path, the all_gather a concatenation.  Here on CPU tensors, the local SpMV played by the oracle."""
    from hispmv_amd.dist import LoopbackWorld
    alpha, beta = 0.85, -2.06
    mats = make_matrices()
    per_rank = []
    for rank in range(world):
        local = []
        for m in mats:
            sh = shard_csr(m["rp"], m["ci"], m["va"], world, rank)
            local.append(dict(shard=sh, y=torch.zeros(sh.n_rows, dtype=torch.float32)))
        per_rank.append(local)
    lw = LoopbackWorld(per_rank, torch.device("cpu"))
    for _ in range(2):                        # the second step must not double count
        for local in per_rank:
            for ent, m in zip(local, mats):
                sh = ent["shard"]
                if sh.n_rows:
                    ent["y"].copy_(torch.from_numpy(oracle.cpu_spmv(sh.row_ptr, sh.col_idx, sh.values, m["x"], sh.local_bias(m["b"]), alpha, beta, 1)))
        lw.exchange()
    for i, m in enumerate(mats):
        y = np.full(m["rows"], np.nan, np.float32)
        cover = np.zeros(m["rows"], int)
        for local in per_rank:
            sh = local[i]["shard"]
            n_own = sh.n_rows - (1 if sh.tail_open else 0)
            y[sh.row_begin:sh.row_begin + n_own] = local[i]["y"][:n_own].numpy()
            cover[sh.row_begin:sh.row_begin + n_own] += 1
        assert (cover == 1).all(), "every row has exactly one owner"
        y64, mag = oracle.spmv_f64(m["rp"], m["ci"], m["va"], m["x"], m["b"], alpha, beta)
        assert bwd_err(y, y64, mag) < 1e-5


def test_rank_breakdown_summary_and_host_threads_per_rank():
    """bench.py's `rank_breakdown`: min / max / mean over ranks, the rank holding the maximum, the imbalance
    figure; and the per-rank share of the host CPUs (quota / LOCAL_WORLD_SIZE)."""
    import importlib.util
    import sys
    from pathlib import Path
    spec = importlib.util.spec_from_file_location("bench_mod", Path(__file__).resolve().parents[1] / "bench.py")
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    per_rank = [dict(rank=1, batch_us=300.0, exchange_us=14.0, step_us_eager=314.0, nnz=100, rows=10, formats={"slices/1024t": 3}, host_threads=2, prep_upload_s=1.0),
                dict(rank=0, batch_us=200.0, exchange_us=18.0, step_us_eager=218.0, nnz=120, rows=12, formats={"slices/1024t": 3}, host_threads=2, prep_upload_s=1.5)]
    out = bench.summarize_ranks(per_rank)
    assert out["ranks"] == 2 and [q["rank"] for q in out["per_rank"]] == [0, 1]
    assert out["batch_us"] == {"min": 200.0, "max": 300.0, "mean": 250.0, "argmax_rank": 1}
    assert out["exchange_us"]["argmax_rank"] == 0 and out["nnz"]["max"] == 120
    assert out["batch_imbalance_max_over_mean"] == 1.2
    one = bench.summarize_ranks([dict(rank=0, batch_us=290.0, exchange_us=None, step_us_eager=290.0, nnz=5, rows=1, formats={}, host_threads=None, prep_upload_s=0.1)])
    assert "exchange_us" not in one and one["batch_us"]["mean"] == 290.0
    n1, n8 = bench.host_threads_per_rank(1), bench.host_threads_per_rank(8)
    assert n1 >= 1 and n8 == max(1, n1 // 8)


def test_host_threads_are_private_to_the_library():
    """creating handles must not change the process-global OpenMP setting; HISPMV_HOST_THREADS sets the library's own
    count (bench.py --gpus N: quota / LOCAL_WORLD_SIZE per rank).  In a child process: the count is decided once per process."""
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parents[1]
    code = ("import os, ctypes; os.environ['HISPMV_HOST_THREADS'] = '3'\n"
            "import numpy as np\n"
            "from hispmv_amd._lib import lib\n"
            "from hispmv_amd.prep import prep_from_coo\n"
            "gomp = ctypes.CDLL('libgomp.so.1'); before = gomp.omp_get_max_threads()\n"
            "r = np.arange(5000, dtype=np.int32) % 100; c = np.arange(5000, dtype=np.int32) % 77\n"
            "p = prep_from_coo(r, c, np.ones(5000, np.float32), 100, 77)\n"
            "assert p.nnz == 5000\n"
            "print(lib.hispmv_host_threads(), before, gomp.omp_get_max_threads())\n")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=str(root), timeout=120)
    assert out.returncode == 0, out.stderr[-500:]
    mine, before, after = (int(v) for v in out.stdout.split()[-3:])
    assert mine == 3 and before == after
