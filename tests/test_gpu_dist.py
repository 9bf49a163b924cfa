"""Multi-GPU path with the real device pieces: world_size 2 and 3 ranks (all on GPU 0, `gloo` rendezvous -- a one-GPU
box has no second device for RCCL; the RCCL calls themselves run in a one-rank communicator) run their shard through hispmv_spmv_device and the boundary exchange through
hispmv_boundary_pack / hispmv_boundary_apply around the all_gather (hispmv_amd/dist.py).  Same matrices and the same
acceptance as the CPU test (tests/test_dist_gloo.py): every row has exactly one owner, y within 1e-5 of the fp64 result."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle
from test_dist_gloo import _free_port, make_matrices
from util import bwd_err

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, alpha, beta, out_q, backend="gloo"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    if backend == "nccl":       # as bench.py initialises it
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    import pyhispmv
    from hispmv_amd.dist import BoundaryExchange, shard_csr
    fpga = pyhispmv.FpgaHandle("none", 0, 24, 1, 1, 2, 5, True, False, True)
    mats = make_matrices()
    local = []
    for m in mats:
        sh = shard_csr(m["rp"], m["ci"], m["va"], world, rank)
        ent = dict(shard=sh, idx=-1, y=torch.zeros(sh.n_rows, dtype=torch.float32, device=dev))
        if sh.n_rows:
            ent["idx"] = fpga.create_sparse_handle_from_csr(sh.row_ptr, sh.col_idx, sh.values, sh.n_rows, m["cols"])
            ent["x"] = torch.from_numpy(m["x"]).to(dev)
            ent["b"] = torch.from_numpy(sh.local_bias(m["b"])).to(dev)
        local.append(ent)
    fpga.load_matrices()
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    ex = BoundaryExchange(len(mats), dev, fpga=fpga)
    for _ in range(2):                        # the second run must not double count
        for ent in local:
            if ent["idx"] >= 0:
                fpga.spmv_device(ent["idx"], ent["x"].data_ptr(), ent["b"].data_ptr(), ent["y"].data_ptr(), alpha, beta,
                                 stream.cuda_stream)
        ex.run(local, alpha)
    torch.cuda.synchronize()
    fpga.synchronize()
    res = []
    for ent in local:
        sh = ent["shard"]
        n_own = sh.n_rows - (1 if sh.tail_open else 0)
        res.append((sh.row_begin, n_own, ent["y"][:n_own].cpu().numpy().copy(), sh.head_open, sh.tail_open))
    out_q.put((rank, res))
    dist.barrier()
    fpga.close()
    dist.destroy_process_group()


def test_boundary_exchange_through_rccl_single_rank():
    """The `nccl` (= RCCL) code path of bench.py and dist.py on the one GPU of the test box: a communicator of one rank
    (RCCL refuses two ranks on one device), initialised with device_id like bench.py, barrier, and the step's
    all_gather_into_tensor on device tensors between the two boundary kernels.  Proves that the calls, dtypes and
    devices are what RCCL accepts -- not that anything scales."""
    if not dist.is_nccl_available():
        pytest.skip("torch built without nccl/RCCL")
    _run_world(1, "nccl")


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_spmv_on_device_with_boundary_kernels(world):
    _run_world(world, "gloo")


def _run_world(world, backend):
    alpha, beta = 0.85, -2.06
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, alpha, beta, q, backend)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
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


def test_null_stream_orders_spmv_and_boundary_kernels():
    """VERDICT r3 weak #9 as a regression test (gpurun_out/r3e: backward error 0.108 on PFlow_742).  include/hispmv.h gives a NULL
    stream ONE meaning -- the context's own stream -- for hispmv_spmv_device* AND hispmv_boundary_pack / _apply: a caller that
    passes NULL everywhere and names no stream gets its launches in order.  Until round 3 the boundary kernels took NULL as HIP's
    null stream and did not wait for SpMVs queued on the context's (non-blocking) stream."""
    import pyhispmv
    from hispmv_amd import matrices as M
    dev = torch.device("cuda", 0)
    fpga = pyhispmv.FpgaHandle("none", 0, 24, 1, 1, 2, 5, True, False, True)
    try:
        fpga.set_arena_bytes(8 << 30)
        rows = 600000                                   # long enough that a boundary kernel on another queue would overtake it
        rp, ci, va = M.zipf_csr(rows, rows, 12000000, 1.2, 3)
        idx = fpga.create_sparse_handle_from_csr(rp, ci, va, rows, rows)
        fpga.load_matrices()
        x = np.linspace(0.5, 1.5, rows).astype(np.float32)
        b = np.linspace(-1.0, 1.0, rows).astype(np.float32)
        dx, db = torch.from_numpy(x).to(dev), torch.from_numpy(b).to(dev)
        dy = torch.zeros(rows, dtype=torch.float32, device=dev)
        send = torch.full((1,), float("nan"), dtype=torch.float32, device=dev)
        recv = torch.tensor([0.25, 100.0], dtype=torch.float32, device=dev)       # world 2: weights pick rank 0's tail only
        w = torch.tensor([1.0, 0.0], dtype=torch.float32, device=dev)
        mask = torch.ones(1, dtype=torch.float32, device=dev)
        last = torch.tensor([dy.data_ptr() + 4 * (rows - 1)], dtype=torch.int64, device=dev)
        first = torch.tensor([dy.data_ptr()], dtype=torch.int64, device=dev)
        torch.cuda.synchronize()
        for _ in range(3):                               # every step rewrites y: an apply that ran early would be overwritten
            fpga.spmv_device(idx, dx.data_ptr(), db.data_ptr(), dy.data_ptr(), 0.85, -2.06, 0)
            fpga.boundary_pack(last.data_ptr(), mask.data_ptr(), send.data_ptr(), 1, 0)
            fpga.boundary_apply(first.data_ptr(), recv.data_ptr(), w.data_ptr(), 1, 2, 0)
        fpga.synchronize()                               # waits for the context's stream: everything above
        y = dy.cpu().numpy()
        y64, mag = oracle.spmv_f64(rp, ci, va, x, b, 0.85, -2.06)
        y64[0] += 0.25
        assert float(np.max(np.abs(y - y64) / np.maximum(mag, 1e-30))) < 1e-5
        assert abs(float(send.cpu()[0]) - float(y[rows - 1])) == 0.0         # the tail that was packed is the SpMV's last row
    finally:
        fpga.close()
