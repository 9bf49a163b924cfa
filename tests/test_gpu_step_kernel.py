"""The step kernel (hispmv_kernels.hip: spmv_step_kernel) against the grids it replaces.

A batch call that shares the chip between its matrices runs every slice group and every tile as an item of one queue drawn by
persistent workgroups; an item executes the very body of the multi-matrix kernels, so the result of every matrix must be
BIT-IDENTICAL to the call issued as separate grids (HISPMV_STEP_KERNEL=0) -- whatever the queue order, for 1024-thread groups,
four 256-thread groups hosted side by side in one workgroup, tiles, column-tiled matrices (partial vectors + tail), parts with
stray slots (the other instantiation), beta == 0, and back-to-back calls without host synchronisation (the queue rearms itself).
The arithmetic itself is pinned elsewhere (tests/test_gpu_parity.py: bit-equal to oracle.emu_spmv / emu_tts; test_gpu_bench_set.py:
the 1e-5 gates against fp64 and MKL with the step kernel as the default path).  Reference counterpart: none -- the reference runs
one matrix at a time (pyhispmv/src/fpga_handle.cpp:286-321)."""
import os

import numpy as np
import pytest

from conftest import ALPHA, BETA, ref_vectors

pytestmark = pytest.mark.gpu

HW = ("tests.xclbin", 0, 24, 1, 1, 2, 5, True, False, True)


def _run(torch, mats, env, reps=1, beta=BETA):
    """One context created under `env`, the matrices loaded, `reps` batch calls back to back; -> list of y (numpy), plan info."""
    import pyhispmv
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        h = pyhispmv.FpgaHandle(*HW)          # (the switches are read when the context is created)
    finally:
        for k, v in old.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v
    dev = torch.device("cuda", 0)
    h.set_arena_bytes(64 << 30)
    try:
        idx = []
        for m in mats:
            idx.append(h.create_sparse_handle_from_csr(m["rp"], m["ci"], m["va"], m["rows"], m["cols"]))
            assert idx[-1] >= 0
        h.load_matrices()
        dx, db, dy = [], [], []
        for m in mats:
            x, b = ref_vectors(m["rows"], m["cols"])
            dx.append(torch.from_numpy(x).to(dev))
            db.append(torch.from_numpy(b).to(dev))
            dy.append(torch.full((m["rows"],), float("nan"), dtype=torch.float32, device=dev))
        batch = h.prepare_batch(idx, [t.data_ptr() for t in dx], [t.data_ptr() for t in db] if beta != 0.0 else None, [t.data_ptr() for t in dy])
        s = torch.cuda.Stream(device=dev)
        torch.cuda.synchronize()
        with torch.cuda.stream(s):
            for _ in range(reps):                       # no host synchronisation in between: the queue must rearm itself
                h.spmv_device_batch(batch, ALPHA, beta, s.cuda_stream)
        h.synchronize()
        torch.cuda.synchronize()
        info = [h.matrix_info(i) for i in idx]
        # the library says how it issued the call (hispmv_batch_call_info): the step kernel exactly when the context asked for it
        call = h.batch_call_info()
        want_step = env.get("HISPMV_STEP_KERNEL", "1") != "0"
        assert call["step_kernel"] == want_step, (env, call)
        if want_step:
            assert call["launches"] <= 2 and call["streams"] == 1 and call["items"] > 0, call        # ONE main launch (+ the tail) on the caller's stream
        else:
            assert call["items"] == 0 and call["launches"] >= 3, call                                  # a grid per class (+ the tail)
        return [t.cpu().numpy() for t in dy], info
    finally:
        h.close()


def _same_bits(a, b):
    return a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32))


@pytest.fixture(scope="module")
def torch_mod():
    import torch
    assert torch.cuda.is_available()
    return torch


@pytest.mark.parametrize("family", ["structured", "uniform"])
def test_step_kernel_gives_the_bits_of_the_grids(torch_mod, family):
    from hispmv_amd import matrices as M
    mats = [m for m in M.benchmark_set(None, family == "uniform") if "rp" in m]
    assert sum(m["nnz"] for m in mats) * 8 >= (256 << 20)         # the call shares the chip: the step kernel is taken
    grids, info = _run(torch_mod, mats, {"HISPMV_STEP_KERNEL": "0"})
    formats = {(i["format"], i["block_threads"], i["col_tiles"] > 1) for i in info}
    # the call really holds every kind of item: 1024-thread groups, 256-thread groups, tiles, a column-tiled matrix
    assert {(0, 1024), (0, 256)} <= {(f, t) for f, t, _ in formats} and any(f == 1 for f, _, _ in formats)
    if family == "structured":
        assert any(ct for _, _, ct in formats)
    for order in ("", "lpt", "grid"):
        step, _ = _run(torch_mod, mats, {"HISPMV_STEP_KERNEL": "1", "HISPMV_STEP_ORDER": order}, reps=3)
        for m, a, b in zip(mats, grids, step):
            assert np.all(np.isfinite(b)), f'{m["name"]}: non-finite y from the step kernel (order {order or "default"})'
            assert _same_bits(a, b), f'{m["name"]}: step kernel (order {order or "default"}) differs from the grids'
    # beta == 0: no bias table
    g0, _ = _run(torch_mod, mats[:8], {"HISPMV_STEP_KERNEL": "0"}, beta=0.0)
    s0, _ = _run(torch_mod, mats[:8], {"HISPMV_STEP_KERNEL": "1"}, reps=2, beta=0.0)
    for m, a, b in zip(mats[:8], g0, s0):
        assert _same_bits(a, b), f'{m["name"]}: beta = 0'


def test_step_kernel_with_stray_slots(torch_mod):
    """Mesh-origin matrices with 2 % of their entries re-drawn at random columns keep compact groups with stray slots
    (hispmv_plan.h): the call takes the step kernel's other instantiation, next to parts without strays and to a tile stream."""
    from hispmv_amd import matrices as M
    names = ["PFlow_742", "TSOPF_RS_b2383", "crankseg_2", "nd6k", "crystk03"]
    mats = []
    for n in names:
        rows, cols, rp, ci, va = M.standin_variant(n, "stray2")
        mats.append(dict(name=n + ":stray2", rows=rows, cols=cols, nnz=int(rp[-1]), rp=rp, ci=ci, va=va))
    mats += [m for m in M.benchmark_set(["soc-Pokec", "mouse_gene", "ford2", "trans5"], False) if "rp" in m]
    assert sum(m["nnz"] for m in mats) * 8 >= (256 << 20)
    grids, info = _run(torch_mod, mats, {"HISPMV_STEP_KERNEL": "0"})
    # (the host-only packer says so for one of them: groups with stray slots exist under the plan the loader takes)
    from hispmv_amd import prep
    nd = mats[3]
    lay = prep.device_layout_from_coo(np.repeat(np.arange(nd["rows"], dtype=np.int32), np.diff(nd["rp"])), nd["ci"], nd["va"], nd["rows"], nd["cols"])
    assert lay["stray_slices"] > 0 and lay["compact_slices"] == lay["n_slices"]
    step, _ = _run(torch_mod, mats, {"HISPMV_STEP_KERNEL": "1"}, reps=2)
    for m, a, b in zip(mats, grids, step):
        assert np.all(np.isfinite(b)) and _same_bits(a, b), f'{m["name"]}: step kernel differs from the grids'
