"""Multi-GPU SpMV: nnz-balanced split of the CSR element range over the ranks of one node, x
replicated, and the exchange of the partial sums of rows cut by a rank boundary -- the reference's
"shared row" idea (rows split over PEs and merged by the reduction network, spmv-helper.cpp:265-347,
base_functions.cpp:356-437) one level up, over RCCL/xGMI.  The reference itself is single-device
(no collectives anywhere), so this module has no reference counterpart.

Only the cut rows travel: one all_gather of `n_matrices` floats per rank per step.  y is never
all-reduced (a ring all-reduce of the largest y would cost more than the whole 1-GPU SpMV).

Ownership rule (same as between wavefront slices): a row belongs to the rank that holds its LAST
element; that rank adds beta*bias.  Ranks holding an earlier part compute alpha*partial only (their
local bias entry for that row is zero) and publish it as their "tail".
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np


@dataclass
class Shard:
    rank: int
    world: int
    elem_begin: int        # global element range [begin, end) of this rank
    elem_end: int
    row_begin: int         # global index of local row 0
    n_rows: int            # local rows (first/last may be partial)
    row_ptr: np.ndarray    # int32 [n_rows+1], local
    col_idx: np.ndarray    # int32, global column ids (x is replicated)
    values: np.ndarray
    head_open: bool        # local row 0 continues a row begun on an earlier rank
    tail_open: bool        # last local row continues on a later rank (this rank does not own it)

    def local_bias(self, bias: np.ndarray) -> np.ndarray:
        """bias restricted to the local rows, zeroed where this rank is not the owner."""
        b = np.array(bias[self.row_begin:self.row_begin + self.n_rows], dtype=np.float32, copy=True)
        if self.tail_open and self.n_rows:
            b[-1] = 0.0
        return b


def split_points(n_elems: int, world: int) -> np.ndarray:
    """Equal cut of [0, n_elems) into `world` ranges (the first ranges take the remainder)."""
    base, rem = divmod(int(n_elems), world)
    sizes = np.full(world, base, dtype=np.int64)
    sizes[:rem] += 1
    return np.concatenate([[0], np.cumsum(sizes)])


def shard_csr(row_ptr, col_idx, values, world: int, rank: int) -> Shard:
    """The rank's share of a global CSR matrix.  As in the slice stream, every row counts at least one
    element (an empty row counts one filler), and the element sequence is cut into `world` equal ranges;
    a cut may fall inside a row.  The local CSR keeps global column ids (x is replicated)."""
    rp = np.asarray(row_ptr, dtype=np.int64)
    rows = rp.size - 1
    lens = np.diff(rp)
    eoff = np.concatenate([[0], np.cumsum(np.maximum(lens, 1))])          # element offsets incl. fillers
    cuts = split_points(int(eoff[-1]), world)
    b, e = int(cuts[rank]), int(cuts[rank + 1])
    if e == b:
        return Shard(rank, world, b, e, 0, 0, np.zeros(1, np.int32), np.zeros(0, np.int32), np.zeros(0, np.float32), False, False)
    r0 = int(np.searchsorted(eoff[1:], b, side="right"))                  # first row ending after b
    r1 = int(np.searchsorted(eoff[:-1], e, side="left"))                  # one past the last row starting before e
    # real (non-filler) elements of local row i: global rows r0+i, clipped to the element range
    lo = np.clip(b - eoff[r0:r1], 0, None)                               # elements of the row that lie before b
    hi = np.clip(eoff[r0 + 1:r1 + 1] - e, 0, None)                       # ... after e
    real = lens[r0:r1]
    k0 = rp[r0:r1] + np.minimum(lo, real)
    k1 = rp[r0:r1] + np.maximum(np.minimum(real, np.maximum(lens[r0:r1], 1) - hi), np.minimum(lo, real))
    k1 = np.where(real == 0, k0, k1)
    cnt = k1 - k0
    loc = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int32)
    if cnt.sum():          # the rows' element ranges are adjacent in the global arrays: one contiguous slice
        nz = np.nonzero(cnt)[0]
        idx = slice(int(k0[nz[0]]), int(k1[nz[-1]]))
        assert idx.stop - idx.start == int(cnt.sum())
    else:
        idx = slice(0, 0)
    return Shard(rank, world, b, e, r0, r1 - r0, loc, np.asarray(col_idx)[idx].astype(np.int32),
                 np.asarray(values)[idx].astype(np.float32), bool(eoff[r0] < b), bool(eoff[r1] > e))


def shard_of_stacked_blocks(cur, nxt, rows: int, cols: int, rank: int, world: int) -> Shard:
    """Weak-scaling workload of bench.py: the global matrix is `world` row blocks stacked (block b = the base
    matrix generated with seed+b, its columns shifted by b*cols; x is replicated), and rank k owns the element
    range of block k shifted by `delta` elements, so that every rank boundary cuts through a row and the
    boundary exchange carries real partial sums.  Rank k therefore needs block k and the first rows of block
    k+1 only.  cur / nxt = (row_ptr, col_idx, values) of blocks rank and rank+1 (nxt None on the last rank)."""
    def cut_of(rp):                      # a cut in the middle of the first row (after row 3) with >= 2 elements
        lens = np.diff(rp)
        cand = np.nonzero(lens[4:] >= 2)[0]
        r = int(cand[0]) + 4 if cand.size else 0
        return (int(rp[r] + lens[r] // 2), r) if lens[r] >= 2 else (0, 0)
    rp, ci, va = (np.asarray(a) for a in cur)
    d0, r0 = cut_of(rp) if rank > 0 else (0, 0)           # this block's first d0 elements belong to rank-1
    row_ptr = [np.asarray(rp[r0:], dtype=np.int64) - d0]
    row_ptr[0][0] = 0
    cols_l = [ci[d0:].astype(np.int64) + rank * cols]
    vals_l = [va[d0:]]
    head_open = rank > 0 and d0 > rp[r0]
    tail_open = False
    n_rows = rows - r0
    if nxt is not None:
        rp2, ci2, va2 = (np.asarray(a) for a in nxt)
        d1, r1 = cut_of(rp2)
        if d1 > 0:
            ext = np.asarray(rp2[1:r1 + 1], dtype=np.int64)
            ext = np.concatenate([ext, [d1]]) if d1 > rp2[r1] else ext
            row_ptr.append(ext + row_ptr[0][-1])
            cols_l.append(ci2[:d1].astype(np.int64) + (rank + 1) * cols)
            vals_l.append(va2[:d1])
            tail_open = d1 > rp2[r1]
            n_rows += ext.size
    rp_loc = np.concatenate(row_ptr)
    return Shard(rank, world, 0, int(rp_loc[-1]), rank * rows + r0, int(n_rows), rp_loc.astype(np.int32),
                 np.concatenate(cols_l).astype(np.int32), np.concatenate(vals_l).astype(np.float32),
                 bool(head_open), bool(tail_open))


def chain_weights(flags: np.ndarray, rank: int) -> np.ndarray:
    """flags[world, 3] = (head_open, tail_open, single_row) of one matrix on every rank.  Returns
    w[world] in {0,1}: the ranks whose tails are parts of this rank's first row -- rank-1 if its tail is
    open, and further down while the rank in between holds nothing but a piece of that same row."""
    w = np.zeros(flags.shape[0], dtype=np.float32)
    if not flags[rank, 0]:
        return w
    j = rank - 1
    while j >= 0 and flags[j, 1]:
        w[j] = 1.0
        if not (flags[j, 0] and flags[j, 2]):
            break
        j -= 1
    return w


class BoundaryExchange:
    """Per-step exchange of the cut rows.  `mats` is the list the caller keeps: each entry has a tensor
    "y" (the rank's local rows, device or CPU) and optionally "shard" (a Shard); entries without a shard
    are row-aligned blocks (nothing is cut) and contribute zeros -- the collective still runs, it is the
    path's exchange step.  Works on any torch.distributed backend (nccl = RCCL on the GPUs, gloo in the
    CPU tests).  With `world`/`rank` given explicitly the object touches no process group: it is one of the
    virtual ranks of a LoopbackWorld (several ranks in one process, the all_gather a concatenation).
    `fpga` is the FpgaHandle whose SpMVs produce the y vectors: on the GPU the two boundary kernels are launches of ITS context
    (include/hispmv.h: one meaning of a NULL stream for SpMVs and boundary kernels alike); `stream` is the raw stream handle
    both go to -- None = the current torch stream at prepare/run time, 0 = the context's own stream."""

    def __init__(self, n_mats: int, device, world: int | None = None, rank: int | None = None, fpga=None, stream: int | None = None):
        import torch
        self.torch = torch
        self.fpga = fpga
        self.stream_arg = stream
        self.loopback = world is not None
        if self.loopback:
            self.dist = None
            self.world, self.rank = int(world), int(rank)
        else:
            import torch.distributed as dist
            self.dist = dist
            self.world, self.rank = dist.get_world_size(), dist.get_rank()
        self.device = device
        self.n = n_mats
        self.send = torch.zeros(n_mats, dtype=torch.float32, device=device)
        self.recv = torch.zeros(self.world * n_mats, dtype=torch.float32, device=device)
        self.ready = False

    def _all_gather(self, out, inp):
        if self.loopback:
            raise RuntimeError("a virtual rank has no collective of its own: drive it through LoopbackWorld.step")
        if self.dist.get_backend() == "gloo":
            parts = list(out.view(self.world, -1).unbind(0))
            self.dist.all_gather(parts, inp.reshape(-1))
        else:
            self.dist.all_gather_into_tensor(out, inp)

    def local_flags(self, mats) -> np.ndarray:
        """flags[n, 3] = (head_open, tail_open, single_row) of this rank's shard of every matrix."""
        flags = np.zeros((self.n, 3), dtype=np.float32)
        for i, m in enumerate(mats):
            sh = m.get("shard")
            if sh is not None:
                flags[i] = (sh.head_open, sh.tail_open, sh.n_rows == 1)
        return flags

    def finish_setup(self, mats, flags: np.ndarray, allf: np.ndarray) -> None:
        """allf[world, n, 3]: every rank's flags -> chain weights, tail mask, head list."""
        torch = self.torch
        w = np.stack([chain_weights(allf[:, i, :], self.rank) for i in range(self.n)])   # [n, world]
        self.weights = torch.from_numpy(w).to(self.device)
        self.tail_mask = torch.from_numpy(flags[:, 1].copy()).to(self.device)
        self.heads = [i for i in range(self.n) if flags[i, 0]]
        self.has_rows = [bool(m["y"].numel()) for m in mats]
        self.zero = torch.zeros((), dtype=torch.float32, device=self.device)
        self.ready = True

    def _setup(self, mats):
        torch = self.torch
        flags = self.local_flags(mats)
        f = torch.from_numpy(flags.reshape(-1)).to(self.device)
        allf = torch.zeros(self.world * f.numel(), dtype=torch.float32, device=self.device)
        self._all_gather(allf, f)
        self.finish_setup(mats, flags, allf.cpu().numpy().reshape(self.world, self.n, 3))

    def _device_tables(self, mats):
        """Device path: pointer tables for hispmv_boundary_pack / hispmv_boundary_apply (include/hispmv.h): the last
        and the first entry of every matrix's local y.  Rebuilt when a y tensor moved."""
        torch = self.torch
        ptrs = tuple(m["y"].data_ptr() if ok else 0 for m, ok in zip(mats, self.has_rows))
        if getattr(self, "_ptrs", None) == ptrs:
            return
        self._ptrs = ptrs
        last = [p + 4 * (m["y"].numel() - 1) if p else 0 for p, m in zip(ptrs, mats)]
        heads = set(self.heads)
        first = [p if (p and i in heads) else 0 for i, p in enumerate(ptrs)]
        self._d_last = torch.tensor(last, dtype=torch.int64, device=self.device)
        self._d_first = torch.tensor(first, dtype=torch.int64, device=self.device)
        self._w = self.weights.to(torch.float32).contiguous()

    def prepare(self, mats) -> None:
        """Everything `run` needs that does not change from step to step (flags, chain weights, the device tables of
        pointers, the stream handle): call once, then `run(..., prepared=True)` does no Python work per matrix."""
        if not self.ready:
            self._setup(mats)
        if self.send.is_cuda:
            self._device_tables(mats)
            self._stream = self._pick_stream()

    def _pick_stream(self) -> int:
        if self.fpga is None:
            raise RuntimeError("BoundaryExchange on a GPU needs the FpgaHandle whose context launches the boundary kernels (fpga=...)")
        return int(self.stream_arg) if self.stream_arg is not None else self.torch.cuda.current_stream(self.device).cuda_stream

    def pack(self, mats, prepared: bool = False) -> None:
        """First half of a step: the tails (y_local[-1] of rows this rank does not own: alpha*partial, its bias entry
        was zeroed) -> self.send.  On the GPU one tiny launch of libhispmv."""
        torch = self.torch
        if self.send.is_cuda:
            if not prepared:
                self._device_tables(mats)
                self._stream = self._pick_stream()
            self.fpga.boundary_pack(self._d_last.data_ptr(), self.tail_mask.data_ptr(), self.send.data_ptr(), self.n, self._stream)
            return
        last = torch.stack([m["y"][-1] if ok else self.zero for m, ok in zip(mats, self.has_rows)])
        torch.mul(last, self.tail_mask, out=self.send)

    def apply(self, mats) -> None:
        """Second half: self.recv (every rank's tails) -> the chain of this rank's first rows, added in rank order."""
        if self.send.is_cuda:
            self.fpga.boundary_apply(self._d_first.data_ptr(), self.recv.data_ptr(), self._w.data_ptr(), self.n, self.world, self._stream)
            return
        if self.heads:
            incoming = (self.recv.view(self.world, self.n).t() * self.weights).sum(dim=1)
            for i in self.heads:
                mats[i]["y"][0] += incoming[i]

    def run(self, mats, alpha: float = 1.0, prepared: bool = False) -> None:
        """After every rank's local SpMV of every matrix: publish the tails, gather them, add the chain into the
        owner's first row.  On the GPU: two tiny launches of libhispmv around ONE all_gather (a chain of ~30 torch
        element ops would cost about half a step of the 20-matrix set).  prepared=True: `prepare` was called and
        neither the y tensors nor the current stream changed since.  (`alpha` is already inside the tails.)"""
        if not self.ready:
            self._setup(mats)
        self.pack(mats, prepared)
        self._all_gather(self.recv, self.send)
        self.apply(mats)


class LoopbackWorld:
    """`world` virtual ranks in ONE process: every rank keeps its own list of matrices (handles, vectors, Shard) and
    its own BoundaryExchange; a step is `local_step(rank)` for every rank, then pack on every rank, the all_gather as a
    concatenation of the send buffers, then apply on every rank -- the same two kernels and the same weights as the
    process-per-GPU path, without a process group.  For checking the sharded path at world sizes a one-GPU box cannot
    host as processes (tests/test_gpu_dist_full.py: world 8)."""

    def __init__(self, per_rank_mats, device, fpgas=None, stream: int | None = None):
        import torch
        self.torch = torch
        self.world = len(per_rank_mats)
        self.mats = per_rank_mats
        n = len(per_rank_mats[0])
        # fpgas: one FpgaHandle per virtual rank (GPU runs: the context whose SpMVs fill that rank's y vectors)
        self.ex = [BoundaryExchange(n, device, world=self.world, rank=r, fpga=fpgas[r] if fpgas else None, stream=stream) for r in range(self.world)]
        flags = [e.local_flags(m) for e, m in zip(self.ex, per_rank_mats)]
        allf = np.stack(flags)
        for e, m, f in zip(self.ex, per_rank_mats, flags):
            e.finish_setup(m, f, allf)

    def exchange(self) -> None:
        for e, m in zip(self.ex, self.mats):
            e.pack(m)
        allsend = self.torch.cat([e.send for e in self.ex])
        for e, m in zip(self.ex, self.mats):
            e.recv.copy_(allsend)
            e.apply(m)
