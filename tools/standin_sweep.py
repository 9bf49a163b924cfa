#!/usr/bin/env python3
"""Stand-in sweep: the eight mesh-origin shapes of the SuiteSparse set under perturbations of their
structure -- half / double band, 2 / 5 / 10 % stray couplings, 4 K-row block shuffle (hispmv_amd.matrices.standin_variant) -- each
launched ALONE on the MI355X: time between two HIP events with a cache flush in between (cold Infinity Cache), the plan the loader
chose, algorithmic GB/s and the fraction of the 8 TB/s peak.  A variant that drops below 0.45 while its neighbours stay above 0.6
is a planner cliff.  Writes a JSON table (default gpurun_out/standin_sweep.json).
  python tools/standin_sweep.py [--names a,b] [--variants base,stray5] [--out file]"""
import argparse
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
ALPHA, BETA = 0.55, -2.05
HW = ("sweep.xclbin", 0, 24, 1, 1, 2, 5, True, False, True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--names", default="")
    ap.add_argument("--variants", default="")
    ap.add_argument("--out", default=str(ROOT / "gpurun_out" / "standin_sweep.json"))
    ap.add_argument("--reps", type=int, default=7)
    args = ap.parse_args()
    import torch
    import pyhispmv
    from hispmv_amd import matrices as M
    dev = torch.device("cuda", 0)
    names = [n for n in args.names.split(",") if n] or [q[0] for q in M.SUITESPARSE_SET if q[3] == "fem"]
    variants = [v for v in args.variants.split(",") if v] or list(M.STANDIN_VARIANTS)
    flush = torch.zeros(160 << 20, dtype=torch.float32, device=dev)          # 640 MB: more than the 256 MiB Infinity Cache
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    rows_out = []
    for name in names:
        for var in variants:
            t0 = time.time()
            rows, cols, rp, ci, va = M.standin_variant(name, var)
            h = pyhispmv.FpgaHandle(*HW)
            h.set_arena_bytes(32 << 30)
            idx = h.create_sparse_handle_from_csr(rp, ci, va, rows, cols)
            h.load_matrices()
            info = h.matrix_info(idx)
            x = torch.rand(cols, dtype=torch.float32, device=dev)
            b = torch.rand(rows, dtype=torch.float32, device=dev)
            y = torch.zeros(rows, dtype=torch.float32, device=dev)
            ts = []
            for _ in range(args.reps + 2):
                flush.add_(1.0)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                h.spmv_device(idx, x.data_ptr(), b.data_ptr(), y.data_ptr(), ALPHA, BETA, stream.cuda_stream)
                e1.record(stream)
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e-3)
            h.synchronize()
            t = float(np.median(ts[2:]))
            warm = h.time_device(idx, x.data_ptr(), b.data_ptr(), y.data_ptr(), ALPHA, BETA, 20) * 1e-3
            # one check per variant: fp64 backward error of the result (scipy, not the oracle)
            import scipy.sparse as sp
            A = sp.csr_matrix((va.astype(np.float64), ci, rp), shape=(rows, cols))
            xs, bs = x.cpu().numpy().astype(np.float64), b.cpu().numpy().astype(np.float64)
            y64 = ALPHA * (A @ xs) + BETA * bs
            A.data = np.abs(A.data)
            mag = ALPHA * (A @ np.abs(xs)) + np.abs(BETA * bs)
            err = float(np.max(np.abs(y.cpu().numpy() - y64) / np.maximum(mag, 1e-300)))
            ab = M.algorithmic_bytes(rows, cols, int(rp[-1]))
            lens = np.diff(rp)
            off = np.abs(ci.astype(np.int64) - np.repeat(np.arange(rows, dtype=np.int64), lens))
            rec = dict(name=name, variant=var, rows=rows, nnz=int(rp[-1]), offset_p998=int(np.percentile(off, 99.8)), us=round(t * 1e6, 2), us_back_to_back=round(warm * 1e6, 2),
                       alg_gbs=round(ab / t / 1e9, 1), frac=round(ab / t / 1e9 / 8000.0, 4), backward_error=err,
                       format="tile_stream" if info["format"] == 1 else "slices", tile_kind=info["tile_kind"], parts=info["col_tiles"],
                       plan=f'{info["block_threads"]}t/{info["group_slices"]}s/{info["lds_bytes"] // 1024}KiB/{info["col_tiles"]}ct/'
                            f'{100 * info["compact_slices"] // max(1, info["n_slices"])}%c', prep_s=round(info["prep_seconds"], 3))
            assert err < 1e-5, rec
            rows_out.append(rec)
            print(json.dumps(rec), f"({time.time() - t0:.1f} s)", flush=True)
            h.close()
            del x, b, y
    # cliffs: a variant below 0.45 whose base (or a neighbouring strength of the same perturbation) is above 0.6
    by = {(r["name"], r["variant"]): r for r in rows_out}
    cliffs = []
    neighbours = {"half_band": ["base"], "double_band": ["base"], "stray2": ["base", "stray5"], "stray5": ["stray2", "stray10"], "stray10": ["stray5"], "shuffle4k": ["base"]}
    for (n, v), r in by.items():
        nb = [by[(n, q)] for q in neighbours.get(v, []) if (n, q) in by]
        if r["frac"] < 0.45 and any(q["frac"] >= 0.6 for q in nb):
            cliffs.append(dict(name=n, variant=v, frac=r["frac"], plan=r["plan"], neighbours={q["variant"]: q["frac"] for q in nb}))
    out = dict(how="each variant launched alone, HIP events around one launch behind a 640 MB cache flush, median of %d; frac = algorithmic bytes "
                   "(8 nnz + 16 rows + 4) / time / 8 TB/s" % args.reps, rows=rows_out, cliffs=cliffs)
    Path(args.out).parent.mkdir(parents=True, exist_ok=True)
    Path(args.out).write_text(json.dumps(out, indent=1) + "\n")
    print(f"{len(rows_out)} variants, {len(cliffs)} cliffs -> {args.out}")


if __name__ == "__main__":
    main()
