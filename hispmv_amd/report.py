"""Result comparison the way the reference's drivers print it: ``print_error_stats`` follows
``HiSpmvHandle::printErrorStats`` (common/src/spmv-helper.cpp:835-895) -- relative errors |(|out| - |ref|)| / |ref|,
exact matches dropped, at most ten listed one by one, otherwise a ten-bin histogram between the smallest and the
largest error with the reference's line formats.  Differences, on inputs the reference mishandles (SURVEY.md
Appendix B.10): a zero reference entry is skipped instead of producing inf/NaN, and equal errors (bin width 0) print
a single bin instead of tripping an assert."""
from __future__ import annotations

import sys

import numpy as np


def error_stats_lines(cpu_ref, out) -> list:
    cpu_ref, out = np.asarray(cpu_ref), np.asarray(out)
    if cpu_ref.shape != out.shape:
        raise ValueError("Error: Vector sizes do not match!")          # spmv-helper.cpp:837-839
    fp, cp = np.abs(out.astype(np.float64)), np.abs(cpu_ref.astype(np.float64))
    with np.errstate(divide="ignore", invalid="ignore"):
        rel = np.abs(fp - cp) / cp
    rel = rel[np.isfinite(rel) & (rel != 0)]
    if rel.size == 0:
        return ["No mismatch found"]
    if rel.size <= 10:
        return ["Found atmost 10 mismatches, Relative Errors:"] + [f"\t{e:.6g}" for e in rel]     # ostream default: 6 significant digits
    lo, hi = float(rel.min()), float(rel.max())
    bw = (hi - lo) / 10
    lines = ["Relative Error Range:\tCount"]
    if bw <= 0:
        return lines + [f"[{lo:.3e}, {lo:.3e}]:\t{rel.size}"]
    idx = np.minimum(((rel - lo) / bw).astype(np.int64), 9)               # :883-886
    counts = np.bincount(idx, minlength=10)
    for k in range(10):
        start = lo + k * bw
        lines.append(f"[{start:.3e}, {start + bw:.3e}):\t{counts[k]}")
    return lines


def print_error_stats(cpu_ref, out, file=None) -> None:
    print("\n".join(error_stats_lines(cpu_ref, out)), file=file or sys.stdout)
