"""bench.py --gpus 2 rehearsed on the one GPU of the test box (HISPMV_BENCH_REHEARSAL=1: both ranks on GPU 0, gloo backend): the
N > 1 code path of the benchmark -- launcher, sharding, boundary exchange, per-rank self-check -- with the fields the first real
multi-GPU run will be diagnosed with.  The numbers of such a run mean nothing; the structure does."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]


@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_two_rank_rehearsal_line(scaling):
    env = dict(os.environ, HISPMV_BENCH_REHEARSAL="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("HISPMV_HOST_THREADS", None)
    cmd = [sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--preheat", "0.01", "--no-cpu-baseline",
           "--per-matrix-reps", "0", "--matrices", "crankseg_2,crystk03,trans5,ford2", "--scaling", scaling, "--strong-gb", ""]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=str(ROOT))
    assert p.returncode == 0, (p.stdout[-400:], p.stderr[-1200:])
    line = [q for q in p.stdout.splitlines() if q.startswith("{") and '"metric"' in q]
    assert len(line) == 1
    d = json.loads(line[0])
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["backend"] == "gloo" and d["scaling"] == scaling
    assert d["y_checked"] is True and d["y_check"]["matrices_without_host_csr"] == 0
    rb = d["rank_breakdown"]
    assert rb["ranks"] == 2 and [q["rank"] for q in rb["per_rank"]] == [0, 1]
    for q in rb["per_rank"]:
        assert q["batch_us"] > 0 and q["exchange_us"] > 0 and q["nnz"] > 0 and q["formats"] and q["host_threads"] >= 1
    for key in ("batch_us", "exchange_us", "nnz"):
        assert {"min", "max", "mean", "argmax_rank"} <= set(rb[key])
    assert rb["batch_imbalance_max_over_mean"] >= 1.0
    # every rank boundary cuts through a row in both layouts: the exchange carried real partial sums
    assert sum(q["cut_heads"] for q in rb["per_rank"]) >= 1 and sum(q["cut_tails"] for q in rb["per_rank"]) >= 1
