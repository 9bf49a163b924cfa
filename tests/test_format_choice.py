"""The loader's FORMAT AND TILING DECISION as host code under test: hispmv_amd/csrc/hispmv_choose.cpp through
the host-only entry hispmv_prep_choose_format -- the MI355X analogue of the reference's per-matrix configuration search
(/root/reference/automation_tool/src/dse.py:23-95).  No device needed.

* today's choices for every matrix the benchmark and the parity tests run -- the 20 shapes of the SuiteSparse set in both
  stand-in families, C3, the sparse C4 layers -- are pinned in tests/golden/format_choices.json
  (tests/golden/make_format_choices.py regenerates it, on purpose, when the planner changes);
* the decision's branches on small matrices built for them, with the environment switches the loader honours."""
import json
import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]
GOLD = json.loads((ROOT / "tests" / "golden" / "format_choices.json").read_text())
sys.path.insert(0, str(ROOT / "tests" / "golden"))


def test_pinned_choices_for_every_benchmarked_matrix():
    import make_format_choices as G
    seen = set()
    bad = []
    for name, rows, cols, rp, ci, va in G.cases():
        got = G.decide(rows, cols, rp, ci, va)
        seen.add(name)
        if got != GOLD[name]:
            bad.append((name, {k: (GOLD[name][k], got[k]) for k in got if got[k] != GOLD[name][k]}))
    assert not bad, f"format choice changed (golden, now): {bad}"
    assert seen == set(GOLD), sorted(set(GOLD) ^ seen)
    # what the pins say, in words (the plans that carry the headline number)
    s = {k.split(":", 1)[1]: v for k, v in GOLD.items() if k.startswith("structured:")}
    assert [n for n, v in s.items() if v["format"] == 1] == ["soc-Pokec", "ASIC_680k", "nxp1", "analytics", "boyd2", "language"]
    assert s["mouse_gene"]["tile_kind"] == 1 and s["mouse_gene"]["parts"] == 2 and s["mouse_gene"]["lds_floats"] > 0      # two LDS-window column tiles
    assert GOLD["uniform:PFlow_742"]["tile_kind"] == 2 and GOLD["uniform:Si41Ge41H72"]["tile_kind"] == 2                  # band tiles
    assert all(v["threads"] == 256 for n, v in s.items() if v["n_elems"] < (3 << 20) and v["format"] == 0)


def _band(rows, per_row, half, seed=3):
    rng = np.random.default_rng(seed)
    r = np.repeat(np.arange(rows, dtype=np.int64), per_row)
    c = np.clip(r + rng.integers(-half, half + 1, size=r.size), 0, rows - 1)
    order = np.lexsort((c, r))
    rp = np.arange(rows + 1, dtype=np.int64) * per_row
    return rp, c[order].astype(np.int32), np.ones(r.size, np.float32)


def _choose_in_child(env_extra, expr):
    """The switches are read from the environment at decision time; a child process keeps this one's environment clean."""
    code = ("import sys, json; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import numpy as np\n"
            "from test_format_choice import _band\n"
            "from hispmv_amd.prep import choose_format_from_csr\n"
            "rp, ci, va = %s\n"
            "print(json.dumps(choose_format_from_csr(rp, ci, va, len(rp) - 1, len(rp) - 1, 256)))\n") % (str(ROOT), str(ROOT / "tests"), expr)
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, **env_extra), timeout=300)
    assert p.returncode == 0, p.stderr[-800:]
    return json.loads(p.stdout.strip().splitlines()[-1])


def test_branches_and_switches():
    from hispmv_amd.prep import choose_format_from_csr
    # a narrow band: one slice stream with an LDS window, nothing gathers through L2
    rp, ci, va = _band(200000, 12, 400)
    d = choose_format_from_csr(rp, ci, va, 200000, 200000, 256)
    assert (d["format"], d["tile_kind"], d["parts"]) == (0, 0, 1) and d["lds_floats"] > 0 and d["l2_gather_elems"] == 0
    # a wide unstructured band of 5 M entries: cut along the diagonal (band tiles), every part with a clean window
    wide = "_band(250000, 20, 30000)"
    d = _choose_in_child({}, wide)
    assert d["format"] == 0 and d["tile_kind"] == 2 and d["parts"] >= 2 and d["lds_floats"] > 0 and d["tile_base"] < 0
    # ... the same matrix without band tiles: the tile stream (its gathers share cache lines); without either: slices through L2
    d = _choose_in_child({"HISPMV_BAND_TILES": "0"}, wide)
    assert d["format"] == 1 and d["parts"] == 1 and 0 < d["lines_per_gather_x1000"] <= 32000
    d = _choose_in_child({"HISPMV_BAND_TILES": "0", "HISPMV_FORMAT": "slices"}, wide)
    assert d["format"] == 0 and d["tile_kind"] in (0, 1)
    # below 1 M entries a scattered matrix stays a slice stream of 256-thread groups that gather through L2 ...
    small = "_band(100000, 8, 45000)"
    d = _choose_in_child({}, small)
    assert (d["format"], d["threads"], d["lds_floats"]) == (0, 256, 0) and d["l2_gather_elems"] == d["n_slices"] * 1024
    # ... unless the threshold is lowered (HISPMV_TTS_MIN_NNZ) or the tile stream is forced
    assert _choose_in_child({"HISPMV_TTS_MIN_NNZ": "100000"}, small)["format"] == 1
    assert _choose_in_child({"HISPMV_FORMAT": "tts"}, small)["format"] == 1
    # the tall geometry of a tile stream: two column parts
    d = _choose_in_child({"HISPMV_BAND_TILES": "0", "HISPMV_TTS_GEOMETRY": "tall"}, wide)
    assert d["format"] == 1 and d["parts"] == 2 and d["tile_kind"] == 1 and d["tile_width"] > 0
    # ... and the experiment knob for other shapes of the column parts (rows, slots, tiles per part, zero fill, parts)
    d = _choose_in_child({"HISPMV_BAND_TILES": "0", "HISPMV_TTS_GEOMETRY": "tall", "HISPMV_TTS_TALL_SHAPE": "8192,28672,256,0,4"}, wide)
    assert d["format"] == 1 and d["parts"] == 4 and d["group"] == 28
    # a gather of 64 column-sorted elements that touches 32 - 48 lines of x: a tile stream since round 4 (HISPMV_TTS_MAX_LINES, default
    # 48: ASIC_680k's case -- fewer, longer-lived workgroups cost the step of the benchmark set less than 645 L2-gather groups)
    sparse = "_band(800000, 2, 400000)"
    d = _choose_in_child({}, sparse)
    assert d["format"] == 1 and 32000 < d["lines_per_gather_x1000"] <= 48000
    assert _choose_in_child({"HISPMV_TTS_MAX_LINES": "32"}, sparse)["format"] == 0
    # stray couplings: 3 % of a narrow band's entries at random columns -> split into the windowed part and the strays (tile_kind 3);
    # 12 % strays of the same band still split (<= 15 %), and the switch turns it off
    strays = "(lambda t: (t[0], np.sort(np.where(np.random.default_rng(5).random(t[1].size) < %s, np.random.default_rng(6).integers(0, 300000, t[1].size), t[1]).reshape(300000, 16), axis=1).reshape(-1).astype(np.int32), t[2]))(_band(300000, 16, 1500))"
    # 3 % (31 per slice): the kernel's stray slots serve them (<= 64 per slice) -- one stream, every slice compact; without the slots:
    # the split; 12 % (123 per slice): beyond the slots, split; and the switch turns the split off
    d = _choose_in_child({}, strays % "0.03")
    assert (d["format"], d["tile_kind"], d["parts"]) == (0, 0, 1) and d["lds_floats"] > 0 and d["l2_gather_elems"] > 0
    d = _choose_in_child({"HISPMV_STRAY_SLOTS": "0"}, strays % "0.03")
    assert (d["format"], d["tile_kind"], d["parts"]) == (0, 3, 2) and d["lds_floats"] > 0
    assert _choose_in_child({}, strays % "0.12")["tile_kind"] == 3
    d = _choose_in_child({"HISPMV_STRAY_SPLIT": "0"}, strays % "0.12")
    assert d["tile_kind"] == 0 and d["parts"] == 1 and d["l2_gather_elems"] > 0
    # fewer compute units, another plan: the decision is a function of (matrix, n_cus) only
    rp, ci, va = _band(200000, 12, 400)
    a = choose_format_from_csr(rp, ci, va, 200000, 200000, 256)
    b = choose_format_from_csr(rp, ci, va, 200000, 200000, 256)
    assert a == b
    assert choose_format_from_csr(rp, ci, va, 200000, 200000, 8)["format"] == 0


def test_window_membership_is_the_split_criterion():
    """hispmv_prep_window_membership: the entries outside their workgroup's x window are exactly the strays of a banded matrix with
    10 % of its entries re-drawn (more than the kernel's stray slots take) -- and the loader's two parts hold exactly those two sets
    (element counts of the decision)."""
    from hispmv_amd.prep import choose_format_from_csr, window_membership
    rng = np.random.default_rng(5)
    rows = 300000
    rp, ci, va = _band(rows, 16, 1500)
    ci = ci.reshape(rows, 16).astype(np.int64)
    far = rng.random(ci.shape) < 0.10
    ci = np.sort(np.where(far, rng.integers(0, rows, ci.shape), ci), axis=1)
    r = np.repeat(np.arange(rows, dtype=np.int32), 16)
    inside, order = window_membership(r, ci.reshape(-1).astype(np.int32), va, rows, rows, 256)
    assert np.array_equal(order, np.arange(r.size))                       # the triplets were in CSR order already
    out_share = 1.0 - inside.mean()
    assert 0.08 < out_share < 0.12
    # an entry within the band of its row is inside (its block is used by many rows of the group); a far one is not, up to the few
    # re-drawn columns that land inside the band by chance
    near = np.abs(ci - np.arange(rows)[:, None]) <= 1500
    assert inside.reshape(rows, 16)[near].mean() > 0.999 and inside.reshape(rows, 16)[~near].mean() < 0.01
    d = choose_format_from_csr(rp, ci.reshape(-1).astype(np.int32), va, rows, rows, 256)
    assert d["tile_kind"] == 3 and d["parts"] == 2
    # elements of the two parts = inside + outside entries + one filler per row that has no stray (part 1) / no inside entry (part 0)
    n_out = int((inside == 0).sum())
    rows_with_out = np.unique(r[inside == 0]).size
    expected = int(inside.sum()) + n_out + (rows - rows_with_out)
    assert expected <= d["n_elems"] <= 1.07 * expected                   # (+ the zero-valued elements of row-aligned slices, <= 6 %)
