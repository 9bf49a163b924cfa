#!/usr/bin/env python3
"""Generates tests/golden/apps_calls.json IN THE BUILD CONTAINER (needs /root/reference): runs the reference's
apps/general_test.py and apps/model_test.py unchanged, seeded, against the recording stand-in tests/recording_pyhispmv
(numpy/scipy arithmetic) and the sparse_dot_mkl stand-in of this repo, and keeps what the drop-in claim is checked against:
the sequence of FpgaHandle calls with argument shapes / dtypes / scalars, and the verdict lines the scripts print.
Nothing of the reference's text is stored -- call records and printed results only.

    python3 tests/golden/make_apps_calls.py
"""
import json
import os
import re
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
APPS = Path("/root/reference/apps")
RUNNER = """
import runpy, sys
import numpy as np
np.random.seed(0)
try:
    import torch
    torch.manual_seed(0)
except Exception:
    pass
sys.argv = [sys.argv[1]] + sys.argv[2:]
runpy.run_path(sys.argv[0], run_name="__main__")
"""


def run(script, args=()):
    with tempfile.TemporaryDirectory() as tmp:
        rec = os.path.join(tmp, "calls.json")
        env = dict(os.environ)
        env["PYTHONPATH"] = os.pathsep.join([str(ROOT / "tests" / "recording_pyhispmv"), str(APPS), str(ROOT)])
        env["HISPMV_RECORD_OUT"] = rec
        env["OMP_NUM_THREADS"] = "8"
        p = subprocess.run([sys.executable, "-c", RUNNER, str(APPS / script), *args], capture_output=True, text=True, env=env, timeout=3000, cwd=tmp)
        if p.returncode != 0:
            raise SystemExit(f"{script} failed:\n{p.stderr[-2000:]}")
        return json.load(open(rec)), p.stdout


def main():
    out = {"generated_by": "tests/golden/make_apps_calls.py", "seed": "np.random.seed(0); torch.manual_seed(0) before the script runs",
           "reference_scripts": ["apps/general_test.py", "apps/model_test.py"]}
    calls, stdout = run("general_test.py")
    out["general_test"] = {"calls": calls,
                           "verdicts": [l.strip() for l in stdout.splitlines() if re.search(r"result is (correct|incorrect)", l)],
                           "max_errors": [l.strip() for l in stdout.splitlines() if l.startswith("Maximum ")]}
    calls, stdout = run("model_test.py")
    mx = {k: float(v) for k, v in re.findall(r"(Max Absolute Error|Max Relative Error): ([0-9.eE+-]+)", stdout)}
    out["model_test"] = {"calls": calls, "printed_max_errors": mx,
                         "densities_printed": [float(v) for v in re.findall(r"Density: ([0-9.]+)", stdout)]}
    (ROOT / "tests" / "golden" / "apps_calls.json").write_text(json.dumps(out, indent=1) + "\n")
    print(json.dumps(out, indent=1)[:3000])


if __name__ == "__main__":
    main()
