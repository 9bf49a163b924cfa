#!/usr/bin/env python3
"""Generates tests/golden/collect_patterns.json from the reference's log scraper (run in the build container, where
/root/reference exists): imports builds/collect_data.py and dumps its METRIC_PATTERNS and SAMPLE_PATTERN (the regexes
of :8-23).  The fixture is data -- pattern strings -- not source text; tests/test_log_schema.py checks that the logs
of examples/spmv_host.py are parsed by every pattern."""
import importlib.util
import json
import sys
from pathlib import Path

ref = Path(sys.argv[1] if len(sys.argv) > 1 else "/root/reference") / "builds" / "collect_data.py"
spec = importlib.util.spec_from_file_location("ref_collect_data", ref)
mod = importlib.util.module_from_spec(spec)
spec.loader.exec_module(mod)          # module level only defines the patterns and functions (main() is guarded)
out = {"source": "builds/collect_data.py:8-23", "metric_patterns": mod.METRIC_PATTERNS, "sample_pattern": mod.SAMPLE_PATTERN}
dst = Path(__file__).resolve().parent / "collect_patterns.json"
dst.write_text(json.dumps(out, indent=1) + "\n")
print(f"wrote {dst} ({len(mod.METRIC_PATTERNS)} patterns)")
