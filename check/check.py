#!/usr/bin/env python3
"""`python check/check.py --ref-av-vels-file=... --ref-final-state-file=... --av-vels-file=...
--final-state-file=... [--tolerance=1]` — same flags and exit codes as the reference's Python-2
`check/check.py`; the implementation is mpilattice-boltzmann_amd/checker.py (Python 3).
Reference files may be gzip'd (tests/golden/check/*.dat.gz)."""
import importlib.util
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
# load the checker module alone (no need for the native library just to compare files)
spec = importlib.util.spec_from_file_location(
    "lbm_checker", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mpilattice-boltzmann_amd", "checker.py"))
mod = importlib.util.module_from_spec(spec)
sys.modules["lbm_checker"] = mod
spec.loader.exec_module(mod)
sys.exit(mod.main())
