#!/usr/bin/env python3
"""Run a uDALES case on the device core:  python run_case.py path/to/namoptions.NNN [--steps N] ...
(thin launcher for u-dales_amd/udcore/run.py; see its docstring)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "u-dales_amd"))
from udcore.run import main  # noqa: E402

if __name__ == "__main__":
    raise SystemExit(main())
