#!/usr/bin/env python3
"""Run one of the GPU-side Python tools on the simulator: `python tools/wavesim/run_on_sim.py tools/gemm_chain_bench.py --batch 2`.
For checking that a tool's Python runs end to end before it spends GPU minutes; every time it prints is simulator wall-clock."""
import os
import runpy
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from tests import fakecuda  # noqa: E402

fakecuda.install()
sys.argv = sys.argv[1:]
runpy.run_path(sys.argv[0], run_name="__main__")
