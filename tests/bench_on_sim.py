"""Helper for tests/test_bench_dryrun.py: bench.py's REAL step, profile pass and JSON line (not the CACO_BENCH_DRYRUN stand-ins)
executed on tools/wavesim through tests/fakecuda.py, at 2 pairs per rank and one layer per tower so that it takes seconds.
The numbers it prints are simulator wall-clock and mean nothing; the test reads the line's structure only."""
import os
import sys
from dataclasses import replace

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from tests import fakecuda  # noqa: E402

fakecuda.install()
from cacophony_amd import config as Cfg  # noqa: E402

_a, _t = Cfg.default_audio_config, Cfg.default_text_config
Cfg.default_audio_config = lambda: replace(_a(), num_layers=1)
Cfg.default_text_config = lambda: replace(_t(), num_hidden_layers=1)
import bench  # noqa: E402

bench.B_PER_GPU = 2
sys.argv = ["bench.py", "--steps", "2", "--warmup", "1", "--profile-steps", "1", "--no-cpu-baseline", "--no-extra-configs",
            "--no-live-traffic"]          # the counter pass spawns rocprofv3 children: hardware only (tests/test_bench_traffic.py covers its plumbing)
bench.main()
