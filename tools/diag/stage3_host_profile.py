#!/usr/bin/env python
"""cProfile of tools/stage3_step.py's timed iterations: where the host time of a Stage-3 iteration goes."""
import cProfile, pstats, os, sys, io, runpy
sys.argv = ["stage3_step.py", "bf16"]
pr = cProfile.Profile()
pr.enable()
runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "stage3_step.py"), run_name="__main__")
pr.disable()
for key in ("tottime", "cumulative"):
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats(key).print_stats(45); print(s.getvalue()[:9000])
