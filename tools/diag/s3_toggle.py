import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import sos_wsod_amd.frcnn as F
import stage3_step as S
# usage: s3_toggle.py FLAG_NAME   — the Stage-3 iteration with a module switch of frcnn.py on / off, alternating, same process
name = sys.argv[1] if len(sys.argv) > 1 else "MASKS_IN_PRODUCERS"
for rep in range(3):
    for flag in (True, False):
        setattr(F, name, flag)
        ms, _ = S.time_step(torch.bfloat16, n=10)
        print(name, flag, round(ms, 2), "ms", flush=True)
