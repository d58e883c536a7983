import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import sos_wsod_amd.frcnn as F
import stage3_step as S
for rep in range(2):
    for flag in (True, False):
        F.GROUP_LINEAR_WGRADS = flag
        ms, _ = S.time_step(torch.bfloat16, n=10)
        print("GROUP_LINEAR_WGRADS", flag, round(ms, 2), "ms", flush=True)
