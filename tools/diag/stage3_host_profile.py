"""cProfile of the steady-state Stage-3 iterations only (tools/stage3_step.py's step and batches; set-up and warm-up excluded):
where the issuing thread's time goes.  The backward pass is issued by autograd's worker thread and shows up as `run_backward` here."""
import cProfile, pstats, os, sys, io
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import stage3_step as S
dev = torch.device("cuda", 0)
step = S.make_step(torch.bfloat16, dev)
n = int(os.environ.get("ITERS", 20))
batches = S.make_batches(4 + n, 800, 1216, dev)
for i in range(4):
    step.run_step(batches[i])
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for i in range(n):
    step.run_step(batches[4 + i])
pr.disable()
torch.cuda.synchronize()
for key in ("tottime", "cumulative"):
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats(key).print_stats(60); print(s.getvalue()[:14000])
