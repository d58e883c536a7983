import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import sos_wsod_amd.ops as ops
torch.manual_seed(11); dev = "cuda"
co, ci = 72, 130
w = torch.randn(co, ci, device=dev)
bn = (torch.rand(co, device=dev) + 0.5, torch.randn(co, device=dev), torch.randn(co, device=dev), torch.rand(co, device=dev) + 0.1)
e = dict(kind=0, w=w, dst=torch.empty(co * ci, device=dev), bn=bn, scale=torch.empty(co, device=dev), shift=torch.empty(co, device=dev))
ops.StagePlan([e], torch.float32).run(); torch.cuda.synchronize()
W, B, M, V = (t.cpu() for t in bn)
for name, sc in [("1/sqrt", W * (1.0 / torch.sqrt(V + 1e-5))), ("rsqrt", W * torch.rsqrt(V + 1e-5)), ("w/sqrt", W / torch.sqrt(V + 1e-5)),
                 ("f64", (W.double() * (1.0 / torch.sqrt(V.double() + 1e-5))).float())]:
    d = (e["scale"].cpu() - sc)
    print(name, "scale mismatches", int((d != 0).sum()), "max", float(d.abs().max()))
sc = W * (1.0 / torch.sqrt(V + 1e-5))
print("eps add", float((V + 1e-5)[0]), float(V[0]), "kernel scale[0]", float(e["scale"][0]), "cpu", float(sc[0]))
