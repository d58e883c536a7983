cd $GRAFT_REPO_ROOT
export SW_DIST_BACKEND=gloo SW_BENCH_DEVICE=0 HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=$PWD SW_TEST_ITER_SIZE=2
python -m torch.distributed.run --nnodes=1 --nproc-per-node=2 --master-addr 127.0.0.1 --master-port 29533 tests/ddp_real_worker.py /tmp/acc bf16 2>&1 | tail -5
python - <<'PY'
import torch
for r in range(2):
    res = torch.load(f"/tmp/acc.rank{r}")
    ge = res["grad_err"]
    print(r, "n", len(ge), "max", max(ge), "sorted top", sorted(ge)[-5:], "replica_err", res["replica_err"], res["same_across_ranks"])
PY
