export SW_DIST_BACKEND=gloo SW_BENCH_DEVICE=0 HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=$PWD SW_DDP_FC1_PANELS=4
python -m torch.distributed.run --nnodes=1 --nproc-per-node=2 --master-addr 127.0.0.1 --master-port 29517 tests/ddp_real_worker.py /tmp/ddpp bf16 full 2>&1 | grep -v "Gloo\|amdgpu" | grep -B2 -A25 "rank1\]\|Error\|error\|abort\|terminate" | head -80
