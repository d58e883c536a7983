"""The REAL model under data parallelism before anyone hands it 8 GPUs: two ranks share cuda:0 (gloo backend, so no second
GPU is needed) and run tests/ddp_real_worker.py; plus bench.py's own multi-rank launch (train_net_multi.py:76-78,
engine/launch.py:55-73)."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _env():
    return dict(os.environ, SW_DIST_BACKEND="gloo", SW_BENCH_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=ROOT)


def test_real_model_two_ranks_update_after_the_backward(tmp_path):
    """SW_DDP_OVERLAP_UPDATE=0: the reference's order (all-reduce everything, then one optimizer step) gives the same parameters"""
    out = str(tmp_path / "ddpseq")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "ddp_real_worker.py"), out, "bf16"]
    r = subprocess.run(cmd, env=dict(_env(), SW_DDP_OVERLAP_UPDATE="0", SW_DDP_NATIVE="0"), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-6000:]
    for rank in range(2):
        res = torch.load(f"{out}.rank{rank}")
        assert not res["overlap_update"] and all(n > 0 for n in res["left_for_step"])
        assert max(res["grad_err"]) <= 1e-6 and res["same_across_ranks"] and res["replica_err"] <= 1e-6


@pytest.mark.parametrize("mode", ["native", "native-graph", "torch-ddp"])
@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_real_model_two_ranks_gradients_and_parameters(tmp_path, dtype, mode):
    """native: trainer._NativeDDP (own flat buckets, the backward in four stages, all-reduces between them, per-bucket updates on a
    side stream); native-graph: the same with every stage replayed as a captured hipGraph; torch-ddp: DistributedDataParallel with
    the per-bucket update in its communication hook (SW_DDP_NATIVE=0).  All three: reduced gradients == the mean of the ranks'
    single-process gradients, parameters bit-identical across ranks and equal to the single-process replica."""
    out = str(tmp_path / "ddp")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "ddp_real_worker.py"), out, dtype]
    extra = {"native": {}, "native-graph": {"SW_STEP_GRAPH": "1"}, "torch-ddp": {"SW_DDP_NATIVE": "0"}}[mode]
    r = subprocess.run(cmd, env=dict(_env(), **extra), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-6000:]
    for rank in range(2):
        res = torch.load(f"{out}.rank{rank}")
        assert res["native"] == (mode != "torch-ddp")
        if mode == "native-graph":
            assert res["replays"] >= 2, res["replays"]              # step 0 eager, step 1 captures and replays, step 2 replays
        # DDP divides by the world size before the all-reduce, the replica after the sum: equal up to the f32 rounding of /2,
        # i.e. exactly, except for denormals; bf16 activations do not enter (both sides run the same kernels)
        assert max(res["grad_err"]) <= 1e-6, max(res["grad_err"])
        assert res["same_across_ranks"]
        assert res["replica_err"] <= 1e-6, res["replica_err"]
        assert res["moved"] > 0
        assert res["dropout_seeds_differ"]
        assert len(res["metrics"]) == 9 and all(v == v for v in res["metrics"].values())
        # the update ran bucket by bucket inside DDP's communication hook: nothing was left for the step() after the backward
        assert res["overlap_update"] and set(res["left_for_step"]) == {0}, res["left_for_step"]


@pytest.mark.parametrize("mode", ["native", "native-graph", "torch-ddp"])
def test_real_model_two_ranks_gradient_accumulation(tmp_path, mode):
    """ITER_SIZE = 2 (what auto_scale_workers sets when fewer GPUs than the recipe's run it, train_net_multi.py:306-324): the optimizer
    steps at iterations 0, 2, 4 on the sum of the micro-steps' rank-mean gradients.  trainer._NativeDDP accumulates in a second flat
    buffer per bucket and all-reduces only at the stepping iterations (round 6; the reference and torch DDP all-reduce every micro-step):
    reduced gradients at every stepping iteration == the single-process replica's accumulated mean, parameters identical across the
    ranks and equal to the replica's after 5 iterations — eager stages, stage graphs, and the torch-DDP path."""
    out = str(tmp_path / "ddpacc")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "ddp_real_worker.py"), out, "bf16"]
    extra = {"native": {}, "native-graph": {"SW_STEP_GRAPH": "1"}, "torch-ddp": {"SW_DDP_NATIVE": "0"}}[mode]
    r = subprocess.run(cmd, env=dict(_env(), SW_TEST_ITER_SIZE="2", **extra), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-6000:]
    for rank in range(2):
        res = torch.load(f"{out}.rank{rank}")
        assert res["native"] == (mode != "torch-ddp")
        if mode == "native-graph":
            assert res["replays"] >= 2, res["replays"]
        ge = res["grad_err"]
        assert len(ge) > 0 and len(ge) % 3 == 0
        per = len(ge) // 3                                             # stepping iterations 0, 2, 4
        # iteration 0 (one micro-step) is the ITER_SIZE 1 arithmetic; iteration 2 sums two micro-steps in another order than the replica
        # (mean of sums vs sum of means): f32 rounding.  From there the two weight sets differ in last bits, and bf16 activations turn a
        # last-bit weight difference into a visible gradient difference two iterations later (measured 1.6e-3 at iteration 4)
        assert max(ge[:per]) <= 1e-6 and max(ge[per:2 * per]) <= 2e-6, (max(ge[:per]), max(ge[per:2 * per]))
        assert max(ge[2 * per:]) <= 1e-2, max(ge[2 * per:])
        assert res["same_across_ranks"]
        assert res["replica_err"] <= 2e-4, res["replica_err"]
        assert res["moved"] > 0


@pytest.mark.parametrize("panels", [0, 4])
def test_real_model_two_ranks_at_config3_per_gpu_size(tmp_path, panels):
    """the same check once at BASELINE config #3's per-rank shape (512x512 views, R = 2000, fc 4096/4096, bf16): 543 MB of
    gradients through the real bucket layout (fc6 a 411 MB bucket of its own).  panels = 4: SW_DDP_FC1_PANELS — fc1.weight's gradient
    is computed and all-reduced in four row panels (each leaves while the next is still in its GEMM); same gradients, same parameters"""
    out = str(tmp_path / "ddpfull")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "ddp_real_worker.py"), out, "bf16", "full"]
    r = subprocess.run(cmd, env=dict(_env(), SW_DDP_FC1_PANELS=str(panels)), capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-6000:]
    for rank in range(2):
        res = torch.load(f"{out}.rank{rank}")
        assert max(res["grad_err"]) <= 1e-6, max(res["grad_err"])
        assert res["same_across_ranks"] and res["replica_err"] <= 1e-6 and res["moved"] > 0


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it prints one line with n_gpus 2 (both ranks on cuda:0 over gloo here)"""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"]
    env = _env()
    env.pop("WORLD_SIZE", None); env.pop("RANK", None)
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-6000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    rec = json.loads(line)
    assert rec["n_gpus"] == 2 and rec["steps"] == 2 and rec["value"] > 0 and rec["config"]["parallelism"] == "dp2"
    # round 6: N > 1 times the stage-graph mode, gives the launch-by-launch time beside it, and proves the replicas stayed identical
    assert rec["ranks_in_sync"] is True and rec["rccl_ranks"] == 2
    assert rec["step_launch"]["mode"].startswith("data parallel: 4 stage graphs") and rec["step_launch"]["graph_replays"] > 0
    assert rec["step_launch"]["eager_ms_per_step"] > 0


def test_stage3_semisup_step_two_ranks(tmp_path):
    """the Unbiased-Teacher iteration with the student detector inside DistributedDataParallel (two student forward passes, one
    backward): all-reduced gradients == the mean of the ranks' single-process gradients, student and EMA teacher identical across ranks"""
    out = str(tmp_path / "s3ddp")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "ddp_stage3_worker.py"), out]
    r = subprocess.run(cmd, env=_env(), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-6000:]
    for rank in range(2):
        res = torch.load(f"{out}.rank{rank}")
        assert res["n_grads"] >= 60 and max(res["grad_err"]) <= 1e-6, max(res["grad_err"])
        assert res["same_across_ranks"] and res["finite"]
        assert any(k.endswith("_pseudo") for k in res["losses"])


@pytest.mark.parametrize("mode", ["native", "native-graph", "torch-ddp"])
@pytest.mark.parametrize("dtype", ["bf16", "fp32"])
def test_ddp_trainer_over_rccl_world_size_one(tmp_path, dtype, mode):
    """the multi-GPU code path on the real RCCL backend (ProcessGroupNCCL streams / futures, DDP reducer, the per-bucket update hook,
    the metrics all-reduce) with the one GPU a test box has: bit-identical parameters to the plain Trainer"""
    out = str(tmp_path / "rccl1.pt")
    cmd = [sys.executable, os.path.join(ROOT, "tests", "ddp_rccl_single_worker.py"), out, dtype, str(_free_port())]
    if mode == "native-graph":
        cmd.append("graph")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=ROOT, SW_DDP_NATIVE="0" if mode == "torch-ddp" else "1")
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-6000:]
    res = torch.load(out)
    assert res["backend"] == "nccl" and res["same"] and res["moved"] > 0
    assert res["native"] == (mode != "torch-ddp")
    if mode == "native-graph":
        assert res["replays"] >= 3, res["replays"]                # RCCL all-reduces between the replayed stage graphs
    assert res["overlap_update"] and res["left_for_step"] == [0] * res["n_steps"], res["left_for_step"]
    assert res["stale_staged"] == [], res["stale_staged"]        # every bucket's update stamped its weight copies; none re-staged
    assert res["metrics"] == res["metrics_ref"] and len(res["metrics"]) == 9
