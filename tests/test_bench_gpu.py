"""bench.py end to end on the GPU box: the self-launch path (--gpus N without torchrun's environment starts N ranks before any
GPU call), the JSON contract of the single line rank 0 prints, and the refusal to report n_gpus it does not have."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--steps", "2", "--warmup", "1", "--batch", "2", "--lr-size", "16", "--num-rrdb", "1", "--no-cpu-baseline"]


def _run(args, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)


def _line(out):
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout + out.stderr[-2000:]
    return json.loads(lines[0])


def test_bench_self_launches_two_ranks_over_gloo():
    """--gpus 2 with no WORLD_SIZE: two child ranks (sharing the box's one GPU, gloo collectives), one line, n_gpus == 2,
    both workloads in it, weak scaling (global batch = 2 x per-GPU batch)."""
    out = _run(["--gpus", "2", "--dist-backend", "gloo"] + SMALL)
    assert out.returncode == 0, out.stderr[-3000:]
    r = _line(out)
    assert r["n_gpus"] == 2 and r["config"]["global_batch"] == 4 and r["config"]["parallelism"] == "dp2" and r["scaling"] == "weak"
    assert r["steps"] == 2 and r["warmup"] == 1 and r["unit"] == "img/s" and r["value"] > 0 and r["dtype"] == "f16"
    assert "gan" in r and r["gan"]["value"] > 0 and "roofline" in r and r["roofline"]["bound"] in ("hbm", "mfma")
    for k in ("achieved", "peak", "unit", "frac", "traffic"):
        assert k in r["roofline"]
    # exposed communication (N > 1): how long the main stream stalled at the generator's / discriminator's gradient exchange
    assert r["exposed_comm_ms_per_step"]["g_grad_exchange"] >= 0.0 and "d_grad_exchange_and_adam" in r["gan"]["exposed_comm_ms_per_step"]
    assert "distinct seeded batches" in r["data"]


def test_bench_two_ranks_over_rccl_through_the_driver_command():
    """The multi-GPU lease's exact command -- ``python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1
    --master-port P bench.py --gpus 2 ...`` with the default backend (nccl = RCCL over xGMI) -- on two real GPUs: one line, n_gpus 2,
    weak scaling, and the exposed-communication object (main-stream stall at the two gradient exchanges).  Skipped on boxes with fewer
    than two GPUs (the one-GPU test boxes): it runs the first time a multi-GPU node executes this suite."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (RCCL with one rank per GPU)")
    from tests.util import free_port
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", "2"] + SMALL
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    r = _line(out)
    assert r["n_gpus"] == 2 and r["config"]["global_batch"] == 4 and r["config"]["parallelism"] == "dp2" and r["scaling"] == "weak"
    assert r["value"] > 0 and r["gan"]["value"] > 0
    for obj in (r, r["gan"]):
        e = obj["exposed_comm_ms_per_step"]
        assert e["g_grad_exchange"] >= 0.0 and e["d_grad_exchange_and_adam"] >= 0.0


def test_bench_refuses_more_gpus_than_visible():
    """RCCL ranks need a GPU each: asking for more than the box has must fail loudly, not fall back to one rank."""
    import torch
    n = torch.cuda.device_count()
    out = _run(["--gpus", str(n + 1)] + SMALL)
    assert out.returncode != 0 and not [l for l in out.stdout.splitlines() if l.startswith("{")]


def test_bench_rejects_mismatched_world_size():
    out = _run(["--gpus", "4"] + SMALL, {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert out.returncode != 0


def test_bench_single_gpu_line():
    out = _run(["--workload", "g_only", "--dtype", "bf16"] + SMALL)
    assert out.returncode == 0, out.stderr[-3000:]
    r = _line(out)
    assert r["n_gpus"] == 1 and r["dtype"] == "bf16" and "gan" not in r and r["config"]["workload"].startswith("BSRGAN RRDBNet x4 generator-only")


def test_bench_default_line_carries_the_module_level_legs():
    """the default run times the reference's loop statements over the drop-in modules beside the fused trainers"""
    out = _run(SMALL)
    assert out.returncode == 0, out.stderr[-3000:]
    r = _line(out)
    ml = r["extra"]["module_loop"]
    for wl in ("g_only", "gan"):
        assert ml[wl]["ms_per_step"] > 0 and ml[wl]["fused_ms_per_step"] > 0 and ml[wl]["module_over_fused"] > 0
        assert all(v == v for v in ml[wl]["last_step_scalars"])
    assert ml["gan"]["loss_scale"]["enabled"] and ml["gan"]["loss_scale"]["scale"] > 0
    # first-class fields of the line: the module-level step times, the whole-step MFMA fraction and the largest
    # single symbol beside the dominant class, the bf16 leg BASELINE.json's metric string names
    for wl in ("g_only", "gan"):
        m = r["module_loop"][wl]
        assert m["dropin_3_imports_ms_per_step"] > 0 and m["model_import_only_ms_per_step"] > 0 and m["fused_trainer_ms_per_step"] > 0
    for obj in (r, r["gan"]):
        rf = obj["roofline"]
        assert 0 < rf["step_mfma_frac"] < 1 and rf["largest_symbol"]["avg_us"] > 0 and 0 <= rf["largest_symbol"]["mfma_frac"] < 1
    assert r["bf16"]["ms_per_step"] > 0 and 0 < r["bf16"]["step_mfma_frac"] < 1 and all(v == v for v in r["bf16"]["last_step_scalars"])
    out = _run(["--workload", "gan", "--module-loop"] + SMALL)
    assert out.returncode == 0, out.stderr[-3000:]
    assert "module-level" in _line(out)["config"]["loop"]
