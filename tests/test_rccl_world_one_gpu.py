"""The data-parallel exchange through REAL RCCL on the one GPU a test box has: a one-rank "nccl" process group.  The sum over one
rank is the value and 1/world is 1, so every step must equal the step without a process group to the bit; what is exercised is
everything the stand-in tests (test_gan_gpu.py) replace: RCCL's own stream and events behind torch.distributed, the side-stream
all-reduce of three generator buckets while the backward pass runs, the discriminator's all-reduce + Adam on its side stream, the
caching allocator's record_stream bookkeeping, and the SyncBatchNorm table exchange.  Runs in a child process (the process group
must not outlive the test)."""
import os
import subprocess
import sys
import textwrap

import pytest

from tests.util import free_port

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, %(root)r)
    import numpy as np, torch, torch.distributed as dist
    from tests.util import scaled_init
    from sr_gan_fd_amd import model as M
    from sr_gan_fd_amd.gan import GanTrainer
    from sr_gan_fd_amd.trainer import GeneratorTrainer

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    pg = dist.group.WORLD
    one = torch.ones(1, device=dev); dist.all_reduce(one); assert one.item() == 1.0

    def nets(dtype):
        torch.manual_seed(0)
        d = M.discriminator_unet(in_channels=3, out_channels=1, channels=64)
        g = M.bsrgan_x4(in_channels=3, out_channels=3, channels=64, growth_channels=32, num_rrdb=2)
        scaled_init(g, 3.0, 0.5)
        d.compute_dtype = g.compute_dtype = dtype
        return g.cuda().train(), d.cuda().train()

    torch.manual_seed(21)
    batches = [(torch.rand(2, 3, 16, 16).cuda(), torch.rand(2, 3, 64, 64).cuda()) for _ in range(3)]
    for dtype in (torch.float32, torch.float16):
        runs = []
        for group in (None, pg):
            g, d = nets(dtype)
            tr = GanTrainer(g, d, None, process_group=group)
            assert (tr.g_reducer.stream is not None) == (group is not None)
            sc = [tr.step(x, y).cpu().numpy().copy() for x, y in batches]
            torch.cuda.synchronize()
            runs.append((sc, tr.g_opt.flat.clone(), tr.d_opt.flat.clone(), tr.g_opt.ema.clone()))
            if group is not None:
                assert len(tr.g_reducer.sizes) == 3 and sum(tr.g_reducer.sizes) == tr.g_opt.flat.numel()
        assert all(np.array_equal(a, b) for a, b in zip(runs[0][0], runs[1][0])), (dtype, runs[0][0], runs[1][0])
        assert all(torch.equal(a, b) for a, b in zip(runs[0][1:], runs[1][1:])), dtype
        assert np.isfinite(np.stack(runs[0][0])).all()
        runs = []
        for group in (None, pg):
            g, _ = nets(dtype)
            tr = GeneratorTrainer(g, lr=1e-4, process_group=group)
            losses = [tr.step(x, y).item() for x, y in batches]
            torch.cuda.synchronize()
            runs.append((losses, tr.flat.clone(), tr.opt.ema.clone()))
        assert runs[0][0] == runs[1][0] and torch.equal(runs[0][1], runs[1][1]) and torch.equal(runs[0][2], runs[1][2]), dtype
    dist.destroy_process_group()
    print("RCCL-WORLD-ONE-OK")
""")


def test_one_rank_rccl_group_leaves_every_step_bit_identical():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "RCCL-WORLD-ONE-OK" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]
