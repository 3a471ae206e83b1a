"""world_size-2 gloo tests (CPU) of the data-parallel glue used by bench.py / the trainers for N > 1."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)          # 2-8 ranks x the host's default intra-op threads would oversubscribe its cores
    import torch.distributed as dist
    from sr_gan_fd_amd import parallel as P
    r, lr, w, pg = P.init_from_env("gloo")
    assert (r, w) == (rank, world) and pg is not None
    # 1. flat-gradient exchange: sum over ranks, 1/world returned for the optimizer
    torch.manual_seed(100 + rank)
    g = torch.randn(1003)
    mine = g.clone()
    scale = P.allreduce_sum_(g, pg)
    torch.manual_seed(100)
    a = torch.randn(1003)
    torch.manual_seed(101)
    b = torch.randn(1003)
    ok1 = scale == 0.5 and torch.allclose(g * scale, (a + b) / 2, atol=1e-6)
    # 2. parameters identical on every rank after the broadcast
    p = torch.full((17,), float(rank + 1))
    P.broadcast_(p, pg)
    ok2 = bool(torch.all(p == 1.0))
    # 3. shards tile the global batch exactly
    batch = torch.arange(8 * 3).view(8, 3)
    sh = P.shard(batch, rank, world)
    gathered = [torch.empty_like(sh) for _ in range(world)]
    dist.all_gather(gathered, sh, group=pg)
    ok3 = torch.equal(torch.cat(gathered), batch)
    # 4. data-parallel mean-loss gradient == single-process gradient on the concatenated batch
    torch.manual_seed(7)
    wgt = torch.randn(5, requires_grad=True)
    x = torch.randn(8, 5)
    full = (x @ wgt).abs().mean()
    gfull, = torch.autograd.grad(full, wgt)
    local = (P.shard(x, rank, world) @ wgt).abs().mean()
    gl, = torch.autograd.grad(local, wgt)
    gl = gl.clone()
    sc = P.allreduce_sum_(gl, pg)
    ok4 = torch.allclose(gl * sc, gfull, atol=1e-6)
    t = P.max_over_ranks(float(rank + 1), pg, "cpu")
    ok5 = t == float(world)
    # 6. ranks that share a GPU keep the per-layer launches (the dense-block launch needs a GPU's compute units to itself)
    from sr_gan_fd_amd import ops
    before = ops.DENSE_CHAIN
    apart = P.dense_chain_needs_its_own_gpu(None, pg, identity=("host", 0, rank, 0))
    ok6 = not apart and ops.DENSE_CHAIN == before and not P.dense_chain_needs_its_own_gpu("cpu", pg)
    shared = P.dense_chain_needs_its_own_gpu(None, pg, identity=("host", 0, 5, 0))
    ok6 = ok6 and shared and ops.DENSE_CHAIN == "0" and not ops.dense_chain_wanted(1, 16, 16)
    q.put((rank, ok1, ok2, ok3, ok4, ok5, ok6))
    dist.destroy_process_group()


def test_two_rank_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]      # generous: on a cold page cache two spawned ranks importing torch take minutes
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in res:
        assert all(r[1:]), f"rank {r[0]} failed: {r}"


def test_shard_rejects_uneven_batch():
    from sr_gan_fd_amd import parallel as P
    with pytest.raises(ValueError):
        P.shard(torch.zeros(5, 3), 0, 2)


def _trainer_worker(rank, world, port, q):
    """The fused trainers' data-parallel exchange on two gloo ranks, with the HIP library in dry-run mode (no kernels run: the
    values are meaningless, what is checked is that both ranks issue the same collectives in the same order on tensors of
    the same size, once per network per iteration, and that the 1/world scale reaches the optimizer)."""
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)          # 2-8 ranks x the host's default intra-op threads would oversubscribe its cores
    import torch.distributed as dist
    from sr_gan_fd_amd import _abi as A, model as M, parallel as P
    from sr_gan_fd_amd.gan import GanTrainer
    from sr_gan_fd_amd.trainer import GeneratorTrainer
    A.set_dry_run(True)
    r, lr_, w, pg = P.init_from_env("gloo")
    calls = []
    orig = dist.all_reduce

    def spy(t, *a, **k):
        calls.append(t.numel())
        return orig(t, *a, **k)
    dist.all_reduce = spy
    torch.manual_seed(0)
    g = M.bsrgan_x4(num_rrdb=1)
    tr = GeneratorTrainer(g, lr=1e-4, process_group=pg)
    lr_img, gt = torch.rand(2, 3, 16, 16), torch.rand(2, 3, 64, 64)
    tr.step(lr_img, gt)
    # the generator's gradient goes out in contiguous buckets that follow its backward pass (tail first); they cover the buffer once
    ok1 = sum(calls) == tr.flat.numel() and calls == tr.g_reducer.sizes and 1 <= len(calls) <= 3
    calls.clear()
    gan = GanTrainer(M.bsrgan_x4(num_rrdb=1), M.discriminator_unet(in_channels=3, out_channels=1, channels=64), None, process_group=pg)
    gan.step(lr_img, gt)
    ok2 = calls[0] == gan.de.fp.total and sum(calls[1:]) == gan.ge.fp.total and calls[1:] == gan.g_reducer.sizes   # D after its second backward, then G's buckets (gan.py)
    dist.barrier()
    q.put((rank, ok1, ok2, calls))
    dist.destroy_process_group()


def test_trainers_exchange_one_flat_gradient_per_network():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_trainer_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = []
    import queue, time
    t0 = time.time()
    while len(res) < len(procs) and time.time() - t0 < 900:      # generous: this test failed once in a 492 s suite run that shared the host with a compile job (85 s alone); cause not captured, 300 s was the limit then
        try:
            res.append(q.get(timeout=2))
        except queue.Empty:
            assert all(p.exitcode in (None, 0) for p in procs), "a rank died: " + str([p.exitcode for p in procs])
    assert len(res) == len(procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in res:
        assert r[1] and r[2], f"rank {r[0]}: {r}"


def _world8_worker(rank, world, port, q):
    """BASELINE configs[3] (global batch 256 over 8 ranks, full GAN) and configs[4] (A-ESRGAN, SyncBatchNorm) as a gloo dry run: the
    REAL networks (23 RRDB: 16,697,987 generator / 4,376,897 discriminator parameters) on tiny images, HIP library in dry-run mode.
    Checked per rank and across ranks: the global batch partitions into 8 x 32 with no image lost or doubled, every rank issues the SAME
    sequence of collectives (sizes and order), the buckets of a network cover its flat gradient exactly once, the discriminator's
    exchange comes before the generator's, the SyncBatchNorm table reduce sits inside the discriminator passes, 1/8 reaches Adam."""
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)          # 2-8 ranks x the host's default intra-op threads would oversubscribe its cores
    import torch.distributed as dist
    from sr_gan_fd_amd import _abi as A, model as M, parallel as P
    from sr_gan_fd_amd.gan import GanTrainer
    A.set_dry_run(True)
    r, _, w, pg = P.init_from_env("gloo")
    # configs[3]'s partition: global batch 256 -> 8 contiguous shards of 32
    ids = torch.arange(256)
    mine = P.shard(ids, rank, world)
    got = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(got, mine, group=pg)
    ok_part = mine.numel() == 32 and torch.equal(torch.cat(got), ids)
    calls = []
    orig = dist.all_reduce

    def spy(t, *a, **k):
        calls.append(int(t.numel()))
        return orig(t, *a, **k)
    dist.all_reduce = spy
    torch.manual_seed(0)
    g = M.bsrgan_x4(in_channels=3, out_channels=3, channels=64, growth_channels=32, num_rrdb=23)
    d = M.discriminator_unet(in_channels=3, out_channels=1, channels=64)
    n_g, n_d = sum(p.numel() for p in g.parameters()), sum(p.numel() for p in d.parameters())
    tr = GanTrainer(g, d, None, process_group=pg)
    lr_img, gt = torch.rand(2, 3, 16, 16), torch.rand(2, 3, 64, 64)
    scales = []
    for opt in (tr.g_opt, tr.d_opt):
        step0 = opt.step
        opt.step = (lambda grad, grad_scale=1.0, *a, _s=step0, **k: (scales.append(grad_scale), _s(grad, grad_scale, *a, **k))[1])
    tr.step(lr_img, gt)
    gan_calls = list(calls)
    ok_sizes = (n_g, n_d) == (16697987, 4376897)
    ok_gan = (gan_calls[0] == tr.de.fp.total and gan_calls[1:] == tr.g_reducer.sizes and sum(gan_calls[1:]) == tr.ge.fp.total
              and len(gan_calls) == 4 and tr.ge.fp.total >= n_g and tr.de.fp.total >= n_d)
    ok_scale = len(scales) == 2 and all(abs(s - 1.0 / world) < 1e-12 for s in scales)
    # configs[4]: the attention U-Net discriminator with whole-batch BatchNorm statistics (parallel.SyncBatchNormReduce)
    calls.clear()
    torch.manual_seed(0)
    g2 = M.bsrgan_x4(num_rrdb=1)
    d2 = M.uNetDiscriminatorAesrgan()
    tr2 = GanTrainer(g2, d2, None, g_lr=5e-5, d_lr=1e-5, pixel_weight=10.0, adversarial_weight=0.1, process_group=pg, sync_batchnorm=True)
    tr2.step(torch.rand(2, 3, 32, 32), torch.rand(2, 3, 128, 128))
    aes_calls = list(calls)
    n_bn = sum(1 for c in aes_calls if c not in (tr2.de.fp.total,) and c not in tr2.g_reducer.sizes)
    ok_aes = tr2.de.fp.total in aes_calls and sum(tr2.g_reducer.sizes) == tr2.ge.fp.total and n_bn > 0
    # every rank must have issued the same collectives in the same order: compare with rank 0's record
    rec = torch.tensor(gan_calls + [-1] + aes_calls, dtype=torch.int64)
    n = torch.tensor([rec.numel()])
    dist.broadcast(n, src=0, group=pg)
    ref = rec.clone() if rank == 0 else torch.zeros(int(n.item()), dtype=torch.int64)
    dist.broadcast(ref, src=0, group=pg)
    ok_same = rec.numel() == ref.numel() and torch.equal(rec, ref)
    dist.barrier()
    q.put((rank, ok_part, ok_sizes, ok_gan, ok_scale, ok_aes, ok_same, gan_calls, len(aes_calls)))
    dist.destroy_process_group()


def test_world8_dry_run_of_the_configs3_and_configs4_partition():
    """no 8-GPU node has been available in any round, so the N = 8 path is rehearsed here on the CPU (gloo, dry run)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    world = 8
    procs = [ctx.Process(target=_world8_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = []
    import queue, time
    t0 = time.time()
    while len(res) < len(procs) and time.time() - t0 < 1200:
        try:
            res.append(q.get(timeout=2))
        except queue.Empty:
            assert all(p.exitcode in (None, 0) for p in procs), "a rank died: " + str([p.exitcode for p in procs])
    assert len(res) == len(procs)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for r in res:
        assert all(r[1:7]), f"rank {r[0]}: partition/sizes/gan/scale/aesrgan/same-order = {r[1:7]}, GAN collectives {r[7]}, A-ESRGAN collectives {r[8]}"
