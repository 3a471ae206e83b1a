"""Data parallelism: one process per GPU (torchrun), RCCL over xGMI (backend "nccl" on ROCm).

The reference has no distributed code; the step is data-parallel by construction: every
loss is a mean over independent images and no layer of the benchmarked networks mixes samples (spectral-norm
u/v depend on the weights only), so equal shards + one gradient all-reduce(sum) x 1/world is exact.  The
BatchNorm discriminators (A-ESRGAN/model.py:233, ESRGAN/model.py:98-126) are the exception: per-rank batch
statistics are what DistributedDataParallel would give the unconverted reference; SyncBatchNormReduce below
is the opt-in whole-batch form (statistics identical to one process holding the full batch).
Exchange steps per iteration: D's flat gradient (17.5 MB fp32) after its second backward, G's flat
gradient (66.8 MB fp32) after its backward -- two large collectives, sized for per-link-bound xGMI
rings, instead of per-tensor buckets.  The 1/world factor is folded into the Adam kernel.
"""
from __future__ import annotations

import os
from typing import Optional, Tuple

import torch
import torch.distributed as dist


def init_from_env(backend: Optional[str] = None):
    """(rank, local_rank, world, process_group or None) from torchrun's environment."""
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1:
        return rank, local_rank, world, None
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    kw = {}
    if backend == "nccl":
        torch.cuda.set_device(local_rank)
        kw["device_id"] = torch.device("cuda", local_rank)
    if not dist.is_initialized():
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    dense_chain_needs_its_own_gpu(torch.device('cuda', local_rank) if torch.cuda.is_available() else None, dist.group.WORLD)
    return rank, local_rank, world, dist.group.WORLD


def dense_chain_needs_its_own_gpu(device, pg, identity=None) -> bool:
    """The LDS-resident dense-block launch (csrc/dense_chain.hip) needs every workgroup of a pass resident at once, one per compute
    unit: two processes that share a GPU could hold compute units each other's launches wait for.  Ranks that find another rank of
    ``pg`` on their device (same host, same PCI address) keep the per-layer launches -- unless SRGANFD_DENSE_CHAIN=1 insists.
    Called by init_from_env and by the trainers' reducers (plans are made at the first step, after this).  True: switched off.
    ``identity`` replaces (host, PCI address) in the host tests."""
    from . import ops
    if pg is None or os.environ.get("SRGANFD_DENSE_CHAIN") == "1":
        return False
    if identity is None:
        if device is None or torch.device(device).type != "cuda":
            return False
        import socket
        p = torch.cuda.get_device_properties(device)
        pci = tuple(getattr(p, k, None) for k in ("pci_domain_id", "pci_bus_id", "pci_device_id"))
        index = torch.device(device).index
        identity = (socket.gethostname(),) + (pci if pci[1] is not None else ("index", torch.cuda.current_device() if index is None else index))
    everyone = [None] * dist.get_world_size(pg)
    if torch.device(device if device is not None else "cpu").type == "cuda":
        with torch.cuda.device(device):        # an RCCL object collective stages through the CURRENT device
            dist.all_gather_object(everyone, identity, group=pg)
    else:
        dist.all_gather_object(everyone, identity, group=pg)
    if everyone.count(identity) > 1:
        ops.DENSE_CHAIN = "0"
        return True
    return False


def allreduce_sum_(flat_grad: torch.Tensor, pg) -> float:
    """In-place all-reduce(sum) of a flat gradient; returns the scale (1/world) the optimizer applies."""
    if pg is None:
        return 1.0
    dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, group=pg)
    return 1.0 / dist.get_world_size(pg)


class SyncBatchNormReduce:
    """Sums a BatchNorm partial-sum table over the ranks between the two phases of srganfd_batchnorm_{fwd,bwd}_sync
    (include/srganfd.h): (sum x, sum x^2) forward, (sum dy, sum dy*xhat) backward, 2 * 1024 * C floats each (2 MB at 256 channels).
    Every rank holds the same number of pixels (equal shards), so the whole-batch pixel count is world * npix."""

    def __init__(self, pg):
        self.pg = pg
        self.world = dist.get_world_size(pg) if pg is not None else 1

    def all_reduce(self, table: torch.Tensor) -> None:
        if self.pg is not None:
            dist.all_reduce(table, op=dist.ReduceOp.SUM, group=self.pg)


class WaitMeter:
    """Exposed communication as a number: HIP event pairs around the main stream's waits for the reducers' side streams
    (BucketReducer.finish, SideStreamReducer.wait).  The first event fires when the main stream REACHES the wait point, the second
    when it gets past it, so their distance is what the exchange cost the critical path (zero when the side stream finished first).
    Off unless a meter is installed in ``parallel.METER`` (bench.py does, for N > 1)."""

    def __init__(self):
        self.pairs = {}

    def bracket(self, name: str, wait_fn) -> None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        wait_fn()
        e1.record()
        self.pairs.setdefault(name, []).append((e0, e1))

    def report(self, steps: int) -> dict:
        """ms per step per wait point (synchronises)"""
        torch.cuda.synchronize()
        return {k: sum(a.elapsed_time(b) for a, b in v) / max(1, steps) for k, v in self.pairs.items()}


METER: Optional[WaitMeter] = None


class BucketReducer:
    """The generator's gradient exchange in buckets that follow its backward pass: the engine reports a contiguous range of the
    flat gradient as final (tail convs first, then the upper half of the trunk, then the rest -- three ranges of 2 / 32 / 33 MB for
    the 23-block RRDBNet) and the all-reduce of that range starts on a side HIP stream while the data- and weight-gradient kernels
    of the earlier layers still run; ``finish()`` makes the main stream wait for the last one before the Adam kernel.  A ring over
    xGMI is per-link bound and needs few CUs, so the exchange costs the backward pass next to nothing and only the last bucket's
    tail is exposed.  No process group (or CPU tensors: the dry-run host tests): ``bucket`` reduces inline / does nothing.

        flat_grad, _ = engine.backward(sp, token, dout, False, on_ready=reducer.bucket)
        scale = reducer.finish()          # 1 / world for the optimizer"""

    def __init__(self, device, pg):
        self.pg = pg
        self.world = dist.get_world_size(pg) if pg is not None else 1
        self.stream = torch.cuda.Stream(device=device) if pg is not None and torch.device(device).type == "cuda" else None
        self.sizes = []                   # elements per bucket of the last backward pass (tests look at it)
        dense_chain_needs_its_own_gpu(device, pg)

    def begin(self) -> None:
        self.sizes = []

    def bucket(self, flat_grad: torch.Tensor, lo: int, hi: int) -> None:
        if self.pg is None or hi <= lo:
            return
        part = flat_grad[lo:hi]
        self.sizes.append(hi - lo)
        if self.stream is None:
            dist.all_reduce(part, op=dist.ReduceOp.SUM, group=self.pg)
            return
        self.stream.wait_stream(torch.cuda.current_stream())      # the kernels that produced [lo, hi) are enqueued before this point
        flat_grad.record_stream(self.stream)
        with torch.cuda.stream(self.stream):
            dist.all_reduce(part, op=dist.ReduceOp.SUM, group=self.pg)

    def finish(self) -> float:
        if self.stream is not None:
            wait = lambda: torch.cuda.current_stream().wait_stream(self.stream)
            if METER is not None:
                METER.bracket("g_grad_exchange", wait)
            else:
                wait()
        return 1.0 / self.world


class SideStreamReducer:
    """All-reduce + optimizer step of one network on a side HIP stream, so that it overlaps whatever the main stream does next
    that does not read that network's parameters (GAN iteration: the discriminator's 17.5 MB reduce + Adam run beside the
    generator's pixel loss and the two VGG-19 forwards, train_bsrgan.py:436-452).  RCCL's ring over xGMI is per-link bound and
    needs no compute units to speak of; the point is to take the exchange off the critical path, not to shorten it.

        ev = reducer.launch(lambda: scaler.step(opt, grad, allreduce_sum_(grad, pg)))   # enqueued behind the main stream's work
        ...                                                                               # main stream: independent work
        reducer.wait()                                                                    # before the first reader of the parameters

    Single-process runs (pg is None) execute the function inline on the main stream: nothing to overlap, nothing to get wrong."""

    def __init__(self, device, pg):
        self.pg = pg
        # the dry-run host tests drive the trainers with CPU tensors over gloo: no stream there, the exchange runs inline
        self.stream = torch.cuda.Stream(device=device) if pg is not None and torch.device(device).type == "cuda" else None
        self.event = None

    def launch(self, fn, tensors=()) -> None:
        """``tensors``: buffers allocated on the main stream that ``fn`` reads or writes (the flat gradient): the caching allocator
        must not hand their memory out again before the side stream is done with it."""
        if self.stream is None:
            fn()
            return
        self.stream.wait_stream(torch.cuda.current_stream())        # the gradient is complete on the main stream first
        for t in tensors:
            t.record_stream(self.stream)
        with torch.cuda.stream(self.stream):
            fn()                                                    # collective + Adam kernels go to the side stream (A.stream_ptr())
            self.event = torch.cuda.Event()
            self.event.record(self.stream)

    def wait(self) -> None:
        if self.event is not None:
            ev = self.event
            wait = lambda: torch.cuda.current_stream().wait_event(ev)
            if METER is not None:
                METER.bracket("d_grad_exchange_and_adam", wait)
            else:
                wait()
            self.event = None


def broadcast_(flat: torch.Tensor, pg, src: int = 0) -> None:
    """Make parameters / optimizer state / spectral-norm buffers identical on every rank."""
    if pg is not None:
        dist.broadcast(flat, src=src, group=pg)


def shard(batch: torch.Tensor, rank: int, world: int) -> torch.Tensor:
    """Equal contiguous shards of a global batch (global batch must divide by world)."""
    n = batch.shape[0]
    if n % world:
        raise ValueError(f"global batch {n} is not divisible by world size {world}")
    per = n // world
    return batch[rank * per:(rank + 1) * per]


def max_over_ranks(seconds: float, pg, device) -> float:
    if pg is None:
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=pg)
    return float(t.item())
