"""One contiguous fp32 view over a module's parameters (or their gradients), when they already lie in one buffer.

The engines keep every network's parameters as views of one flat buffer (engine.FlatParams) and return a backward pass's gradients as
views of one flat gradient; the drop-in optimizer / EMA classes (optim.py, swa_utils.py) use that to run ONE kernel over the whole
network instead of one per tensor, and fall back to torch's own per-tensor code whenever the layout is anything else.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import torch
from torch import Tensor


MAX_GAP = 3        # FlatParams pads every tensor to 16 bytes: at most 3 fp32 elements between neighbours


def flat_span(tensors: Sequence[Optional[Tensor]]) -> Optional[Tuple[Tensor, List[int]]]:
    """(flat view covering every tensor, element offset of each tensor inside it) if all tensors are contiguous fp32 views of ONE
    storage on one device that do not overlap and leave nothing but alignment padding between them; else None.  (A subset of a
    network's parameters -- e.g. ``filter(lambda p: p.requires_grad, ...)`` -- has the others in its gaps: a whole-buffer kernel would
    update those too, so such a list is not a span.)"""
    if not tensors or any(t is None for t in tensors):
        return None
    t0 = tensors[0]
    if t0.dtype != torch.float32:
        return None
    st0 = t0.untyped_storage().data_ptr()
    lo, hi, ptrs = None, 0, []
    for t in tensors:
        if t.dtype != torch.float32 or t.device != t0.device or not t.is_contiguous() or t.untyped_storage().data_ptr() != st0:
            return None
        p = t.data_ptr()
        ptrs.append(p)
        lo = p if lo is None or p < lo else lo
        hi = max(hi, p + 4 * t.numel())
    order = sorted(range(len(tensors)), key=lambda i: ptrs[i])
    for a, b in zip(order[:-1], order[1:]):
        end = ptrs[a] + 4 * tensors[a].numel()
        if end > ptrs[b] or ptrs[b] - end > 4 * MAX_GAP:
            return None                       # overlapping views, or other tensors in between: not (all of) a parameter layout
    n = (hi - lo) // 4
    flat = torch.empty(0, dtype=torch.float32, device=t0.device).set_(t0.untyped_storage(), (lo - st0) // 4, (n,))
    return flat, [(p - lo) // 4 for p in ptrs]


def engine_flatten(module) -> None:
    """If ``module`` is one of this package's networks on a GPU, make its parameters views of the engine's flat buffer now (the engines
    do this at the first forward; an optimizer or EMA copy built before it would otherwise see 702 separate tensors once)."""
    from . import engine as E
    from . import model as M
    try:
        p = next(module.parameters())
    except StopIteration:
        return
    if not p.is_cuda:
        return
    eng = None
    if isinstance(module, M._RRDBGenerator):
        eng = E.generator_engine(module)
    elif isinstance(module, M.DiscriminatorUNet):
        from .engine_d import discriminator_engine
        eng = discriminator_engine(module)
    elif isinstance(module, M.UNetDiscriminatorAesrgan):
        from .engine_a import aesrgan_engine
        eng = aesrgan_engine(module)
    elif isinstance(module, M.Discriminator):
        from .engine_e import esrgan_discriminator_engine
        eng = esrgan_discriminator_engine(module)
    if eng is not None:
        eng.fp.sync(p.device)
