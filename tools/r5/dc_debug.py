"""EXPERIMENT: where does the dense-chain launch differ from the five launches?  per-channel / per-pixel error map of the block buffer"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sr_gan_fd_amd import ops
from tests.test_dense_chain_gpu import _build
n, h, w = (int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (1, 16, 16)
a_ref, (buf_ref, out_ref), k1 = _build(torch.float16, n, h, w, 1, False)
a_dc, (buf_dc, out_dc), k2 = _build(torch.float16, n, h, w, 1, False)
for a in a_ref:
    ops.conv2d(a)
ch = ops.DenseChain(a_dc, "cuda"); ch.run(); torch.cuda.synchronize()
br, bd = buf_ref.view(n, 6, h, w, 32).float(), buf_dc.view(n, 6, h, w, 32).float()
e = (br - bd).abs()
print("max err per group:", [round(e[:, g].max().item(), 4) for g in range(6)])
g = 3
print("group 3, err per row:", [round(v, 3) for v in e[0, g].amax(dim=(1, 2)).tolist()])
print("group 3, err per col:", [round(v, 3) for v in e[0, g].amax(dim=(0, 2)).tolist()])
print("group 3, err per channel:", [round(v, 3) for v in e[0, g].amax(dim=(0, 1)).tolist()])
print("group 3, pixel (3,5): ref", [round(v, 3) for v in br[0, g, 3, 5, :16].tolist()])
print("group 3, pixel (3,5): dc ", [round(v, 3) for v in bd[0, g, 3, 5, :16].tolist()])
print("group 3, pixel (3,5): ref", [round(v, 3) for v in br[0, g, 3, 5, 16:].tolist()])
print("group 3, pixel (3,5): dc ", [round(v, 3) for v in bd[0, g, 3, 5, 16:].tolist()])
print("out err", (out_ref.float() - out_dc.float()).abs().max().item())
