"""Calibration of the L2-side request counters on this kernel's own access pattern (MI355X_MICROARCH.md, HBM section:
FETCH_SIZE / TCC_EA0_RDREQ x 64 B is exact for some widths and half for wide streaming reads on gfx950).
Runs the 3x3 conv once per configuration on inputs larger than the Infinity Cache, after streaming 1 GiB through the
caches, so every input byte must come from HBM exactly once (plus halo re-reads that L2 should absorb).
    rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum -- python tools/pmc_calibrate.py ; python tools/pmc_summary.py <dir>"""
import sys, torch
sys.path.insert(0, '.')
from sr_gan_fd_amd import _abi as A, ops
N, H, W = 64, 128, 128
flush = torch.empty(1 << 28, device='cuda', dtype=torch.float32)      # 1 GiB
for cin, xC in ((128, 192), (128, 128), (64, 64)):
    x = torch.randn(N, H, W, xC, device='cuda').bfloat16()
    y = torch.empty(N, H, W, 32, device='cuda', dtype=torch.bfloat16)
    wp = ops.pack_single(torch.randn(32, cin, 3, 3, device='cuda') * 0.05, A.BF16)
    a = ops.conv_args(A.BF16, A.view(x), A.view(y), wp, N, H, W, cin, 32)
    flush.fill_(1.0)
    torch.cuda.synchronize()
    ops.conv2d(a)
    torch.cuda.synchronize()
    print(f"cin={cin} of a {xC}-channel buffer: algorithmic read {N*H*W*cin*2/1e6:.1f} MB, write {N*H*W*32*2/1e6:.1f} MB")
