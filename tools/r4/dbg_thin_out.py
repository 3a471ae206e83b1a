import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn.functional as F
from sr_gan_fd_amd import _abi as A, ops
torch.manual_seed(2)
dt = torch.float16
n, h, w, cs = 1, 16, 16, 3
big = torch.randn(n, h, w, 64, device="cuda").to(dt)
big_r = big.permute(0, 3, 1, 2).float()
W = torch.randn(cs, 64, 3, 3, device="cuda") * 0.1
b = torch.zeros(cs, device="cuda")
out = torch.full((n, h, w, 4), 5.0, device="cuda")
want = F.conv2d(big_r, W.to(dt).float(), b, padding=1)
ops.thin_out(ops.thin_args(A.F16, n, h, w, cs, W, A.view(big), w_big_is_cout=False, bias=b, thin_out=out))
torch.cuda.synchronize()
got = out.permute(0, 3, 1, 2)
for c in range(4):
    print("channel", c, "got", got[0, c, 4, :6].tolist(), "want", (want[0, c, 4, :6].tolist() if c < cs else None))
    if c < cs:
        print("   err", (got[0, c] - want[0, c]).abs().max().item())
# which single-weight probes land where: weight only at (co, ci=0, ky, kx) = 1
for co in range(cs):
    for t in (0, 4, 8):
        Wp = torch.zeros(cs, 64, 3, 3, device="cuda"); Wp[co, 5, t // 3, t % 3] = 1.0
        ops.thin_out(ops.thin_args(A.F16, n, h, w, cs, Wp, A.view(big), w_big_is_cout=False, bias=b, thin_out=out))
        wantp = F.conv2d(big_r, Wp, None, padding=1)
        torch.cuda.synchronize()
        print("probe co", co, "tap", t, "per-channel err", [(out[..., c] - (wantp[0, c] if c < cs else 0)).abs().max().item() for c in range(4)])
