import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn.functional as F
from sr_gan_fd_amd import _abi as A, ops
torch.manual_seed(2)
dt = torch.float16
n, h, w, cs = 1, 16, 16, 1
small = torch.randn(n, cs, h, w, device="cuda")
thin = torch.zeros(n, h, w, 4, dtype=dt, device="cuda"); thin[..., :cs] = small.permute(0, 2, 3, 1).to(dt)
small_r = thin[..., :cs].permute(0, 3, 1, 2).float()
W = torch.zeros(64, cs, 3, 3, device="cuda")
for c in range(64): W[c, 0, 1, 1] = c + 1          # output channel c = (c + 1) * x
y = torch.zeros(n, h, w, 64, dtype=dt, device="cuda")
bias = torch.full((64,), 1000.0, device="cuda") if os.environ.get("DBG_BIAS") else None
ops.thin_in(ops.thin_args(A.F16, n, h, w, cs, W, A.view(y), w_big_is_cout=True, thin=thin, bias=bias))
if bias is not None: y = (y.float() - 1000.0)
torch.cuda.synchronize()
ratio = (y.float() / thin[..., :1].float())
print("pixel (3,0..15) channel 0 ratio:", ratio[0, 3, :, 0].tolist())
print("pixel (3,5) channels ratio:", [round(v, 2) for v in ratio[0, 3, 5, :].tolist()])
print("pixel (3,12) channels ratio:", [round(v, 2) for v in ratio[0, 3, 12, :].tolist()])
