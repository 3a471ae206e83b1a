import sys, torch
sys.path.insert(0, '.')
from sr_gan_fd_amd import _abi as A, ops
for dtype in (torch.float32, torch.bfloat16):
    dt = ops.DT[dtype]
    n, h, w, cin, cout = 1, 16, 16, 64, 32
    x = torch.zeros(n, h, w, cin, dtype=dtype, device='cuda')
    wt = torch.zeros(cout, cin, 3, 3).cuda()
    wp = ops.pack_single(wt, dt)
    r1 = (torch.arange(n*h*w).view(n, h, w, 1) + torch.arange(cout).view(1, 1, 1, cout) / 64.0).to(dtype).cuda().contiguous()
    y = torch.full((n, h, w, cout), -1.0, dtype=dtype, device='cuda')
    a = ops.conv_args(dt, A.view(x), A.view(y), wp, n, h, w, cin, cout, r1=A.view(r1), r1_scale=1.0)
    ops.conv2d(a); torch.cuda.synchronize()
    d = (y.float() - r1.float()).abs()
    print(dtype, 'max diff', d.max().item())
    if d.max() > 0:
        bad = (d > 0).nonzero()
        print(bad[:10].tolist(), y[0,0,0,:8].tolist(), r1[0,0,0,:8].tolist(), y[0,0,1,:8].tolist())
