"""Raw timeline of persistent conv_igemm workgroups (two tiles each): SRGANFD_LIB=build_exp/libsrganfd_exp.so python tools/r3/conv_stamps2.py [cin] [cout] [mask]
Prints the mean time of every stamp relative to the workgroup's entry (wave 0), first-round workgroups only."""
import os, sys, ctypes as C, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sr_gan_fd_amd import _abi as A, ops
cin = int(sys.argv[1]) if len(sys.argv) > 1 else 64
cout = int(sys.argv[2]) if len(sys.argv) > 2 else 32
mask = len(sys.argv) > 3
L = A.lib()
L.srganfd_set_stamp_buffer.argtypes = [C.c_void_p]
n, h, w = 32, 128, 128
DT = torch.float16
x = (torch.randn(n, h, w, 192, device='cuda') * 0.5).to(DT)
y = torch.empty(n, h, w, 192, device='cuda', dtype=DT)
m = (torch.randn(n, h, w, 192, device='cuda')).to(DT)
wp = ops.pack_single(torch.randn(cout, cin, 3, 3, device='cuda') * 0.05, ops.DT[DT])
kw = dict(mask=A.view(m, c0=160, planar=1)) if mask else dict(act=A.ACT_LRELU)
a = ops.conv_args(ops.DT[DT], A.view(x, c0=0, planar=1), A.view(y, c0=0 if cout == 64 else 160, planar=1), wp, n, h, w, cin, cout, **kw)
big = torch.empty(1 << 28, dtype=torch.uint8, device='cuda')
for _ in range(3):
    big.fill_(1); ops.conv2d(a)
buf = torch.zeros(256 * 2 * 32, dtype=torch.int64, device='cuda')
big.fill_(2); torch.cuda.synchronize()
L.srganfd_set_stamp_buffer(buf.data_ptr())
ops.conv2d(a); torch.cuda.synchronize()
L.srganfd_set_stamp_buffer(None)
s = buf.cpu().view(256, 2, 32).double()
s = s[s[:, 0, 0] > 0]
nst = int((s[0, 0] > 0).sum())
t0 = s[:, :, 0].min()
first = s[:, 0, 0] - t0 < 300
ss = s[first]
rel = (ss[:, 0, :nst] - ss[:, 0, :1]) / 100.0
print(f"cin {cin} cout {cout} mask {mask}: {int(first.sum())} workgroups, {nst} stamps; mean time since entry (us), wave 0:")
print("  " + " ".join(f"{v:6.2f}" for v in rel.mean(0).tolist()))
print("  deltas: " + " ".join(f"{v:5.2f}" for v in (rel[:, 1:] - rel[:, :-1]).mean(0).tolist()))
