"""Timeline of conv_igemm tiles from in-kernel s_memtime stamps (SRGANFD_EXPERIMENT build):
    SRGANFD_LIB=build_exp/libsrganfd_exp.so python tools/r3/conv_stamps.py [cin] [cout]
Stamps (wave 0 and the last wave of 256 sampled workgroups): 0 entry, 1 prologue done, then per chunk c: 2+4c barrier passed (everyone's
previous MFMA phase over), 3+4c own loads arrived + written to LDS, 4+4c stage published, 5+4c MFMA phase done; then stores issued,
stores acknowledged.  s_memrealtime: 100 MHz constant clock, printed in microseconds."""
import os, sys, ctypes as C, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sr_gan_fd_amd import _abi as A, ops
cin = int(sys.argv[1]) if len(sys.argv) > 1 else 96
cout = int(sys.argv[2]) if len(sys.argv) > 2 else 32
L = A.lib()
L.srganfd_set_stamp_buffer.argtypes = [C.c_void_p]
n, h, w = 32, 128, 128
DT = torch.float16
x = (torch.randn(n, h, w, 192, device='cuda') * 0.5).to(DT)
y = torch.empty(n, h, w, 192, device='cuda', dtype=DT)
wp = ops.pack_single(torch.randn(cout, cin, 3, 3, device='cuda') * 0.05, ops.DT[DT])
a = ops.conv_args(ops.DT[DT], A.view(x, c0=0, planar=1), A.view(y, c0=0 if cout == 64 else 160, planar=1), wp, n, h, w, cin, cout, act=A.ACT_LRELU)
big = torch.empty(1 << 28, dtype=torch.uint8, device='cuda')
for _ in range(3):
    big.fill_(1)       # evict the caches: the step runs these kernels on cold inputs
    ops.conv2d(a)
buf = torch.zeros(256 * 2 * 32, dtype=torch.int64, device='cuda')
big.fill_(2)
torch.cuda.synchronize()
L.srganfd_set_stamp_buffer(buf.data_ptr())
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); ops.conv2d(a); e1.record()
torch.cuda.synchronize()
L.srganfd_set_stamp_buffer(None)
print(f"kernel {e0.elapsed_time(e1) * 1e3:.1f} us (with stamps); cin {cin} cout {cout}, {cin // 32} chunks")
s = buf.cpu().view(256, 2, 32).double()
nst = int((s[0, 0] > 0).sum())
t0 = s[:, :, 0].min()
names = ["entry", "prologue"] + sum([[f"c{c} barrier", f"c{c} committed", f"c{c} published", f"c{c} mfma done"] for c in range(cin // 32)], []) + ["stores issued", "stores acked"]
# workgroups of the first dispatch round start near t0; the second round starts when a slot frees up
first = s[:, 0, 0] - t0 < 300          # ticks of 10 ns
for label, sel in (("first-round workgroups", first), ("later workgroups", ~first)):
    if sel.sum() == 0:
        continue
    ss = s[sel]
    print(f"{label}: {int(sel.sum())} sampled; entry at {((ss[:, 0, 0] - t0).mean() / 100):.2f} us after the first")
    prev = ss[:, :, 0]
    for i in range(1, nst):
        d = (ss[:, :, i] - prev) / 100.0     # us
        print(f"   {names[i]:16s} +{d[:, 0].mean():6.2f} us (wave 0; last wave +{d[:, 1].mean():6.2f})   p10 {d[:, 0].quantile(0.1):5.2f} p90 {d[:, 0].quantile(0.9):5.2f}")
        prev = ss[:, :, i]
    tot = (ss[:, :, nst - 1] - ss[:, :, 0]) / 100.0
    print(f"   tile total {tot[:, 0].mean():.2f} us")
