"""In-kernel s_memtime stamps of the stream conv kernel (SRGANFD_EXPERIMENT build): where a stage's cycles go.
SRGANFD_LIB=build_exp/libsrganfd_exp.so python tools/stamps.py [mode] [cin] [cout]"""
import os, sys, ctypes as C, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sr_gan_fd_amd import _abi as A, ops
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 4
cin = int(sys.argv[2]) if len(sys.argv) > 2 else 128
cout = int(sys.argv[3]) if len(sys.argv) > 3 else 32
L = A.lib()
L.srganfd_set_stamp_buffer.argtypes = [C.c_void_p]
n, h, w = 32, 128, 128
DT = torch.bfloat16
x = (torch.randn(n, h, w, 192, device='cuda') * 0.5).to(DT)
y = torch.empty(n, h, w, 192, device='cuda', dtype=DT)
wp = ops.pack_single(torch.randn(cout, cin, 3, 3, device='cuda') * 0.05, ops.DT[DT])
a = ops.conv_args(ops.DT[DT], A.view(x, c0=0, planar=1), A.view(y, c0=0 if cout == 64 else 160, planar=1), wp, n, h, w, cin, cout, act=A.ACT_LRELU)
L.srganfd_set_ring_mode(mode)
for _ in range(5):
    ops.conv2d(a)
buf = torch.zeros(8 * 8 * 64 * 4, dtype=torch.int64, device='cuda')
L.srganfd_set_stamp_buffer(buf.data_ptr())
ops.conv2d(a)
torch.cuda.synchronize()
L.srganfd_set_stamp_buffer(None)
s = buf.cpu().view(8, 8, 64, 4)
for wg in (0, 3):
    t0 = s[wg, :, 0, 0].min().item()
    print(f"workgroup {wg}: per stage [wait-start, wait-done, barrier-done, compute-done] relative cycles, waves 0 and 7; stage lengths")
    for it in range(0, 18):
        if s[wg, 0, it, 3] == 0:
            break
        r0 = [(s[wg, 0, it, j].item() - t0) for j in range(4)]
        r7 = [(s[wg, 7, it, j].item() - t0) for j in range(4)]
        wait = [max(0, (s[wg, wv, it, 1] - s[wg, wv, it, 0]).item()) for wv in range(8)]
        bar = [(s[wg, wv, it, 2] - s[wg, wv, it, 1]).item() for wv in range(8)]
        comp = [(s[wg, wv, it, 3] - s[wg, wv, it, 2]).item() for wv in range(8)]
        print(f"  stage {it:2d}: w0 {r0}  w7 {r7} | vmcnt-wait {wait} | barrier {bar} | compute {comp}")
