"""Do two half-batch launch chains on two HIP streams beat one full-batch chain?  (round-3 experiment: hides launch boundaries, ramp and tail)
    python tools/r3/two_stream.py [--dtype f16]
Chain = the four growth convs + conv5 of one dense block (forward), repeated REPS times; variant A: batch 32 on one stream; variant B: images
0-15 on stream 0 and 16-31 on stream 1 (same buffers, per-image independent work)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sr_gan_fd_amd import _abi as A, ops
import ctypes as C

DT = torch.float16
dtc = ops.DT[DT]
N, H, W, Cc, G = 32, 128, 128, 64, 32
Ccat = Cc + 4 * G
bufs = [(torch.randn(N, H, W, Ccat, device="cuda") * 0.5).to(DT) for _ in range(2)]
wts = [ops.pack_single(torch.randn((Cc if k == 5 else G), Cc + (k - 1) * G, 3, 3, device="cuda") * 0.05, dtc) for k in range(1, 6)]


def chain(n0, n):
    """launch structs of one dense block over images [n0, n0+n)"""
    out = []
    per_img = H * W * Ccat * 2
    class V:  # view with offset base pointer
        pass
    def view(t, c0=0):
        v = A.view(t, c0=c0, planar=1)
        v.ptr = t.data_ptr() + n0 * per_img
        return v
    x, y = bufs
    for k in range(1, 5):
        out.append(ops.conv_args(dtc, view(x), view(x, c0=Cc + (k - 1) * G), wts[k - 1], n, H, W, Cc + (k - 1) * G, G, act=A.ACT_LRELU))
    out.append(ops.conv_args(dtc, view(x), view(y), wts[4], n, H, W, Ccat, Cc, post_scale=0.2, r1=view(x), r1_scale=1.0))
    return out


L = A.lib()
s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()
full, h0, h1 = chain(0, N), chain(0, N // 2), chain(N // 2, N // 2)
q0, q1, q2, q3 = chain(0, 8), chain(8, 8), chain(16, 8), chain(24, 8)
s2, s3 = torch.cuda.Stream(), torch.cuda.Stream()


def run(lists_streams, reps):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _, s in lists_streams:
        s.wait_event(e0)
    for _ in range(reps):
        for lst, s in lists_streams:
            sp = s.cuda_stream
            for a in lst:
                rc = L.srganfd_conv2d(C.byref(a), sp)
                assert rc == 0
    cur = torch.cuda.current_stream()
    for _, s in lists_streams:
        ev = torch.cuda.Event(); ev.record(s); cur.wait_event(ev)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


res = {}
for rnd in range(5):
    for name, ls in (("1 stream x 32", [(full, s0)]), ("2 streams x 16", [(h0, s0), (h1, s1)]), ("4 streams x 8", [(q0, s0), (q1, s1), (q2, s2), (q3, s3)]),
                     ("1 stream, 2 x 16 back to back", [(h0, s0), (h1, s0)])):
        t = run(ls, 3 if rnd == 0 else 20)
        if rnd:
            res.setdefault(name, []).append(t)
for k, v in res.items():
    v.sort()
    print(f"{k:32s}: dense block forward {v[len(v)//2]:8.1f} us (min {v[0]:.1f})")
