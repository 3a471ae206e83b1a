"""conv1 -> conv2 of a dense block: the fused experiment kernel (tools/experiments/conv_pair.hip) against the two product launches.
    SRGANFD_LIB=build_exp/libsrganfd_exp.so python tools/r3/pair_bench.py [--cold]
Checks y1 / y2 bit for bit, then times both forms interleaved in one process (HIP events; --cold: a 256 MiB fill between launches so that
neither form finds its input in the caches, as in the training step)."""
import os, sys, ctypes as C, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sr_gan_fd_amd import _abi as A, ops

COLD = "--cold" in sys.argv
DT = torch.float16
dtc = ops.DT[DT]
N, H, W = 32, 128, 128
L = A.lib()
L.srganfd_exp_conv_pair.restype = C.c_int
L.srganfd_exp_conv_pair.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_int32, C.c_void_p]
torch.manual_seed(0)
x = (torch.randn(N, H, W, 192, device="cuda") * 0.5).to(DT)            # planar view: the same bytes, groups of 32 channels per image plane
w1 = torch.randn(32, 64, 3, 3, device="cuda") * 0.05
w2 = torch.randn(32, 96, 3, 3, device="cuda") * 0.05
b1, b2 = torch.randn(32, device="cuda") * 0.1, torch.randn(32, device="cuda") * 0.1
p1, p2 = ops.pack_single(w1, dtc), ops.pack_single(w2, dtc)
bufa, bufb = x.clone(), x.clone()
a1 = ops.conv_args(dtc, A.view(bufa, c0=0, planar=1), A.view(bufa, c0=64, planar=1), p1, N, H, W, 64, 32, bias=b1, act=A.ACT_LRELU, slope=0.2)
a2 = ops.conv_args(dtc, A.view(bufa, c0=0, planar=1), A.view(bufa, c0=96, planar=1), p2, N, H, W, 96, 32, bias=b2, act=A.ACT_LRELU, slope=0.2)


def two():
    ops.conv2d(a1); ops.conv2d(a2)


def fused():
    A.check(L.srganfd_exp_conv_pair(bufb.data_ptr(), N, H, W, p1.data_ptr(), p2.data_ptr(), b1.data_ptr(), b2.data_ptr(), 0.2, dtc, A.stream_ptr()), "exp_conv_pair")


two(); fused(); torch.cuda.synchronize()
va, vb = bufa.view(N, 6, H, W, 32), bufb.view(N, 6, H, W, 32)          # planar groups
for g, name in ((2, "y1"), (3, "y2")):
    same = torch.equal(va[:, g].view(torch.int16), vb[:, g].view(torch.int16))
    print(f"{name}: fused == two launches bit for bit: {same}; max |diff| {(va[:, g].float() - vb[:, g].float()).abs().max().item():.3e}, max |value| {va[:, g].float().abs().max().item():.3f}")
assert torch.equal(va[:, :2], vb[:, :2]) and torch.equal(va[:, 4:], vb[:, 4:])      # nothing else touched
big = torch.empty(1 << 28, dtype=torch.uint8, device="cuda")
res = {"two launches": [], "fused": []}
for rnd in range(7):
    for name, fn in (("two launches", two), ("fused", fused)):
        reps = 1 if COLD else 20
        tot = 0.0
        for _ in range(10 if COLD else 1):
            if COLD:
                big.fill_(rnd)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record(); torch.cuda.synchronize()
            tot += e0.elapsed_time(e1) * 1e3 / reps
        if rnd:
            res[name].append(tot / (10 if COLD else 1))
for k, v in res.items():
    v.sort()
    print(f"{'cold' if COLD else 'warm'} {k:14s}: median {v[len(v) // 2]:7.1f} us  min {v[0]:7.1f}")
