"""Round 5: the round-4 probe with the shape on the command line (python tools/r5/chain_probe.py N H W), for the reference-default
shapes where a launch is one round of tiles or less.
EXPERIMENT (variant library: `git apply tools/experiments/r4_chain.diff && make -C sr_gan_fd_amd/csrc OUT=../libsrganfd_chain.so EXTRA=-DSRGANFD_CHAIN_BUILD=1
&& git checkout sr_gan_fd_amd/csrc`): the four growth convs of one dense block (64/96/128/160 -> 32,
bias + LeakyReLU) at batch 32, 128 x 128 as four launches and as ONE layer-persistent launch with per-tile neighbour flags.
    SRGANFD_LIB=.../libsrganfd_chain.so python tools/r4/chain_probe.py"""
import ctypes as C
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sr_gan_fd_amd import _abi as A, ops

L = A.lib()
L.srganfd_conv2d_chain.restype = C.c_int
L.srganfd_conv2d_chain.argtypes = [C.POINTER(A.ConvArgs), C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
torch.manual_seed(0)
N, H, W = (int(v) for v in (sys.argv[1:4] if len(sys.argv) >= 4 else (32, 128, 128)))
Cc, G = 64, 32
dt, dtype = A.F16, torch.float16
planar = 1
def new():
    return torch.zeros(N, H, W, Cc + 4 * G, device="cuda", dtype=dtype)
x0 = torch.randn(N, (Cc // 32), H, W, 32, device="cuda", dtype=dtype)        # planar group planes of the 64 input channels
ws = [torch.randn(G, Cc + k * G, 3, 3, device="cuda") / (3.0 * (Cc + k * G) ** 0.5) for k in range(4)]
bs = [torch.randn(G, device="cuda") * 0.1 for _ in range(4)]
wp = [ops.pack_single(w, dt) for w in ws]

def fill(buf):
    buf.zero_()
    buf.view(-1)[: x0.numel()].view_as(x0).copy_(x0)      # planar layout: group g of image n at ((n * 6 + g) * H * W) * 32 ... see below
# the planar buffer is addressed as image base + group * (H*W*32) + pixel * 32: per image 6 planes
def plane_fill(buf):
    b = buf.view(N, 6, H, W, 32)
    b.zero_()
    b[:, :2].copy_(x0)

def args_for(buf):
    out = (A.ConvArgs * 4)()
    for k in range(4):
        a = ops.conv_args(dt, A.view(buf, c0=0, planar=planar), A.view(buf, c0=Cc + k * G, planar=planar), wp[k], N, H, W, Cc + k * G, G,
                          bias=bs[k], act=A.ACT_LRELU, slope=0.2)
        C.memmove(C.byref(out, k * C.sizeof(A.ConvArgs)), C.byref(a), C.sizeof(A.ConvArgs))
    return out

bufA, bufB = new(), new()
plane_fill(bufA); plane_fill(bufB)
aA, aB = args_for(bufA), args_for(bufB)
ntiles = N * ((H + 15) // 16) * ((W + 31) // 32)
print("shape: batch %d, %d x %d -> %d tiles of 16 x 32" % (N, H, W, ntiles), flush=True)
flags = torch.zeros(4 * ntiles, dtype=torch.int32, device="cuda")
err = torch.zeros(2, dtype=torch.int32, device="cuda")          # [timed-out waits, workgroups not on XCD blockIdx % 8]
epoch = [0]

def four():
    for k in range(4):
        A.check(L.srganfd_conv2d(C.byref(aA[k]), A.stream_ptr()), "conv")
def one():
    epoch[0] += 1
    A.check(L.srganfd_conv2d_chain(aB, 4, flags.data_ptr(), epoch[0], err.data_ptr(), A.stream_ptr()), "chain")

four(); one(); torch.cuda.synchronize()
print("timeouts / off-XCD workgroups:", err.tolist(), " equal:", torch.equal(bufA, bufB), " max|y|:", float(bufA.float().abs().max()), flush=True)
if not torch.equal(bufA, bufB):
    d = (bufA.float() - bufB.float()).abs().view(N, 6, H, W, 32)
    for g in range(6):
        print("  group", g, "max diff", float(d[:, g].max()))
for name, fn in (("four launches", four), ("one launch", one), ("four launches", four), ("one launch", one)):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    print("%-14s %8.1f us per dense-block forward (4 convs)" % (name, e0.elapsed_time(e1) * 50), flush=True)
print("timeouts / off-XCD workgroups:", err.tolist(), " equal after timing:", torch.equal(bufA, bufB))
# the same launch while another stream keeps the chip busy with the four plain launches on the other buffer (partial residency)
s2 = torch.cuda.Stream()
for rep in range(3):
    bufB.view(N, 6, H, W, 32)[:, 2:].zero_()
    with torch.cuda.stream(s2):
        for _ in range(10): four()
    for _ in range(10): one()
    torch.cuda.synchronize()
    print("concurrent with a second stream: timeouts / off-XCD", err.tolist(), " equal:", torch.equal(bufA, bufB), flush=True)
