"""srganfd_conv2d_chain (the four growth convs of a dense block as ONE kernel) against four srganfd_conv2d launches: bitwise equality and time.
    python tools/r3/chain_bench.py [--mask] [--reps 50]"""
import ctypes as C
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sr_gan_fd_amd import _abi as A, ops

MASK = "--mask" in sys.argv
REPS = int(sys.argv[sys.argv.index("--reps") + 1]) if "--reps" in sys.argv else 50
N = int(sys.argv[sys.argv.index("--batch") + 1]) if "--batch" in sys.argv else 32
H = W = int(sys.argv[sys.argv.index("--size") + 1]) if "--size" in sys.argv else 128
DT = torch.float16
dt = ops.DT[DT]
torch.manual_seed(0)
L = A.lib()
if not hasattr(L, "srganfd_conv2d_chain"):
    raise SystemExit("chain_bench: experiment build needed (SRGANFD_LIB=build_exp/libsrganfd_exp.so)")
L.srganfd_conv2d_chain.restype = C.c_int
L.srganfd_conv2d_chain.argtypes = [C.POINTER(A.ConvArgs), C.c_int, C.c_void_p, C.c_void_p]


def build():
    x = (torch.randn(N, H, W, 192, device="cuda") * 0.5).to(DT)
    m = torch.randn(N, H, W, 192, device="cuda").to(DT)
    args, keep = [], [x, m]
    for k in range(4):
        cin = 64 + 32 * k
        wt = torch.randn(32, cin, 3, 3, device="cuda") / (cin * 9) ** 0.5
        wp = ops.pack_single(wt, dt)
        b = torch.randn(32, device="cuda") * 0.1
        kw = dict(bias=b, act=A.ACT_LRELU) if not MASK else dict(mask=A.view(m, c0=64 + 32 * k, planar=1), mask_slope=0.2)
        args.append(ops.conv_args(dt, A.view(x, planar=1), A.view(x, c0=64 + 32 * k, planar=1), wp, N, H, W, cin, 32, **kw))
        keep += [wt, wp, b]
    return x, args, keep


x1, a1, k1 = build()
torch.manual_seed(0)
x2, a2, k2 = build()
assert torch.equal(x1, x2)
ctr = torch.zeros(2560, dtype=torch.int32, device="cuda")
arr = (A.ConvArgs * 4)(*a2)
st = A.stream_ptr()
for a in a1:
    ops.conv2d(a)
A.check(L.srganfd_conv2d_chain(arr, 4, ctr.data_ptr(), st), "chain")
torch.cuda.synchronize()
same = torch.equal(x1.view(torch.int16), x2.view(torch.int16))
print("bitwise equal:", same, " max |diff| %.3g" % (x1.float() - x2.float()).abs().max().item(), " finite:", bool(torch.isfinite(x2.float()).all()))


def timeit(fn):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(5):
        e0.record()
        for _ in range(REPS):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / REPS)
    ts.sort()
    return ts[len(ts) // 2]


def four():
    for a in a1:
        L.srganfd_conv2d(C.byref(a), st)


def chain():
    L.srganfd_conv2d_chain(arr, 4, ctr.data_ptr(), st)


t4, tc = timeit(four), timeit(chain)
print("four launches %.1f us   chain %.1f us   (%+.1f %%)" % (t4, tc, 100 * (tc / t4 - 1)))
for nn in (1, 2, 3):
    tn = timeit(lambda: [L.srganfd_conv2d(C.byref(a), st) for a in a1[:nn]])
    tcn = timeit(lambda: L.srganfd_conv2d_chain(arr, nn, ctr.data_ptr(), st))
    print("first %d: launches %.1f us   chain %.1f us" % (nn, tn, tcn))
if not same:
    sys.exit(1)
