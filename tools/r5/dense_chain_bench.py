"""The LDS-resident dense-block launch (srganfd_dense_chain) against the five separate srganfd_conv2d launches, same buffers, alternating
in one process (forward form: bias + LeakyReLU growth convs, residual closing conv; and the data-gradient form).
    python tools/r5/dense_chain_bench.py [N H W ...]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sr_gan_fd_amd import ops
from tests.test_dense_chain_gpu import _build

shapes = [(16, 72, 72), (16, 48, 48), (16, 32, 32), (8, 60, 60), (4, 128, 128), (32, 128, 128)]
if len(sys.argv) >= 4:
    v = [int(t) for t in sys.argv[1:]]
    shapes = [tuple(v[i:i + 3]) for i in range(0, len(v), 3)]


def timed(fn, reps=30):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


for (n, h, w) in shapes:
    for bwd in (False, True):
        args, _, keep = _build(torch.float16, n, h, w, 1, bwd)
        chain = ops.DenseChain(args, "cuda")
        if not chain.ok:
            print(f"n{n} {h}x{w}: refused"); continue
        five = lambda: [ops.conv2d(a) for a in args]
        res = []
        for rnd in range(3):
            res.append((timed(five), timed(chain.run)))
        a5 = sorted(r[0] for r in res)[1]; a1 = sorted(r[1] for r in res)[1]
        gf = chain.flops / 1e9
        print(f"n{n:3d} {h:3d}x{w:3d} {'dgrad' if bwd else 'fwd  '}: five launches {a5:7.1f} us   dense chain {a1:7.1f} us   x{a5 / a1:4.2f}   ({gf / a1 / 1e3:6.1f} TF/s, hand-off give-ups {chain.errors()})", flush=True)
