"""Micro-benchmark the dense-block weight-gradient launch (five convs of one RDB, B=32, 128x128, planar buffers) over the
pixel-split count: python tools/wgbench.py [--splits 0,36,32] [--dtype bf16|f16]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sr_gan_fd_amd import _abi as A, ops


def arg(name, default):
    return sys.argv[sys.argv.index(name) + 1] if name in sys.argv else default


DT = {"bf16": torch.bfloat16, "f16": torch.float16}[arg("--dtype", "bf16")]
SPLITS = [int(x) for x in arg("--splits", "0").split(",")]
VARIANTS = [int(x) for x in arg("--variants", "0,1,2").split(",")]
DBG = int(arg("--dbg", "0"))     # experiment builds: 1 no global loads, 16 no LDS reads / MFMA, 4 no slab store (results are wrong)
if DBG:
    A.lib().srganfd_set_debug(DBG)
N, H, W, Cc, G = int(arg("--batch", "32")), int(arg("--size", "128")), int(arg("--size", "128")), 64, 32
Ccat = Cc + 4 * G
dtc = ops.DT[DT]
x = (torch.randn(N, H, W, Ccat, device="cuda") * 0.5).to(DT)
dy = (torch.randn(N, H, W, Ccat, device="cuda") * 0.01).to(DT)
convs, off = [], 0
for k in range(1, 6):
    cin, cout = Cc + (k - 1) * G, (Cc if k == 5 else G)
    convs.append(dict(ci_lo=0, cin=cin, co_lo=(0 if k == 5 else Cc + (4 - k) * G), cout=cout, dw_off=off, db_off=off + cout * cin * 9, co_dst=cout, ci_dst=cin))
    off += cout * cin * 9 + cout
    off = (off + 3) // 4 * 4
grads = torch.zeros(off, device="cuda")
res = {}
plans = {s: ops.WgradPlan(x.device, dtc, N, H, W, Ccat, Ccat, convs, splits=s) for s in SPLITS}
ws = torch.empty(max(p.workspace_bytes for p in plans.values()), dtype=torch.uint8, device="cuda")
def set_variant(v):
    f = getattr(A.lib(), "srganfd_set_ring_mode", None)      # experiment builds; the product library runs variant 3 only
    if f is not None:
        f(0x1000 | (v << 9))


for rnd in range(4):
  for var in VARIANTS:
    set_variant(var)
    for s0, p in plans.items():
        s = (s0, var)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 3 if rnd == 0 else 10
        e0.record()
        for _ in range(reps):
            p.run(A.view(x, planar=1), A.view(dy, planar=1), grads, ws)
        e1.record(); torch.cuda.synchronize()
        if rnd:
            res.setdefault(s, []).append(e0.elapsed_time(e1) * 1e3 / reps)
ref = None
for s, v in res.items():
    v.sort()
    set_variant(s[1])
    plans[s[0]].run(A.view(x, planar=1), A.view(dy, planar=1), grads, ws)
    torch.cuda.synchronize()
    set_variant(0)
    g = grads.clone()
    ref = g if ref is None else ref
    print(f"splits {s[0]:3d} variant {s[1]}: median {v[len(v)//2]:7.1f} us  min {v[0]:7.1f}  {plans[s[0]].flops / v[len(v)//2] / 1e6:7.1f} TFLOP/s   max |dW - dW(first)| / max|dW| = {((g - ref).abs().max() / ref.abs().max()).item():.2e}")
