"""Does the 1.18x read over-fetch of the 32-channel dense-block convs come from 128-byte line granularity?  A conv reading the
first `cin` channels of a 192-channel-pitch NHWC buffer touches cin*2 bytes of each 384-byte pixel: 96 channels = 1.5 lines,
160 = 2.5 lines.  One launch per configuration on cold caches (1 GiB streamed in between), N=64 so the input exceeds the
Infinity Cache.
    rocprofv3 --output-format csv --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum -d DIR -- python tools/pmc_pitch.py
    python tools/pmc_pitch.py --parse DIR      # per-dispatch counters of the conv launches, in launch order"""
import csv, glob, sys
CONFIGS = ((64, 192), (96, 192), (128, 192), (160, 192), (192, 192), (96, 96), (160, 160))
N, H, W = 64, 128, 128
if len(sys.argv) > 2 and sys.argv[1] == "--parse":
    rows = []
    for f in glob.glob(sys.argv[2] + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "conv_igemm" in r["Kernel_Name"]:
                rows.append((int(r["Dispatch_Id"]), r["Counter_Name"], float(r["Counter_Value"])))
    disp = sorted({d for d, _, _ in rows})
    for d, (cin, xC) in zip(disp, CONFIGS):
        v = {c: val for dd, c, val in rows if dd == d}
        rd, wr = v.get("TCC_EA0_RDREQ_sum", 0), v.get("TCC_EA0_WRREQ_sum", 0)
        alg = N * H * W * cin * 2
        lines = N * H * W * ((cin * 2 + 127) // 128) * 128
        print(f"cin={cin:3d} pitch={xC:3d}: RDREQ {rd:.4g} WRREQ {wr:.4g}; algorithmic read {alg/1e6:6.1f} MB, whole-line read {lines/1e6:6.1f} MB, "
              f"2*RDREQ*64 = {2*rd*64/1e6:6.1f} MB ({2*rd*64/alg:.3f}x algorithmic, {2*rd*64/lines:.3f}x whole lines)")
    sys.exit(0)
import torch
sys.path.insert(0, '.')
from sr_gan_fd_amd import _abi as A, ops
import os
if os.environ.get('SRGANFD_DBG'):
    A.lib().srganfd_set_debug(int(os.environ['SRGANFD_DBG']))
flush = torch.empty(1 << 28, device='cuda', dtype=torch.float32)      # 1 GiB
for cin, xC in CONFIGS:
    x = torch.randn(N, H, W, xC, device='cuda').bfloat16()
    y = torch.empty(N, H, W, 32, device='cuda', dtype=torch.bfloat16)
    wp = ops.pack_single(torch.randn(32, cin, 3, 3, device='cuda') * 0.05, A.BF16)
    a = ops.conv_args(A.BF16, A.view(x), A.view(y), wp, N, H, W, cin, 32)
    flush.fill_(1.0)
    torch.cuda.synchronize()
    ops.conv2d(a)
    torch.cuda.synchronize()
    print(f"cin={cin} of a {xC}-channel buffer launched")
