"""Is the 1.1-1.26x read excess of the 3x3 convs halo re-fetch?  Same input (first `cin` channels of a 192-channel-pitch buffer),
1x1 conv (no halo) vs 3x3 conv, one cold launch each.
    rocprofv3 --output-format csv --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum -d DIR -- python tools/pmc_halo.py ; python tools/pmc_halo.py --parse DIR"""
import csv, glob, sys
CONFIGS = ((1, 64), (3, 64), (1, 128), (3, 128), (1, 192), (3, 192))
N, H, W = 64, 128, 128
if len(sys.argv) > 2 and sys.argv[1] == "--parse":
    rows = []
    for f in glob.glob(sys.argv[2] + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "conv_igemm" in r["Kernel_Name"]:
                rows.append((int(r["Dispatch_Id"]), r["Counter_Name"], float(r["Counter_Value"])))
    for d, (ks, cin) in zip(sorted({d for d, _, _ in rows}), CONFIGS):
        v = {c: val for dd, c, val in rows if dd == d}
        alg = N * H * W * cin * 2
        print(f"{ks}x{ks} cin={cin:3d}: 2*RDREQ*64 = {2*v['TCC_EA0_RDREQ_sum']*64/1e6:6.1f} MB = {2*v['TCC_EA0_RDREQ_sum']*64/alg:.3f}x the {alg/1e6:.1f} MB read once")
    sys.exit(0)
import torch
sys.path.insert(0, '.')
from sr_gan_fd_amd import _abi as A, ops
flush = torch.empty(1 << 28, device='cuda', dtype=torch.float32)
for ks, cin in CONFIGS:
    x = torch.randn(N, H, W, 192, device='cuda').bfloat16()
    y = torch.empty(N, H, W, 32, device='cuda', dtype=torch.bfloat16)
    wp = ops.pack_single(torch.randn(32, cin, ks, ks, device='cuda') * 0.05, A.BF16)
    a = ops.conv_args(A.BF16, A.view(x), A.view(y), wp, N, H, W, cin, 32, ksize=ks, pad=ks // 2)
    flush.fill_(1.0)
    torch.cuda.synchronize()
    ops.conv2d(a)
    torch.cuda.synchronize()
