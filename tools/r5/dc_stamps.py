"""EXPERIMENT: in-kernel timeline of the LDS-resident dense-block launch (variant library built with -DSRGANFD_EXPERIMENT):
    git apply tools/experiments/r5_dense_chain_stamps.diff && make -C sr_gan_fd_amd/csrc -j8 OUT=../libsrganfd_dcx.so EXTRA=-DSRGANFD_EXPERIMENT && git checkout sr_gan_fd_amd/csrc/dense_chain.hip
    SRGANFD_LIB=$PWD/sr_gan_fd_amd/libsrganfd_dcx.so SRGANFD_DC_STAMPS=1 python tools/r5/dc_stamps.py [N H W]
prints, per layer, the median over workgroups of wave 0's s_memtime differences (in shader-clock cycles and in us)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sr_gan_fd_amd import ops
from tests.test_dense_chain_gpu import _build

n, h, w = (int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (16, 32, 32)
args, _, keep = _build(torch.float16, n, h, w, 1, False)
chain = ops.DenseChain(args, "cuda")
assert chain.ok
for _ in range(5):
    chain.run()
torch.cuda.synchronize()
off = 64 + 4 * 4 * 16384
st = chain.ws[off:off + 8 * 256 * 1024].view(torch.int64).view(-1, 256).cpu()
ntiles = min(1024, n * -(-h // 8) * -(-w // 32)) if os.environ.get("DC_TILE") is None else None
st = st[:256]
st = st[st[:, 0] != 0]
print(f"{st.shape[0]} workgroups stamped; n{n} {h}x{w}")
tot = (st[:, 60] - st[:, 0]).float()
real = (st[:, 62] - st[:, 61]).float() * 10e-3     # us (100 MHz)
ghz = (tot / real).median().item() / 1e3
print(f"kernel body (wave 0): median {tot.median().item():.0f} cycles = {real.median().item():.2f} us  -> shader clock counter {ghz:.3f} GHz")
med = lambda a: a.float().median().item()
print(f"prologue of the LAST pass ends {med(st[:, 1] - st[:, 0]):.0f} cycles after the kernel entry (one pass: the patch load + first weight pieces)")
for l in range(6):
    b = 2 + 8 * l
    if st[0, b] == 0:
        break
    start, spin, steps_e, epi_e, wacc = st[:, b], st[:, b + 1], st[:, b + 2], st[:, b + 3], st[:, b + 4]
    print(f"         behind step 1: drain of the previous layer's stores {med(st[:, b + 6] - st[:, b + 5]):6.0f}, output address + operand requests {med(st[:, b + 7] - st[:, b + 6]):6.0f}")
    print(f"layer {l}: steps {med(steps_e - start):7.0f} cycles (of which at the step barriers {med(wacc):7.0f}, waiting for the neighbours' flags {med(spin):6.0f})   epilogue {med(epi_e - steps_e):6.0f}")
d = (st[:, 101:178] - st[:, 100:177]).float().median(dim=0).values
print("cycles from step barrier to step barrier, steps 0..76 of the last pass (median over workgroups; layer boundaries at 6, 15, 27, 42, 60):")
print(" ".join(f"{int(v)}" for v in d.tolist()))
# the layer-boundary step in detail (last pass): last step's barrier -> its MFMAs done -> epilogue done -> next layer's set-up done -> next layer's first barrier passed
bounds = [6, 15, 27, 42, 60, 78]
for l in range(5):
    b = 2 + 8 * l
    last_bar = st[:, 100 + bounds[l] - 1]
    steps_e, epi_e, nstart, nbar = st[:, b + 2], st[:, b + 3], st[:, b + 8], st[:, 100 + bounds[l]]
    print(f"boundary {l}->{l+1}: last step (barrier -> MFMAs issued) {med(steps_e - last_bar):6.0f}, epilogue {med(epi_e - steps_e):6.0f}, next layer's set-up {med(nstart - epi_e):6.0f}, to its first barrier passed {med(nbar - nstart):6.0f}")
lw = st[:, 178:256].float().median(dim=0).values
print("cycles loader wave 4 waited for its pieces of step t + 1 before step t's barrier, steps 0..77 of the last pass:")
print(" ".join(f"{int(v)}" for v in lw.tolist()))
