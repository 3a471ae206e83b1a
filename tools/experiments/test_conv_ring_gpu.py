"""GPU parity of the LDS-DMA ring kernel (csrc/conv3x3_ring.hip) that serves the 3x3 stride-1 16-bit convolutions of the hot
path (BSRGAN/model.py:42-46 dense-block convs, :340-355 tail): every ring configuration against (a) torch's conv2d in fp64 on
the same 16-bit-rounded operands and (b) the conv_igemm tiles (mode 0) -- same packed weights, same epilogue contract.
Only the accumulation order differs, so the bound is one rounding of the 16-bit store."""
import pytest
import torch
import torch.nn.functional as F

import os

pytestmark = pytest.mark.gpu

FORCE = 0x100
# The ring / stream kernels were measured and rejected; they are compiled in -DSRGANFD_EXPERIMENT builds only
# (tools/build_variant.sh, selected with SRGANFD_LIB), and so is this module's subject.
MODES = [1, 2, 3, 4, 5]
if not os.environ.get("SRGANFD_LIB"):
    pytest.skip("ring / stream conv kernels exist in experiment builds only (SRGANFD_LIB=...)", allow_module_level=True)


def _planar(t_nchw, dtype, cbuf, c0):
    """NCHW cpu tensor -> planar 32-channel-group cuda buffer (n, cbuf/32, h, w, 32) with the data at channels [c0, c0+C)"""
    n, c, h, w = t_nchw.shape
    full = torch.randn(n, cbuf, h, w) * 3.0
    full[:, c0:c0 + c] = t_nchw
    return full.view(n, cbuf // 32, 32, h, w).permute(0, 1, 3, 4, 2).contiguous().to(dtype).cuda()


def _from_planar(buf, c0, c):
    n, g, h, w, _ = buf.shape
    return buf.permute(0, 1, 4, 2, 3).reshape(n, g * 32, h, w)[:, c0:c0 + c]


def _nhwc(t_nchw, dtype, cbuf, c0):
    n, c, h, w = t_nchw.shape
    full = torch.randn(n, h, w, cbuf) * 3.0
    full[..., c0:c0 + c] = t_nchw.permute(0, 2, 3, 1)
    return full.to(dtype).cuda()


CASES = [
    dict(n=2, h=64, w=64, cin=64, cout=32, act=1, bias=True, planar=1),                 # dense-block conv1
    dict(n=1, h=40, w=70, cin=160, cout=32, act=1, bias=True, planar=1),                # ragged edges, 5 chunks
    dict(n=1, h=33, w=33, cin=192, cout=64, res=True, bias=True, planar=1),             # conv5 + two residuals
    dict(n=1, h=64, w=32, cin=192, cout=32, mask=True, planar=1),                       # dgrad-like + LeakyReLU' mask
    dict(n=1, h=24, w=24, cin=64, cout=64, up=1, act=1, bias=True, planar=0),           # nearest x2 gather, NHWC
    dict(n=1, h=48, w=48, cin=64, cout=128, act=1, bias=True, planar=0, y2=True),       # two channel blocks + second output
    dict(n=3, h=32, w=32, cin=32, cout=64, act=2, planar=0, alpha_dev=True),            # one chunk, ReLU, device alpha
]


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("case", CASES)
def test_ring_conv_matches_torch_and_igemm(dtype, mode, case):
    from sr_gan_fd_amd import _abi as A, ops
    if not hasattr(A.lib(), "srganfd_set_ring_mode"):
        pytest.skip("SRGANFD_LIB does not point at an experiment build")
    torch.manual_seed(3)
    dt = ops.DT[dtype]
    A.lib().srganfd_set_mfma16(0)      # the ring kernels read 32x32x16-order weights: pack and run both kernels under that setting
    n, h, w, cin, cout = case["n"], case["h"], case["w"], case["cin"], case["cout"]
    up, planar = case.get("up", 0), case["planar"]
    x = torch.randn(n, cin, h, w)
    wt = torch.randn(cout, cin, 3, 3) / (cin * 9) ** 0.5
    b = torch.randn(cout) if case.get("bias") else None
    mk = _planar if planar else _nhwc
    xbuf = mk(x, dtype, cin + 32, 32)
    wp = ops.pack_single(wt.cuda(), dt)
    ho, wo = h << up, w << up
    rt = lambda t: t.to(dtype).double()
    ref = F.conv2d(F.interpolate(rt(x), scale_factor=2, mode="nearest") if up else rt(x), rt(wt), b.double() if b is not None else None, padding=1)
    alpha = 1.0
    kw = dict(up=up, act=case.get("act", 0), bias=b.cuda() if b is not None else None)
    keep = []
    if case.get("alpha_dev"):
        ad = torch.tensor([0.75], device="cuda")
        keep.append(ad)
        kw.update(alpha=2.0, alpha_dev=ad)
        ref = ref * 1.5
    if case.get("act") == 1:
        ref = F.leaky_relu(ref, 0.2)
    elif case.get("act") == 2:
        ref = F.relu(ref)
    pre = ref
    V = lambda buf, c0: A.view(buf, cstride=(buf.shape[1] * 32 if planar else buf.shape[-1]), c0=c0, planar=planar)
    if case.get("res"):
        r1, r2 = torch.randn(n, cout, ho, wo), torch.randn(n, cout, ho, wo)
        r1b, r2b = mk(r1, dtype, cout + 64, 64), mk(r2, dtype, cout, 0)
        keep += [r1b, r2b]
        kw.update(post_scale=0.04, r1=V(r1b, 64), r1_scale=0.2, r2=V(r2b, 0), r2_scale=1.0)
        ref = ref * 0.04 + 0.2 * rt(r1) + rt(r2)
    if case.get("mask"):
        m = torch.randn(n, cout, ho, wo)
        mb = mk(m, dtype, cout + 32, 32)
        keep.append(mb)
        kw.update(mask=V(mb, 32), mask_slope=0.2)
        ref = ref * torch.where(rt(m) > 0, 1.0, 0.2)

    def run(ring_mode):
        A.lib().srganfd_set_ring_mode(ring_mode)
        ybuf = mk(torch.zeros(n, cout, ho, wo), dtype, cout + 64, 32)
        ybuf.fill_(7.0)
        y2buf = None
        kw2 = dict(kw)
        if case.get("y2"):
            y2buf = mk(torch.zeros(n, cout, ho, wo), dtype, cout, 0)
            kw2.update(y2=V(y2buf, 0), r1=V(xbuf_r, 0), r1_scale=1.0)
        args = ops.conv_args(dt, V(xbuf, 32), V(ybuf, 32), wp, n, h, w, cin, cout, **kw2)
        ops.conv2d(args)
        torch.cuda.synchronize()
        return ybuf, y2buf

    xbuf_r = None
    if case.get("y2"):
        rr = torch.randn(n, cout, ho, wo)
        xbuf_r = mk(rr, dtype, cout, 0)
        ref = ref + rt(rr)
    try:
        got_buf, got_y2 = run(FORCE | mode)
        old_buf, old_y2 = run(0)              # conv_igemm tiles (16x16x32 form for the 32-channel cases: the library default)
    finally:
        A.lib().srganfd_set_ring_mode(-1)
        A.lib().srganfd_set_mfma16(3)
    take = (lambda bf, c0: _from_planar(bf, c0, cout)) if planar else (lambda bf, c0: bf[..., c0:c0 + cout].permute(0, 3, 1, 2))
    got, old = take(got_buf, 32).double().cpu(), take(old_buf, 32).double().cpu()
    scale = ref.abs().max().item()
    ulp = 2.0 ** (-8 if dtype == torch.bfloat16 else -11)
    assert (got - ref).abs().max().item() <= 2.5 * ulp * scale, f"ring vs torch: {(got - ref).abs().max().item() / scale:.3e}"
    assert (got - old).abs().max().item() <= 2.0 * ulp * scale, f"ring vs conv_igemm: {(got - old).abs().max().item() / scale:.3e}"
    # nothing outside the output view is written
    outside = torch.ones_like(got_buf, dtype=torch.bool)
    if planar:
        outside[:, 1:1 + cout // 32] = False
    else:
        outside[..., 32:32 + cout] = False
    assert torch.all(got_buf[outside] == 7.0)
    if got_y2 is not None:
        g2 = take(got_y2, 0).double().cpu()
        assert (g2 - pre).abs().max().item() <= 2.5 * ulp * pre.abs().max().item()
