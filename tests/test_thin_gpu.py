"""The thin-side convolution kernels (csrc/conv_thin.hip: 1..4 channels against 64, 3x3 stride 1 pad 1) through the C ABI against
torch's fp32 conv2d / conv2d gradients on the same 16-bit-rounded operands.  These are the layers BSRGAN/model.py:102,135,325,355
(discriminator / generator conv1 and conv4) and VGG-19 features.0 run on; every orientation the engines use is covered: forward
and data gradient of both directions, with bias / activation / LeakyReLU' mask, NHWC and planar 64-channel views, ragged sizes
(tile edges: widths that are not multiples of 14 / 16 / 32 / 64, heights that are not multiples of 32), one to four thin channels.
Tolerance: operands are rounded to the 16-bit type first, accumulation is fp32 -> 2e-3 relative for f16 outputs (one f16 rounding of
the result), 2e-5 for fp32 outputs; bf16 outputs 8e-3."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DTS = [torch.float16, torch.bfloat16]


def _rel(a, b):
    return ((a.double() - b.double()).abs().max() / (b.double().abs().max() + 1e-30)).item()


def _nhwc4(x_nchw, dt):
    """(N, c<=4, H, W) fp32 -> (N, H, W, 4) 16-bit with zero padding channels"""
    n, c, h, w = x_nchw.shape
    out = torch.zeros(n, h, w, 4, dtype=dt, device=x_nchw.device)
    out[..., :c] = x_nchw.permute(0, 2, 3, 1).to(dt)
    return out


SHAPES = [(2, 16, 16), (1, 37, 52), (2, 33, 70), (1, 64, 129), (3, 5, 7)]


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("cs", [1, 3, 4])
@pytest.mark.parametrize("mode", ["fwd", "dgrad_mask", "fwd_planar_relu"])
def test_thin_in(dt, cs, mode):
    """thin_in: 1..4 -> 64.  fwd: y = lrelu(conv(x, W) + b); dgrad_mask: dX = conv_transpose(dY, W) * lrelu'(act) for a 64 -> cs conv"""
    from sr_gan_fd_amd import _abi as A, ops
    torch.manual_seed(1)
    tol = 2e-3 if dt == torch.float16 else 8e-3
    for (n, h, w) in SHAPES:
        small = torch.randn(n, cs, h, w, device="cuda")
        thin = _nhwc4(small, dt)
        small_r = thin[..., :cs].permute(0, 3, 1, 2).float()
        if mode == "dgrad_mask":
            W = torch.randn(cs, 64, 3, 3, device="cuda") * 0.2          # the 64 -> cs conv's weight
            Wr = W.to(dt).float()
            act = torch.randn(n, h, w, 64, device="cuda").to(dt)
            want = F.conv_transpose2d(small_r, Wr, padding=1) * torch.where(act.permute(0, 3, 1, 2).float() > 0, 1.0, 0.2)
            y = torch.empty(n, h, w, 64, dtype=dt, device="cuda")
            a = ops.thin_args(ops.DT[dt], n, h, w, cs, W, A.view(y), w_big_is_cout=False, flip=True, mask=A.view(act), mask_slope=0.2, thin=thin)
            ops.thin_in(a)
            got = y.permute(0, 3, 1, 2).float()
        else:
            W = torch.randn(64, cs, 3, 3, device="cuda") * 0.3
            b = torch.randn(64, device="cuda")
            Wr = W.to(dt).float()
            pre = F.conv2d(small_r, Wr, b, padding=1)
            if mode == "fwd":
                want = F.leaky_relu(pre, 0.2)
                y = torch.full((n, h, w, 96), 7.0, dtype=dt, device="cuda")           # a channel slice of a wider NHWC buffer
                a = ops.thin_args(ops.DT[dt], n, h, w, cs, W, A.view(y, c0=32), w_big_is_cout=True, bias=b, act=A.ACT_LRELU, slope=0.2, thin=thin)
                ops.thin_in(a)
                got = y[..., 32:96].permute(0, 3, 1, 2).float()
                assert (y[..., :32] == 7.0).all()
            else:
                want = F.relu(pre)
                y = torch.zeros(n, 6, h, w, 32, dtype=dt, device="cuda")               # planar 32-channel groups of a 192-channel buffer
                a = ops.thin_args(ops.DT[dt], n, h, w, cs, W, A.View(y.data_ptr(), 192, 64, 1, 0), w_big_is_cout=True, bias=b, act=A.ACT_RELU, thin=thin)
                ops.thin_in(a)
                got = y[:, 2:4].permute(0, 1, 4, 2, 3).reshape(n, 64, h, w).float()
                assert (y[:, :2] == 0).all() and (y[:, 4:] == 0).all()
        torch.cuda.synchronize()
        e = _rel(got, want)
        assert e < tol, (mode, cs, (n, h, w), e)


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("cs", [1, 3, 4])
@pytest.mark.parametrize("mode", ["fwd", "dgrad"])
def test_thin_out(dt, cs, mode):
    """thin_out: 64 -> 1..4, fp32 output.  fwd: y = conv(x, W) + b; dgrad: dX = conv_transpose(dY, W) for a cs -> 64 conv"""
    from sr_gan_fd_amd import _abi as A, ops
    torch.manual_seed(2)
    for (n, h, w) in SHAPES:
        big = torch.randn(n, h, w, 64, device="cuda").to(dt)
        big_r = big.permute(0, 3, 1, 2).float()
        pitch = 1 if (cs == 1 and mode == "fwd") else 4
        out = torch.full((n, h, w, pitch), 5.0, dtype=torch.float32, device="cuda")
        if mode == "fwd":
            W = torch.randn(cs, 64, 3, 3, device="cuda") * 0.1
            b = torch.randn(cs, device="cuda")
            want = F.conv2d(big_r, W.to(dt).float(), b, padding=1)
            a = ops.thin_args(ops.DT[dt], n, h, w, cs, W, A.view(big), w_big_is_cout=False, bias=b, thin_out=out, thin_out_pitch=pitch)
        else:
            W = torch.randn(64, cs, 3, 3, device="cuda") * 0.1
            want = F.conv_transpose2d(big_r, W.to(dt).float(), padding=1)
            a = ops.thin_args(ops.DT[dt], n, h, w, cs, W, A.view(big), w_big_is_cout=True, flip=True, thin_out=out, thin_out_pitch=pitch)
        ops.thin_out(a)
        torch.cuda.synchronize()
        got = out[..., :cs].permute(0, 3, 1, 2)
        e = _rel(got, want)
        assert e < 2e-5, (mode, cs, (n, h, w), e)
        if pitch == 4 and cs < 4:
            assert (out[..., cs:] == 0).all()


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("cs", [1, 3, 4])
@pytest.mark.parametrize("big_is_cout", [True, False])
def test_thin_wgrad(dt, cs, big_is_cout):
    """weight + bias gradient of a cs -> 64 conv (big = dy) and of a 64 -> cs conv (big = x) against autograd; bitwise run-to-run"""
    from sr_gan_fd_amd import _abi as A, ops
    torch.manual_seed(3)
    ws = torch.empty(ops.thin_wgrad_workspace_bytes(), dtype=torch.uint8, device="cuda")
    for (n, h, w) in SHAPES + [(4, 96, 160)]:
        small = torch.randn(n, cs, h, w, device="cuda")
        thin = _nhwc4(small, dt)
        small_r = thin[..., :cs].permute(0, 3, 1, 2).float()
        big = torch.randn(n, h, w, 64, device="cuda").to(dt)
        big_r = big.permute(0, 3, 1, 2).float()
        if big_is_cout:      # conv: small (cs) -> big (64); x = small, dy = big
            Wp = torch.zeros(64, cs, 3, 3, device="cuda", requires_grad=True)
            bp = torch.zeros(64, device="cuda", requires_grad=True)
            (F.conv2d(small_r, Wp, bp, padding=1) * big_r).sum().backward()
        else:                # conv: big (64) -> small (cs); x = big, dy = small
            Wp = torch.zeros(cs, 64, 3, 3, device="cuda", requires_grad=True)
            bp = torch.zeros(cs, device="cuda", requires_grad=True)
            (F.conv2d(big_r, Wp, bp, padding=1) * small_r).sum().backward()
        dw = torch.full_like(Wp, 3.0).detach()
        db = torch.full_like(bp, 3.0).detach()
        a = ops.thin_args(ops.DT[dt], n, h, w, cs, Wp.detach(), A.view(big), w_big_is_cout=big_is_cout, thin=thin)
        ops.thin_wgrad(a, dw, db, ws)
        dw2, db2 = torch.empty_like(dw), torch.empty_like(db)
        ops.thin_wgrad(a, dw2, db2, ws)
        torch.cuda.synchronize()
        e, eb = _rel(dw, Wp.grad), _rel(db, bp.grad)
        assert e < 2e-5 and eb < 2e-5, (cs, big_is_cout, (n, h, w), e, eb)
        assert torch.equal(dw, dw2) and torch.equal(db, db2)


def test_thin_argument_checks():
    from sr_gan_fd_amd import _abi as A, ops
    y = torch.empty(1, 8, 8, 64, dtype=torch.float16, device="cuda")
    thin = torch.zeros(1, 8, 8, 4, dtype=torch.float16, device="cuda")
    W = torch.zeros(64, 3, 3, 3, device="cuda")
    with pytest.raises(A.SrganfdError, match="16-bit"):
        ops.thin_in(ops.thin_args(A.F32, 1, 8, 8, 3, W, A.view(y), w_big_is_cout=True, thin=thin))
    with pytest.raises(A.SrganfdError, match="thin channels"):
        ops.thin_in(ops.thin_args(A.F16, 1, 8, 8, 5, W, A.view(y), w_big_is_cout=True, thin=thin))
    with pytest.raises(A.SrganfdError, match="out of range"):
        ops.thin_in(ops.thin_args(A.F16, 1, 8, 8, 3, W, A.view(y, c0=32), w_big_is_cout=True, thin=thin))
    with pytest.raises(A.SrganfdError, match="pitch"):
        ops.thin_out(ops.thin_args(A.F16, 1, 8, 8, 3, W, A.view(y), w_big_is_cout=False, thin_out=torch.empty(64, device="cuda"), thin_out_pitch=3))
