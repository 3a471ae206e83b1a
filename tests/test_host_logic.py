"""CPU-only host-logic tests: the C ABI loads and exports every declared symbol, argument validation
works, and the engines' launch plans (views, channel slices, shapes, workspaces) are self-consistent.
Kernels are NOT executed here (library dry-run mode): numerics are covered by the -m gpu tests."""
import ctypes as C
import re
import os

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_header_symbol():
    from sr_gan_fd_amd import _abi as A
    hdr = open(os.path.join(ROOT, "include", "srganfd.h")).read()
    declared = set(re.findall(r"\b(srganfd_[a-z0-9_]+)\s*\(", hdr))
    lib = C.CDLL(A.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/srganfd.h but not exported"
    assert declared == set(A.SYMBOLS), f"binding/header mismatch: {declared ^ set(A.SYMBOLS)}"
    assert A.lib().srganfd_abi_version() == 1


def test_product_path_fails_loudly_without_gpu():
    from sr_gan_fd_amd import _abi as A, model as M
    net = M.bsrgan_x4(num_rrdb=1)
    with pytest.raises(A.SrganfdError):
        net(torch.rand(1, 3, 8, 8))


def test_conv_argument_validation():
    from sr_gan_fd_amd import _abi as A, ops
    A.set_dry_run(True)
    try:
        x = torch.zeros(1, 8, 8, 64)
        y = torch.zeros(1, 8, 8, 32)
        w = torch.zeros(64 * 32 * 9 * 2, dtype=torch.uint8)
        ok = ops.conv_args(A.BF16, A.view(x), A.view(y), w, 1, 8, 8, 64, 32)
        ops.conv2d(ok)
        bad = ops.conv_args(A.BF16, A.view(x), A.view(y), w, 1, 8, 8, 48, 32)   # cin not a multiple of 32
        with pytest.raises(A.SrganfdError):
            ops.conv2d(bad)
        bad = ops.conv_args(A.BF16, A.view(x, c0=40), A.view(y), w, 1, 8, 8, 64, 32)  # view exceeds buffer
        with pytest.raises(A.SrganfdError):
            ops.conv2d(bad)
        bad = ops.conv_args(A.BF16, A.view(x), A.view(y), w, 1, 8, 8, 64, 32)
        bad.h_out = 9
        with pytest.raises(A.SrganfdError):
            ops.conv2d(bad)
    finally:
        A.set_dry_run(False)


@pytest.mark.parametrize("fac,kw", [("bsrgan_x4", dict(num_rrdb=2)), ("bsrgan_x2", dict(num_rrdb=1)),
                                    ("rrdbnet_x1", dict(num_blocks=1)), ("rrdbnet_x8", dict(num_blocks=1))])
def test_generator_plan_dry_run(fac, kw):
    """forward + backward launch lists validate against the C ABI's shape checks (no kernels run)"""
    from sr_gan_fd_amd import _abi as A, model as M
    A.set_dry_run(True)
    try:
        net = getattr(M, fac)(**kw)
        for dt in (torch.float32, torch.bfloat16):
            net.compute_dtype = dt
            x = torch.rand(2, 3, 12, 20)
            sr = net(x)
            s = net.upscale_factor if fac.startswith("rrdb") else max(net.upscale_factor, 2)
            assert sr.shape == (2, 3, 12 * s, 20 * s)
            sr.sum().backward()
            for n, p in net.named_parameters():
                assert p.grad is not None and p.grad.shape == p.shape, n
            net.zero_grad(set_to_none=True)
            with torch.no_grad():
                assert net(x).shape == sr.shape
    finally:
        A.set_dry_run(False)


def test_flat_params_alias_module_parameters():
    from sr_gan_fd_amd import _abi as A, model as M, engine as E
    A.set_dry_run(True)
    try:
        net = M.bsrgan_x4(num_rrdb=1)
        sd0 = {k: v.clone() for k, v in net.state_dict().items()}
        net(torch.rand(1, 3, 8, 8))
        eng = E.generator_engine(net)
        flat = eng.fp.flat
        for (n, p), o in zip(net.named_parameters(), eng.fp.offsets):
            assert p.data_ptr() == flat.data_ptr() + 4 * o
            assert torch.equal(p.detach(), sd0[n]), n        # flattening must not change values
        assert list(net.state_dict().keys()) == list(sd0.keys())
    finally:
        A.set_dry_run(False)
