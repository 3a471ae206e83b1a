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
    # declarations inside `#ifdef SRGANFD_EXPERIMENT` belong to the timing-experiment builds only (tools/build_variant.sh)
    experiment = re.findall(r"#ifdef SRGANFD_EXPERIMENT(.*?)#endif", hdr, re.S)
    product_hdr = re.sub(r"#ifdef SRGANFD_EXPERIMENT.*?#endif", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(srganfd_[a-z0-9_]+)\s*\(", product_hdr))
    lib = C.CDLL(A.LIB_PATH)
    for name in re.findall(r"\b(srganfd_[a-z0-9_]+)\s*\(", " ".join(experiment)):
        assert not hasattr(lib, name), f"{name} (experiment hook) must not be exported by the product library"
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/srganfd.h but not exported"
    assert declared == set(A.SYMBOLS), f"binding/header mismatch: {declared ^ set(A.SYMBOLS)}"
    assert A.lib().srganfd_abi_version() == A.ABI_VERSION == A._ABI_VERSION_BUILT == 7
    assert A.lib().srganfd_get_mfma16() == 3


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


@pytest.mark.parametrize("which", ["unet", "aesrgan", "esrgan"])
def test_discriminator_plans_dry_run(which):
    """the three discriminators' forward / backward launch lists pass the C ABI's argument checks (no kernels run)"""
    from sr_gan_fd_amd import _abi as A, model as M
    A.set_dry_run(True)
    try:
        d = {"unet": lambda: M.discriminator_unet(in_channels=3, out_channels=1, channels=64), "aesrgan": M.uNetDiscriminatorAesrgan,
             "esrgan": M.discriminator}[which]()
        size = 128 if which == "esrgan" else 64
        for dt in (torch.float32, torch.bfloat16):
            d.compute_dtype = dt
            d.train()
            x = torch.rand(2, 3, size, size, requires_grad=True)
            out = d(x)
            assert out.shape == ((2, 1) if which == "esrgan" else (2, 1, size, size))
            out.sum().backward()
            assert x.grad is not None and x.grad.shape == x.shape
            for n, p in d.named_parameters():
                assert p.grad is not None and p.grad.shape == p.shape, n
            d.zero_grad(set_to_none=True)
        if which == "esrgan":
            with pytest.raises(A.SrganfdError):
                d(torch.rand(1, 3, 64, 64))            # the classifier fixes the input size (ESRGAN/model.py:129)
    finally:
        A.set_dry_run(False)


def test_content_losses_and_fused_trainers_dry_run():
    from sr_gan_fd_amd import _abi as A, model as M
    from sr_gan_fd_amd.gan import GanTrainer
    from sr_gan_fd_amd.trainer import GeneratorTrainer
    nodes, mean, std = ["features.2", "features.7", "features.16", "features.25", "features.34"], [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]
    A.set_dry_run(True)
    try:
        cl5 = M.ContentLoss(nodes, mean, std)
        cl1 = M.ContentLoss("features.34", mean, std)
        sr = torch.rand(2, 3, 32, 48, requires_grad=True)
        gt = torch.rand(2, 3, 32, 48)
        v5 = cl5(sr, gt)
        assert v5.shape == (1, 5) and not v5.requires_grad           # detached like torch.Tensor([losses]) (model.py:552)
        v1 = cl1(sr, gt)
        assert v1.dim() == 0 and v1.requires_grad                    # ESRGAN's stays in the graph
        v1.backward()
        assert sr.grad.shape == sr.shape
        with pytest.raises(A.SrganfdError):
            cl1(torch.rand(1, 3, 30, 32), torch.rand(1, 3, 30, 32))   # four 2x2 pools
        g = M.bsrgan_x4(num_rrdb=1)
        assert g.compute_dtype is None
        t = GeneratorTrainer(g, lr=1e-4)
        # a trainer built over default modules, outside autocast, trains as the reference's loop does:
        # float16 with the loss scaler on -- not in the exact-fp32 parity mode the modules run in outside autocast
        assert g.compute_dtype == torch.float16 and t.scaler.enabled
        assert t.step(torch.rand(2, 3, 16, 16), torch.rand(2, 3, 64, 64)).shape == (1,)
        for dfac in (lambda: M.discriminator_unet(in_channels=3, out_channels=1, channels=64), M.uNetDiscriminatorAesrgan):
            tr = GanTrainer(M.bsrgan_x4(num_rrdb=1), dfac(), M.ContentLoss(nodes, mean, std))
            assert tr.g.compute_dtype == tr.d.compute_dtype == tr.content.compute_dtype == torch.float16 and tr.scaler.enabled
            assert tr.step(torch.rand(2, 3, 16, 16), torch.rand(2, 3, 64, 64)).shape == (8,)
            assert tr.content_vals.shape == (1, 5)
        # an explicit dtype is kept (the parity tests' float32, configs[0]), and a bfloat16 autocast region is followed
        g32 = M.bsrgan_x4(num_rrdb=1)
        g32.compute_dtype = torch.float32
        assert not GeneratorTrainer(g32, lr=1e-4).scaler.enabled and g32.compute_dtype == torch.float32
        if torch.cuda.is_available():
            with torch.autocast("cuda", dtype=torch.bfloat16):
                gb = M.bsrgan_x4(num_rrdb=1)
                assert not GeneratorTrainer(gb, lr=1e-4).scaler.enabled and gb.compute_dtype == torch.bfloat16
    finally:
        A.set_dry_run(False)


def test_planar_views_are_validated():
    """srganfd_view.planar (32-channel group planes): conv2d / conv2d_wgrad accept it for aligned channel ranges only"""
    from sr_gan_fd_amd import _abi as A, ops
    A.set_dry_run(True)
    try:
        L = A.lib()
        x = torch.zeros(1, 8, 8, 192)
        y = torch.zeros(1, 8, 8, 192)
        wp = torch.zeros(ops.packed_bytes(A.BF16, 3, 96, 32), dtype=torch.uint8)
        ok = ops.conv_args(A.BF16, A.view(x, planar=1), A.view(y, c0=96, planar=1), wp.data_ptr(), 1, 8, 8, 96, 32)
        assert L.srganfd_conv2d(C.byref(ok), None) == 0
        bad = ops.conv_args(A.BF16, A.view(x, planar=1), A.view(y, c0=104, planar=1), wp.data_ptr(), 1, 8, 8, 96, 32)
        assert L.srganfd_conv2d(C.byref(bad), None) != 0 and b"planar" in L.srganfd_last_error()
        f32out = ops.conv_args(A.BF16, A.view(x, planar=1), A.view(torch.zeros(1, 8, 8, 32), planar=1), wp.data_ptr(), 1, 8, 8, 96, 32, y_f32=True)
        assert L.srganfd_conv2d(C.byref(f32out), None) != 0
    finally:
        A.set_dry_run(False)


def test_validation_side_argument_checks():
    from sr_gan_fd_amd import _abi as A
    A.set_dry_run(True)
    try:
        L = A.lib()
        buf = torch.zeros(64)
        p = buf.data_ptr()
        assert L.srganfd_crop_nchw(p, p, 1, 3, 8, 8, 2, 2, 4, 4, None) == 0
        assert L.srganfd_crop_nchw(p, p, 1, 3, 8, 8, 6, 2, 4, 4, None) != 0          # window leaves the image
        assert L.srganfd_psnr(p, p, 1, 3, 8, 8, 2, 1, p, p, None) == 0
        assert L.srganfd_psnr(p, p, 1, 1, 8, 8, 2, 1, p, p, None) != 0              # luma needs RGB
        assert L.srganfd_psnr(p, p, 1, 3, 8, 8, 4, 0, p, p, None) != 0              # nothing left after the crop
        assert L.srganfd_ssim_workspace_doubles(2, 3, 40, 56, 4, 1, 11) == 2 * 1 * 2 * 3   # 22x38 map -> 2x3 tiles of 16x16
        assert L.srganfd_ssim_workspace_doubles(2, 3, 40, 56, 4, 0, 11) == 2 * 3 * 2 * 3
        assert L.srganfd_ssim_workspace_doubles(1, 3, 12, 12, 1, 1, 11) == 0               # window larger than the cropped image
        assert L.srganfd_ssim(p, p, 1, 3, 16, 16, 2, 1, p, 11, p, p, None) == 0
        assert L.srganfd_ssim(p, p, 1, 3, 12, 12, 1, 1, p, 11, p, p, None) != 0            # window does not fit
        assert L.srganfd_ssim(p, p, 1, 1, 16, 16, 0, 1, p, 11, p, p, None) != 0            # luma needs RGB
        assert L.srganfd_ssim(p, p, 1, 3, 32, 32, 0, 0, p, 17, p, p, None) != 0            # window above the kernel's LDS tile
        # on-device degradation stages (Real_ESRGAN/imgproc.py)
        assert L.srganfd_filter2d(p, p, 1, 2, 3, 32, 32, 21, p, None) == 0
        assert L.srganfd_filter2d(p, p, 2, 2, 3, 32, 32, 21, p, None) == 0
        assert L.srganfd_filter2d(p, p, 1, 2, 3, 32, 32, 4, p, None) != 0 and b"Wrong kernel size." in L.srganfd_last_error()
        assert L.srganfd_filter2d(p, p, 3, 2, 3, 32, 32, 21, p, None) != 0                   # 3 kernels for 2 images
        assert L.srganfd_filter2d(p, p, 1, 2, 3, 10, 32, 21, p, None) != 0                   # reflect padding 10 needs > 10 rows
        assert L.srganfd_filter2d(p, p, 1, 2, 3, 64, 64, 53, p, None) != 0                   # above the LDS tile
        assert L.srganfd_filter2d_separable(p, p, 1, 2, 3, 64, 64, 51, p, None) == 0
        assert L.srganfd_usm_sharp(p, p, 0, 1, 3, 64, 64, 51, 0.5, 10.0, p, p, None) == 0 and L.srganfd_usm_sharp(p, p, 1, 1, 3, 64, 64, 51, 0.5, 10.0, p, p, None) == 0
        assert L.srganfd_usm_sharp(p, p, 0, 1, 3, 64, 64, 51, 0.5, 10.0, p, None, None) != 0    # workspace required
        assert L.srganfd_diff_jpeg_table_floats() == 2 * 4096 + 4 * 64
        assert L.srganfd_diff_jpeg(p, 2, 3, 17, 33, p, 0, 0, p, p, None) == 0
        assert L.srganfd_diff_jpeg(p, 2, 1, 16, 16, p, 0, 0, p, p, None) != 0                # RGB only
        assert L.srganfd_resize(p, 6, 32, 32, 8, 8, 0, 0.0, 0.0, p, None) == 0
        assert L.srganfd_resize(p, 6, 32, 32, 8, 8, 3, 0.0, 0.0, p, None) != 0               # unknown mode
        assert L.srganfd_resize(p, 6, 32, 32, 0, 8, 1, 0.0, 0.0, p, None) != 0
        assert L.srganfd_gaussian_noise(p, p, None, p, None, 2, 3, 8, 8, 1, 0, p, None) == 0
        assert L.srganfd_gaussian_noise(p, p, p, p, None, 2, 3, 8, 8, 1, 0, p, None) != 0    # grey field without the grey flags
        assert L.srganfd_poisson_prepare(p, 2, 3, 8, 8, 0, p, None, p, None, p, None) == 0
        assert L.srganfd_poisson_prepare(p, 2, 1, 8, 8, 1, p, p, p, p, p, None) != 0         # grey needs RGB
        assert L.srganfd_poisson_apply(p, p, None, p, None, p, None, p, None, 2, 3, 8, 8, 1, 0, p, None) == 0
        assert L.srganfd_poisson_apply(p, p, None, p, p, p, None, p, None, 2, 3, 8, 8, 1, 0, p, None) != 0
        assert L.srganfd_quantize_u8(p, p, 64, None) == 0 and L.srganfd_quantize_u8(p, p, 0, None) != 0
        assert L.srganfd_crop_rot_flip(p, p, 6, 8, 8, 2, 2, 4, 4, 1, None) == 0
        assert L.srganfd_crop_rot_flip(p, p, 6, 8, 8, 2, 2, 4, 6, 1, None) != 0              # quarter turn of a non-square window
        assert L.srganfd_crop_rot_flip(p, p, 6, 8, 8, 6, 2, 4, 4, 0, None) != 0              # window leaves the image
        assert L.srganfd_crop_rot_flip(p, p, 6, 8, 8, 0, 0, 8, 8, 6, None) != 0              # unknown op
    finally:
        A.set_dry_run(False)


def test_modules_deepcopy_and_pickle_like_the_train_scripts_do():
    """AveragedModel deep-copies the generator (train_bsrgan.py:291) and mlflow.pytorch.log_model pickles both networks
    (:203-213): copies must carry the same state_dict, own their parameters, and still run"""
    import copy
    import pickle
    from sr_gan_fd_amd import _abi as A, model as M
    A.set_dry_run(True)
    try:
        nodes, mean, std = ["features.2", "features.34"], [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]
        mods = [M.bsrgan_x4(num_rrdb=1), M.rrdbnet_x4(num_blocks=1), M.discriminator_unet(in_channels=3, out_channels=1, channels=64),
                M.uNetDiscriminatorAesrgan(), M.discriminator(), M.ContentLoss(nodes, mean, std)]
        for m in mods:
            if not isinstance(m, M.ContentLoss):
                m(torch.rand(1, 3, 128, 128) if isinstance(m, M.Discriminator) else torch.rand(1, 3, 64, 64))   # build the engine first
            for clone in (copy.deepcopy(m), pickle.loads(pickle.dumps(m))):
                assert type(clone) is type(m)
                sd, sc = m.state_dict(), clone.state_dict()
                assert list(sd) == list(sc) and all(torch.equal(sd[k], sc[k]) for k in sd)
                p0, c0 = next(m.parameters()), next(clone.parameters())
                assert p0.data_ptr() != c0.data_ptr()
                with torch.no_grad():
                    c0.add_(1.0)
                assert not torch.equal(p0, c0)                       # no aliasing of the flat parameter buffers
                if isinstance(m, M.ContentLoss):
                    assert clone(torch.rand(1, 3, 32, 32), torch.rand(1, 3, 32, 32)).shape == (1, 2)
                elif isinstance(m, M.Discriminator):
                    assert clone(torch.rand(1, 3, 128, 128)).shape == (1, 1)
                else:
                    clone(torch.rand(1, 3, 64, 64))
    finally:
        A.set_dry_run(False)


def test_engines_release_their_buffers_by_reference_counting():
    """A plan closure that captures its engine or its own plan object is a reference cycle: the activation buffers (hundreds of
    GB at the benchmark sizes) then outlive the module until a cyclic collection that HIP allocations never trigger -- an
    out-of-memory error between two full-size tests.  After dropping the modules, the cyclic collector must find no tensor
    and no engine."""
    import gc
    from sr_gan_fd_amd import _abi as A, model as M
    from sr_gan_fd_amd.gan import GanTrainer
    nodes, mean, std = ["features.2", "features.7", "features.16", "features.25", "features.34"], [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]
    A.set_dry_run(True)
    gc.collect()
    gc.disable()
    try:
        for dfac in (lambda: M.discriminator_unet(in_channels=3, out_channels=1, channels=64), M.uNetDiscriminatorAesrgan, M.discriminator):
            g, d, cl, cl1 = M.bsrgan_x4(num_rrdb=1), dfac(), M.ContentLoss(nodes, mean, std), M.ContentLoss("features.34", mean, std)
            if dfac is M.discriminator:
                x = torch.rand(2, 3, 128, 128, requires_grad=True)
                d(x).sum().backward()
                sr = torch.rand(2, 3, 32, 32, requires_grad=True)
                cl1(sr, torch.rand(2, 3, 32, 32)).backward()
                y = g(torch.rand(2, 3, 16, 16))
                y.sum().backward()
                del x, sr, y
            else:
                tr = GanTrainer(g, d, cl)
                tr.step(torch.rand(2, 3, 16, 16), torch.rand(2, 3, 64, 64))
                del tr
            del g, d, cl, cl1
            gc.set_debug(gc.DEBUG_SAVEALL)
            gc.collect()
            leaked = []
            for o in gc.garbage:
                name = type(o).__name__            # (type(), not isinstance(): dead weakref proxies sit in the garbage list too)
                if type(o) is torch.Tensor or "Engine" in name or name in ("_Shape", "FlatParams"):
                    leaked.append(name)
            gc.garbage.clear()
            gc.set_debug(0)
            assert not leaked, f"kept alive only by reference cycles: {sorted(set(leaked))} ({len(leaked)} objects)"
    finally:
        gc.set_debug(0)
        gc.garbage.clear()
        gc.enable()
        A.set_dry_run(False)


def test_fused_optimizer_checkpoint_round_trip(tmp_path):
    """FlatAdamEMA.state_dict() / load_state_dict() speak torch.optim.Adam's per-parameter format and AveragedModel's
    ``module.<name>`` + ``n_averaged`` layout (the reference's checkpoint, train_bsrnet.py:124-130): a state written by
    torch's own classes loads into the flat buffers and comes back identical."""
    from torch.optim.swa_utils import AveragedModel
    from sr_gan_fd_amd import model as M, utils
    from sr_gan_fd_amd.engine import FlatParams
    from sr_gan_fd_amd.trainer import FlatAdamEMA
    torch.manual_seed(0)
    net = M.bsrgan_x4(num_rrdb=1)
    opt = torch.optim.Adam(net.parameters(), 1e-4, (0.9, 0.99), 1e-4, 0.0)
    ema = AveragedModel(net, avg_fn=lambda a, p, n: 0.001 * a + 0.999 * p)
    for _ in range(3):
        for p in net.parameters():
            p.grad = torch.randn_like(p) * 1e-3
        opt.step()
        ema.update_parameters(net)
    path = str(tmp_path / "g_last.pth.tar")
    torch.save({"epoch": 7, "best_psnr": 1.0, "best_ssim": 2.0, "state_dict": net.state_dict(), "ema_state_dict": ema.state_dict(),
                "optimizer": opt.state_dict()}, path)

    net2 = M.bsrgan_x4(num_rrdb=1)
    fp = FlatParams(list(net2.named_parameters()))
    flat = fp.sync(torch.device("cpu"))
    fused = FlatAdamEMA(flat, 5e-5, (0.9, 0.999), 1e-8, 0.0, ema_decay=0.999, layout=fp)
    out = utils.load_state_dict(net2, path, ema_model=fused, optimizer=fused, load_mode="resume")
    assert out[2] == 7 and fused.t == 3 and fused.n_averaged == 3 and fused.lr == 1e-4 and fused.eps == 1e-4 and fused.betas == (0.9, 0.99)
    for (k, a), b in zip(net.state_dict().items(), net2.state_dict().values()):
        assert torch.equal(a, b), k
    want = opt.state_dict()["state"]
    got = fused.state_dict()
    assert got["param_groups"][0]["params"] == list(range(len(want)))
    for i in want:
        assert torch.equal(want[i]["exp_avg"], got["state"][i]["exp_avg"]) and torch.equal(want[i]["exp_avg_sq"], got["state"][i]["exp_avg_sq"])
        assert float(got["state"][i]["step"]) == 3.0
    for k, v in ema.state_dict().items():
        assert torch.equal(v, fused.ema_state_dict()[k]), k
    # and back into torch's own classes (what the reference's resume path does with a file the fused trainer wrote)
    utils.save_checkpoint(str(tmp_path / "again.pth.tar"), net2, fused, ema=fused, epoch=8)
    net3 = M.bsrgan_x4(num_rrdb=1)
    opt3 = torch.optim.Adam(net3.parameters(), 1.0)
    ema3 = AveragedModel(net3)
    utils.load_state_dict(net3, str(tmp_path / "again.pth.tar"), ema_model=ema3, optimizer=opt3, load_mode="resume")
    assert opt3.state_dict()["param_groups"][0]["lr"] == 1e-4
    for i in want:
        assert torch.equal(opt3.state_dict()["state"][i]["exp_avg_sq"], want[i]["exp_avg_sq"])
    assert int(ema3.n_averaged) == 3


def test_plan_cache_keeps_training_plan():
    from sr_gan_fd_amd.engine import PlanCache
    c = PlanCache(cap=3, cap_pinned=2)
    c.put("train", "T", pinned=True)
    for i in range(10):
        c.put(("eval", i), i)
    assert c.get("train") == "T" and len(c) == 4 and c.get(("eval", 9)) == 9 and c.get(("eval", 0)) is None
    c.put("train2", "T2", pinned=True); c.put("train3", "T3", pinned=True)
    assert c.get("train") is None and c.get("train3") == "T3"


def test_multistep_lr_mirror_follows_torch_and_exchanges_state():
    """trainer.MultiStepLR == torch.optim.lr_scheduler.MultiStepLR (bsrgan_config.py:153-155 milestones / gamma, one step per
    epoch: train_bsrgan.py:193-195) on the fused optimizer, and the two load each other's state_dict (checkpoint key "scheduler")."""
    from sr_gan_fd_amd.trainer import FlatAdamEMA, MultiStepLR
    flat = torch.zeros(8)
    fused = FlatAdamEMA(flat, 8e-5, (0.9, 0.999), 1e-4)
    ref_p = torch.nn.Parameter(torch.zeros(8))
    ref_opt = torch.optim.Adam([ref_p], 8e-5)
    milestones, gamma = [2, 5, 5, 9], 0.5          # a repeated milestone applies gamma twice (torch keeps a Counter)
    mine, ref = MultiStepLR(fused, milestones, gamma), torch.optim.lr_scheduler.MultiStepLR(ref_opt, milestones, gamma)
    for epoch in range(12):
        assert abs(fused.lr - ref_opt.param_groups[0]["lr"]) < 1e-18 and mine.get_last_lr() == ref.get_last_lr(), epoch
        if epoch == 6:
            # resume in the middle: each implementation continues from the other's state
            fused2 = FlatAdamEMA(torch.zeros(8), 1.0, (0.9, 0.999), 1e-4)
            mine2 = MultiStepLR(fused2, [1], 0.1)
            mine2.load_state_dict(ref.state_dict())
            ref_opt2 = torch.optim.Adam([torch.nn.Parameter(torch.zeros(8))], 1.0)
            ref2 = torch.optim.lr_scheduler.MultiStepLR(ref_opt2, [1], 0.1)
            ref2.load_state_dict(mine.state_dict())
            ref_opt2.param_groups[0]["lr"] = ref2.get_last_lr()[0]      # torch restores the rate with the optimizer's own state
            assert fused2.lr == fused.lr and mine2.last_epoch == ref.last_epoch
        ref_opt.step()
        mine.step()
        ref.step()
        if epoch >= 6:
            ref_opt2.step()
            mine2.step()
            ref2.step()
            assert abs(fused2.lr - fused.lr) < 1e-18 and abs(ref_opt2.param_groups[0]["lr"] - fused.lr) < 1e-18, epoch
