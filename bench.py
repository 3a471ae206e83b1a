#!/usr/bin/env python3
"""bench.py -- SR training images/sec of the MI355X-native hot path (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload g_only|gan|aesrgan_gan] [--batch B] [--lr-size S]

One "step" = one training iteration of the reference (``train()`` body) on one synthetic batch that is
already resident in HBM.  Default run = BASELINE.json configs[1] AND configs[2] in one process: the headline fields are the
BSRGAN RRDBNet x4 (23 RRDB) generator-only step (L1 pixel loss, batch 32 per GPU, 128x128 -> 512x512, 16-bit MFMA with fp32 master
weights, Adam + EMA inside the timed region); the object under ``"gan"`` is the full GAN step of train_bsrgan.py:387-483 (U-Net
discriminator + VGG-19 content loss, same batch) timed the same way right after it (``--workload g_only`` / ``gan`` time one only).
N > 1: one process per GPU, weak scaling (batch 32 per GPU), one RCCL all-reduce of the flat gradient per network and step.  Under
torchrun (RANK / WORLD_SIZE set) the process is one rank; WITHOUT them ``--gpus N`` starts the N ranks itself -- fresh child
processes through torch.distributed.run, before this process has touched a GPU -- and exits non-zero if fewer than N GPUs are
visible.  ``--workload aesrgan_gan`` = configs[4] per GPU (RRDBNet + A-ESRGAN attention U-Net discriminator, 192 -> 768).

Prints ONE JSON line (rank 0) with the contract fields plus
  roofline     -- dominant kernel class (largest total time) from HIP-event brackets around its launches in the timed
                  region, against the roof its arithmetic intensity puts it under: algorithmic bytes / time vs the
                  8 TB/s HBM peak when FLOP/byte is below the ridge (312), else algorithmic FLOP / time vs the dense
                  bf16 MFMA peak; both fractions are in the object (events bracket every 7th launch of a class: timing every
                  launch serialised the queue and cost 8 % of the step),
  sr_parity    -- PSNR (Y, 4-pixel border) and max abs error of the bf16 SR against the fp32 CPU oracle on one image,
  cpu_baseline -- the CPU oracle (oracle/srgan_oracle.py, torch-CPU fp32) timed on this host on a
                  bounded sample (batch 1) of the same workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0   # MI355X dense bf16 MFMA peak (MI355X_MICROARCH.md)
PEAK_HBM_GBPS = 8000.0      # HBM3E peak (same guide)
# algorithmic FLOP per image, SURVEY.md 8(d): 1 MAC = 2 FLOP, backward = 2x forward
# esrgan_gan (ESRGAN/train_esrgan.py:364-431 at 32 -> 128, the discriminator's fixed input size): generator fwd+bwd
# 1762.3/16 GFLOP, five discriminator forwards + three backwards of ~1.0 GFLOP each, VGG-19[:35] fwd on SR and GT + bwd on SR
FLOP_PER_IMG = {"g_only": 1762.3e9, "gan": 3828.9e9, "aesrgan_gan": 10566.0e9, "esrgan_gan": 1762.3e9 / 16 + 11 * 1.0e9 + 4 * 5.1e9,
                "realesrgan_gan": 3828.9e9 / 4}      # the BSRGAN GAN networks at 64 -> 256 (realesrgan_config.py:116); degradation FLOPs not counted


def flop_per_image(workload: str, h: int, upscale: int, num_rrdb: int) -> float:
    """Algorithmic FLOP of one training iteration per image for an LR side ``h`` and scale factor ``upscale`` (SURVEY 8(d) convention).
    At the BASELINE shapes (x4) this is FLOP_PER_IMG scaled by area; for the x2 networks of bsrgan_config.py:62 / aesrgan_config.py:62
    the generator's tail has ONE nearest-x2 stage (BSRGAN/model.py:335-352), so the generator term is rebuilt from the layer list:
    MAC per LR pixel = 3->64 (1728) + num_rrdb x 3 x 239,616 + 64->64 (36,864) + upsampling convs at 2h (and 4h) + conv3 + conv4 at s*h."""
    if upscale == 4:
        return FLOP_PER_IMG[workload] * (h / BASE_LR_SIZE[workload]) ** 2 * (num_rrdb / 23.0 if workload == "g_only" else 1.0)
    if upscale != 2 or workload not in ("g_only", "gan", "aesrgan_gan"):
        raise SystemExit("--upscale 2: g_only / gan / aesrgan_gan only")
    c = 36864
    g_mac = (1728 + num_rrdb * 3 * 239616 + c + 4 * c + 4 * c + 4 * 1728) * h * h
    g_train = 3 * 2 * g_mac
    hr = (2 * h) ** 2
    d_mac_px = {"gan": 103.7e9 / 512 ** 2, "aesrgan_gan": 355.3e9 / 768 ** 2}.get(workload, 0.0)
    vgg_mac_px = 101.9e9 / 512 ** 2                                                                       # row A7
    return g_train if workload == "g_only" else g_train + 8 * 2 * d_mac_px * hr + 2 * 2 * vgg_mac_px * hr


BASE_LR_SIZE = {"g_only": 128, "gan": 128, "aesrgan_gan": 192, "esrgan_gan": 32, "realesrgan_gan": 64}     # the input size those figures are quoted at


# Counter-derived fields of the roofline object (HBM-side traffic per launch, MFMA-pipe utilisation) come from
# profiles/r05_pmc.json, which tools/pmc_to_json.py writes from separate `rocprofv3 --pmc` passes of this command (the guide's
# gfx950 rule: 16-byte-per-lane reads are tallied at half their bytes, so traffic = (2 x RDREQ + WRREQ) x 64 B; MFMA utilisation =
# SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 256 CUs x 4 SIMDs)).  The file records the SHA of the kernel sources it was
# measured on; when the sources differ the fields are null instead of stale.
PMC_FILE = os.path.join(ROOT, "profiles", "r05_pmc.json")


def csrc_sha() -> str:
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "sr_gan_fd_amd", "csrc", "*.h*"))) + [os.path.join(ROOT, "include", "srganfd.h")]:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def pmc_fields(workload: str, kernel: str, B: int, h: int) -> dict:
    try:
        rec = json.load(open(PMC_FILE))
    except Exception:
        return {"traffic": None, "mfma_busy_pmc": None, "pmc_source": None}
    if rec.get("csrc_sha") != csrc_sha() or rec.get("batch") != B or rec.get("lr_size") != h:
        return {"traffic": None, "mfma_busy_pmc": None, "pmc_source": "profiles/r05_pmc.json is from other kernel sources / another shape"}
    w = rec.get("workloads", {}).get(workload, {}).get(kernel, {})
    return {"traffic": w.get("traffic_bytes_per_launch"), "mfma_busy_pmc": w.get("mfma_util"), "pmc_source": "profiles/r05_pmc.json (counters taken on another box of the pool, tools/profile_round.sh)"}


REALESRGAN_DEGRADATION = dict(      # realesrgan_config.py:67-90
    first_blur_probability=1.0, resize_probability1=[0.2, 0.7, 0.1], resize_range1=[0.15, 1.5], gray_noise_probability1=0.4,
    gaussian_noise_probability1=0.5, noise_range1=[1, 30], poisson_scale_range1=[0.05, 3], jpeg_range1=[30, 95],
    second_blur_probability=0.8, resize_probability2=[0.3, 0.4, 0.3], resize_range2=[0.3, 1.2], gray_noise_probability2=0.4,
    gaussian_noise_probability2=0.5, noise_range2=[1, 25], poisson_scale_range2=[0.05, 2.5], jpeg_range2=[30, 95])
NODES = ["features.2", "features.7", "features.16", "features.25", "features.34"]   # bsrgan_config.py:130-132
MEAN, STD = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]


def conv_flops(a) -> float:
    """algorithmic FLOP of one fused-conv launch: 2 * pixels_out * k*k * cin * cout_store"""
    return 2.0 * a.n * a.h_out * a.w_out * a.ksize * a.ksize * a.cin * a.cout_store * (4 if a.out_classes == 4 else 1)


def log(msg):
    """progress on stderr (rank 0 prints the single JSON result line on stdout)"""
    print("[bench %6.1fs] %s" % (time.perf_counter() - _T0, msg), file=sys.stderr, flush=True)


_T0 = time.perf_counter()


def host_cores() -> int:
    """CPU share of this process: min(affinity, cgroup quota); capped at 16 (the GPU box's share per GPU)"""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except Exception:
        pass
    return max(1, min(n, 16))


def visible_gpus() -> int:
    """GPUs this process may use, counted WITHOUT touching the HIP runtime (the parent must not initialise a GPU before it starts its ranks):
    KFD topology nodes that have SIMDs, narrowed by HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES when set.  Falls back to
    torch.cuda.device_count() (which does not initialise the GPU on this image) where the topology is not readable."""
    import glob
    n = 0
    try:
        for f in glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties"):
            props = dict(line.split(None, 1) for line in open(f).read().splitlines() if " " in line)
            if int(props.get("simd_count", "0")) > 0:
                n += 1
    except Exception:
        n = 0
    if n == 0:
        import torch
        return torch.cuda.device_count()
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([t for t in v.split(",") if t.strip() != ""]))
    return n


def self_launch(args) -> int:
    """``--gpus N`` without torchrun's environment: start the N ranks as fresh child processes (torch.distributed.run) BEFORE this
    process initialises any GPU, pass rank 0's JSON line through, return the children's exit status."""
    import subprocess
    n_vis = visible_gpus()
    if args.dist_backend == "nccl" and n_vis < args.gpus:
        print("bench.py: --gpus %d but only %d GPU(s) are visible" % (args.gpus, n_vis), file=sys.stderr)
        return 2
    if args.dist_backend == "gloo" and n_vis < 1:
        print("bench.py: no GPU visible", file=sys.stderr)
        return 2
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log("starting %d ranks: %s" % (args.gpus, " ".join(cmd)))
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="both", choices=["both", "g_only", "gan", "aesrgan_gan", "esrgan_gan", "realesrgan_gan"],
                    help="both (default) = configs[1] generator-only as the headline + configs[2] full GAN step under \"gan\"")
    ap.add_argument("--batch", type=int, default=32, help="images per GPU")
    ap.add_argument("--esrgan-module-loop", action="store_true", help="esrgan_gan: run the script's own autograd loop over the drop-in modules instead of the fused trainer")
    ap.add_argument("--lr-size", type=int, default=0, help="LR image side (default 128; 192 for aesrgan_gan)")
    ap.add_argument("--upscale", type=int, default=4, choices=[2, 4],
                    help="generator scale factor: 4 = the BASELINE configs; 2 = bsrgan_config.py:62 / aesrgan_config.py:62 (reference-default shapes)")
    ap.add_argument("--num-rrdb", type=int, default=23)
    ap.add_argument("--dtype", default="f16", choices=["f16", "bf16", "f32"],
                    help="compute dtype of activations / packed weights / gradients (fp32 master weights and accumulation); f16 is what the "
                         "reference's amp.autocast() computes in and the mode whose SR meets the 1e-3 tolerance (with its GradScaler)")
    ap.add_argument("--module-loop", action="store_true",
                    help="g_only / gan: time the reference's own loop statements (torch.optim.Adam, AveragedModel, amp.autocast + GradScaler, "
                         "autograd) over the drop-in modules instead of the fused trainer; the default run reports both")
    ap.add_argument("--no-module-loop", action="store_true", help="default run: skip the module-level legs under \"extra\"")
    ap.add_argument("--dropin-optim", action="store_true", help="--module-loop with sr_gan_fd_amd.optim.Adam / swa_utils.AveragedModel behind the scripts' names")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-bf16", action="store_true", help="default run: skip the bf16 leg (\"bf16\" sub-object)")
    ap.add_argument("--no-kernel-events", action="store_true")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="collective backend for N > 1 (nccl = RCCL; gloo only to rehearse the multi-rank path on one GPU)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch one rank per GPU, or drop the torchrun environment)" % (args.gpus, world))
    if args.dist_backend == "gloo":
        local_rank %= max(1, torch.cuda.device_count())     # rehearsal: several ranks may share one GPU
    elif local_rank >= torch.cuda.device_count():
        raise SystemExit("bench.py: rank %d has no GPU (%d visible)" % (local_rank, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    pg = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        pg = dist.group.WORLD
        ones = torch.ones(1, device=dev if args.dist_backend == "nccl" else "cpu")
        dist.all_reduce(ones)                                # every rank is really there: the sum of ones is the world size
        if int(ones.item()) != world or dist.get_world_size() != world:
            raise SystemExit("bench.py: all-reduce of ones gave %d, world size %d" % (int(ones.item()), world))

    workloads = ["g_only", "gan"] if args.workload == "both" else [args.workload]
    results = [run_workload(args, wl, rank, world, dev, pg, module_loop=args.module_loop, dropin_optim=args.dropin_optim) for wl in workloads]
    out = results[0]
    if len(results) > 1 and rank == 0:
        out["gan"] = {k: results[1][k] for k in ("value", "unit", "ms_per_step", "config", "step_tflops_per_gpu", "loss_scale", "last_step_scalars", "roofline", "kernel_classes",
                                                 "exposed_comm_ms_per_step")
                      if k in results[1]}
    if args.workload == "both" and world == 1 and not args.module_loop and not args.no_module_loop:
        # the module-level (drop-in) path next to the fused trainers: the reference's loop statements over the mirror modules, a short
        # timed region of its own (train_bsrnet.py:244-272, train_bsrgan.py:387-483)
        import copy
        margs = copy.copy(args)
        margs.steps, margs.warmup, margs.no_kernel_events = min(args.steps, 5), min(args.warmup, 2), True
        extra = {}
        for wl, fused in zip(workloads, results):
            r = run_workload(margs, wl, rank, world, dev, pg, module_loop=True)
            r2 = run_workload(margs, wl, rank, world, dev, pg, module_loop=True, dropin_optim=True)
            extra[wl] = {"ms_per_step": r["ms_per_step"], "value": r["value"], "unit": r["unit"], "steps": margs.steps, "warmup": margs.warmup,
                         "fused_ms_per_step": fused["ms_per_step"], "module_over_fused": round(r["ms_per_step"] / fused["ms_per_step"], 4),
                         "loss_scale": r.get("loss_scale"), "last_step_scalars": r.get("last_step_scalars"),
                         "sections_ms": r.get("module_loop_sections_ms"),
                         "with_dropin_optim_and_ema": {"ms_per_step": r2["ms_per_step"], "value": r2["value"],
                                                       "module_over_fused": round(r2["ms_per_step"] / fused["ms_per_step"], 4),
                                                       "sections_ms": r2.get("module_loop_sections_ms"), "last_step_scalars": r2.get("last_step_scalars")}}
        # first-class field: what a user of the unchanged scripts gets.  "dropin_3_imports" = the documented
        # default (model + optim + AveragedModel imports), "model_import_only" = the one-line change
        out["module_loop"] = {wl: {"dropin_3_imports_ms_per_step": e["with_dropin_optim_and_ema"]["ms_per_step"],
                                   "dropin_3_imports_img_per_s": e["with_dropin_optim_and_ema"]["value"],
                                   "model_import_only_ms_per_step": e["ms_per_step"], "model_import_only_img_per_s": e["value"],
                                   "fused_trainer_ms_per_step": e["fused_ms_per_step"]} for wl, e in extra.items()}
        out["extra"] = {"module_loop": extra,
                        "what": "the reference's loop statements (amp.autocast + GradScaler, torch.optim.Adam, AveragedModel, autograd) over the drop-in "
                                "modules with nothing set on them; with_dropin_optim_and_ema = the same statements with sr_gan_fd_amd.optim.Adam / "
                                "sr_gan_fd_amd.swa_utils.AveragedModel behind the scripts' names; the headline fields are the fused trainers"}
    if args.workload == "both" and world == 1 and not args.module_loop and args.dtype == "f16" and not args.no_bf16:
        # BASELINE.json's metric string names bf16; the reference's autocast computes in f16, which is what the headline runs (bf16 misses the
        # 1e-3 SR tolerance).  The same generator-only step in bf16, a short timed region of its own, with its own SR parity figure.
        import copy
        bargs = copy.copy(args)
        bargs.dtype, bargs.steps, bargs.warmup, bargs.no_kernel_events = "bf16", min(args.steps, 8), min(args.warmup, 2), True
        rb = run_workload(bargs, "g_only", rank, world, dev, pg)
        out["bf16"] = {"ms_per_step": rb["ms_per_step"], "value": rb["value"], "unit": rb["unit"], "steps": bargs.steps, "warmup": bargs.warmup,
                       "step_mfma_frac": round(rb["step_tflops_per_gpu"] / PEAK_BF16_TFLOPS, 4), "last_step_scalars": rb.get("last_step_scalars"),
                       "what": "configs[1] in bfloat16 (no loss scaling needed); the headline dtype is float16 = the reference's amp.autocast() dtype"}
        if not args.no_cpu_baseline:
            out["bf16"]["sr_parity"] = sr_parity(args.lr_size or BASE_LR_SIZE["g_only"], args.num_rrdb, dev, "bf16")
    if rank == 0:
        h = args.lr_size or BASE_LR_SIZE[workloads[0]]
        if not args.no_cpu_baseline and world == 1:      # host-side legs: rank 0 of the single-GPU run only
            out["sr_parity"] = sr_parity(h, args.num_rrdb, dev, args.dtype)
            out["cpu_baseline"] = cpu_baseline(workloads[0], h, args.num_rrdb)
            if len(results) > 1:
                out["gan"]["cpu_baseline"] = cpu_baseline("gan", h, args.num_rrdb)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


DTYPES = {"f16": "float16", "bf16": "bfloat16", "f32": "float32"}


def run_workload(args, workload, rank, world, dev, pg, module_loop=False, dropin_optim=False):
    """W warm-up steps, then exactly K timed steps between barrier + synchronize on both sides; max over ranks."""
    import torch
    import torch.distributed as dist
    from sr_gan_fd_amd import model as M
    from sr_gan_fd_amd import ops, profiling
    from sr_gan_fd_amd.trainer import GeneratorTrainer

    cdt = getattr(torch, DTYPES[args.dtype])
    h = args.lr_size or BASE_LR_SIZE[workload]
    B = args.batch
    S = getattr(args, "upscale", 4)
    flop_img = flop_per_image(workload, h, S, args.num_rrdb)
    torch.manual_seed(0)                      # identical weights on every rank (bsrgan_config.py:35-37 seeds at import)
    g = (M.bsrgan_x4 if S == 4 else M.bsrgan_x2)(in_channels=3, out_channels=3, channels=64, growth_channels=32, num_rrdb=args.num_rrdb)
    g.compute_dtype = cdt
    g.to(dev)
    trainer = None
    if module_loop:
        if workload not in ("g_only", "gan") or world > 1:
            raise SystemExit("--module-loop: g_only / gan on one GPU")
        g.compute_dtype = None                    # the modules follow the loop's own autocast, as in the unchanged scripts
        step_fn = module_level_loop(workload, M, g, dev, cdt, dropin_optim)
    elif workload == "esrgan_gan":
        if world > 1 or h != 32:
            raise SystemExit("esrgan_gan: single GPU, 32 -> 128 only (the discriminator's classifier fixes the 128x128 input, ESRGAN/model.py:118-122)")
        if args.esrgan_module_loop:
            step_fn = esrgan_loop(M, g, dev, cdt)     # round-1/2 form: the script's own autograd loop over the drop-in modules
        else:
            # the fused trainer (gan_esrgan.py): every loss, seed and optimizer step a HIP kernel; esrgan_config.py:75-111 hyper-parameters
            from sr_gan_fd_amd.gan_esrgan import EsrganGanTrainer
            d = M.discriminator()
            cl = M.ContentLoss("features.34", MEAN, STD)          # a str node selects ESRGAN's differentiable single-tap loss
            d.compute_dtype = cl.compute_dtype = cdt
            d.to(dev).train()
            cl.to(dev)
            g.train()
            trainer = EsrganGanTrainer(g, d, cl)
            step_fn = trainer.step
    elif workload == "g_only":
        # BSRGAN/bsrnet_config.py:86-96 hyper-parameters
        trainer = GeneratorTrainer(g, lr=1e-4, betas=(0.9, 0.99), eps=1e-4, ema_decay=0.999, process_group=pg)
        step_fn = trainer.step
    else:
        from sr_gan_fd_amd.gan import GanTrainer
        aes = workload == "aesrgan_gan"
        d = M.uNetDiscriminatorAesrgan() if aes else M.discriminator_unet(in_channels=3, out_channels=1, channels=64)
        cl = M.ContentLoss(NODES, MEAN, STD)      # seeded random VGG-19 weights (no ImageNet download offline)
        d.compute_dtype = cl.compute_dtype = cdt
        d.to(dev)
        cl.to(dev)
        # bsrgan_config.py:137-151 defaults / aesrgan_config.py:137-155
        kw = dict(g_lr=5e-5, d_lr=1e-5, pixel_weight=10.0, adversarial_weight=0.1) if aes else {}
        if workload == "realesrgan_gan":
            # Real_ESRGAN/train_realesrgan.py:383-476 per batch: on-device second-order degradation of the GT batch, then the
            # generator-first GAN iteration (realesrgan_config.py:67-90, 138-151)
            from sr_gan_fd_amd import imgproc
            kw = dict(g_lr=1e-4, d_lr=1e-4, betas=(0.9, 0.99), pixel_weight=1.0, content_weight=[0.1, 0.1, 1.0, 1.0, 1.0], adversarial_weight=0.1,
                      generator_first=True)
            jpeg, usm = imgproc.DiffJPEG().to(dev), imgproc.USMSharp().to(dev)
            kgen = torch.Generator(device=dev).manual_seed(7)
            k21 = torch.rand(B, 21, 21, device=dev, generator=kgen) ** 4
            k21 = k21 / k21.sum(dim=(1, 2), keepdim=True)
            import random as _random
            import numpy as _np
            _random.seed(rank)
            _np.random.seed(rank)
        trainer = GanTrainer(g, d, cl, process_group=pg, **kw)
        step_fn = trainer.step
        if workload == "realesrgan_gan":
            def step_fn(_lr_unused, gt_batch):
                gt_usm, gt_, lr_ = imgproc.degradation_process(gt_batch, k21, k21, k21, 4, REALESRGAN_DEGRADATION, jpeg, usm)
                return trainer.step(lr_, gt_, gt_usm)

    # a fresh batch per step from the seeded generator -- all of them resident in HBM before the timed region starts
    # (at most 32 distinct batches, 3.4 GB at the default size; longer runs cycle through them)
    gen = torch.Generator(device=dev)
    gen.manual_seed(1234 + rank)
    n_batches = max(1, min(args.steps, 32))
    batches = [(torch.rand(B, 3, h, h, device=dev, generator=gen), torch.rand(B, 3, S * h, S * h, device=dev, generator=gen)) for _ in range(n_batches)]
    lr_img, gt = batches[0]

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    log("%s: model + inputs ready (B=%d/GPU, %dx%d -> %dx%d, %s), warm-up" % (workload, B, h, h, S * h, S * h, args.dtype))
    for i in range(args.warmup):
        step_fn(lr_img, gt)
        torch.cuda.synchronize()
        log("warm-up step %d done" % i)
    from sr_gan_fd_amd import parallel
    barrier()
    rec = None if args.no_kernel_events else profiling.enable()
    meter = parallel.WaitMeter() if world > 1 else None       # N > 1: how long the main stream stalls for each gradient exchange
    parallel.METER = meter
    t0 = time.perf_counter()
    last = None
    for i in range(args.steps):
        last = step_fn(*batches[i % n_batches])
    barrier()
    dt = time.perf_counter() - t0
    parallel.METER = None
    profiling.disable()
    log("%s: timed region done: %.1f ms/step" % (workload, dt / args.steps * 1e3))
    if world > 1:
        t = torch.tensor([dt], device=dev if args.dist_backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    ms_per_step = dt / args.steps * 1e3
    value = B * world * args.steps / dt

    out = {
        "metric": "SR training images/sec (%d->%d x%d, %s)" % (h, S * h, S, args.dtype), "value": round(value, 3), "unit": "img/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype,
        "data": "synthetic (%d distinct seeded batches, resident in HBM before the timed region)" % n_batches,
        "config": {"workload": {"g_only": "BSRGAN RRDBNet x4 generator-only (L1), %d RRDB, batch %d/GPU, %d->%d",
                                "gan": "BSRGAN full GAN step (RRDBNet %d RRDB + U-Net D + VGG19 content), batch %d/GPU, %d->%d",
                                "aesrgan_gan": "A-ESRGAN full GAN step (RRDBNet %d RRDB + attention U-Net D + VGG19 content), batch %d/GPU, %d->%d",
                                "realesrgan_gan": "Real-ESRGAN iteration: on-device second-order degradation of the GT batch + generator-first GAN step "
                                                  "(RRDBNet %d RRDB + U-Net D + VGG19 content), batch %d/GPU, %d->%d",
                                "esrgan_gan": "ESRGAN relativistic GAN step, fused trainer (RRDBNet %d RRDB + BatchNorm D "
                                              "+ differentiable VGG19 content), batch %d/GPU, %d->%d",
                                }[workload].replace("RRDBNet x4", "RRDBNet x%d" % S) % (args.num_rrdb, B, h, S * h),
                   "global_batch": B * world, "num_rrdb": args.num_rrdb, "parallelism": "dp%d" % world,
                   "flop_per_image": flop_img},
        "step_tflops_per_gpu": round(value / world * flop_img / 1e12, 2),
    }
    if meter is not None:
        # exposed communication: time the main stream waited at BucketReducer.finish() / SideStreamReducer.wait() (HIP event pairs),
        # max over ranks like the step time
        exp = meter.report(args.steps)
        t = torch.tensor([exp.get("g_grad_exchange", 0.0), exp.get("d_grad_exchange_and_adam", 0.0)], device=dev if args.dist_backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        out["exposed_comm_ms_per_step"] = {"g_grad_exchange": round(t[0].item(), 4), "d_grad_exchange_and_adam": round(t[1].item(), 4),
                                           "what": "main-stream stall at the reducers' wait points, max over ranks"}
    if module_loop:
        out["config"]["loop"] = "module-level: the reference's statements over the drop-in modules (autocast + GradScaler + torch.optim.Adam + AveragedModel)"
        # where the module-level step spends its time: two more steps with an event at each section boundary (outside the timed region)
        sect = {}
        for i in range(2):
            marks = [("start", torch.cuda.Event(enable_timing=True))]
            marks[0][1].record()

            def mark(name):
                e = torch.cuda.Event(enable_timing=True)
                e.record()
                marks.append((name, e))
            step_fn(*batches[i % n_batches], mark=mark)
            torch.cuda.synchronize()
            for (_, e0), (name, e1) in zip(marks[:-1], marks[1:]):
                sect[name] = sect.get(name, 0.0) + e0.elapsed_time(e1) / 2
        out["module_loop_sections_ms"] = {k: round(v, 3) for k, v in sect.items()}
        sc = getattr(step_fn, "scaler", None)
        if sc is not None and sc.is_enabled():
            out["loss_scale"] = {"enabled": True, "scale": sc.get_scale()}
    scaler = getattr(trainer, "scaler", None)
    if scaler is not None:
        out["loss_scale"] = scaler.report()
    if torch.is_tensor(last):
        # the step's own scalars after the timed region (read once, outside it): a step that overflowed its 16-bit range or was
        # skipped by the loss scaler every time would show here, not as a fast number
        vals = [float(v) for v in last.detach().float().reshape(-1)[:6].cpu()]
        out["last_step_scalars"] = [round(v, 6) for v in vals]
        if not all(v == v and abs(v) != float("inf") for v in vals):
            raise SystemExit("bench.py: non-finite training scalars %r" % (vals,))
    if rank == 0 and rec is not None:
        out["roofline"] = profiling.roofline(rec, PEAK_BF16_TFLOPS, PEAK_HBM_GBPS)
        out["roofline"].update(pmc_fields(workload, out["roofline"]["kernel"], B, h))
        # the whole step against the dense 16-bit MFMA peak (north_star's ">= 40 % MFMA utilisation" is about this number): algorithmic
        # FLOP of the iteration x images/s / GPUs / 2.5 PF
        out["roofline"]["step_mfma_frac"] = round(out["step_tflops_per_gpu"] / PEAK_BF16_TFLOPS, 4)
        out["kernel_classes"] = profiling.summary(rec)
    # the LDS-resident dense-block launch (small batches): a neighbour hand-off that gave up would mean wrong results, not a slow step
    giveups = ops.dense_chain_giveups(dev)
    if giveups is not None:
        out["dense_chain"] = {"handoff_waits_given_up": giveups}
        if giveups:
            raise SystemExit("bench.py: %d dense-chain hand-off waits gave up (results are wrong)" % giveups)
    del trainer, step_fn, g
    import gc
    gc.collect()
    torch.cuda.empty_cache()
    return out


def module_level_loop(workload, M, g, dev, cdt, dropin_optim=False):
    """The reference's loop bodies as written -- BSRGAN/train_bsrnet.py:244-272 (g_only) and train_bsrgan.py:387-483 (gan): amp.autocast
    around the forwards, ONE GradScaler, torch.optim.Adam, AveragedModel with the scripts' avg_fn, autograd incl. retain_graph -- over
    the drop-in modules, which take their precision from that autocast (nothing is set on them).  The scripts' five ``.item()`` reads per
    iteration (train_bsrgan.py:479-483) are deferred: the step returns the device scalars."""
    import torch
    from torch import amp
    if dropin_optim:
        # the same statements with the package's drop-in classes behind the scripts' names: optim.Adam and AveragedModel run one kernel
        # over a network's flat parameter buffer where torch's run one (or several) per tensor
        from sr_gan_fd_amd import optim
        from sr_gan_fd_amd.swa_utils import AveragedModel
    else:
        from torch import optim
        from torch.optim.swa_utils import AveragedModel
    g.train()
    use_amp = cdt != torch.float32
    ac = lambda: amp.autocast("cuda", dtype=cdt, enabled=use_amp)
    scaler = amp.GradScaler("cuda", enabled=cdt == torch.float16)      # bf16 / f32: inert, as on the reference's CPU path
    ema = AveragedModel(g, avg_fn=lambda a, p, n: (1 - 0.999) * a + 0.999 * p)
    l1 = torch.nn.L1Loss()
    if workload == "g_only":
        g_opt = optim.Adam(g.parameters(), 1e-4, (0.9, 0.99), 1e-4, 0.0)      # bsrnet_config.py:86-96
        pw = torch.Tensor([1.0]).to(dev)

        def step(lr, gt, mark=lambda name: None):
            g.zero_grad(set_to_none=True)
            with ac():
                sr = g(lr)
                loss = torch.sum(torch.mul(pw, l1(sr, gt)))
            scaler.scale(loss).backward()
            mark("forward_backward")
            scaler.step(g_opt)
            scaler.update()
            mark("gradscaler_and_torch_adam")
            ema.update_parameters(g)
            mark("averaged_model_update")
            return loss.detach().reshape(1)
        step.scaler = scaler
        return step
    d = M.discriminator_unet(in_channels=3, out_channels=1, channels=64).to(dev).train()
    cl = M.ContentLoss(NODES, MEAN, STD).to(dev)
    d_opt = optim.Adam(d.parameters(), 2e-4, (0.9, 0.999), 1e-4, 0.0)         # bsrgan_config.py:147-151
    g_opt = optim.Adam(g.parameters(), 8e-5, (0.9, 0.999), 1e-4, 0.0)
    bce = torch.nn.BCEWithLogitsLoss()
    pw, cw, aw = (torch.Tensor(w).to(dev) for w in ([20.0], [1.0], [0.5]))          # bsrgan_config.py:137-143

    def step(lr, gt, mark=lambda name: None):
        B, _, H, W = gt.shape
        real = torch.full([B, 1, H, W], 1.0, dtype=gt.dtype, device=dev)
        fake = torch.full([B, 1, H, W], 0.0, dtype=gt.dtype, device=dev)
        for p in d.parameters():
            p.requires_grad = True
        d.zero_grad(set_to_none=True)
        with ac():
            gt_output = d(gt)
            d_loss_hr = bce(gt_output, real)
        scaler.scale(d_loss_hr).backward(retain_graph=True)
        with ac():
            sr = g(lr)
            sr_output = d(sr.detach().clone())
            d_loss_sr = bce(sr_output, fake)
        scaler.scale(d_loss_sr).backward()
        d_loss = d_loss_hr + d_loss_sr
        mark("forward_backward")
        scaler.step(d_opt)
        scaler.update()
        mark("gradscaler_and_torch_adam")
        for p in d.parameters():
            p.requires_grad = False
        g.zero_grad(set_to_none=True)
        with ac():
            pixel = l1(sr, gt)
            content = cl(sr, gt)
            adv = bce(d(sr), real)
            pixel = torch.sum(torch.mul(pw, pixel))
            content = torch.sum(torch.mul(cw, content))
            adv = torch.sum(torch.mul(aw, adv))
            g_loss = pixel + content + adv
        scaler.scale(g_loss).backward()
        mark("forward_backward")
        scaler.step(g_opt)
        scaler.update()
        mark("gradscaler_and_torch_adam")
        ema.update_parameters(g)
        mark("averaged_model_update")
        d_gt = torch.mean(torch.sigmoid_(gt_output.detach()))
        d_sr = torch.mean(torch.sigmoid_(sr_output.detach()))
        return torch.stack([d_loss.detach(), pixel.detach(), content.detach(), adv.detach(), d_gt, d_sr])
    step.scaler = scaler
    return step


def esrgan_loop(M, g, dev, cdt):
    """ESRGAN/train_esrgan.py:364-431 as the script writes it -- torch.optim.Adam, BCEWithLogitsLoss / L1Loss, autograd with
    retain_graph and three live discriminator forwards -- over the drop-in modules (esrgan_config.py:75-92 hyper-parameters)."""
    import torch
    d = M.discriminator()
    cl = M.ContentLoss("features.34", MEAN, STD)          # a str node selects ESRGAN's differentiable single-tap loss
    d.compute_dtype = cl.compute_dtype = cdt
    d.to(dev).train()
    cl.to(dev)
    g.train()
    d_opt = torch.optim.Adam(d.parameters(), 1e-4, (0.9, 0.99), 1e-8, 0.0)
    g_opt = torch.optim.Adam(g.parameters(), 1e-4, (0.9, 0.99), 1e-8, 0.0)
    bce, l1 = torch.nn.BCEWithLogitsLoss(), torch.nn.L1Loss()

    def step(lr, gt):
        B = gt.shape[0]
        real, fake = torch.full([B, 1], 1.0, device=dev), torch.full([B, 1], 0.0, device=dev)
        for p in d.parameters():
            p.requires_grad = False
        g.zero_grad(set_to_none=True)
        sr = g(lr)
        gt_output = d(gt.detach().clone())
        sr_output = d(sr)
        loss = 0.01 * l1(sr, gt) + 1.0 * cl(sr, gt) + 0.005 * (bce(gt_output - torch.mean(sr_output), fake) * 0.5 +
                                                              bce(sr_output - torch.mean(gt_output), real) * 0.5)
        loss.backward()
        g_opt.step()
        for p in d.parameters():
            p.requires_grad = True
        d.zero_grad(set_to_none=True)
        gt_output = d(gt)
        sr_output = d(sr.detach().clone())
        (bce(gt_output - torch.mean(sr_output), real) * 0.5).backward(retain_graph=True)
        sr_output = d(sr.detach().clone())
        (bce(sr_output - torch.mean(gt_output), fake) * 0.5).backward()
        d_opt.step()
    return step


def sr_parity(h: int, num_rrdb: int, dev, dtype_name: str = "f16"):
    """The metric's "PSNR vs ref" leg: SR of the HIP path in the benchmark dtype against the CPU oracle (fp32) for one
    image of the workload's size, same weights (seed 0, the x3 / bias 0.5 init recipe of the parity tests: the default
    init gives a near-constant SR) and same input.  PSNR through the product's own kernel, on Y with a 4-pixel border crop
    as the reference's validate() does."""
    import torch
    from oracle import srgan_oracle as O
    from sr_gan_fd_amd import model as M
    from sr_gan_fd_amd.image_quality_assessment import PSNR
    torch.manual_seed(0)
    g = M.bsrgan_x4(num_rrdb=num_rrdb)
    with torch.no_grad():
        for p in g.parameters():
            if p.dim() == 4:
                p.mul_(3.0)
        g.conv4.bias.fill_(0.5)
    P = {k: v.detach().clone() for k, v in g.state_dict().items()}
    x = torch.rand(1, 3, h, h)
    torch.set_num_threads(host_cores())
    with torch.no_grad():
        want = O.rrdbnet_forward(x, P, 4)
    g.compute_dtype = getattr(torch, DTYPES[dtype_name])
    g.to(dev).eval()
    with torch.no_grad():
        got = g(x.to(dev))
        psnr = PSNR(4, True)(got, want.to(dev)).item()
    err = (got.cpu() - want).abs().max().item()
    log("SR parity: PSNR(Y) %.2f dB, max abs err %.2e" % (psnr, err))
    return {"psnr_y_db": round(psnr, 2), "max_abs_err": float("%.3g" % err), "what": "%s HIP SR vs fp32 CPU oracle, 1 image %d->%d, scaled init" % (dtype_name, h, 4 * h), "tolerance": 1e-3,
            "within_tolerance": bool(err <= 1e-3)}


CPU_WARMUP, CPU_TIMED = 3, 5      # "3 warm-up + >= 5 timed iterations, median"


def _cpu_timed(step, what: str):
    """BASELINE.md 4's protocol: CPU_WARMUP untimed + CPU_TIMED timed iterations of ``step``; returns (median seconds, sample text)."""
    import statistics
    times = []
    for it in range(CPU_WARMUP + CPU_TIMED):
        t0 = time.perf_counter()
        step()
        dt = time.perf_counter() - t0
        log("cpu baseline iteration %d%s: %.1f s" % (it, " (warm-up)" if it < CPU_WARMUP else "", dt))
        if it >= CPU_WARMUP:
            times.append(dt)
    return statistics.median(times), "%s, fp32, %d warm-up + %d timed iterations (median)" % (what, CPU_WARMUP, CPU_TIMED)


def cpu_baseline(workload: str, h: int, num_rrdb: int):
    """The CPU oracle's training iteration (oracle/srgan_oracle.py, torch-CPU fp32) on a bounded sample -- batch 1 of the same workload --
    timed by BASELINE.md 4's protocol (3 warm-up + 5 timed iterations, median)."""
    import torch
    from oracle import srgan_oracle as O
    from sr_gan_fd_amd import model as M
    cores = host_cores()
    torch.set_num_threads(cores)
    log("cpu baseline: oracle on %d threads" % cores)
    torch.manual_seed(0)
    g = M.bsrgan_x4(num_rrdb=num_rrdb)
    G = {k: v.detach().clone() for k, v in g.state_dict().items()}
    opt = O.AdamState(G, O.g_param_names(G))
    lr_img, gt = torch.rand(1, 3, h, h), torch.rand(1, 3, 4 * h, 4 * h)
    d_forward, hp = None, dict(g_lr=8e-5, d_lr=2e-4, pixel_weight=20.0, adversarial_weight=0.5)
    if workload == "realesrgan_gan":
        import random as _random
        import numpy as _np
        from oracle import degradation_oracle as DO
        _random.seed(0)
        _np.random.seed(0)
        d = M.discriminator_unet(in_channels=3, out_channels=1, channels=64)
        D = {k: v.detach().clone() for k, v in d.state_dict().items()}
        d_opt = O.AdamState(D, O.d_param_names(D))
        cl = M.ContentLoss(NODES, MEAN, STD)
        VP = {"features." + k: v.detach().clone() for k, v in cl.features.state_dict().items()}
        gt = torch.rand(1, 3, 4 * h, 4 * h)
        k21 = torch.rand(1, 21, 21) ** 4
        k21 = k21 / k21.sum()
        def one():
            gt_usm, gt_, lr_ = DO.degradation_process(gt, k21, k21, k21, 4, REALESRGAN_DEGRADATION, usm=(DO.usm_kernel(), 0.5, 10))
            O.realesrgan_gan_step(G, D, opt, d_opt, lr_, gt_, gt_usm, content_fn=lambda sr, gt__: O.content_loss(sr, gt__, VP, NODES, MEAN, STD))
        t, sample = _cpu_timed(one, "batch 1, degradation + GAN step, %d->%d" % (h, 4 * h))
        return {"value": round(1.0 / t, 4), "unit": "img/s", "cores": cores, "kind": "port", "sample": sample}
    if workload == "esrgan_gan":
        bsz = 4
        lr_img, gt = torch.rand(bsz, 3, h, h), torch.rand(bsz, 3, 4 * h, 4 * h)
        d = M.discriminator()
        D = {k: v.detach().clone() for k, v in d.state_dict().items()}
        d_opt = O.AdamState(D, [k for k in D if k.endswith((".weight", ".bias"))])
        cl = M.ContentLoss("features.34", MEAN, STD)
        VP = {"features." + k: v.detach().clone() for k, v in cl.features.state_dict().items()}
        t, sample = _cpu_timed(lambda: O.esrgan_gan_step(G, D, opt, d_opt, lr_img, gt,
                                                         content_fn=lambda sr, gt_: O.content_loss_single(sr, gt_, VP, "features.34", MEAN, STD)),
                               "batch %d, %d->%d" % (bsz, h, 4 * h))
        return {"value": round(bsz / t, 4), "unit": "img/s", "cores": cores, "kind": "port", "sample": sample}
    if workload != "g_only":
        if workload == "aesrgan_gan":
            d_forward, hp = O.aesrgan_unet_forward, dict(g_lr=5e-5, d_lr=1e-5, pixel_weight=10.0, adversarial_weight=0.1)
        d = M.uNetDiscriminatorAesrgan() if workload == "aesrgan_gan" else M.discriminator_unet(in_channels=3, out_channels=1, channels=64)
        D = {k: v.detach().clone() for k, v in d.state_dict().items()}
        d_opt = O.AdamState(D, O.d_param_names(D))
        cl = M.ContentLoss(NODES, MEAN, STD)
        VP = {"features." + k: v.detach().clone() for k, v in cl.features.state_dict().items()}
        content_fn = lambda sr, gt_: O.content_loss(sr, gt_, VP, NODES, MEAN, STD)
    def one():
        if workload == "g_only":
            O.g_only_step(G, opt, lr_img, gt, upscale=4, lr=1e-4, betas=(0.9, 0.99), eps=1e-4)
        else:
            O.gan_step(G, D, opt, d_opt, lr_img, gt, upscale=4, betas=(0.9, 0.999), eps=1e-4, content_weight=1.0,
                       content_fn=content_fn, d_forward=d_forward, **hp)
    t, sample = _cpu_timed(one, "batch 1, %d->%d" % (h, 4 * h))
    return {"value": round(1.0 / t, 4), "unit": "img/s", "cores": cores, "kind": "port", "sample": sample}


if __name__ == "__main__":
    main()
