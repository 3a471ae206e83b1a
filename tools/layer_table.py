"""Per-shape table of a training step: every conv / weight-gradient launch timed with an event pair (eager step, every launch bracketed),
grouped by kernel template AND shape, sorted by share of the step's kernel time.  Tells which layers of the GAN step run far from either roof.
    python tools/layer_table.py [--workload gan|g_only] [--batch 32] [--steps 2]
(the event brackets serialise the queue: totals are ~8 % above an unbracketed step; the ranking is what this is for)"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from sr_gan_fd_amd import profiling  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="gan")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--top", type=int, default=45)
    ap.add_argument("--lr-size", type=int, default=0)
    ap.add_argument("--upscale", type=int, default=4)
    o = ap.parse_args()
    orig = profiling.conv_label

    def shaped(a):
        ep = "".join(t for t, v in (("+r1", a.r1.ptr), ("+r2", a.r2.ptr), ("+mask", a.mask.ptr), ("+y2", a.y2.ptr)) if v)
        return "%s | %d->%d k%d s%d up%d %dx%d%s" % (orig(a).replace("conv_igemm_kernel", "conv"), a.cin, a.cout_store, a.ksize, a.stride, a.up, a.h_out, a.w_out, ep)
    profiling.conv_label = shaped
    bracket = profiling.Recorder.bracket

    def bracket_shaped(self, label, work, fn):
        if "|" not in label:
            f = work[0] if isinstance(work, tuple) else work
            label = "%s | %.4g GF" % (label, f / 1e9)
        return bracket(self, label, work, fn)
    profiling.Recorder.bracket = bracket_shaped

    args = argparse.Namespace(gpus=1, steps=o.steps, warmup=2, workload=o.workload, batch=o.batch, esrgan_module_loop=False, lr_size=o.lr_size, upscale=o.upscale, num_rrdb=23,
                              dtype="f16", no_cpu_baseline=True, no_kernel_events=False, dist_backend="nccl", module_loop=False, no_module_loop=True, dropin_optim=False)
    enable = profiling.enable
    profiling.enable = lambda every=7: enable(1)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    out = bench.run_workload(args, o.workload, 0, 1, dev, None)
    per = out["kernel_classes"]
    tot = sum(d["ms"] for d in per.values())
    print("step (bracketed): %.2f ms; kernels bracketed: %.2f ms/step" % (out["ms_per_step"], tot / o.steps))
    print("%6s %5s %9s %8s %8s  %s" % ("share", "n/st", "avg us", "TF/s", "GB/s", "kernel | shape"))
    for lab, d in sorted(per.items(), key=lambda kv: -kv[1]["ms"])[:o.top]:
        print("%5.1f%% %5.1f %9.1f %8.1f %8.1f  %s" % (100 * d["ms"] / tot, d["launches"] / o.steps, d["avg_us"], d["tflops"], d["gbps"], lab))


if __name__ == "__main__":
    main()
