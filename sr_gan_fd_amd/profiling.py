"""HIP-event brackets around kernel launches (bench.py's live per-kernel timing).

When enabled, the engines record one event pair per fused-conv / weight-gradient launch on the stream
the kernel is launched on, labelled with the kernel template it dispatches to and its algorithmic FLOP
count.  ``roofline()`` aggregates the class with the largest total time (the dominant kernel)."""
from __future__ import annotations

from typing import Dict, List, Optional

import torch

REC: Optional["Recorder"] = None


class Recorder:
    """Every launch is counted (FLOP, bytes); every ``every``-th launch of a class is timed with a HIP event pair.
    Timing all ~930 launches of a generator step cost 6 ms of an 84 ms step (event records serialise the queue), which
    would distort the very throughput the bench reports; a 1-in-7 sample (7 is coprime with the 4- and 5-launch shape
    cycles of a dense block, so every shape of a class is sampled) still gives hundreds of timings per class."""

    def __init__(self, every: int = 7):
        self.items: List[tuple] = []
        self.every = max(1, every)
        self.count: Dict[str, int] = {}
        self.work: Dict[str, list] = {}

    def bracket(self, label: str, work, fn) -> None:
        """work = (algorithmic FLOP, algorithmic HBM bytes) of the launch"""
        flops, nbytes = work if isinstance(work, tuple) else (work, 0.0)
        n = self.count.get(label, 0)
        self.count[label] = n + 1
        w = self.work.setdefault(label, [0.0, 0.0])
        w[0] += flops
        w[1] += nbytes
        if n % self.every:
            fn()
            return
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        self.items.append((label, flops, nbytes, e0, e1))


def enable(every: int = 7) -> Recorder:
    global REC
    REC = Recorder(every)
    return REC


def disable() -> None:
    global REC
    REC = None


def conv_label(a) -> str:
    """kernel template the launch dispatches to (asked from the library: srganfd_conv2d_describe)"""
    lab = getattr(a, "_kernel_label", None)      # launch structs live in the engines' plans: ask once
    if lab is None:
        import ctypes as C
        from . import _abi as A
        buf = C.create_string_buffer(160)
        A.check(A.lib().srganfd_conv2d_describe(C.byref(a), buf, 160), "conv2d_describe")
        lab = a._kernel_label = buf.value.decode()
    return lab


def conv_flops(a) -> float:
    return 2.0 * a.n * a.h_out * a.w_out * a.ksize * a.ksize * a.cin * a.cout_store * (4 if a.out_classes == 4 else 1)


def conv_work(a) -> tuple:
    """(algorithmic FLOP, algorithmic bytes) of one fused-conv launch: every input pixel's cin channels read once, every
    output pixel's cout_store channels written once, plus the epilogue tensors (residuals, mask read; pre-skip copy
    written); weights are negligible.  Element size 2 (bf16) / 4 (f32; also the fp32 SR / logits outputs)."""
    es = 4 if a.dtype == 1 else 2
    pin = a.n * a.h_in * a.w_in * a.cin * es
    pout = a.n * a.h_out * a.w_out * a.cout_store * (4 if a.out_classes == 4 else 1)    # a class launch reads dy once and writes all of dx
    nb = pin + pout * (4 if a.y_f32 else es)
    for v in (a.r1, a.r2, a.mask, a.y2):
        if v.ptr:
            nb += pout * es
    return conv_flops(a), float(nb)


def summary(rec: Recorder) -> Dict[str, dict]:
    torch.cuda.synchronize()
    agg: Dict[str, dict] = {}
    for label, flops, nbytes, e0, e1 in rec.items:
        d = agg.setdefault(label, {"launches": 0, "ms": 0.0, "flop": 0.0, "bytes": 0.0, "timed_launches": 0})
        d["timed_launches"] += 1
        d["ms"] += e0.elapsed_time(e1)
        d["flop"] += flops
        d["bytes"] += nbytes
    for label, d in agg.items():
        # throughputs come from the timed sample (its own FLOP / bytes / time); `launches` and `ms` are scaled to all launches
        d["launches"] = rec.count[label]
    for d in agg.values():
        d["avg_us"] = round(d["ms"] * 1e3 / d["timed_launches"], 2)
        d["tflops"] = round(d["flop"] / (d["ms"] * 1e-3) / 1e12, 1) if d["ms"] > 0 else 0.0
        d["gbps"] = round(d["bytes"] / (d["ms"] * 1e-3) / 1e9, 1) if d["ms"] > 0 else 0.0
        d["flop_per_byte"] = round(d["flop"] / d["bytes"], 1) if d["bytes"] > 0 else None
        scale = d["launches"] / d["timed_launches"]
        d["ms"] = round(d["ms"] * scale, 3)                      # estimated total of the class over the timed region
    for label, d in agg.items():
        d["flop"] = float(f"{rec.work[label][0]:.6g}")          # exact totals over all launches of the class
        d["bytes"] = float(f"{rec.work[label][1]:.6g}")
    return agg


MEASURED_READ_GBPS = 5600.0      # midpoint of the read-only probe runs in profiles/r02_hbm_bw_probe.txt


def roofline(rec: Recorder, peak_tflops: float, peak_gbps: float = 8000.0) -> dict:
    """Dominant kernel class (largest total time) against the roof that bounds it: its arithmetic intensity (algorithmic
    FLOP / algorithmic bytes) below the ridge peak_tflops / peak_gbps means HBM-bound, else MFMA-bound.  Both fractions
    are reported; `achieved` / `peak` / `frac` are those of the binding roof."""
    import re
    per_label = summary(rec)
    # the epilogue kinds of one tile shape (",E0" plain, ",E4" masked, ...: conv_igemm_kernel's EK) share the staging and MFMA loop: one class
    agg: Dict[str, dict] = {}
    for lab, v in per_label.items():
        fam = re.sub(r",E\d+", "", lab)
        d = agg.setdefault(fam, {"launches": 0, "ms": 0.0, "flop": 0.0, "bytes": 0.0, "kinds": {}})
        d["launches"] += v["launches"]; d["ms"] += v["ms"]; d["flop"] += v["flop"]; d["bytes"] += v["bytes"]
        d["kinds"][lab] = {"launches": v["launches"], "avg_us": v["avg_us"]}
    for d in agg.values():
        d["avg_us"] = round(d["ms"] * 1e3 / d["launches"], 2) if d["launches"] else 0.0
        d["tflops"] = round(d["flop"] / (d["ms"] * 1e-3) / 1e12, 1) if d["ms"] > 0 else 0.0
        d["gbps"] = round(d["bytes"] / (d["ms"] * 1e-3) / 1e9, 1) if d["ms"] > 0 else 0.0
        d["flop_per_byte"] = round(d["flop"] / d["bytes"], 1) if d["bytes"] > 0 else None
    label, d = max(agg.items(), key=lambda kv: kv[1]["ms"])
    ridge = peak_tflops * 1e12 / (peak_gbps * 1e9)
    ai = d["flop_per_byte"]
    hbm = ai is not None and ai < ridge
    out = {"bound": "hbm" if hbm else "mfma", "kernel": label}
    if hbm:
        out.update(achieved=d["gbps"], peak=peak_gbps, unit="GB/s", frac=round(d["gbps"] / peak_gbps, 4))
    else:
        out.update(achieved=d["tflops"], peak=peak_tflops, unit="TFLOP/s", frac=round(d["tflops"] / peak_tflops, 4))
    if len(d["kinds"]) > 1:
        out["kinds"] = d["kinds"]
    # the largest single kernel symbol of the step (a class above may be several epilogue kinds of one tile shape) against the MFMA roof
    slab, sv = max(per_label.items(), key=lambda kv: kv[1]["ms"])
    tot_ms = sum(v["ms"] for v in per_label.values())
    out["largest_symbol"] = {"kernel": slab, "avg_us": sv["avg_us"], "launches": sv["launches"], "tflops": sv["tflops"],
                             "mfma_frac": round(sv["tflops"] / peak_tflops, 4), "share_of_bracketed_kernel_time": round(sv["ms"] / tot_ms, 4) if tot_ms > 0 else None}
    out.update(traffic=None, avg_launch_us=d["avg_us"], launches=d["launches"],
               flop_per_launch=float(f"{d['flop'] / d['launches']:.6g}"),
               bytes_per_launch=float(f"{d['bytes'] / d['launches']:.6g}"), flop_per_byte=ai, ridge_flop_per_byte=round(ridge, 1),
               mfma_frac=round(d["tflops"] / peak_tflops, 4), hbm_frac=round(d["gbps"] / peak_gbps, 4),
               # what a pure streaming kernel reaches on this part (tools/probes/bw_probe.hip, profiles/r02_hbm_bw_probe.txt: reads
               # 5.3-6.2 TB/s, copies 4.3-5.1): context for hbm_frac, which the contract prices against the 8 TB/s datasheet figure
               hbm_measured_read_gbps=MEASURED_READ_GBPS, hbm_frac_of_measured=round(d["gbps"] / MEASURED_READ_GBPS, 4))
    return out
