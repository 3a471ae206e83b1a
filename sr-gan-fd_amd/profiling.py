"""HIP-event brackets around kernel launches (bench.py's live per-kernel timing).

When enabled, the engines record one event pair per fused-conv / weight-gradient launch on the stream
the kernel is launched on, labelled with the kernel template it dispatches to and its algorithmic FLOP
count.  ``roofline()`` aggregates the class with the largest total time (the dominant kernel)."""
from __future__ import annotations

from typing import Dict, List, Optional

import torch

REC: Optional["Recorder"] = None


class Recorder:
    def __init__(self):
        self.items: List[tuple] = []

    def bracket(self, label: str, flops: float, fn) -> None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        self.items.append((label, flops, e0, e1))


def enable() -> Recorder:
    global REC
    REC = Recorder()
    return REC


def disable() -> None:
    global REC
    REC = None


def conv_label(a) -> str:
    """kernel template the launch dispatches to (conv_igemm.hip: dispatch_conv)"""
    dt = "bf16" if a.dtype == 0 else "f32"
    wide = a.cout % 64 == 0
    if a.ksize in (3, 2):
        mr, wr, wn = (2, 4, 2) if wide else (2, 8, 1)
    elif a.ksize == 4:
        mr, wr, wn = (1, 4, 2) if (wide and a.dtype == 0) else (1, 4, 1)
    else:
        mr, wr, wn = 2, 4, 1
    return f"conv_igemm_kernel<{dt},KS={a.ksize},S={a.stride},MR={mr},WR={wr},WN={wn}>"


def conv_flops(a) -> float:
    return 2.0 * a.n * a.h_out * a.w_out * a.ksize * a.ksize * a.cin * a.cout_store


def summary(rec: Recorder) -> Dict[str, dict]:
    torch.cuda.synchronize()
    agg: Dict[str, dict] = {}
    for label, flops, e0, e1 in rec.items:
        d = agg.setdefault(label, {"launches": 0, "ms": 0.0, "flop": 0.0})
        d["launches"] += 1
        d["ms"] += e0.elapsed_time(e1)
        d["flop"] += flops
    for d in agg.values():
        d["avg_us"] = round(d["ms"] * 1e3 / d["launches"], 2)
        d["tflops"] = round(d["flop"] / (d["ms"] * 1e-3) / 1e12, 1) if d["ms"] > 0 else 0.0
        d["ms"] = round(d["ms"], 3)
        d["flop"] = float(f"{d['flop']:.6g}")
    return agg


def roofline(rec: Recorder, peak_tflops: float) -> dict:
    agg = summary(rec)
    label, d = max(agg.items(), key=lambda kv: kv[1]["ms"])
    return {"bound": "mfma", "kernel": label, "achieved": d["tflops"], "peak": peak_tflops, "unit": "TFLOP/s",
            "frac": round(d["tflops"] / peak_tflops, 4), "traffic": None,
            "avg_launch_us": d["avg_us"], "launches": d["launches"],
            "flop_per_launch": float(f"{d['flop'] / d['launches']:.6g}")}
