"""CPU emulation of the storage-precision options of the generator forward (which roundings cost what).
Usage: python tools/precision_emul.py [num_rrdb] [h]
Compares an fp32 reference against emulations where conv inputs / stored activations are rounded to
bf16 or fp16, with the residual stream kept in bf16, bf16 hi+lo, or fp32.  Test infrastructure / design study only."""
import sys, torch, torch.nn.functional as F
sys.path.insert(0, '.')
torch.manual_seed(0)
R = int(sys.argv[1]) if len(sys.argv) > 1 else 23
h = int(sys.argv[2]) if len(sys.argv) > 2 else 32
torch.set_num_threads(8)

def mk():
    P = {}
    def conv(name, co, ci):
        w = torch.empty(co, ci, 3, 3); torch.nn.init.kaiming_normal_(w); P[name + ".weight"] = w * 0.1 * 3.0; P[name + ".bias"] = torch.zeros(co)
    conv("conv1", 64, 3)
    for i in range(R):
        for r in (1, 2, 3):
            for k in range(1, 6):
                conv(f"trunk.{i}.rdb{r}.conv{k}", 64 if k == 5 else 32, 64 + 32 * (k - 1))
    for n in ("conv2", "upsampling1.0", "upsampling2.0", "conv3.0"): conv(n, 64, 64)
    conv("conv4", 3, 64); P["conv4.bias"].fill_(0.5)
    return P

def rnd(x, mode):
    if mode == "f32": return x
    if mode == "bf16": return x.bfloat16().float()
    if mode == "f16": return x.half().float()
    raise ValueError(mode)

def fwd(x, P, act, wt, resid):
    """act: storage dtype of activations fed to convs; wt: weight dtype; resid: 'same' (residual stream stored in act dtype),
    'hilo' (act dtype hi + act dtype lo), 'f32'."""
    W = {k: (rnd(v, wt) if k.endswith("weight") else v) for k, v in P.items()}
    def conv(t, name): return F.conv2d(t, W[name + ".weight"], W[name + ".bias"], padding=1)
    def store_resid(v):
        # returns (value used as conv input, value used in later residual adds)
        hi = rnd(v, act)
        if resid == "same": return hi, hi
        if resid == "hilo": return hi, hi + rnd(v - hi, act)
        return hi, v
    xin = rnd(x, act)
    o1_in, o1_res = store_resid(conv(xin, "conv1"))
    cur_in, cur_res = o1_in, o1_res
    for i in range(R):
        blk_in, blk_res = cur_in, cur_res
        for r in (1, 2, 3):
            feats = [cur_in]
            for k in range(1, 5):
                y = F.leaky_relu(conv(torch.cat(feats, 1), f"trunk.{i}.rdb{r}.conv{k}"), 0.2)
                feats.append(rnd(y, act))
            o5 = conv(torch.cat(feats, 1), f"trunk.{i}.rdb{r}.conv5")
            if r < 3:
                cur_in, cur_res = store_resid(o5 * 0.2 + cur_res)
            else:
                cur_in, cur_res = store_resid((o5 * 0.2 + cur_res) * 0.2 + blk_res)
    out = rnd(conv(cur_in, "conv2") + o1_res, act)
    for u in (1, 2):
        out = F.interpolate(out, scale_factor=2, mode="nearest")
        out = rnd(F.leaky_relu(conv(out, f"upsampling{u}.0"), 0.2), act)
    out = rnd(F.leaky_relu(conv(out, "conv3.0"), 0.2), act)
    return conv(out, "conv4").clamp(0, 1)

P = mk()
x = torch.rand(1, 3, h, h)
with torch.no_grad():
    ref = fwd(x.double(), {k: v.double() for k, v in P.items()}, "f32", "f32", "f32").float()
    print(f"R={R} h={h}: SR mean {ref.mean():.3f} std {ref.std():.3f} clamped {(ref==0).float().mean():.3f}/{(ref==1).float().mean():.3f}")
    for act, wt, resid in [("f32", "f32", "f32"), ("bf16", "bf16", "same"), ("bf16", "bf16", "hilo"), ("bf16", "bf16", "f32"), ("bf16", "f32", "f32"),
                           ("f16", "f16", "same"), ("f16", "f16", "hilo"), ("f16", "f16", "f32"), ("f16", "bf16", "same")]:
        y = fwd(x, P, act, wt, resid)
        e = (y - ref).abs()
        rel = (e / ref.abs().clamp_min(1e-3)).max()
        print(f"act={act:5s} w={wt:5s} resid={resid:5s}: max abs {e.max():.2e}  rms {e.pow(2).mean().sqrt():.2e}  max rel {rel:.2e}")
