"""Per-basic-block instruction census of one kernel in a hipcc -S listing.
    python tools/isa_sections.py build/isa/conv_igemm.s <mangled-name substring> [--dump]
Prints, per label block: instructions, MFMA, VALU, SALU, LDS reads/writes, global/buffer loads, stores, waitcnt, barriers, branches."""
import re, sys
path, key = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l and re.match(r"^_Z\S+:", l))
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
# the kernel may have several s_endpgm (early exits): run to .Lfunc_end
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
def cat(op):
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith("ds_read") or op.startswith("ds_load"): return "ldsr"
    if op.startswith("ds_"): return "ldsw"
    if op.startswith("buffer_load") or op.startswith("global_load") or op.startswith("scratch_load"): return "vld"
    if op.startswith("buffer_store") or op.startswith("global_store") or op.startswith("scratch_store"): return "vst"
    if op.startswith("s_waitcnt"): return "wait"
    if op.startswith("s_barrier"): return "bar"
    if op.startswith("s_cbranch") or op.startswith("s_branch"): return "br"
    if op.startswith("s_load") or op.startswith("s_buffer_load"): return "smem"
    if op.startswith("s_nop"): return "nop"
    if op.startswith("v_"): return "valu"
    if op.startswith("s_"): return "salu"
    return "other"
cats = ["n", "mfma", "valu", "salu", "ldsr", "ldsw", "vld", "vst", "smem", "wait", "bar", "br", "nop"]
blocks, cur = [], {"label": "entry", "line": start}
tot = {c: 0 for c in cats}
for i in range(start + 1, end):
    l = lines[i].strip()
    if not l or l.startswith(";") or l.startswith("."):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            blocks.append(cur); cur = {"label": m.group(1), "line": i}
        continue
    op = l.split()[0]
    c = cat(op)
    cur["n"] = cur.get("n", 0) + 1; cur[c] = cur.get(c, 0) + 1
    tot["n"] += 1; tot[c] = tot.get(c, 0) + 1
blocks.append(cur)
print("%-12s %7s " % ("block", "line") + " ".join("%5s" % c for c in cats))
for b in blocks:
    if b.get("n", 0) >= int(sys.argv[sys.argv.index("--min") + 1]) if "--min" in sys.argv else True:
        print("%-12s %7d " % (b["label"], b["line"] + 1) + " ".join("%5d" % b.get(c, 0) for c in cats))
print("%-12s %7s " % ("TOTAL", "") + " ".join("%5d" % tot.get(c, 0) for c in cats))
for l in lines[end:end + 80]:
    if any(k in l for k in ("vgpr_count", "sgpr_count", "spill", "scratch", "lds_size", "Occupancy", "NumVgprs", "ScratchSize")):
        print(l.strip())
