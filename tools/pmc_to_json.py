"""rocprofv3 outputs of tools/profile_round.sh -> profiles/: per-kernel stats CSVs, a PMC summary, and <tag>_pmc.json (--tag, default r03) (what bench.py
reads for roofline.traffic / mfma_busy_pmc, stamped with the SHA of the kernel sources it was measured on).

    python tools/pmc_to_json.py gpurun_out/prof_r02 [--batch 32 --lr-size 128]

traffic per launch  = (2 x TCC_EA0_RDREQ + TCC_EA0_WRREQ) x 64 B   (gfx950: 16-byte-per-lane reads are tallied at half their bytes)
MFMA utilisation    = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 256 CUs x 4 SIMDs)"""
import collections, csv, glob, json, os, re, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench


def arg(name, default):
    return sys.argv[sys.argv.index(name) + 1] if name in sys.argv else default


TY = {"unsigned short": "bf16", "_Float16": "f16", "float": "f32"}


MTY = {"t": "unsigned short", "DF16_": "_Float16", "f": "float"}


def demangle(kernel: str) -> str:
    """rocprofv3 leaves names with _Float16 template arguments mangled (_ZN7srganfd17conv_igemm_kernelIDF16_Li3ELi1E...): rebuild
    `name<type, ints...>` so that one parser serves both forms"""
    m = re.match(r"_ZN7srganfd\d+(\w+?_kernel)I(DF16_|t|f)((?:Lin?\d+E|Lb[01]E)*)E", kernel)
    if not m:
        return kernel
    args = [MTY[m.group(2)]] + [(a[2:-1].replace("n", "-") if a.startswith("Li") else ("true" if a[2] == "1" else "false")) for a in re.findall(r"Lin?\d+E|Lb[01]E", m.group(3))]
    return "%s<%s>" % (m.group(1), ", ".join(args))


def label(kernel: str):
    """rocprof kernel name -> bench.py class label (profiling.conv_label / WgradPlan.label)"""
    kernel = demangle(kernel)
    m = re.search(r"conv_igemm_kernel<([^,]+), (\d+), (\d+), (\d+), (\d+), (\d+)(?:, (true|false))?(?:, (\d+))?(?:, (-?\d+))?(?:, (true|false))?>", kernel)
    if m:
        return "conv_igemm_kernel<%s,KS=%s,S=%s,MR=%s,WR=%s,WN=%s%s%s%s%s>" % (TY.get(m.group(1), m.group(1)), *m.groups()[1:6], ",M16" if m.group(7) == "true" else "",
                                                                             ",TS=%s" % m.group(8) if m.group(8) and int(m.group(8)) > 1 else "",
                                                                             "",      # epilogue kinds (EK) of one tile shape: one class, as profiling.roofline groups them
                                                                             "")      # non-temporal-store twin (NT) of a kind: same class
    m = re.search(r"conv3x3_ring_kernel<([^,]+), (\d+), (\d+), (\d+), (\d+), (\d+)>", kernel)
    if m:
        return "conv3x3_ring_kernel<%s,MR=%s,WR=%s,NR=%s,SCH=%s,NBUF=%s>" % (TY.get(m.group(1), m.group(1)), *m.groups()[1:])
    m = re.search(r"(?<!thin_)wgrad_kernel<([^,]+), (\d+), (\d+),", kernel)
    if m:
        return "wgrad_kernel<%s,KS=%s,S=%s>+reduce" % (TY.get(m.group(1), m.group(1)), m.group(2), m.group(3))
    m = re.search(r"thin_in_kernel<([^,]+), (true|false)", kernel)          # ops.ThinLaunch.label
    if m:
        return "thin_in_kernel<%s%s>" % (TY.get(m.group(1), m.group(1)), ",mask" if m.group(2) == "true" else "")
    m = re.search(r"dense_chain_kernel<([^,>]+), (\d+)>", kernel)              # ops.DenseChain (rows per compute wave -> tile rows)
    if m:
        return "dense_chain_kernel<%s,tile=%dx16>" % (TY.get(m.group(1), m.group(1)), 4 * int(m.group(2)))
    m = re.search(r"(thin_out|thin_wgrad)_kernel<([^,>]+)>", kernel)
    if m:
        return "%s_kernel<%s>" % (m.group(1), TY.get(m.group(2), m.group(2)))
    return None


def counters(d):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.defaultdict(lambda: collections.defaultdict(int))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            agg[row["Kernel_Name"]][row["Counter_Name"]] += float(row["Counter_Value"])
            n[row["Kernel_Name"]][row["Counter_Name"]] += 1
    return agg, n


src = sys.argv[1]
TAG = arg("--tag", "r04")
out = {"csrc_sha": bench.csrc_sha(), "batch": int(arg("--batch", "32")), "lr_size": int(arg("--lr-size", "128")), "workloads": {},
       "how": "tools/profile_round.sh + tools/pmc_to_json.py; traffic = (2*RDREQ + WRREQ)*64 B per launch, mfma_util = MFMA_BUSY / (GUI_ACTIVE/8 * 1024)"}
text = []
for wl in ("g_only", "gan"):
    st = glob.glob(f"{src}/stats_{wl}/**/*kernel_stats.csv", recursive=True)
    if st:
        shutil.copy(st[0], os.path.join(ROOT, "profiles", f"{TAG}_{wl}_b32_kernel_stats.csv"))
    tcc, ntcc = counters(f"{src}/pmc_tcc_{wl}")
    sq, nsq = counters(f"{src}/pmc_sq_{wl}")
    W = out["workloads"].setdefault(wl, {})
    rows = collections.defaultdict(lambda: collections.defaultdict(float))
    for k, c in tcc.items():
        lab = label(k)
        if lab and "reduce_kernel" not in k:
            rows[lab]["rd"] += c.get("TCC_EA0_RDREQ_sum", 0.0); rows[lab]["wr"] += c.get("TCC_EA0_WRREQ_sum", 0.0)
            rows[lab]["n"] += ntcc[k].get("TCC_EA0_RDREQ_sum", 0)
    for k, c in sq.items():
        lab = label(k)
        if lab and "reduce_kernel" not in k:
            for cn in ("SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
                rows[lab][cn] += c.get(cn, 0.0)
    text.append(f"== {wl}")
    for lab, r in sorted(rows.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0)):
        e = {}
        if r["n"]:
            e["traffic_bytes_per_launch"] = round((2 * r["rd"] + r["wr"]) * 64 / r["n"])
            e["launches_profiled"] = int(r["n"])
        if r["GRBM_GUI_ACTIVE"]:
            e["mfma_util"] = round(r["SQ_VALU_MFMA_BUSY_CYCLES"] / (r["GRBM_GUI_ACTIVE"] / 8 * 1024), 4)
            wc = r["SQ_WAVE_CYCLES"] or 1.0
            e["wave_cycle_split"] = {"wait_any": round(r["SQ_WAIT_ANY"] / wc, 3), "issue_stall": round(r["SQ_WAIT_INST_ANY"] / wc, 3),
                                     "issuing": round(r["SQ_ACTIVE_INST_ANY"] / wc, 3)}
        W[lab] = e
        text.append(f"  {lab}: {json.dumps(e)}")
json.dump(out, open(os.path.join(ROOT, "profiles", f"{TAG}_pmc.json"), "w"), indent=1)
open(os.path.join(ROOT, "profiles", f"{TAG}_pmc_summary.txt"), "w").write("\n".join(text) + "\n")
print("\n".join(text))
