"""profiles/r3_traffic.json from the rocprofv3 --pmc CSVs of scripts/r3_profile.sh: HBM bytes per launch of the
hot kernels (FETCH_SIZE x 2 as MI355X_MICROARCH.md prescribes for gfx950, + WRITE_SIZE; KB -> bytes), per box size.
usage: python scripts/r3_traffic.py <n>=<pmc dir> [<n>=<pmc dir> ...]"""
import collections, csv, glob, json, re, sys

KERNELS = {  # name in the JSON -> substrings that all must occur in the (demangled) kernel name
    "k_spmv<DOT>": ("k_spmv<true, false>",), "k_spmv<DINV>": ("k_spmv<false, true>",), "k_spmv": ("k_spmv<false, false>",),
    "k_assemble_rows_lagpf<ContOp>": ("k_assemble_rows_lagpf<", "ContOp,"), "k_assemble_rows_lagpf<ScalOp>": ("k_assemble_rows_lagpf<", "ScalOp,"),
    "k_assemble_rows_lag<ContOp>": ("k_assemble_rows_lag<", "ContOp,"), "k_assemble_rows_burst<ContOp>": ("k_assemble_rows_burst<", "ContOp,"),
    "k_assemble_rows_lag<ScalOp>": ("k_assemble_rows_lag<", "ScalOp,"), "k_assemble_rows_burst<ScalOp>": ("k_assemble_rows_burst<", "ScalOp,"),
    "k_assemble_rows_march<ContOp>": ("k_assemble_rows_march<", "ContOp,"), "k_assemble_rows<ScalOp>": ("k_assemble_rows<", "ScalOp,"),
    "k_cg_update": ("k_cg_update(",), "k_cg_p": ("k_cg_p(",), "k_cg_flush_x": ("k_cg_flush_x(",),
    "k_gm_dots_all": ("k_gm_dots_all(",), "k_gm_upd_dots": ("k_gm_upd_dots<",), "k_gm_update": ("k_gm_update(",),
    "k_flow_sweep<FWD>": ("k_flow_sweep<true,",), "k_flow_sweep<BWD>": ("k_flow_sweep<false,",),
}


def classify(name):
    for k, subs in KERNELS.items():
        if all(s in name for s in subs):
            return k
    return None


def collect(d):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in sorted(glob.glob(d + "/*_counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            k = classify(r["Kernel_Name"])
            if k:
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out = {}
    for k, cs in acc.items():
        c = {name: sum(v) / len(v) for name, v in cs.items()}
        if "FETCH_SIZE" not in c or "WRITE_SIZE" not in c:
            continue
        out[k] = {"FETCH_SIZE_KB": c["FETCH_SIZE"], "WRITE_SIZE_KB": c["WRITE_SIZE"], "TCC_REQ": c.get("TCC_REQ_sum"),
                  "TCC_MISS": c.get("TCC_MISS_sum"), "TCC_HIT": c.get("TCC_HIT_sum"),
                  "hbm_bytes": (2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024, "launches_seen": len(cs["FETCH_SIZE"])}
    return out


res = {"note": "rocprofv3 --pmc passes (scripts/r3_profile.sh over scripts/pmc_r3.py) on MI355X, round 3; FETCH_SIZE doubled as "
               "MI355X_MICROARCH.md prescribes for gfx950 (calibrated in round 1 with scripts/membench.hip: a 1 GiB streaming "
               "read reports 524,299 KB); bytes per launch, mean over the launches seen (the GMRES kernels: over basis sizes "
               "1..30)", "boxes": {}}
for arg in sys.argv[1:]:
    n, d = arg.split("=")
    res["boxes"][n] = collect(d)
print(json.dumps(res, indent=1))
