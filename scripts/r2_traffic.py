"""profiles/r2_traffic.json from the rocprofv3 --pmc CSVs of scripts/r2_profile.sh: HBM bytes per launch
of SpMV and of the continuity edge assembly (FETCH_SIZE x 2 as MI355X_MICROARCH.md prescribes for gfx950,
+ WRITE_SIZE; the marching assembly is two kernels per launch -- pattern runs + the remaining slices)."""
import collections, csv, glob, json, sys
d = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(d + "/*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        k = next((t for t in ("k_assemble_rows_march", "k_assemble_rows_burst", "k_assemble_rows", "k_spmv") if t in k), None)
        if k:
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
mean = {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in acc.items()}
out = {"note": "rocprofv3 --pmc passes (scripts/r2_profile.sh -> scripts/pmc_short.sh over scripts/pmc_asm.py 256) on MI355X, "
               "round 2; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950 (calibrated in round 1 with "
               "scripts/membench.hip: a 1 GiB streaming read reports 524,299 KB); bytes per launch", "n": 256}
def rec(c):
    return {"FETCH_SIZE_KB": c["FETCH_SIZE"], "WRITE_SIZE_KB": c["WRITE_SIZE"], "TCC_REQ": c["TCC_REQ_sum"],
            "TCC_MISS": c["TCC_MISS_sum"], "TCC_HIT": c["TCC_HIT_sum"],
            "hbm_bytes": (2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024}
out["k_spmv"] = rec(mean["k_spmv"])
parts = [k for k in ("k_assemble_rows_march", "k_assemble_rows_burst", "k_assemble_rows") if k in mean]
tot = collections.Counter()
for k in parts:
    tot.update({c: v for c, v in mean[k].items()})
out["k_assemble_rows"] = rec(tot)
out["k_assemble_rows"]["kernels"] = parts
print(json.dumps(out, indent=1))
