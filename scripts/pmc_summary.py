"""Summarise rocprofv3 --pmc CSVs: mean counter value per kernel (per dispatch)."""
import csv, glob, sys, collections
d = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(d + "/*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        k = next((t for t in ("k_assemble_rows_march", "k_assemble_rows_burst", "k_assemble_rows", "k_assemble_entries", "k_assemble_chain", "k_assemble_edges", "k_spmv", "k_cg_update", "k_cg_p") if t in k), None)
        if k: acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()):
        print(f"  {c:42s} {sum(v)/len(v):16.1f}  (n={len(v)})")
