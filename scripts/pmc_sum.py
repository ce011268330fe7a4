"""mean counter values per kernel (assembly kernels only) of rocprofv3 --pmc output directories"""
import collections, csv, glob, sys
for d in sys.argv[1:]:
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in sorted(glob.glob(d + "/**/*_counter_collection.csv", recursive=True)):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            if "k_assemble_rows" not in n:
                continue
            n = n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
            acc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(d)
    for k, cs in acc.items():
        print("  " + k)
        for c, v in sorted(cs.items()):
            print("    %-24s %16.1f (n=%d)" % (c, sum(v) / len(v), len(v)))
