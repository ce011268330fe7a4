R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; cd /tmp
timeout -k 5 60 $R/scripts/dupbench.bin 8 && timeout -k 5 60 $R/scripts/dupbench.bin 9 || exit 1
timeout -s KILL 60 rocprofv3 --pmc WRITE_SIZE TCC_REQ_sum --output-format csv -d $R/gpurun_out/dup_8 -o p -- $R/scripts/dupbench.bin 8 > $R/gpurun_out/dup_8.log 2>&1 || { tail -3 $R/gpurun_out/dup_8.log; exit 1; }
timeout -s KILL 60 rocprofv3 --pmc TCC_REQ_sum --output-format csv -d $R/gpurun_out/dup_0r -o p -- $R/scripts/dupbench.bin 9 > $R/gpurun_out/dup_0r.log 2>&1
python3 - $R/gpurun_out/dup_8/p_counter_collection.csv $R/gpurun_out/dup_0r/p_counter_collection.csv <<'PY'
import csv, sys, collections
for f in sys.argv[1:]:
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Kernel_Name"].startswith("k("):
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(f.split("/")[-2], {k: round(sum(v) / len(v) / 1e6, 2) for k, v in acc.items()})
PY
