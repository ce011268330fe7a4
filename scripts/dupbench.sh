# usage (on the GPU box): bash scripts/dupbench.sh   -> FETCH_SIZE / TCC hits per mode
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; cd /tmp
for m in 0 1 2 3 4 5 6; do
  timeout -k 5 60 $R/scripts/dupbench.bin $m || exit 1
  timeout -s KILL 60 rocprofv3 --pmc FETCH_SIZE TCC_MISS_sum --output-format csv -d $R/gpurun_out/dup_$m -o p -- $R/scripts/dupbench.bin $m > $R/gpurun_out/dup_$m.log 2>&1 || { echo "pmc pass failed"; tail -3 $R/gpurun_out/dup_$m.log; exit 1; }
  python3 - $R/gpurun_out/dup_$m/p_counter_collection.csv <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("   ", {k: round(sum(v) / len(v) / 1e6, 3) for k, v in acc.items()}, "(millions; FETCH_SIZE in KB/1e6: x2 = GB)")
PY
done
