# duration of one dataflow SGS sweep against the number of chunks on the critical path
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/flowprobe
for dims in "256 8 8" "256 16 8" "256 64 8" "256 8 64" "256 64 64"; do
  tag=$(echo $dims | tr ' ' 'x')
  timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/flowprobe/$tag -o p -- python3 scripts/sgsflow_probe.py $dims 10 > gpurun_out/flowprobe/$tag.log 2>&1
  grep "ms per CG" gpurun_out/flowprobe/$tag.log
  f=$(find gpurun_out/flowprobe/$tag -name "*kernel_stats.csv" | head -1)
  grep -E "k_flow_sweep" $f | cut -c1-200
done
