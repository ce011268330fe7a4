set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/flowprobe2
for dims in "256 256 32" "256 256 64" "256 256 128"; do
  tag=$(echo $dims | tr ' ' 'x')
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/flowprobe2/$tag -o p -- python3 scripts/sgsflow_probe.py $dims 10 > gpurun_out/flowprobe2/$tag.log 2>&1
  grep "ms per CG" gpurun_out/flowprobe2/$tag.log | cut -c1-160
  grep -E "k_flow_sweep" gpurun_out/flowprobe2/$tag/p_kernel_stats.csv | awk -F'",' '{print "   ", substr($1,36,30), $2}' | cut -c1-100
done
