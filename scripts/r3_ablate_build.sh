# builds ablation variants of the library (timing only, results are WRONG): scripts/ablate/libnalu_gpu_<tag>.so
# usage: bash scripts/r3_ablate_build.sh tag1:-DFLAG1[,-DFLAG2] tag2:...
set -e
cd "$(dirname "$0")/../nalu_amd/csrc"
mkdir -p ../../scripts/ablate
for spec in "$@"; do
  tag=${spec%%:*}; flags=$(echo ${spec#*:} | tr ',' ' ')
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -I../../include -Wall -Wno-unused-function $flags -c assemble.hip -o /tmp/assemble_$tag.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../scripts/ablate/libnalu_gpu_$tag.so core.o /tmp/assemble_$tag.o solve.o geometry.o precond.o scatter.o -lpthread
  echo built $tag
done
