#!/bin/bash
# Same-box A/B: the tools library with SOME sources taken from an older commit, everything else from the current tree's objects.
#   tools/build_ab_lib.sh <rev> <file.hip> [...]   ->  nmrf_amd/lib/ab/libnmrf_hip_debug.so   (travels with gpurun; kernel_bench.py --lib)
set -e
cd "$(dirname "$0")/.."
rev=$1; shift
mkdir -p nmrf_amd/build_ab nmrf_amd/lib/ab
python -m nmrf_amd.build --debug > /dev/null
objs=""
for o in nmrf_amd/build_debug/*.o; do
  b=$(basename $o .o); skip=0
  for f in "$@"; do [ "$(basename $f .hip)" == "$b" ] && skip=1; done
  [ $skip == 0 ] && objs="$objs $o"
done
for f in "$@"; do
  b=$(basename $f .hip)
  git show $rev:nmrf_amd/csrc/$f > nmrf_amd/csrc/_ab_$b.hip
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -DNMRF_DEBUG_PROBES -c nmrf_amd/csrc/_ab_$b.hip -o nmrf_amd/build_ab/$b.o
  rm nmrf_amd/csrc/_ab_$b.hip
  objs="$objs nmrf_amd/build_ab/$b.o"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o nmrf_amd/lib/ab/libnmrf_hip_debug.so $objs
echo built nmrf_amd/lib/ab/libnmrf_hip_debug.so with $@ from $rev
