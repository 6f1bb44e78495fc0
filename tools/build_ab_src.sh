#!/bin/bash
# A/B build of libnmrf_hip.so with SOME sources taken from an older commit, everything else from the current tree's objects.
#   tools/build_ab_src.sh <name> <rev> file1.hip [...]   ->  nmrf_amd/lib/ab_<name>/libnmrf_hip.so   (bench.py --lib / tools/gpu_ab.sh)
cd "$(dirname "$0")/.." || exit 1
set -e
name=$1; rev=$2; shift 2
python -m nmrf_amd.build --main-only > /dev/null
mkdir -p nmrf_amd/lib/ab_$name nmrf_amd/build_ab/$name
objs=""
for o in nmrf_amd/build/*.o; do
  b=$(basename $o .o); skip=0
  for f in "$@"; do [ "$(basename $f .hip)" == "$b" ] && skip=1; done
  [ $skip == 0 ] && objs="$objs $o"
done
for f in "$@"; do
  b=$(basename $f .hip)
  git show $rev:nmrf_amd/csrc/$f > nmrf_amd/csrc/_ab_$b.hip
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -c nmrf_amd/csrc/_ab_$b.hip -o nmrf_amd/build_ab/$name/$b.o
  rm nmrf_amd/csrc/_ab_$b.hip
  objs="$objs nmrf_amd/build_ab/$name/$b.o"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o nmrf_amd/lib/ab_$name/libnmrf_hip.so $objs
echo built nmrf_amd/lib/ab_$name/libnmrf_hip.so with $@ from $rev
