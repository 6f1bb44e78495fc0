#!/bin/bash
# A/B build of libnmrf_hip.so with extra compiler flags on some sources, everything else from the current tree's objects.
#   tools/build_ab_flag.sh <name> "<flags>" file1 file2 ...   ->  nmrf_amd/lib/ab_<name>/libnmrf_hip.so  (bench.py --lib / tools/gpu_ab.sh)
# Run after `python -m nmrf_amd.build`.
cd "$(dirname "$0")/.." || exit 1
set -e
name=$1; extra=$2; shift 2
mkdir -p nmrf_amd/lib/ab_$name nmrf_amd/build_ab/$name
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function"
for f in "$@"; do
  /opt/rocm/bin/hipcc $FLAGS $extra -c nmrf_amd/csrc/$f.hip -o nmrf_amd/build_ab/$name/$f.o &
done
wait
objs=""
for o in nmrf_amd/build/*.o; do b=$(basename $o); if [ -f nmrf_amd/build_ab/$name/$b ]; then objs="$objs nmrf_amd/build_ab/$name/$b"; else objs="$objs $o"; fi; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o nmrf_amd/lib/ab_$name/libnmrf_hip.so $objs
ls -la nmrf_amd/lib/ab_$name/libnmrf_hip.so
