#!/bin/bash
# A/B build of libnmrf_hip.so: the block / chain kernels compiled WITHOUT packed fp32 VALU (-fno-slp-vectorize, scalar split2u) ->
# nmrf_amd/lib/ab_nopk/libnmrf_hip.so (same ABI; bench.py --lib / tools/gpu_ab.sh).  Run after `python -m nmrf_amd.build`.
cd "$(dirname "$0")/.." || exit 1
set -e
mkdir -p nmrf_amd/lib/ab_nopk nmrf_amd/build_nopk
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function"
for f in ${NOPK_FILES:-nmp_block16 mlp_chain}; do
  /opt/rocm/bin/hipcc $FLAGS -fno-slp-vectorize -DNMRF_SCALAR_SPLIT -c nmrf_amd/csrc/$f.hip -o nmrf_amd/build_nopk/$f.o &
done
wait
objs=""
for o in nmrf_amd/build/*.o; do b=$(basename $o); if [ -f nmrf_amd/build_nopk/$b ]; then objs="$objs nmrf_amd/build_nopk/$b"; else objs="$objs $o"; fi; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o nmrf_amd/lib/ab_nopk/libnmrf_hip.so $objs
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -fno-slp-vectorize -DNMRF_SCALAR_SPLIT -o tools/_ab/mlp_b32_nopk.so tools/ab/mlp_b32.hip 2>/dev/null
ls -la nmrf_amd/lib/ab_nopk/libnmrf_hip.so tools/_ab/mlp_b32_nopk.so
