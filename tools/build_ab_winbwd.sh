#!/bin/bash
# A/B builds of libnmrf_hip.so whose window-attention backward kernel returns after phase k (0: loads only ... 4: before dk | dv) ->
# nmrf_amd/lib/ab_wb<k>/libnmrf_hip.so, for tools/winbwd_phase_bench.py.  Run after `python -m nmrf_amd.build`.
cd "$(dirname "$0")/.." || exit 1
set -e
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function"
for k in 0 1 2 3 4; do
  mkdir -p nmrf_amd/lib/ab_wb$k nmrf_amd/build_wb
  ( /opt/rocm/bin/hipcc $FLAGS -DWB_STOP=$k -c nmrf_amd/csrc/backward.hip -o nmrf_amd/build_wb/backward_$k.o
    objs=""; for o in nmrf_amd/build/*.o; do if [ "$(basename $o)" = backward.o ]; then objs="$objs nmrf_amd/build_wb/backward_$k.o"; else objs="$objs $o"; fi; done
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o nmrf_amd/lib/ab_wb$k/libnmrf_hip.so $objs ) &
done
wait
ls nmrf_amd/lib/ab_wb*/libnmrf_hip.so
