#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
mkdir -p gpurun_out
( timeout 600 python tools/driver_probe.py 2>&1 | grep -v "^$" | tail -90 ) > gpurun_out/driver_probe.log

cat gpurun_out/driver_probe.log gpurun_out/kernel_bench_block.log
