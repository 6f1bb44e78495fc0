#!/bin/bash
# round 6, call R: the whole GPU suite on the tree (stripe LePE on the matrix pipe, tree reduction, reducer tests, dropout rates)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
export HSA_ENABLE_IPC_MODE_LEGACY=0
( timeout 1800 python -m pytest tests -m gpu -q --tb=short -rf -p no:cacheprovider 2>&1 | tail -120 ) > gpurun_out/r06r_pytest.log
grep -h "WTA decisions\|RCCL" gpurun_out/r06r_pytest.log | cut -c1-230
tail -6 gpurun_out/r06r_pytest.log | cut -c1-250
