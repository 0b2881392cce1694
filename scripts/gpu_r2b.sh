#!/bin/bash
# round 2, call B: shared-sigma fp16 engine: parity tests + configs[2] timing
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out; mkdir -p $O
echo "=== pytest gmm"; timeout 1500 python -m pytest tests/test_gpu_gmm.py -m gpu -x -q 2>&1 | tail -25 | tee $O/r02b_pytest_gmm.log
echo "=== cfg2 timing"
for e in 6 4; do CFG3_S=200 CFG3_K=512 CFG3_U=10000 CFG3_ENGINE=$e timeout 600 python scripts/bench_cfg3_shard.py 2>&1 | tail -1; done | tee $O/r02b_cfg2.txt
CFG3_S=200 CFG3_K=512 CFG3_U=2000 CFG3_ENGINE=6 CFG3_FORCE_EXC=1 timeout 600 python scripts/bench_cfg3_shard.py 2>&1 | tail -1 | tee -a $O/r02b_cfg2.txt
