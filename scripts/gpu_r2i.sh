#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
echo "=== pytest shared"; timeout 900 python -m pytest tests/test_gpu_gmm.py -m gpu -x -q -k "shared or h2s or cfg2 or cfg3" 2>&1 | tail -4
export CFG3_S=200 CFG3_K=512 CFG3_ENGINE=6
for u in 3000 10000; do CFG3_U=$u timeout 300 python scripts/bench_cfg3_shard.py 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print($u, d['score_kernel_s'], d['frames_per_s'], d['algorithmic_tflops'], d['checks'])"; done
export CFG3_U=10000
PMC_CMD="python $PWD/scripts/bench_cfg3_shard.py" PMC_SETS="TCC_HIT_sum TCC_MISS_sum;FETCH_SIZE" bash scripts/pmc.sh 2>&1 | grep "h2s_kernel<8, 8, false>"
