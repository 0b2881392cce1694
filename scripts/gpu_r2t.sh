#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 900 python -m pytest tests/test_gpu_gmm.py -m gpu -x -q -k "shared_sigma or h2s or cfg3_shape or cfg2 or hybrid" 2>&1 | tail -3
bash scripts/gpu_r2q.sh base:1 base:2
CFG3_S=209 CFG3_K=512 CFG3_ENGINE=6 CFG3_U=5000 CFG3_ROUNDS=10 CFG3_SHAPE=2 timeout 200 python scripts/bench_cfg3_shard.py 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('S=209+1 (full blocks):', round(d['score_kernel_s'],4))"
