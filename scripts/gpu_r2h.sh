#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export CFG3_S=200 CFG3_K=512 CFG3_ENGINE=6
for u in 3000 10000; do CFG3_U=$u timeout 300 python scripts/bench_cfg3_shard.py 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print($u, d['score_kernel_s'], d['frames_per_s'], d['algorithmic_tflops'], d['checks'])"; done
