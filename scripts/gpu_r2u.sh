#!/bin/bash
# launches per pass of the 12-wave form on the full configs[2] grid (10 M frames): tail of every launch vs L2 phase drift
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export CFG3_S=200 CFG3_K=512 CFG3_ENGINE=6 CFG3_U=10000 CFG3_ROUNDS=6 CFG3_SHAPE=2
for tpl in 0 18432 36864 73728 147456 400000; do
  CFG3_TPL=$tpl timeout 300 python scripts/bench_cfg3_shard.py 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('tiles per launch $tpl:', round(d['score_kernel_s'],4), d['kernel'][-26:])"
done
