#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
export CFG3_S=200 CFG3_K=512 CFG3_ENGINE=6
for u in 1500 3000 5000 7000; do
  export CFG3_U=$u
  echo "U=$u"; PMC_CMD="python $PWD/scripts/bench_cfg3_shard.py" PMC_SETS="TCC_HIT_sum TCC_MISS_sum" bash scripts/pmc.sh 2>&1 | grep "h2s_kernel<8, 8, false>"
done
