#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
export CFG3_S=200 CFG3_K=512 CFG3_ENGINE=6 CFG3_U=10000
for tpl in 1000000 36864 18432 9216 4608; do
  export CFG3_TPL=$tpl
  timeout 300 python scripts/bench_cfg3_shard.py 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('tpl', $tpl, d['score_kernel_s'], d['frames_per_s'], d['checks']['own_speaker_wins'])"
done
for tpl in 18432 9216; do
  export CFG3_TPL=$tpl
  echo "tpl $tpl"; PMC_CMD="python $PWD/scripts/bench_cfg3_shard.py" PMC_SETS="TCC_HIT_sum TCC_MISS_sum" bash scripts/pmc.sh 2>&1 | grep "h2s_kernel<8, 8, false>"
done
