#!/bin/bash
# stall / energy experiment on the h2s kernel: strip parts off one by one, kernel time + clock/power under load
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export CFG3_S=200 CFG3_K=512 CFG3_ENGINE=6 CFG3_U=5000 CFG3_ROUNDS=30 CFG3_COLS=1
for v in "" _NOEPI _E_F _E_F_D _E_F_D_B _D _B; do
  ( while true; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | sed 's/.*(\([0-9]*\)Mhz).*/\1/; s/.*(W): //' | tr '\n' ' '; echo; sleep 0.3; done ) > gpurun_out/smi$v.txt &
  SMI=$!
  SR_PYGMM_LIB=$PWD/speaker-recognition_amd/lib/pygmm$v.so timeout 200 python scripts/bench_cfg3_shard.py 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('variant [$v]', round(d['score_kernel_s'],4), 'wall', round(d['wall_s'],4))"
  kill $SMI; wait $SMI 2>/dev/null
  # the samples under load = those with the highest power
  sort -k2 -n -r gpurun_out/smi$v.txt | head -6 | tr '\n' '|'; echo
done
