#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export CFG3_S=200 CFG3_K=512 CFG3_ENGINE=6 CFG3_U=5000 CFG3_ROUNDS=6
for v in "" _NO_EPILOGUE _ONE_FRAG _BOTH; do
  ( for i in $(seq 1 6); do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | sed 's/.*(\([0-9]*Mhz\)).*/\1/; s/.*(W): //' | tr '\n' ' '; echo; sleep 1; done ) > gpurun_out/smi$v.txt &
  SMI=$!
  SR_PYGMM_LIB=$PWD/speaker-recognition_amd/lib/pygmm$v.so timeout 120 python scripts/bench_cfg3_shard.py 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('variant [$v]', d['score_kernel_s'], d['frames_per_s'])"
  wait $SMI; sed -n '3,5p' gpurun_out/smi$v.txt | tr '\n' '|'; echo
done
