#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export CFG3_S=200 CFG3_K=512 CFG3_ENGINE=6 CFG3_U=5000 CFG3_ROUNDS=30
run() { # lib-suffix shape
  v=$1; export CFG3_SHAPE=$2
  ( while true; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | sed 's/.*(\([0-9]*\)Mhz).*/\1/; s/.*(W): //' | tr '\n' ' '; echo; sleep 0.3; done ) > gpurun_out/smi.txt &
  SMI=$!
  SR_PYGMM_LIB=$PWD/speaker-recognition_amd/lib/pygmm$v.so timeout 200 python scripts/bench_cfg3_shard.py 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('variant [$v shape=$2]', round(d['score_kernel_s'],4), d['checks']['own_speaker_wins'], d['kernel'][:36])"
  kill $SMI; wait $SMI 2>/dev/null
  sort -k2 -n -r gpurun_out/smi.txt | sed -n '3,4p' | sed 's/=* Power Consumption =*//' | tr '\n' '|'; echo
}
for a in "$@"; do v=${a%%:*}; [ "$v" = base ] && v=""; run "$v" ${a##*:}; done
