#!/bin/bash
# Shader clock and socket power (rocm-smi, once a second) while the configs[2] scoring kernel runs back to back:
# evidence for "the split-fp16 shared-sigma kernel is power-limited" (DESIGN.md 2.1).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out; mkdir -p $O
( for i in $(seq 1 16); do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr '\n' ' '; echo; sleep 1; done ) > $O/r02_smi_cfg2.txt &
SMI=$!
CFG3_S=200 CFG3_K=512 CFG3_U=10000 CFG3_ENGINE=${ENGINE:-6} CFG3_ROUNDS=40 timeout 120 python scripts/bench_cfg3_shard.py | tail -1 | cut -c1-400
wait $SMI
echo "--- smi samples"; cut -c1-200 $O/r02_smi_cfg2.txt | sed -n '4,14p'
