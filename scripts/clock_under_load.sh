#!/bin/bash
# Samples the GPU's shader clock and socket power once a second while a forced scoring engine runs
# back to back for ~12 s (evidence for DESIGN.md 2.1: the split-bf16 kernel is power-limited).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/clock; mkdir -p $O
for eng in 3 2 1; do
  ( for i in $(seq 1 14); do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr '\n' ' '; echo; sleep 1; done ) > $O/smi_engine$eng.txt &
  SMI=$!
  timeout 60 python scripts/time_score_engine.py $eng 1 0 ${ROUNDS:-2400} | sed 's/ms .*//' | awk '{n=NF; printf "engine %s: first %s ... last %s %s %s ms\n", $2, $6, $(n-2), $(n-1), $n}' | tee $O/run_engine$eng.txt
  wait $SMI
  echo "--- smi samples, engine $eng"; cat $O/smi_engine$eng.txt | cut -c1-220 | sed -n '3,12p'
done
