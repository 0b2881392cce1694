#!/bin/bash
# round 2, call A: fp16 subnormal check, GMM parity tests with the new engine, accuracy + timing
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out; mkdir -p $O
echo "=== denorm"; timeout 60 scripts/ubench/mfma_f16_denorm 2>&1 | tee $O/r02_f16_denorm.txt
echo "=== pytest gmm"; timeout 1200 python -m pytest tests/test_gpu_gmm.py -m gpu -x -q 2>&1 | tail -25 | tee $O/r02a_pytest_gmm.log
echo "=== accuracy"; timeout 300 python scripts/engine_accuracy.py 2>&1 | tee $O/r02_engine_accuracy.json | tail -40
echo "=== timing"; for e in 3 5; do timeout 200 python scripts/time_score_engine.py $e 1 0 6 2>&1 | tail -1; done | tee $O/r02a_timing.txt
timeout 200 python scripts/time_score_engine.py 5 2 0 6 2>&1 | tail -1 | tee -a $O/r02a_timing.txt
