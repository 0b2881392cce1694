#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_gmm.py -m gpu -x -q -k "shared_sigma or h2s or cfg3_shape" 2>&1 | tail -5
CFG3_S=200 CFG3_K=512 CFG3_ENGINE=6 CFG3_U=5000 CFG3_ROUNDS=1 CFG3_SHAPE=3 SR_PYGMM_LIB=$PWD/speaker-recognition_amd/lib/pygmm_STAMP.so timeout 200 python scripts/bench_cfg3_shard.py > gpurun_out/stamps2.txt 2>&1; sed -n 9,16p gpurun_out/stamps2.txt
bash scripts/gpu_r2q.sh base:1 base:2 base:3
