#!/bin/bash
# round 2, session 2: the two-column h2s kernel -- parity tests, then configs[2]-shaped scoring-only timing of the variants
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_gmm.py -m gpu -x -q -k "shared_sigma or h2s or cfg3_shape" 2>&1 | tail -5
export CFG3_S=200 CFG3_K=512 CFG3_ENGINE=6 CFG3_U=5000 CFG3_ROUNDS=5
run() {  # label lib cols
  CFG3_COLS=$3 SR_PYGMM_LIB=$PWD/speaker-recognition_amd/lib/$2 timeout 200 python scripts/bench_cfg3_shard.py 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('variant [$1]', d['score_kernel_s'], d['frames_per_s'], d['checks'], d['kernel'][:40])"
}
run cols1 pygmm.so 1
run cols2 pygmm.so 2
run cols2_G4 pygmm_G4.so 2
run cols2_PRIO pygmm_PRIO.so 2
run cols1_again pygmm.so 1
