#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export CFG3_S=200 CFG3_K=512 CFG3_ENGINE=6 CFG3_U=5000 CFG3_ROUNDS=10
r() { CFG3_SHAPE=$2 CFG3_TPL=$3 SR_PYGMM_LIB=$PWD/speaker-recognition_amd/lib/pygmm$1.so timeout 200 python scripts/bench_cfg3_shard.py 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('lib[$1] shape $2 tpl $3:', round(d['score_kernel_s'],4), d['checks']['own_speaker_wins'])"; }
r "" 2 0
r _W8 3 0
r _W8G2 3 0
r _W8G8 3 0
r _W8 3 0
r "" 2 0
