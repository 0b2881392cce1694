#!/bin/bash
# round 2, call C: what bounds the h2s kernel -- rounds experiment + L2 hit rate + MFMA busy
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out; mkdir -p $O
export CFG3_S=200 CFG3_K=512 CFG3_ENGINE=6
for u in 96 192 960 3000; do CFG3_U=$u timeout 300 python scripts/bench_cfg3_shard.py 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print($u, d['score_kernel_s'], d['frames_per_s'], d['algorithmic_tflops'])"; done | tee $O/r02c_rounds.txt
export CFG3_U=3000
PMC_CMD="python $PWD/scripts/bench_cfg3_shard.py" PMC_SETS="TCC_HIT_sum TCC_MISS_sum;SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY;GRBM_GUI_ACTIVE;FETCH_SIZE;SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" bash scripts/pmc.sh 2>&1 | grep -v "^$" | tee $O/r02c_pmc.txt | cut -c1-600
