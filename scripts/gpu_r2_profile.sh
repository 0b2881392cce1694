#!/bin/bash
# round 2 evidence: full bench line, rocprofv3 kernel stats of the same command, PMC passes (own runs)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out; mkdir -p $O
rocminfo 2>/dev/null | grep -E "Marketing Name|gfx" | head -4 > $O/r02b_device.txt; nproc >> $O/r02b_device.txt
echo "=== bench (full)"; timeout 1500 python bench.py > $O/r02b_bench.json 2> $O/r02b_bench.err; tail -2 $O/r02b_bench.err
CMD="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-config-blocks"
echo "=== rocprof kernel stats"; (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o r02b -- $CMD > $O/r02b_rocprof_run.log 2>&1)
f=$(find $O/prof -name "*kernel_stats*.csv" | head -1); echo $f; [ -n "$f" ] && cp $f $O/r02b_kernel_stats.csv && head -12 $O/r02b_kernel_stats.csv | cut -c1-200
echo "=== pmc"
PMC_CMD="$CMD" PMC_SETS="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES;SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS;SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM;SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU_TRANS SQ_ACTIVE_INST_MISC;GRBM_GUI_ACTIVE GRBM_COUNT;FETCH_SIZE;WRITE_SIZE;TCC_HIT_sum TCC_MISS_sum" bash scripts/pmc.sh 2>&1 | grep -v "^$" | grep -v "rocclr\|fillBuffer\|copyBuffer" > $O/r02b_pmc.txt
cut -c1-400 $O/r02b_pmc.txt | head -60
