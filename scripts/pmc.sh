#!/bin/bash
# PMC passes (counters only, each in its own rocprofv3 run; no trace domains combined with --pmc except kernel-trace).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/pmc
rm -rf $O; mkdir -p $O
CMD="${PMC_CMD:-python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline}"
cd /tmp
i=0
if [ -n "$PMC_SETS" ]; then IFS=';' read -ra SETS <<< "$PMC_SETS"; else SETS=(); fi
if [ ${#SETS[@]} -eq 0 ]; then SETS=("SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INST_CYCLES_VMEM" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" \
           "GRBM_GUI_ACTIVE GRBM_COUNT" \
           "FETCH_SIZE" \
           "WRITE_SIZE"); fi
for set in "${SETS[@]}"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/p$i -o p$i -- $CMD > $O/p$i.log 2>&1
  f=$(find $O/p$i -name "*counter_collection.csv" | head -1)
  echo "== pass $i: $set -> $f"
  if [ -n "$f" ]; then
    python - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
with open(sys.argv[1]) as fh:
    for row in csv.DictReader(fh):
        k = row["Kernel_Name"].split("(")[0][-48:]
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
        cnt[(k, row["Counter_Name"])] += 1
for k, d in acc.items():
    print("  ", k, {c: v / max(1, cnt[(k, c)]) for c, v in d.items()})
PY
  else tail -5 $O/p$i.log; fi
done
