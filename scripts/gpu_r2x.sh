#!/bin/bash
# last validation of round 2: GPU suite, smoke, kernel trace + phase trace of an EM / MAP pass after the vector-engine fix
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out; mkdir -p $O
echo "=== pytest -m gpu"; timeout 240 python -m pytest tests -x -q -m gpu < /dev/null 2>&1 | tail -6
echo "=== smoke"; timeout 60 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" < /dev/null 2>&1 | tail -2
echo "=== kernel trace of an EM + MAP pass"
(cd /tmp && N=400000 EM_IT=2 MAPS=3 timeout 80 rocprofv3 --kernel-trace --stats --output-format csv -d $O/tprof2 -o tp -- python $R/scripts/debug/train_prof.py > $O/tprof2_run.log 2>&1 < /dev/null)
f=$(find $O/tprof2 -name "*kernel_trace*.csv" < /dev/null | head -1)
if [ -n "$f" ]; then python - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Kernel_Name']
    if 'gmm_score_kernel' in n or 'em_stats' in n:
        print("%-44s %8.3f ms  grid %7s x %s  scratch %5s B/lane" % (n.split('(')[0][:44], (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6, r['Grid_Size_X'], r['Grid_Size_Y'], r['Scratch_Size']))
PY
else echo "no trace"; tail -3 $O/tprof2_run.log; fi
echo "=== phase trace"; SKIP_KMEANS=1 TRAIN_TRACE=1 timeout 80 python scripts/bench_train_scale.py 2048 400000 1 < /dev/null 2>&1 | tail -12
