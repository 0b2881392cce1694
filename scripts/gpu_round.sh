#!/bin/bash
# One gpurun call: tests, smoke, bench, variant sweep, rocprof summary.  Everything lands in gpurun_out/.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
rocminfo 2>/dev/null | grep -E "Marketing Name|gfx" | head -4 > $O/device.txt
nproc >> $O/device.txt
echo "=== smoke" ; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 | tee $O/smoke.log
echo "=== pytest gpu" ; timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -30 | tee $O/pytest_gpu.log
echo "=== tune" ; timeout 600 python scripts/tune_score.py 2>&1 | tail -40 | tee $O/tune.log
echo "=== bench" ; timeout 900 python bench.py --steps 5 --warmup 2 2>&1 | tail -5 | tee $O/bench.log
echo "=== rocprof" ; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$O/prof -o r01 -- python $OLDPWD/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OLDPWD/$O/rocprof_run.log 2>&1)
ls -R $O/prof | head -20
find $O/prof -name "*kernel_stats*.csv" | head -3 | while read f; do echo "== $f"; head -12 "$f"; done
