#!/bin/bash
# One GPU-box session, stage by stage (replaces the per-session scripts of rounds 1-2):
#
#   gpurun --timeout 900 -- 'bash scripts/gpu_session.sh TAG stage [stage ...]'
#
# stages (each under its own `timeout`, stdin from /dev/null; outputs under gpurun_out/TAG_*):
#   test[=EXPR]      pytest -m gpu (-k EXPR)                  smoke            __graft_entry__.smoke()
#   bench[=ARGS]     python bench.py ARGS -> TAG_bench.json   device           rocminfo / nproc / rocm-smi
#   prof[=ARGS]      rocprofv3 --kernel-trace --stats of `bench.py ARGS` (default: 3 steps, headline only)
#   pmc[=SETS]       PMC passes of the same command, each in its own run (scripts/pmc.sh; SETS ';'-separated)
#   py=SCRIPT[,ARG…] python SCRIPT ARG…  (cwd = repo root)   sh=CMD           bash -c CMD
# A stage's time limit: STAGE_TIMEOUT (seconds, default 600).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out; mkdir -p $O
TAG=$1; shift
T=${STAGE_TIMEOUT:-600}
DEF_ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-config-blocks --no-traffic"
for st in "$@"; do
  name=${st%%=*}; arg=""; [ "$st" != "$name" ] && arg=${st#*=}
  echo "=== [$TAG] $name $arg"
  case $name in
    test)   if [ -n "$arg" ]; then timeout $T python -m pytest tests -x -q -m gpu -k "$arg" < /dev/null 2>&1 | tail -15
            else timeout $T python -m pytest tests -x -q -m gpu < /dev/null 2>&1 | tail -15; fi ;;
    smoke)  timeout 120 python -c "import __graft_entry__ as g; g.smoke()" < /dev/null 2>&1 | tail -3 ;;
    device) (rocminfo 2>/dev/null | grep -E "Marketing Name|gfx" | head -4; nproc; rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power" | head -6) > $O/${TAG}_device.txt < /dev/null; cat $O/${TAG}_device.txt ;;
    bench)  timeout $T python bench.py $arg > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err < /dev/null; grep -v '^{' $O/${TAG}_bench.err | tail -5 | cut -c1-300; [ -f bench_blocks.json ] && cp bench_blocks.json $O/${TAG}_blocks.json; tail -1 $O/${TAG}_bench.json | cut -c1-1600 ;;
    prof)   CMD="python $R/bench.py ${arg:-$DEF_ARGS}"
            (cd /tmp && timeout $T rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_prof -o $TAG -- $CMD > $O/${TAG}_rocprof_run.log 2>&1 < /dev/null)
            f=$(find $O/${TAG}_prof -name "*kernel_stats*.csv" < /dev/null | head -1)
            if [ -n "$f" ]; then cp $f $O/${TAG}_kernel_stats.csv; head -14 $O/${TAG}_kernel_stats.csv | cut -c1-220; else tail -5 $O/${TAG}_rocprof_run.log; fi ;;
    pmc)    PMC_CMD="python $R/bench.py $DEF_ARGS" PMC_SETS="$arg" bash scripts/pmc.sh < /dev/null 2>&1 | grep -v "^$" | grep -v "rocclr\|fillBuffer\|copyBuffer" > $O/${TAG}_pmc.txt; cut -c1-400 $O/${TAG}_pmc.txt | head -60 ;;
    py)     IFS=',' read -ra A <<< "$arg"; timeout $T python "${A[@]}" < /dev/null 2>&1 | tail -40 ;;
    sh)     timeout $T bash -c "$arg" < /dev/null 2>&1 | tail -40 ;;
    *)      echo "unknown stage $name" ;;
  esac
done
