#!/bin/bash
# A round's closing session in ONE gpurun call (every figure of profiles/rNN_* from one box, one build):
#   gpurun --timeout 3400 -- 'bash scripts/final_session.sh r06'
# device, the whole GPU suite, smoke, the full bench line (20 steps), rocprofv3 kernel stats of the headline command and of the WHOLE
# bench, the PMC passes, the driver's 8-rank command stacked on this device, and the one-shape scripts of the session notes.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=${1:-r06}; O=gpurun_out; mkdir -p $O
export STAGE_TIMEOUT=1500
bash scripts/gpu_session.sh ${R}f device test smoke "bench=--steps 20 --warmup 5" prof pmc
# (the whole bench under the profiler: its statistics are written ~70 s in; the traced process then may not exit -- fork helpers in the
# profiler's signal handler -- so this stage gets its own, short limit)
STAGE_TIMEOUT=420 bash scripts/gpu_session.sh ${R}g "prof=--steps 3 --warmup 1 --no-cpu-baseline --no-traffic"
REHEARSE_TIMEOUT=900 timeout 1000 bash scripts/rehearse_n8.sh torchrun < /dev/null 2>&1 | tail -8 | cut -c1-1700
{
  echo "== scripts/ab_h2s_small.py"; timeout 300 python scripts/ab_h2s_small.py < /dev/null 2>&1 | tail -12
  echo "== scripts/debug/serving_one.py 400 {1,8,64}"; for u in 1 8 64; do timeout 120 python scripts/debug/serving_one.py 400 $u < /dev/null 2>&1 | tail -1; done
  echo "== scripts/debug/multi_timeline.py run {1,2}"; for c in 1 2; do timeout 300 python scripts/debug/multi_timeline.py run $c < /dev/null 2>&1 | tail -1; done
  echo "== scripts/debug/point256_wall.py"; timeout 120 python scripts/debug/point256_wall.py < /dev/null 2>&1 | tail -2
  echo "== scripts/debug/cfg0_train_trace.py"; timeout 120 python scripts/debug/cfg0_train_trace.py < /dev/null 2>&1 | grep -E "fit|asarray"
  echo "== scripts/debug/map_default_trace.py 512 / 2048"; for k in 512 2048; do timeout 120 python scripts/debug/map_default_trace.py $k < /dev/null 2>&1 | grep "^MAP"; done
  echo "== scripts/debug/em_small_time.py"; timeout 120 python scripts/debug/em_small_time.py < /dev/null 2>&1 | tail -4
  echo "== scripts/debug/em_f64_check.py"; timeout 300 python scripts/debug/em_f64_check.py < /dev/null 2>&1 | grep -E "^MAP K|^EM n|findings"
  echo "== scripts/time_wide_rows.py"; timeout 300 python scripts/time_wide_rows.py < /dev/null 2>&1 | tail -6
} > $O/${R}_session_notes_raw.txt 2>&1
cut -c1-400 $O/${R}_session_notes_raw.txt
