#!/bin/bash
# round 2, call E: whole GPU suite + new bench (1 rank, then 2 ranks stacked on device 0)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out; mkdir -p $O
echo "=== pytest gpu"; timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee $O/r02e_pytest.log
echo "=== bench"; (time timeout 1500 python bench.py --steps 5 --warmup 2) > $O/r02e_bench.json 2> $O/r02e_bench.err; tail -3 $O/r02e_bench.err; python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r02e_bench.json') if l.startswith('{')][-1])
print({k:d[k] for k in ('value','ms_per_step','n_gpus','kernel_ms_per_step','hbm_copy_ceiling_GBps','parity')})
print('roofline', {k:d['roofline'][k] for k in ('kernel','achieved','peak','frac','avg_launch_ms')})
print('mfcc', {k:d['mfcc_roofline'][k] for k in ('achieved','frac','avg_launch_ms')})
print('cpu', d.get('cpu_baseline'))
for k,v in d.get('configs',{}).items(): print(k, json.dumps(v)[:900])
PY
echo "=== bench 2 ranks on device 0"; timeout 900 python bench.py --gpus 2 --device-override 0 --steps 3 --warmup 1 --utts 3000 2> $O/r02e_bench2.err | tee $O/r02e_bench2.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('value','n_gpus','ms_per_step','rank_frames_per_s')})"; tail -2 $O/r02e_bench2.err
