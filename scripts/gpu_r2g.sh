#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out; mkdir -p $O
echo "=== pytest pipeline+mfcc"; timeout 1800 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_mfcc.py -m gpu -x -q 2>&1 | tail -8 | tee $O/r02g_pytest.log
echo "=== bench"; timeout 1500 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/r02g_bench.json 2> $O/r02g_bench.err; tail -3 $O/r02g_bench.err; python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r02g_bench.json') if l.startswith('{')][-1])
print({k:d[k] for k in ('value','ms_per_step','kernel_ms_per_step','kernel_launches_per_step')})
print('unpipelined', d['unpipelined'])
c=d['configs']['configs[1]']
print({k:c[k] for k in c if k not in ('roofline','mfcc_roofline','workload')})
PY
