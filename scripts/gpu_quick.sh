#!/bin/bash
# quick loop: mfcc+pipeline tests and a short bench line
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 600 python -m pytest tests/test_gpu_mfcc.py tests/test_gpu_pipeline.py -m gpu -x -q 2>&1 | tail -5
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['kernel_ms_per_step'], d['roofline']['frac'])"
