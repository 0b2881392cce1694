#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out; mkdir -p $O
echo "=== pytest new"; timeout 1800 python -m pytest tests -m gpu -x -q -k "reference_binding or cfg0_at_stated or cfg3_shape or legacy" 2>&1 | tail -15 | tee $O/r02f_pytest.log
