#!/bin/bash
# rocprofv3 kernel statistics of scripts/ab_kmeans.py (the k-means|| start, exact and fast nearest-centre search).
# Every command fed from /dev/null and guarded: an empty file argument once made `head` wait on stdin for 15 GPU-minutes.
R="${GRAFT_REPO_ROOT:-/root/repo}"
O=$R/gpurun_out/kmprof; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o km -- python $R/scripts/ab_kmeans.py > $O/run.log 2>&1 < /dev/null
f=$(find $O -name "*kernel_stats*.csv" < /dev/null | head -1)
if [ -n "$f" ] && [ -f "$f" ]; then head -14 "$f" < /dev/null | cut -c1-160; else echo "no stats file"; tail -5 $O/run.log < /dev/null; fi
