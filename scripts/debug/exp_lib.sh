#!/bin/bash
# Measurement builds: one library per (source, macro) under build_dbg/, linked from the regular objects.
#   exp_lib.sh gmm_score_split.hip SPLIT_EXP 1 2      -> build_dbg/pygmm_SPLIT_EXP1.so, ..._SPLIT_EXP2.so
# Run on the GPU box with SR_PYGMM_LIB=build_dbg/pygmm_<MACRO><VALUE>.so in front of any script.
# SPLIT_EXP (gmm_score_split.hip): 1 = leave after the frame prologue.  H2S_EXP (gmm_score_h2_shared.hip): 1 = the same.
# SPLITP_OFF (gmm_score_splitp.hip): bit mask of parts compiled out (see the source).
cd "$(dirname "$0")/../../speaker-recognition_amd/csrc" || exit 1
src=$1; macro=$2; shift 2
mkdir -p ../../build_dbg
obj=$(basename "$src" .hip).o
OBJS=$(ls ../build/*.o | grep -v "/$obj\$")
for m in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-slp-vectorize -D$macro=$m $EXP_EXTRA -c "$src" -o ../../build_dbg/${macro}$m.o &
done
wait
for m in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build_dbg/pygmm_${macro}$m.so $OBJS ../../build_dbg/${macro}$m.o
done
ls -la ../../build_dbg/*.so
