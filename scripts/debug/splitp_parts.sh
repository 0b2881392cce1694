#!/bin/bash
# Parts-off builds of gmm_score_splitp_kernel (csrc/gmm_score_splitp.hip, SPLITP_OFF): one library per mask under build_dbg/,
# linked from the regular objects.  Run on the GPU box:  for m in 0 1 2 ...; SR_PYGMM_LIB=build_dbg/pygmm_off$m.so python scripts/ab_split_shape.py cfg1 16 8
cd "$(dirname "$0")/../../speaker-recognition_amd/csrc" || exit 1
mkdir -p ../../build_dbg
OBJS=$(ls ../build/*.o | grep -v gmm_score_splitp.o)
for m in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-slp-vectorize -DSPLITP_OFF=$m -c gmm_score_splitp.hip -o ../../build_dbg/splitp_off$m.o &
done
wait
for m in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build_dbg/pygmm_off$m.so $OBJS ../../build_dbg/splitp_off$m.o
done
ls -la ../../build_dbg/*.so
