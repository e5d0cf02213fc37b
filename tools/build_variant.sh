#!/bin/bash
# tools/build_variant.sh NAME "-DMI_SPMM_..." [FILE]: liblaplace_hip_NAME.so with csrc/FILE.hip (default spmm) recompiled
# under the given macros (A/B runs).
set -e
cd "$(dirname "$0")/.."
P=laplace-gnn-recommendation_amd
python -c "import sys; sys.path.insert(0,'.'); import importlib; b=importlib.import_module('laplace_amd.build'); b.build_hip()"
OBJ=$(python -c "import sys; sys.path.insert(0,'.'); import laplace_amd.build as b; print(b.OBJ_DIR)")
F=${3:-spmm}
EXTRA=$(python -c "import sys; sys.path.insert(0,'.'); import laplace_amd.build as b; print(' '.join(b.EXTRA_FLAGS.get('$F.hip', [])))")
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DNDEBUG $EXTRA $2 -c $P/csrc/$F.hip -o /tmp/${F}_$1.o
OBJS=$(ls $OBJ/*.o | grep -v "/$F.o")
hipcc --offload-arch=gfx950 -shared -fPIC -o $P/liblaplace_hip_$1.so $OBJS /tmp/${F}_$1.o
echo built $P/liblaplace_hip_$1.so
