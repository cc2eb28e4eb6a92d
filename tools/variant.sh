#!/bin/bash
# Build a VARIANT of libdvsof_hip.so for an experiment: the listed sources recompiled with extra
# flags (-D switches), every other object taken from the product build.  The variant travels to the
# GPU box with the snapshot (*.so is git-ignored, not gpurun-ignored); tools load it with
# DVSOF_LIB_PATH=<path>.   usage: tools/variant.sh NAME "-DFOO=1 ..." loss.hip [more.hip ...]
set -e
NAME=$1; FLAGS=$2; shift 2
cd "$(dirname "$0")/../dvs_of_training_framework_amd/csrc"
make -s -j8 > /dev/null
mkdir -p variants/$NAME
OBJS=""
for f in *.hip; do
  o=${f%.hip}.o
  if [[ " $* " == *" $f "* ]]; then
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function \
      -ffp-contract=off -fno-fast-math -mllvm -amdgpu-mfma-vgpr-form $FLAGS -c $f -o variants/$NAME/$o
    OBJS="$OBJS variants/$NAME/$o"
  else
    OBJS="$OBJS $o"
  fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o variants/$NAME/libdvsof_hip.so $OBJS -ldl
echo "dvs_of_training_framework_amd/csrc/variants/$NAME/libdvsof_hip.so"
