#!/bin/bash
# Build a kernel-experiment variant of librpe_hip.so:  tools/build_variant.sh NAME [extra hipcc flags, e.g. -DLK_RPP=2]
# -> robust-pose-estimator_amd/csrc/build/variants/librpe_NAME.so ; run with RPE_HIP_LIBRARY=<that path>.
set -e
name=$1; shift
here=$(cd "$(dirname "$0")/.." && pwd)
src=$here/robust-pose-estimator_amd/csrc
out=$src/build/variants/$name
mkdir -p $out
for f in se3 pose pose_backward geometry corr raft_ops conv conv_direct conv1x1 conv1x1_x3 conv_wino conv_wino_x3 conv_wino1d conv_wino1d_x3 stem unet preprocess; do
  if [ -f $src/build/$f.o ] && [ "$f" != "${VARIANT_SRC:-corr}" ]; then cp $src/build/$f.o $out/$f.o
  else /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wno-unused-function -Wno-pass-failed $( { [ "${f#conv_wino}" != "$f" ] || [ "${f#conv1x1}" != "$f" ]; } && echo -fno-slp-vectorize ) "$@" -c $src/$f.hip -o $out/$f.o; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $src/build/variants/librpe_$name.so $out/*.o
echo $src/build/variants/librpe_$name.so
