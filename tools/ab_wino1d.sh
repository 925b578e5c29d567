#!/bin/bash
# timing-only A/B of conv_wino1d variants on the GRU shapes: tools/ab_wino1d.sh VARIANT...   (CONV_N=2 for the sequential tracker's launches)
V=robust-pose-estimator_amd/csrc/build/variants
for rep in 1 2; do for lib in main "$@"; do
  if [ "$lib" = main ]; then unset RPE_HIP_LIBRARY; else export RPE_HIP_LIBRARY=$PWD/$V/librpe_$lib.so; fi
  echo "=== $lib"; CONV_ONLY=gru python tools/bench_conv_wino.py 2>&1 | grep -E "^zr|^q "
done; done
