#!/bin/bash
# timing-only A/B of conv_wino variants: tools/bench_conv_wino.py per library
V=robust-pose-estimator_amd/csrc/build/variants
for rep in 1 2; do for lib in main "$@"; do
  if [ "$lib" = main ]; then unset RPE_HIP_LIBRARY; else export RPE_HIP_LIBRARY=$PWD/$V/librpe_$lib.so; fi
  echo "=== $lib"; python tools/bench_conv_wino.py 2>&1 | grep -E "^convc2|^conv |^fh1|^layer"
done; done
