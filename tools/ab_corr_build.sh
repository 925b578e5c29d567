#!/bin/bash
# timing-only A/B of k_corr_build variants (tools/build_variant.sh): tools/ab_corr_build.sh VARIANT...
V=robust-pose-estimator_amd/csrc/build/variants
for rep in 1 2 3; do for lib in main "$@"; do
  if [ "$lib" = main ]; then unset RPE_HIP_LIBRARY; else export RPE_HIP_LIBRARY=$PWD/$V/librpe_$lib.so; fi
  echo "=== $lib $(python tools/bench_kernels.py --only build --reps 40 2>&1 | grep corr_build)"
done; done
