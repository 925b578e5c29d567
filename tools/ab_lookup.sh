#!/bin/bash
# timing-only A/B of k_corr_lookup variants (tools/build_variant.sh lk_NAME -DLK_...): tools/ab_lookup.sh NAME...
V=robust-pose-estimator_amd/csrc/build/variants
for rep in 1 2; do for lib in main "$@"; do
  if [ "$lib" = main ]; then unset RPE_HIP_LIBRARY; else export RPE_HIP_LIBRARY=$PWD/$V/librpe_$lib.so; fi
  echo "$lib $(python tools/bench_kernels.py --only lookup --reps 60 2>&1 | grep -i lookup | head -1)"
done; done
