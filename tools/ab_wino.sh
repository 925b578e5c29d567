#!/bin/bash
# A/B of conv_wino variants on the GPU box: correctness (the Winograd parity tests) + timing (tools/bench_conv_wino.py) per library.
V=robust-pose-estimator_amd/csrc/build/variants
for lib in main "$@"; do
  if [ "$lib" = main ]; then unset RPE_HIP_LIBRARY; else export RPE_HIP_LIBRARY=$PWD/$V/librpe_$lib.so; fi
  echo "=== $lib"
  python -m pytest tests/test_gpu_conv.py -q -x -k "winograd_3x3 or winograd_encoder or winograd_small or winograd_random or raw_residual" 2>&1 | tail -2
  python tools/bench_conv_wino.py 2>&1 | grep -v "^$" | grep -v amdgpu.ids
done
