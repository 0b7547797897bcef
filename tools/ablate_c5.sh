#!/bin/bash
# Ablation timing of the C5 cube kernels (variants built with tools/build_variant.sh cabN "-DCUBE_ABLATE=N"); run on the GPU box.
mkdir -p gpurun_out/r3
export TMPDIR=/tmp
ROOT=$(pwd)
cd /tmp
for v in "" $@; do
  if [ -n "$v" ]; then export RADTXFR_LIB=$ROOT/build/$v.so; else unset RADTXFR_LIB; fi
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/r3/c5ab_$v -o run -- python3 $ROOT/tools/time_c5.py --reps 6 > $ROOT/gpurun_out/r3/c5ab_$v.txt 2>&1 || exit 1
  echo "== variant [$v]"; grep -h "pixel_cube\|basis_moments" $ROOT/gpurun_out/r3/c5ab_$v/*kernel_stats.csv | cut -d, -f1-4
done
