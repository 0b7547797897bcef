mkdir -p gpurun_out/r3
export TMPDIR=/tmp
ROOT=$(pwd)
cd /tmp
for v in "" abl1 abl2 abl4; do
  if [ -n "$v" ]; then export RADTXFR_LIB=$ROOT/build/$v.so; else unset RADTXFR_LIB; fi
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/r3/p66_$v -o run -- python3 $ROOT/tools/time_c3.py --layers 66 --n 11000000 --reps 3 > $ROOT/gpurun_out/r3/p66_$v.txt 2>&1
  echo "== variant [$v]"; grep -h "voigt_nodal" $ROOT/gpurun_out/r3/p66_$v/*kernel_stats.csv | cut -d, -f1-4
done
