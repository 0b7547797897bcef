#!/bin/bash
# The reference's DEFAULT compute_TUD call (radiative_transfer.py:152-183: all 66 layers, DVOUT 0.0005 -> 11 M wavenumbers on
# 500-6000 cm^-1) under rocprofv3: kernel trace + the `sq` and `trans` counter passes, each in its own run.
#   gpurun -- 'bash tools/profile_default66.sh r3'   then   python tools/profile_summarize66.py r3
set -o pipefail
TAG=${1:-r3}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/${TAG}_default66
mkdir -p "$OUT"
export TMPDIR=/tmp
CMD="$ROOT/tools/time_c3.py --layers 66 --n 11000000 --reps 5"
timeout -k 10 200 python3 $CMD > "$OUT/time.txt" 2>&1 || exit 1
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o run -- python3 $CMD > /dev/null 2> "$OUT/trace.err" || exit 2
for pass in "sq:SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVES" "trans:SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "busy:SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE"; do
  name=${pass%%:*}; ctrs=${pass#*:}
  timeout -k 10 300 rocprofv3 --pmc $ctrs --output-format csv -d "$OUT/pmc_$name" -o run -- python3 $ROOT/tools/time_c3.py --layers 66 --n 11000000 --reps 1 > /dev/null 2> "$OUT/pmc_$name.err" || exit 3
done
echo done
