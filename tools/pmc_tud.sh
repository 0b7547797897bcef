#!/bin/bash
# Counter passes over the TUD kernel on a thin column: bash tools/pmc_tud.sh TAG SCALE
TAG=${1:-pmct}; SCALE=${2:-1e-5}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp; cd /tmp
i=0
for ctrs in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" \
            "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32" \
            "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_IFETCH"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $ctrs --output-format csv -d "$OUT/p$i" -o run -- python3 "$ROOT/tools/time_c3.py" --reps 2 --mf-scale $SCALE > "$OUT/p$i.log" 2>&1 || echo "pass $i failed"
done
echo done
