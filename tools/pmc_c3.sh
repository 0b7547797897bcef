#!/bin/bash
# Counter passes over tools/time_c3.py (the C3 line-sum + TUD in isolation): bash tools/pmc_c3.sh TAG ; then
# python tools/pmc_sum.py gpurun_out/TAG
TAG=${1:-pmc}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp; cd /tmp
i=0
for ctrs in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" \
            "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU" \
            "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD" \
            "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_SMEM GRBM_GUI_ACTIVE SQ_WAVES"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $ctrs --output-format csv -d "$OUT/p$i" -o run -- python3 "$ROOT/tools/time_c3.py" --reps 2 > "$OUT/p$i.log" 2>&1 || echo "pass $i failed"
done
echo done
