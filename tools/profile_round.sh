#!/bin/bash
# One gpurun call that produces everything profiles/ holds for a round:
#   gpurun -- 'bash tools/profile_round.sh r1'
# then, back in the container:  python tools/profile_summarize.py r1
# Counters are collected in their own passes (one --pmc set per run, never together with a trace).
set -o pipefail
TAG=${1:-r1}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
ARGS="--steps 10 --warmup 3"
timeout -k 10 300 python3 bench.py $ARGS > "$OUT/bench.json" 2> "$OUT/bench.err" || exit 1
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o run -- python3 "$ROOT/bench.py" $ARGS --no-cpu-baseline > "$OUT/bench_prof.json" 2> "$OUT/trace.err" || exit 2
for pass in "fetch:FETCH_SIZE" "write:WRITE_SIZE" "sq:SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVES" "busy:SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE" "trans:SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  name=${pass%%:*}; ctrs=${pass#*:}
  timeout -k 10 300 rocprofv3 --pmc $ctrs --output-format csv -d "$OUT/pmc_$name" -o run -- python3 "$ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> "$OUT/pmc_$name.err" || exit 3
done
cd "$ROOT"
# thin-column TUD timing (tools/time_c3.py --mf-scale 1e-3) and the per-step host overhead
timeout -k 10 200 python3 tools/time_c3.py --reps 5 --mf-scale 1e-3 > "$OUT/time_c3_thin.txt" 2>&1 || exit 4
timeout -k 10 200 python3 tools/time_overhead.py > "$OUT/time_overhead.txt" 2>&1 || exit 5
timeout -k 10 300 python3 tools/time_fused_bound.py > "$OUT/time_fused_bound.txt" 2>&1 || exit 6
# rehearsal of the N = 2 path (two ranks sharing this one GPU over gloo): a record that the code path runs, never a measurement
timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 3 --warmup 1 --backend gloo --no-cpu-baseline > "$OUT/rehearsal_gloo2.json" 2> "$OUT/rehearsal_gloo2.err" || exit 7
echo done
