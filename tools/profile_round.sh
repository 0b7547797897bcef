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
# kernel durations: traced with ONE pipeline, so that kernels of consecutive atmospheres do not overlap and rocprofv3's
# per-kernel averages are the kernels' own durations (what bench.py's roofline.ms_per_launch measures with HIP events);
# a second trace of the default command (two pipelines: kernels of two atmospheres share the chip) is kept beside it
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o run -- python3 "$ROOT/bench.py" $ARGS --no-cpu-baseline --pipelines 1 > "$OUT/bench_prof.json" 2> "$OUT/trace.err" || exit 2
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_pipelined" -o run -- python3 "$ROOT/bench.py" $ARGS --no-cpu-baseline > "$OUT/bench_prof_pipelined.json" 2> "$OUT/trace_pipelined.err" || exit 2
for pass in "fetch:FETCH_SIZE" "write:WRITE_SIZE" "sq:SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVES" "busy:SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE" "trans:SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "stall:SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  name=${pass%%:*}; ctrs=${pass#*:}
  timeout -k 10 300 rocprofv3 --pmc $ctrs --output-format csv -d "$OUT/pmc_$name" -o run -- python3 "$ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --pipelines 1 > /dev/null 2> "$OUT/pmc_$name.err" || exit 3
done
cd "$ROOT"
if [ "$2" != "bench-only" ]; then
# thin-column TUD timing (tools/time_c3.py --mf-scale 1e-3) and the per-step host overhead
timeout -k 10 200 python3 tools/time_c3.py --reps 5 --mf-scale 1e-3 > "$OUT/time_c3_thin.txt" 2>&1 || exit 4
timeout -k 10 200 python3 tools/time_overhead.py > "$OUT/time_overhead.txt" 2>&1 || exit 5
# rehearsal of the N = 2 path (two ranks sharing this one GPU over gloo): a record that the code path runs, never a measurement
timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 3 --warmup 1 --backend gloo --no-cpu-baseline > "$OUT/rehearsal_gloo2.json" 2> "$OUT/rehearsal_gloo2.err" || exit 7
# round 3: balance across ranks, pipelines, clustered table (timing + per-tile spread from the stamp build), drop-in, cross sections
timeout -k 10 400 python3 tools/shard_balance.py > "$OUT/shard_balance.txt" 2>&1 || exit 8
timeout -k 10 200 python3 tools/time_pipeline.py > "$OUT/time_pipeline.txt" 2>&1 || exit 9
for t in uniform clustered; do
  timeout -k 10 200 python3 tools/time_c3.py --table $t --reps 7 > "$OUT/time_c3_$t.txt" 2>&1 || exit 10
  if [ -f build/stamp.so ]; then RADTXFR_LIB=build/stamp.so timeout -k 10 200 python3 tools/tile_spread.py --table $t > "$OUT/tile_spread_$t.txt" 2>&1 || exit 11; fi
done
timeout -k 10 300 python3 tools/time_dropin.py > "$OUT/time_dropin.txt" 2>&1 || exit 12
timeout -k 10 300 python3 tools/time_xs.py > "$OUT/time_xs.txt" 2>&1 || exit 13
if [ -f build/sdslow.so ]; then RADTXFR_LIB=build/sdslow.so timeout -k 10 300 python3 tools/time_xs.py --states 2 > "$OUT/time_xs_round2_kernel.txt" 2>&1; fi
if [ -x tools/ubench_crosslayer ]; then timeout -k 10 100 tools/ubench_crosslayer > "$OUT/ubench_crosslayer.txt" 2>&1; fi
timeout -k 10 200 python3 tools/time_c4.py > "$OUT/c4_timing.txt" 2>&1
timeout -k 10 200 python3 tools/time_c5.py > "$OUT/c5_timing.txt" 2>&1
fi
echo done
