#!/bin/bash
# Runs on the GPU box (through gpurun): default bench, a kernel-trace of the bench, and the PMC
# passes for the dominant kernel (each counter group in a run of its own).  Outputs land under
# gpurun_out/; tools/summarize_profiles.py turns them into the files kept under profiles/.
set -e
export TMPDIR=/tmp
TAG=${1:-r03}
OUT=gpurun_out/prof_$TAG
rm -rf $OUT && mkdir -p $OUT
timeout -k 10 400 python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
tail -c 3000 $OUT/bench_default.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fc --no-f2x4 > $OUT/bench_traced.json 2> $OUT/trace.err
# the same with the weight gradients on the main stream: per-kernel durations without time-sharing
SIV_OVERLAP_WGRAD=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_serial -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fc --no-f2x4 > $OUT/bench_traced_serial.json 2> $OUT/trace_serial.err
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 tools/bench_conv.py --shapes v0_64_64 --ops fwd --reps 3 > $OUT/pmc_fetch.log 2>&1
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 tools/bench_conv.py --shapes v0_64_64 --ops fwd --reps 3 > $OUT/pmc_write.log 2>&1
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq -- python3 tools/bench_conv.py --shapes v0_64_64 --ops fwd --reps 3 > $OUT/pmc_sq.log 2>&1
find $OUT -name "*.csv" | head -30
