#!/bin/bash
# PMC passes over the upsampled input-gradient / forward kernels at 64 -> 64 @80x96x80 (one counter group per run).
set -e
export TMPDIR=/tmp
O=gpurun_out/pmc_ud
rm -rf $O && mkdir -p $O
P="python3 tools/bench_conv.py --shapes v0_64_64_up --ops dgrad fwd --reps 3"
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -- $P > $O/fetch.log 2>&1
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --output-format csv -d $O/ea -- $P > $O/ea.log 2>&1
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCP_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum --output-format csv -d $O/lat -- $P > $O/lat.log 2>&1
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_DRAM_sum --output-format csv -d $O/tcc -- $P > $O/tcc.log 2>&1
