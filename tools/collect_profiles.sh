#!/bin/bash
# final evidence set of a round: bench line (+cpu baseline), kernel stats, PMC traffic passes
set -e
TAG=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$TAG
timeout -k 10 400 python3 bench.py --steps 20 --warmup 3 > gpurun_out/$TAG/bench.json 2> gpurun_out/$TAG/bench.err
echo "bench done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_stats -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/$TAG/stats_bench.json 2> gpurun_out/$TAG/stats.err
cp $(find /tmp/p_stats -name "*kernel_stats.csv" | head -1) gpurun_out/$TAG/kernel_stats.csv
echo "stats done"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/$TAG/pmc_fetch -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> gpurun_out/$TAG/pmc_fetch.err
echo "fetch done"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/$TAG/pmc_write -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> gpurun_out/$TAG/pmc_write.err
echo "write done"
ls gpurun_out/$TAG
