#!/bin/bash
# evidence set of a round (one gpurun call on one box): tools/collect_profiles.sh <tag> [genome_mbp]
#   bench line of the default workload (+cpu baseline, configs1 / lookup / union objects), kernel stats of the same command,
#   PMC traffic passes, and the configs[1] line + stats for continuity with round 1
set -e
TAG=$1
MBP=${2:-1000}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG
mkdir -p $O
timeout -k 10 500 python3 bench.py --genome-mbp $MBP --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
echo "bench done"
rm -rf /tmp/p_stats
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_stats -- python3 bench.py --genome-mbp $MBP --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/stats_bench.json 2> $O/stats.err
# the library's kernels only (the torch kernels that generate the reads have kilobyte-long names)
S=$(find /tmp/p_stats -name "*kernel_stats.csv" | head -1)
head -1 $S > $O/kernel_stats.csv
grep -E "^\"(void )?k_|rocclr" $S >> $O/kernel_stats.csv
echo "stats done"
rm -rf $O/pmc_fetch $O/pmc_write
timeout -k 10 500 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 bench.py --genome-mbp $MBP --steps 20 --warmup 0 --no-cpu-baseline --no-extras > /dev/null 2> $O/pmc_fetch.err
echo "fetch done"
timeout -k 10 500 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 bench.py --genome-mbp $MBP --steps 20 --warmup 0 --no-cpu-baseline --no-extras > /dev/null 2> $O/pmc_write.err
echo "write done"
python3 tools/make_traffic_json.py $O/pmc_fetch $O/pmc_write r02/$TAG $MBP 20 > $O/traffic.txt
cp profiles/traffic.json $O/traffic.json
cp profiles/r02/${TAG}_pmc_traffic.json $O/ 2>/dev/null || true
# the PMC csv files are large (one row per dispatch incl. torch kernels): keep only the library's kernels
for d in pmc_fetch pmc_write; do f=$(find $O/$d -name "*counter_collection.csv" | head -1); head -1 $f > $O/$d.csv; grep -E ",\"(void )?k_" $f >> $O/$d.csv || true; rm -rf $O/$d; done
timeout -k 10 300 python3 bench.py --workload cfg1 --steps 20 --warmup 3 --no-cpu-baseline > $O/cfg1_bench.json 2> $O/cfg1_bench.err
rm -rf /tmp/p_stats1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_stats1 -- python3 bench.py --workload cfg1 --steps 10 --warmup 2 --no-cpu-baseline > /dev/null 2> $O/cfg1_stats.err
cp $(find /tmp/p_stats1 -name "*kernel_stats.csv" | head -1) $O/cfg1_kernel_stats.csv
ls $O
