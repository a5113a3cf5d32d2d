#!/bin/bash
# evidence set of a round, in three gpurun calls on one box each (a call lasts at most 20 minutes):
#   tools/collect_profiles.sh <tag> bench   bench line of the default workload (BASELINE configs[2] at full size; + cpu baseline, configs1 / lookup /
#                                           union / third-scale objects) and the kernel stats of the same command
#   tools/collect_profiles.sh <tag> fetch   PMC pass FETCH_SIZE of the same workload (--warmup 0: the dispatches are exactly the job's)
#   tools/collect_profiles.sh <tag> write   PMC pass WRITE_SIZE; then fold both with tools/make_traffic_json.py (here, after the merge)
set -e
TAG=$1
WHAT=${2:-bench}
MBP=${3:-3000}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG
mkdir -p $O
if [ "$WHAT" = bench ]; then
  timeout -k 10 700 python3 bench.py --genome-mbp $MBP --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
  echo "bench done"
  rm -rf /tmp/p_stats
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_stats -- python3 bench.py --genome-mbp $MBP --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/stats_bench.json 2> $O/stats.err
  # the library's kernels only (the torch kernels that generate the reads have kilobyte-long names)
  S=$(find /tmp/p_stats -name "*kernel_stats.csv" | head -1)
  head -1 $S > $O/kernel_stats.csv
  grep -E "^\"(void )?k_|rocclr" $S >> $O/kernel_stats.csv
  echo "stats done"
  timeout -k 10 300 python3 bench.py --workload cfg1 --steps 20 --warmup 3 --no-cpu-baseline > $O/cfg1_bench.json 2> $O/cfg1_bench.err
  rm -rf /tmp/p_stats1
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_stats1 -- python3 bench.py --workload cfg1 --steps 10 --warmup 2 --no-cpu-baseline > /dev/null 2> $O/cfg1_stats.err
  cp $(find /tmp/p_stats1 -name "*kernel_stats.csv" | head -1) $O/cfg1_kernel_stats.csv
else
  C=FETCH_SIZE; [ "$WHAT" = write ] && C=WRITE_SIZE
  rm -rf $O/pmc_$WHAT
  timeout -k 10 1000 rocprofv3 --pmc $C --output-format csv -d $O/pmc_$WHAT -- python3 bench.py --genome-mbp $MBP --steps 20 --warmup 0 --no-cpu-baseline --no-extras > $O/pmc_${WHAT}_bench.json 2> $O/pmc_$WHAT.err
  # the PMC csv files are large (one row per dispatch incl. torch kernels): keep only the library's kernels
  f=$(find $O/pmc_$WHAT -name "*counter_collection.csv" | head -1); head -1 $f > $O/pmc_$WHAT.csv; grep -E ",\"(void )?k_" $f >> $O/pmc_$WHAT.csv || true; rm -rf $O/pmc_$WHAT
  echo "$WHAT done"
fi
ls $O
