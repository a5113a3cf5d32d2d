#!/bin/bash
# full-size bench + kernel stats with an environment variable set to each of the given values in turn (0 = unset):
#   VAR=KQ_SCRAMBLE_MB tools/bench_extra/ab_env.sh 0 x 0 x        ("x" = leave the variable unset)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in "$@"; do
  if [ "$v" != x ]; then export $VAR=$v; else unset $VAR; fi
  rm -rf /tmp/p_stats
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_stats -- python3 bench.py --genome-mbp ${GENOME_MBP:-3000} --steps 20 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$VAR=$v: %.2f ms/step  %.1f G/s  validate %.1f ms' % (d['ms_per_step'], d['value']/1e9, d['validate']['ms']))"
  python3 - <<PY
import csv, glob
f = glob.glob('/tmp/p_stats/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    n = r['Name']
    if 'k_lv_' in n or 'k_count_regions_q4' in n or 'k_p1_' in n or 'k_lookup' in n or 'k_summary' in n:
        if float(r['TotalDurationNs']) > 2e7: print('   %-40s %5s calls  %8.3f ms avg' % (n.split('(')[0][5:45], r['Calls'], float(r['AverageNs']) / 1e6))
PY
done
