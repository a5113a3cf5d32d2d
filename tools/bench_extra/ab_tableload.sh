#!/bin/bash
# full-size bench + table-pass time for several table loads: tools/bench_extra/ab_tableload.sh 0.8 0.85 0.9
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in "$@"; do
  rm -rf /tmp/p_stats
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_stats -- python3 bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-extras --table-load $v 2>/tmp/err.txt | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('load $v: %.2f ms/step  %.1f G/s  passes %s  table %.1f GB  validate %.1f ms' % (d['ms_per_step'], d['value']/1e9, d['config']['table_passes_per_range'], d['config']['table_bytes']/1e9, d['validate']['ms']))" || tail -3 /tmp/err.txt
  python3 - <<PY
import csv, glob
f = glob.glob('/tmp/p_stats/**/*kernel_stats.csv', recursive=True)
if f:
    for r in csv.DictReader(open(f[0])):
        n = r['Name']
        if ('k_count_regions_q4' in n or 'k_lv_scatter_s' in n) and float(r['TotalDurationNs']) > 2e7: print('   %-40s %5s calls  %8.3f ms avg' % (n.split('(')[0][5:45], r['Calls'], float(r['AverageNs']) / 1e6))
PY
done
