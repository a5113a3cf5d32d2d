#!/bin/bash
# full-size bench + kernel stats for several --slice-cap values (k-mer starts per slice): tools/bench_extra/ab_slicecap.sh 2147483648 2300000000
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for cap in "$@"; do
  rm -rf /tmp/p_stats
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_stats -- python3 bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-extras --slice-cap $cap 2>/tmp/err.txt | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('cap $cap: %.2f ms/step  %.1f G/s  passes %s  slice %s' % (d['ms_per_step'], d['value']/1e9, d['config']['table_passes_per_range'], d['config']['slice_kmers']))" || tail -3 /tmp/err.txt
  python3 - <<PY
import csv, glob
f = glob.glob('/tmp/p_stats/**/*kernel_stats.csv', recursive=True)
if f:
    for r in csv.DictReader(open(f[0])):
        n = r['Name']
        if ('k_lv_' in n or 'k_count_regions_q4' in n or 'k_p1_' in n) and float(r['TotalDurationNs']) > 2e7: print('   %-40s %5s calls  %8.3f ms avg %8.1f ms total' % (n.split('(')[0][5:45], r['Calls'], float(r['AverageNs']) / 1e6, float(r['TotalDurationNs']) / 1e6))
PY
done
