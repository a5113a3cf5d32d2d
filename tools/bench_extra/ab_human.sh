#!/bin/bash
# human-scale bench + kernel stats for library variants ("default" = the shipped library): GENOME_MBP=3000 STEPS=10 tools/bench_extra/ab_human.sh default <variant>...
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in "$@"; do
  if [ "$v" != default ]; then export KQ_LIB=$GRAFT_REPO_ROOT/kreeq_amd/lib/variants/$v.so; else unset KQ_LIB; fi
  rm -rf /tmp/p_stats
  timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_stats -- python3 bench.py --genome-mbp ${GENOME_MBP:-600} --steps ${STEPS:-12} --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v: %.2f ms/step  %.1f G/s  passes %d' % (d['ms_per_step'], d['value']/1e9, d['config']['table_passes']))"
  python3 - <<PY
import csv, glob
f = glob.glob('/tmp/p_stats/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    n = r['Name']
    if n.startswith('void k_') or n.startswith('k_'):
        if float(r['TotalDurationNs']) > 2e7: print('   %-40s %5s calls  %8.3f ms avg  %8.1f ms total' % (n.split('(')[0][5 if n.startswith('void') else 0:45], r['Calls'], float(r['AverageNs']) / 1e6, float(r['TotalDurationNs']) / 1e6))
PY
done
