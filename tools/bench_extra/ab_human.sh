#!/bin/bash
# human-scale bench (600 Mbp, 12 steps) + kernel stats for library variants: tools/bench_extra/ab_human.sh <variant>...
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in "" "$@"; do
  if [ -n "$v" ]; then export KQ_LIB=$GRAFT_REPO_ROOT/kreeq_amd/lib/variants/$v.so; else unset KQ_LIB; fi
  rm -rf /tmp/p_stats
  timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_stats -- python3 bench.py --genome-mbp ${GENOME_MBP:-600} --steps 12 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('${v:-default}: %.2f ms/step  %.1f G/s  passes %d' % (d['ms_per_step'], d['value']/1e9, d['config']['table_passes']))"
  grep -E "^\"(void )?k_" $(find /tmp/p_stats -name "*kernel_stats.csv" | head -1) | sed -E "s/\(.*\)\"/\"/" | cut -d, -f1-4 | head -8
done
