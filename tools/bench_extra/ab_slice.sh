#!/bin/bash
# human-scale bench for several KQ_OPT_SLICE_KMERS values: tools/bench_extra/ab_slice.sh <kmers>...
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in 0 "$@"; do
  rm -rf /tmp/p_stats
  timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_stats -- python3 bench.py --genome-mbp ${GENOME_MBP:-1000} --steps ${STEPS:-20} --warmup 2 --no-cpu-baseline --no-extras --slice-kmers $v 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('slice $v: %.2f ms/step  %.1f G/s  passes %d' % (d['ms_per_step'], d['value']/1e9, d['config']['table_passes']))"
  grep -E "^\"(void )?k_" $(find /tmp/p_stats -name "*kernel_stats.csv" | head -1) | sed -E "s/\(.*\)\"/\"/" | cut -d, -f1-4 | head -7
done
