#!/bin/bash
# full-size bench (BASELINE configs[2], 3 Gbp) for library variants, in the order given ("default" = the shipped library):
#   tools/bench_extra/ab_full.sh default p1v1 default
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in "$@"; do
  if [ "$v" != default ]; then export KQ_LIB=$GRAFT_REPO_ROOT/kreeq_amd/lib/variants/$v.so; else unset KQ_LIB; fi
  timeout -k 10 500 python3 bench.py --no-cpu-baseline --no-extras ${BENCH_ARGS} 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v: %.2f ms/step  %.1f G/s  passes %d' % (d['ms_per_step'], d['value']/1e9, d['config']['table_passes']), d['roofline'].get('stage_ms'))"
done
