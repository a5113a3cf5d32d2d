#!/bin/bash
# A/B of library variants on configs[1]: bench.py --workload cfg1 (empty table) + multibatch steady state
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in "" "$@"; do
  if [ -n "$v" ]; then export KQ_LIB=$GRAFT_REPO_ROOT/kreeq_amd/lib/variants/$v.so; else unset KQ_LIB; fi
  echo "== variant: ${v:-default}"
  timeout -k 10 200 python3 bench.py --workload cfg1 --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('cfg1 empty-table: %.3f ms  %.1f G/s' % (d['ms_per_step'], d['value']/1e9), d['roofline']['stage_ms'])"
  timeout -k 10 200 python3 tools/bench_extra/multibatch.py 2>/dev/null | tail -1
done
