#!/bin/bash
# stage times of configs[1] (clear + one batch) for library variants: tools/bench_extra/ab_cfg1.sh <variant>...
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export KQ_BENCH_NOCHECK=1
for v in "" "$@"; do
  if [ -n "$v" ]; then export KQ_LIB=$GRAFT_REPO_ROOT/kreeq_amd/lib/variants/$v.so; else unset KQ_LIB; fi
  timeout -k 10 200 python3 bench.py --workload cfg1 --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('${v:-default}: %.3f ms  %.1f G/s' % (d['ms_per_step'], d['value']/1e9), d['roofline']['stage_ms'])"
done
