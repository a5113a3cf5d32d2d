#!/bin/bash
# library variants on the workloads that use the OTHER region kernels (k = 31 counts, region-wise lookup, union): tools/bench_extra/ab_p3threads.sh default <variant>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export KQ_BENCH_NOCHECK=1
for v in "$@"; do
  if [ "$v" != default ]; then export KQ_LIB=$GRAFT_REPO_ROOT/kreeq_amd/lib/variants/$v.so; else unset KQ_LIB; fi
  for k in 31 27; do
    timeout -k 10 200 python3 bench.py --workload cfg1 -k $k --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v k=$k cfg1: %.3f ms  %.1f G/s' % (d['ms_per_step'], d['value']/1e9), d['roofline']['stage_ms'])"
  done
  timeout -k 10 300 python3 bench.py --genome-mbp 300 --steps 6 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v extras: lookup %.3f ms, per-base %.3f ms, union %.3f ms, cfg1 steady %.3f ms' % (d['lookup']['ms'], d['lookup']['per_base']['ms'], d['union']['ms'], d['configs1']['steady_state']['ms_per_batch']))"
done
