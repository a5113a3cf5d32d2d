#!/bin/bash
# scrambled buffers (second record array, arena) against plain hipMalloc: configs[1] and the 1000 Mbp workload
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in off on off on; do
  if [ $v = off ]; then export KQ_SCRAMBLE_MB=0 KQ_SCRAMBLE_ARENA=0; else unset KQ_SCRAMBLE_MB KQ_SCRAMBLE_ARENA; fi
  timeout -k 10 200 python3 bench.py --workload cfg1 --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('cfg1 scramble $v: %.3f ms  %.1f G/s' % (d['ms_per_step'], d['value']/1e9), d['roofline']['stage_ms'])"
  timeout -k 10 300 python3 bench.py --genome-mbp 1000 --steps 20 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('1000 Mbp scramble $v: %.2f ms/step  %.1f G/s' % (d['ms_per_step'], d['value']/1e9))"
done
