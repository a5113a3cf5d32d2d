#!/bin/bash
# full --verbose timeline of the CLI on the configs[1] FASTQ: tools/bench_extra/cli_ingest_timeline.sh [-j N]
cd $GRAFT_REPO_ROOT
[ -f /tmp/reads.fastq ] || python3 tools/bench_extra/cli_ingest_sweep.py --make-only
for rep in 1 2 3; do
  KQ_INGEST_TRACE=1 kreeq_amd/bin/kreeq validate -r /tmp/reads.fastq --verbose "$@" 2>&1 | grep -E "^\[|^ingest" | sed -E 's/thread-time sums: //' | tr '\n' ' '; echo
done
