#!/bin/bash
# where the CLI's ingest threads spend their time (KQ_INGEST_TRACE) on the configs[1] FASTQ of cli_ingest_sweep.py:
#   tools/bench_extra/cli_ingest_trace.sh   ("threads cap_MB buffers pack" per line below)
cd $GRAFT_REPO_ROOT
[ -f /tmp/reads.fastq ] || python3 tools/bench_extra/cli_ingest_sweep.py --make-only
while read -r j cap nb pack; do
  for rep in 1 2; do
    KQ_INGEST_TRACE=1 KQ_INGEST_CAP_MB=$cap KQ_INGEST_BUFFERS=$nb KQ_INGEST_PACK=$pack kreeq_amd/bin/kreeq validate -r /tmp/reads.fastq -j $j --verbose 2>&1 | grep -E "^ingest|Loading input|Reads loaded|Summary computed" | sed -E 's/thread-time sums: //' | tr '\n' ' '; echo
  done
done <<CFG
8 8 10 0
16 8 18 0
16 16 10 0
32 8 18 0
64 8 18 0
CFG
echo "-- automatic arena (doubling):"
for rep in 1 2; do KQ_CLI_PENDING_AUTO=1 KQ_INGEST_TRACE=1 KQ_INGEST_CAP_MB=8 KQ_INGEST_BUFFERS=18 kreeq_amd/bin/kreeq validate -r /tmp/reads.fastq -j 16 --verbose 2>&1 | grep -E "^ingest|Loading input|Reads loaded|Summary computed" | sed -E 's/thread-time sums: //' | tr '\n' ' '; echo; done
