"""CLI ingest + count on the configs[1] FASTQ (310 MB, 1 M x 150 bp): time between the CLI's "Loading input reads" and "Reads loaded"
stamps (parse + PCIe copies + count kernels enqueued and the table pass started) for several parser-thread counts."""
import subprocess, sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from kreeq_amd import synth, build
n, ln, k = 1_000_000, 150, 21
genome = synth.genome_codes(5_000_000, seed=1)
reads = synth.reads_batch(genome, n, ln, seed=2, err=0.005).reshape(-1)
rec = np.empty((n, 3 + ln + 3 + ln + 1), dtype=np.uint8)
rec[:, 0:3] = np.frombuffer(b"@r\n", dtype=np.uint8)
rec[:, 3:3 + ln] = np.concatenate([reads, [10]]).reshape(n, ln + 1)[:, :ln]
rec[:, 3 + ln:6 + ln] = np.frombuffer(b"\n+\n", dtype=np.uint8)
rec[:, 6 + ln:6 + 2 * ln] = ord("I")
rec[:, -1] = 10
rec.tofile("/tmp/reads.fastq")
def stamp(lines, what):
    return float([l for l in lines if what in l][0].split("s]")[0].strip("[ "))
for j in (4, 8, 16, 32, 64, 0):
    best = None
    for rep in range(3):
        p = subprocess.run([build.CLI, "validate", "-r", "/tmp/reads.fastq", "--verbose"] + (["-j", str(j)] if j else []), capture_output=True, text=True)
        lines = p.stderr.split("\n")
        dt = stamp(lines, "Summary computed") - stamp(lines, "Loading input reads")
        best = dt if best is None else min(best, dt)
    print(f"-j {j or 'default'}: ingest + count + summary {best * 1e3:.0f} ms = {n * (ln - k + 1) / best / 1e9:.2f} G k-mers/s")
