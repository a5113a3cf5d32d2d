"""`kreeq validate -r reads.fastq -o db.kreeq` on the configs[1] FASTQ: wall time and the CLI's own verbose stamps."""
import subprocess, sys, os, time, shutil, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from kreeq_amd import synth, build
n, ln, k = 1_000_000, 150, 21
fq = "/tmp/reads.fastq"
if not os.path.exists(fq):
    genome = synth.genome_codes(5_000_000, seed=1)
    reads = synth.reads_batch(genome, n, ln, seed=2, err=0.005).reshape(-1)
    rec = np.empty((n, 3 + ln + 3 + ln + 1), dtype=np.uint8)
    rec[:, 0:3] = np.frombuffer(b"@r\n", dtype=np.uint8)
    rec[:, 3:3 + ln] = np.concatenate([reads, [10]]).reshape(n, ln + 1)[:, :ln]
    rec[:, 3 + ln:6 + ln] = np.frombuffer(b"\n+\n", dtype=np.uint8)
    rec[:, 6 + ln:6 + 2 * ln] = ord("I")
    rec[:, -1] = 10
    rec.tofile(fq)
for rep in range(3):
    shutil.rmtree("/tmp/db.kreeq", ignore_errors=True)
    t0 = time.perf_counter()
    p = subprocess.run([build.CLI, "validate", "-r", fq, "-o", "/tmp/db.kreeq", "--verbose"], capture_output=True, text=True)
    dt = time.perf_counter() - t0
    print(f"wall {dt:.2f} s;", " | ".join(l.strip() for l in p.stderr.split("\n") if "s]" in l))
