"""CLI ingest + count on the configs[1] FASTQ for several parser-thread counts, buffer sizes and pinned / pageable buffers
(KQ_INGEST_CAP_MB, KQ_INGEST_PIN): where do the ~70 ms go?"""
import subprocess, sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from kreeq_amd import synth, build
n, ln, k = 1_000_000, 150, 21
if not os.path.exists("/tmp/reads.fastq"):
    genome = synth.genome_codes(5_000_000, seed=1)
    reads = synth.reads_batch(genome, n, ln, seed=2, err=0.005).reshape(-1)
    rec = np.empty((n, 3 + ln + 3 + ln + 1), dtype=np.uint8)
    rec[:, 0:3] = np.frombuffer(b"@r\n", dtype=np.uint8)
    rec[:, 3:3 + ln] = np.concatenate([reads, [10]]).reshape(n, ln + 1)[:, :ln]
    rec[:, 3 + ln:6 + ln] = np.frombuffer(b"\n+\n", dtype=np.uint8)
    rec[:, 6 + ln:6 + 2 * ln] = ord("I")
    rec[:, -1] = 10
    rec.tofile("/tmp/reads.fastq")
if "--make-only" in sys.argv:
    sys.exit(0)
def stamp(lines, what):
    return float([l for l in lines if what in l][0].split("s]")[0].strip("[ "))
def run(j, cap, pin):
    env = dict(os.environ)
    if cap: env["KQ_INGEST_CAP_MB"] = str(cap)
    if pin is not None: env["KQ_INGEST_PIN"] = str(pin)
    best = None
    for rep in range(3):
        p = subprocess.run([build.CLI, "validate", "-r", "/tmp/reads.fastq", "--verbose"] + (["-j", str(j)] if j else []), capture_output=True, text=True, env=env)
        lines = p.stderr.split("\n")
        a, b, c = stamp(lines, "Loading input reads"), stamp(lines, "Reads loaded"), stamp(lines, "Summary computed")
        if best is None or c - a < best[0]: best = (c - a, b - a)
    return best
for j in (8, 16, 32, 64, 0):
    for cap in (0, 4, 16, 32):
        for pin in (0, 1):
            t, tl = run(j, cap, pin)
            print(f"-j {j or 'all':>3} cap {cap or 'auto':>4} MB pin {pin}: load {tl * 1e3:5.0f} ms, + summary {t * 1e3:5.0f} ms = {n * (ln - k + 1) / t / 1e9:.2f} G k-mers/s", flush=True)
