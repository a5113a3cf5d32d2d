"""`kreeq validate -f asm -r hifi.fastq -k 31 -o vcf` at the size of tests/test_gpu_cli.py::test_configs4_shape_hifi_k31_vcf (60 kbp genome,
240 reads of 15 kbp) and at 10 x and 100 x that size: wall time of the CLI, its own verbose stamps, VCF records."""
import os, subprocess, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from kreeq_amd import build

def make(scale, d):
    rng = np.random.default_rng(31)
    G, L = 60000 * scale, 15000
    g = rng.integers(0, 4, G).astype(np.uint8)
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    fq = os.path.join(d, f"hifi{scale}.fastq"); fa = os.path.join(d, f"asm{scale}.fasta")
    with open(fq, "wb") as f:
        for i, s in enumerate(rng.integers(0, G - L, 240 * scale)):
            r = g[s:s + L].copy()
            p = np.nonzero(rng.random(L) < 0.001)[0]
            r[p] = (r[p] + 1 + rng.integers(0, 3, len(p))) % 4
            if rng.random() < 0.5: r = (3 - r)[::-1]
            f.write(b"@m%d\n" % i + lut[r].tobytes() + b"\n+\n" + b"~" * L + b"\n")
    a = g.copy()
    p = rng.choice(np.arange(500, G - 500), 25 * scale, replace=False)
    a[p] = (a[p] + 1 + rng.integers(0, 3, len(p))) % 4                  # substitutions only (the test has indels too)
    with open(fa, "wb") as f:
        for c in range(2 * scale):
            f.write(b">contig%d\n" % c + lut[a[c * 30000:(c + 1) * 30000]].tobytes() + b"\n")
    return fa, fq

d = "/tmp/vcfscale"; os.makedirs(d, exist_ok=True)
for scale in [int(x) for x in (sys.argv[1:] or ["1", "10", "100"])]:
    fa, fq = make(scale, d)
    for rep in range(2):
        t0 = time.perf_counter()
        p = subprocess.run([build.CLI, "validate", "-f", fa, "-r", fq, "-k", "31", "-o", "vcf", "--search-depth", "40", "--max-span", "16", "--verbose"], capture_output=True, text=True)
        dt = time.perf_counter() - t0
        assert p.returncode == 0, p.stderr[-2000:]
        n = sum(1 for l in p.stdout.split("\n") if l.startswith("contig"))
        print(f"scale {scale}: wall {dt:.2f} s, {n} VCF records;", " | ".join(l.strip() for l in p.stderr.split("\n") if "s]" in l or "Candidate" in l)[:2500], flush=True)
