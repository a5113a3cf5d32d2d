"""Host-side FASTA/FASTQ(.gz) ingest (kreeq_amd/host/fastx.cpp): the multi-threaded chunked parser
must yield exactly the sequences of the sequential reader.  No GPU needed."""
import gzip
import subprocess

import numpy as np
import pytest

from kreeq_amd import build
from tests import helpers as H


@pytest.fixture(scope="module")
def cli():
    build.build_lib()
    return build.build_cli()


def seqsum(cli, path, threads=0, batch=1 << 16):
    out = subprocess.run([cli, "dbtool", "seqsum", path, str(threads), str(batch)], capture_output=True, text=True, check=True).stdout
    return tuple(int(x) for x in out.split())


def py_digest(seqs):
    n = bases = dig = 0
    for s in seqs:
        h = 1469598103934665603
        for c in s:
            h = ((h ^ c) * 1099511628211) & (2**64 - 1)
        n += 1
        bases += len(s)
        dig = (dig + h) & (2**64 - 1)
    return n, bases, dig


def make_fastq(path, n, seed, nasty_quals=True):
    rng = np.random.default_rng(seed)
    seqs = []
    with open(path, "wb") as f:
        for i in range(n):
            ln = int(rng.integers(1, 400))
            s = bytes(rng.choice(np.frombuffer(b"ACGTNacgt", dtype=np.uint8), ln))
            q = bytes(rng.integers(33, 74, ln, dtype=np.uint8))
            if nasty_quals and i % 3 == 0:
                q = b"@" + q[1:]                      # quality lines may start with '@' (and with '+')
            if nasty_quals and i % 5 == 0:
                q = b"+" + q[1:]
            f.write(b"@r%d some comment\n" % i + s + b"\n+\n" + q + b"\n")
            seqs.append(s)
    return seqs


def test_fastq_parallel_matches_sequential(cli, tmp_path):
    p = str(tmp_path / "reads.fastq")
    seqs = make_fastq(p, 60000, seed=1)               # ~ 24 MB: several 8 MiB chunks
    want = py_digest(seqs)
    assert seqsum(cli, p, 0) == want
    for threads in (1, 3, 8):
        for batch in (1 << 12, 1 << 20):
            assert seqsum(cli, p, threads, batch) == want, (threads, batch)
    gz = p + ".gz"
    with open(p, "rb") as src, gzip.open(gz, "wb", compresslevel=1) as dst:
        dst.write(src.read())
    assert seqsum(cli, gz, 0) == want
    assert seqsum(cli, gz, 4) == want                 # one inflating producer thread behind the queue


def test_fasta_multiline_parallel(cli, tmp_path):
    rng = np.random.default_rng(2)
    p = str(tmp_path / "asm.fasta")
    seqs = []
    with open(p, "wb") as f:
        for i in range(300):
            ln = int(rng.integers(1, 200000))
            s = bytes(rng.choice(np.frombuffer(b"ACGTN", dtype=np.uint8), ln))
            f.write(b">s%d\n" % i)
            for o in range(0, ln, 70):
                f.write(s[o:o + 70] + b"\n")
            seqs.append(s)
    want = py_digest(seqs)
    assert seqsum(cli, p, 0) == want
    for threads in (1, 4, 8):
        assert seqsum(cli, p, threads, 1 << 18) == want, threads


def test_golden_inputs_all_readers(cli):
    for name in ("random1.fastq", "random1.fastq.gz", "random1.fasta", "random3.N.fastq", "to_correct.fastq"):
        want = py_digest([s for _, s in H.read_fastx(H.golden_input(name))])
        assert seqsum(cli, H.golden_input(name), 0) == want, name
        assert seqsum(cli, H.golden_input(name), 4) == want, name


def test_wrapped_fastq_is_refused(cli, tmp_path):
    """four-line FASTQ only (like the reference's loader): a wrapped record must fail loudly, not be mis-parsed"""
    p = str(tmp_path / "wrapped.fastq")
    with open(p, "wb") as f:
        for i in range(2000):
            f.write(b"@r%d\nACGTACGTAC\nGTACGTACGT\n+\nIIIIIIIIII\nIIIIIIIIII\n" % i)
    r = subprocess.run([cli, "dbtool", "seqsum", p, "4", "4096"], capture_output=True, text=True)
    assert r.returncode != 0 and "malformed FASTQ" in r.stderr


def test_consumer_failure_stops_the_parser_threads(cli, tmp_path):
    """when the batch consumer throws, the parser threads are cancelled and joined before the file mapping
    goes away (was: detached threads reading an unmapped file)"""
    import os

    p = str(tmp_path / "reads.fastq")
    make_fastq(p, 60000, seed=3, nasty_quals=False)
    for threads in (4, 8):
        r = subprocess.run([cli, "dbtool", "seqsum", p, str(threads), "65536"], capture_output=True, text=True,
                           env=dict(os.environ, KQ_TEST_FAIL_AFTER_BATCHES="3"), timeout=60)
        assert r.returncode == 1 and "consumer failed" in r.stderr, (r.returncode, r.stderr)
    gz = p + ".gz"
    with open(p, "rb") as src, gzip.open(gz, "wb", compresslevel=1) as dst:
        dst.write(src.read())
    r = subprocess.run([cli, "dbtool", "seqsum", gz, "4", "65536"], capture_output=True, text=True,
                       env=dict(os.environ, KQ_TEST_FAIL_AFTER_BATCHES="3"), timeout=60)
    assert r.returncode == 1 and "consumer failed" in r.stderr


def test_sink_reader_matches(cli, tmp_path):
    """read_batches_sink (parser threads write into buffers they submit themselves: the CLI's pinned-memory count path)
    yields exactly the sequences of the sequential reader -- FASTQ, multi-line FASTA, .gz, small and large buffers"""
    def seqsink(path, threads, cap):
        out = subprocess.run([cli, "dbtool", "seqsink", path, str(threads), str(cap)], capture_output=True, text=True, check=True).stdout
        return tuple(int(x) for x in out.split())

    fq = str(tmp_path / "reads.fastq")
    want = py_digest(make_fastq(fq, 40000, seed=11))
    for threads, cap in ((1, 1 << 12), (3, 1 << 14), (8, 1 << 20), (16, 1 << 23)):
        assert seqsink(fq, threads, cap) == want, (threads, cap)
    gz = fq + ".gz"
    with open(fq, "rb") as src, gzip.open(gz, "wb", compresslevel=1) as dst:
        dst.write(src.read())
    assert seqsink(gz, 4, 1 << 16) == want
    rng = np.random.default_rng(12)
    fa = str(tmp_path / "asm.fasta")
    seqs = []
    with open(fa, "wb") as f:
        for i in range(200):
            ln = int(rng.integers(1, 50000))
            s = bytes(rng.choice(np.frombuffer(b"ACGTN", dtype=np.uint8), ln))
            f.write(b">s%d\n" % i)
            for o in range(0, ln, 60):
                f.write(s[o:o + 60] + b"\n")
            seqs.append(s)
    assert seqsink(fa, 6, 1 << 17) == py_digest(seqs)
    # a sequence that does not fit a pool buffer travels whole in a one-off buffer ...
    assert seqsink(fa, 2, 1000) == py_digest(seqs)
    assert seqsink(fa, 5, 257) == py_digest(seqs)
    # ... and is refused, not truncated, by a sink that has none
    import os
    r = subprocess.run([cli, "dbtool", "seqsink", fa, "2", "1000"], capture_output=True, text=True, env=dict(os.environ, KQ_TEST_NO_BIG="1"))
    assert r.returncode != 0 and "does not fit" in r.stderr
