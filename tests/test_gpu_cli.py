"""Replays the reference's golden-stdout tests (validateFiles/*.tst, harness semantics of reference
src/validate.cpp:52-122) against OUR `kreeq` CLI on the GPU, plus .kreeq / .bkwig file round trips."""
import os
import subprocess

import numpy as np
import pytest

from kreeq_amd import build
from tests import helpers as H

pytestmark = pytest.mark.gpu

VALIDATE_TESTS = list(range(0, 35))


@pytest.fixture(scope="module")
def cli():
    assert os.path.exists(build.LIB), "libkreeq_amd.so must be built in-tree"
    return build.build_cli()


def remap(argv, golden_dbs):
    out = []
    for a in argv[1:]:
        if a.startswith("testFiles/") and a.endswith(".kreeq"):
            out.append(os.path.join(golden_dbs, os.path.basename(a)))
        elif a.startswith("testFiles/"):
            out.append(H.golden_input(a))
        else:
            out.append(a)
    return out


def run(cli, args, cwd=None, env=None):
    p = subprocess.run([cli] + args, capture_output=True, text=True, cwd=cwd, timeout=300, env=env)
    assert p.returncode == 0, p.stderr
    return p.stdout.split("\n")


@pytest.mark.parametrize("idx", VALIDATE_TESTS + [35])
def test_tst_replay(cli, golden_dbs, idx):
    argv, expected = H.parse_tst(os.path.join(H.GOLDEN, "validateFiles", f"test.{idx}.tst"))
    got = run(cli, remap(argv, golden_dbs))
    while got and got[-1] == "":
        got.pop()
    assert got == expected


def test_db_write_then_validate(cli, tmp_path, golden_dbs):
    """validate -r reads -o db.kreeq ; validate -f asm -d db.kreeq == validate -f asm -r reads"""
    from tests.golden.make_golden import decode_db

    db = str(tmp_path / "r1.kreeq")
    out1 = run(cli, ["validate", "-r", H.golden_input("random1.fastq"), "-o", db])
    _, exp = H.parse_tst(os.path.join(H.GOLDEN, "validateFiles", "test.0.tst"))
    assert [l for l in out1 if l] == exp[:6]                      # stats only, no QV table
    assert decode_db(db) == decode_db(os.path.join(golden_dbs, "test1.kreeq"))
    out2 = run(cli, ["validate", "-f", H.golden_input("random1.fasta"), "-d", db])
    assert [l for l in out2 if l] == exp
    # union of our own databases, written to disk and validated from disk
    db2, dbu = str(tmp_path / "r2.kreeq"), str(tmp_path / "u.kreeq")
    run(cli, ["validate", "-r", H.golden_input("random2.fastq"), "-o", db2])
    outu = run(cli, ["union", "-d", db, db2, "-o", dbu])
    _, exp35 = H.parse_tst(os.path.join(H.GOLDEN, "validateFiles", "test.35.tst"))
    assert [l for l in outu if l] == exp35
    _, exp3 = H.parse_tst(os.path.join(H.GOLDEN, "validateFiles", "test.3.tst"))
    out3 = run(cli, ["validate", "-f", H.golden_input("random1.fasta"), "-d", dbu])
    assert [l for l in out3 if l] == exp3


@pytest.mark.parametrize("asm,reads,bkwig", [("repeat1.fasta", "repeat1.fastq", "decompressor2.bkwig"),
                                            ("decompressor1.fasta", "random1.fastq", "decompressor1.bkwig")])
def test_bkwig_bytes(cli, tmp_path, asm, reads, bkwig):
    """the per-base binary table is byte-identical to the reference's fixture"""
    out = str(tmp_path / "o.bkwig")
    run(cli, ["validate", "-f", H.golden_input(asm), "-r", H.golden_input(reads), "-o", out])
    assert open(out, "rb").read() == open(H.golden_input(bkwig), "rb").read()


def test_kwig_text(cli, tmp_path):
    """.kwig is the text twin; the reference's decompressor 'inflate' golden (test.49) has the same body"""
    out = str(tmp_path / "o.kwig")
    run(cli, ["validate", "-f", H.golden_input("repeat1.fasta"), "-r", H.golden_input("repeat1.fastq"), "-o", out])
    _, exp = H.parse_tst(os.path.join(H.GOLDEN, "validateFiles", "test.49.tst"))
    assert open(out).read().split("\n")[:len(exp)] == exp


def test_cli_errors(cli):
    p = subprocess.run([cli, "union", "-d", "x"], capture_output=True, text=True)
    assert p.returncode != 0
    p = subprocess.run([cli, "bogus", "-h"], capture_output=True, text=True)
    assert p.returncode != 0 and "does not exist" in p.stderr


def test_cli_config1_end_to_end(cli, tmp_path):
    """BASELINE configs[1] through the CLI: FASTQ file -> parallel ingest -> GPU count -> .kreeq on disk,
    then validate the error-free genome from that database.  Checks the closed-form totals."""
    import time

    import numpy as np

    from kreeq_amd import synth

    n, ln, k = 1_000_000, 150, 21
    genome = synth.genome_codes(5_000_000, seed=1)
    reads = synth.reads_batch(genome, n, ln, seed=2, err=0.005).reshape(-1)
    rec = np.empty((n, 3 + ln + 3 + ln + 1), dtype=np.uint8)
    rec[:, 0:3] = np.frombuffer(b"@r\n", dtype=np.uint8)
    rec[:, 3:3 + ln] = np.concatenate([reads, [10]]).reshape(n, ln + 1)[:, :ln]
    rec[:, 3 + ln:6 + ln] = np.frombuffer(b"\n+\n", dtype=np.uint8)
    rec[:, 6 + ln:6 + 2 * ln] = ord("I")
    rec[:, -1] = 10
    fq = str(tmp_path / "reads.fastq")
    rec.tofile(fq)
    fa = str(tmp_path / "genome.fasta")
    with open(fa, "wb") as f:
        f.write(b">chr1\n" + synth.codes_to_ascii(genome).tobytes() + b"\n")
    db = str(tmp_path / "reads.kreeq")
    t0 = time.perf_counter()
    out = run(cli, ["validate", "-r", fq, "-o", db])
    dt = time.perf_counter() - t0
    print(f"\nCLI end to end (310 MB FASTQ -> .kreeq): {dt:.2f} s = {n * (ln - k + 1) / dt / 1e6:.0f} M k-mers/s")
    assert out[:4] == ["DBG Summary statistics:", f"Total kmers: {n * (ln - k + 1)}", out[2], "Distinct kmers: 17733815"]
    out2 = run(cli, ["validate", "-f", fa, "-d", db])
    assert out2[:6] == out[:6]
    assert out2[6] == "Missing\tTotal\tQV\tError\tk\tMethod"
    missing, total = out2[7].split("\t")[:2]
    assert int(total) == 5_000_000 - k + 1 and int(missing) < 2000


@pytest.mark.parametrize("passes", [2, 5, 128])
def test_cli_passes_match_single_pass(cli, tmp_path, golden_dbs, passes):
    """--passes n (memory-bounded map-range passes) prints and writes exactly what one pass does"""
    from tests.golden.make_golden import decode_db

    _, exp = H.parse_tst(os.path.join(H.GOLDEN, "validateFiles", "test.3.tst"))
    reads = [H.golden_input("random1.fastq"), H.golden_input("random2.fastq")]
    out = run(cli, ["validate", "-f", H.golden_input("random1.fasta"), "-r"] + reads + ["--passes", str(passes)])
    assert [l for l in out if l] == exp
    db = str(tmp_path / "u.kreeq")
    run(cli, ["validate", "-r"] + reads + ["-o", db, "--passes", str(passes)])
    single = str(tmp_path / "s.kreeq")
    run(cli, ["validate", "-r"] + reads + ["-o", single])
    assert decode_db(db) == decode_db(single)
    bk = str(tmp_path / "o.bkwig")
    run(cli, ["validate", "-f", H.golden_input("repeat1.fasta"), "-r", H.golden_input("repeat1.fastq"), "-o", bk, "--passes", str(passes)])
    assert open(bk, "rb").read() == open(H.golden_input("decompressor2.bkwig"), "rb").read()


def test_bed_table(cli, tmp_path):
    """.bed per-base table: windows of the k latest cov / fw / bw values (no reference fixture: the
    text is checked against the oracle's per-base values, format restated from src/kreeq-output.cpp:138-241)"""
    import numpy as np

    from oracle import oracle as O
    from tests.test_oracle_golden import per_base_triplets

    out = str(tmp_path / "o.bed")
    run(cli, ["validate", "-f", H.golden_input("decompressor1.fasta"), "-r", H.golden_input("random1.fastq"), "-o", out])
    db = O.OracleDB(21, 128)
    db.count_batch(H.reads_batch([H.golden_input("random1.fastq")]))
    lines = open(out).read().split("\n")
    li = 0
    for hdr, seq in H.read_fastx(H.golden_input("decompressor1.fasta")):
        _, pb = db.validate_sequence(seq, per_base=True)
        trip = per_base_triplets(pb)
        i = 0
        while i < len(seq):
            if seq[i:i + 1].upper() not in (b"A", b"C", b"G", b"T"):
                i += 1
                continue
            j = i
            while j < len(seq) and seq[j:j + 1].upper() in (b"A", b"C", b"G", b"T"):
                j += 1
            hist = np.zeros((20, 3), dtype=np.uint32)
            for p in range(i, j):
                hist = np.vstack([hist, trip[p:p + 1]])
                cols = [":".join(str(x) for x in hist[-21:, c]) for c in range(3)]
                assert lines[li] == "\t".join([hdr, str(p)] + cols), (hdr, p)
                li += 1
            i = j
    assert lines[li:] == [""]


@pytest.mark.parametrize("ext", ["kreeq", "hist", "unknownext"])
def test_report_validates_for_every_extension(cli, tmp_path, ext):
    """DBG::report's first switch has no case for .kreeq / .hist (src/kreeq-output.cpp:62-72): with -f they validate like
    any other extension and, the output name containing a '.', print the QV table (src/kreeq.cpp:78)"""
    _, exp = H.parse_tst(os.path.join(H.GOLDEN, "validateFiles", "test.0.tst"))
    out = str(tmp_path / ("o." + ext))
    got = run(cli, ["validate", "-f", H.golden_input("random1.fasta"), "-r", H.golden_input("random1.fastq"), "-o", out])
    assert [l for l in got if l] == exp
    if ext == "kreeq":
        assert open(os.path.join(out, ".index")).read() == "21\n128\n"
    if ext == "hist":
        assert os.path.getsize(out) > 0


def test_memory_bound_picks_passes(cli, tmp_path):
    """-m <GB> bounds the HBM of the table: when it does not fit, the maps are counted in ranges automatically
    (the reference's computeMapRange loop, src/kreeq.cpp:59-74) and every output stays the same"""
    from tests.golden.make_golden import decode_db

    _, exp = H.parse_tst(os.path.join(H.GOLDEN, "validateFiles", "test.3.tst"))
    reads = [H.golden_input("random1.fastq"), H.golden_input("random2.fastq")]
    p = subprocess.run([cli, "validate", "-f", H.golden_input("random1.fasta"), "-r"] + reads + ["-m", "0.001", "--verbose"],
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr
    assert [l for l in p.stdout.split("\n") if l] == exp
    assert "map ranges" in p.stderr and "Pass 2/" in p.stderr
    db, single = str(tmp_path / "m.kreeq"), str(tmp_path / "s.kreeq")
    run(cli, ["validate", "-r"] + reads + ["-o", db, "-m", "0.001"])
    run(cli, ["validate", "-r"] + reads + ["-o", single])
    assert decode_db(db) == decode_db(single)


def test_union_of_three_goes_through_merge(cli, tmp_path, golden_dbs):
    """union of three databases: the largest is imported, the others are merged region by region on the device;
    the result equals counting all reads together"""
    from tests.golden.make_golden import decode_db

    dbs = []
    for i, name in enumerate(["random1.fastq", "random2.fastq", "random3.N.fastq"]):
        d = str(tmp_path / f"d{i}.kreeq")
        run(cli, ["validate", "-r", H.golden_input(name), "-o", d])
        dbs.append(d)
    u, whole = str(tmp_path / "u.kreeq"), str(tmp_path / "w.kreeq")
    out_u = run(cli, ["union", "-d"] + dbs + ["-o", u])
    out_w = run(cli, ["validate", "-r"] + [H.golden_input(n) for n in ("random1.fastq", "random2.fastq", "random3.N.fastq")] + ["-o", whole])
    assert [l for l in out_u if l] == [l for l in out_w if l]
    assert decode_db(u) == decode_db(whole)


def test_vcf_golden_replay(cli):
    """validateFiles/test.50.tst through the CLI (-o vcf --search-depth 50 --max-span 32): device pre-filter + host search
    with batched device lookups; 30 of 31 records line for line (the 31st: helpers.VCF_GOLDEN_DEVIATION)"""
    argv, expected = H.parse_tst(os.path.join(H.GOLDEN, "validateFiles", "test.50.tst"))
    got = run(cli, remap(argv, None))
    while got and got[-1] == "":
        got.pop()
    assert got == H.vcf_expected(expected)


def test_vcf_random_vs_oracle(cli, tmp_path):
    """a 30 kb genome with planted substitutions, insertions and deletions (some close together), error-free reads at 12x:
    the CLI's VCF (GPU table, device pre-filter, lockstep host searches) equals the Python restatement's on the oracle table"""
    import numpy as np

    from oracle import oracle as O
    from oracle import variants as V

    rng = np.random.default_rng(7)
    acgt = "ACGT"
    genome = "".join(acgt[i] for i in rng.integers(0, 4, 30000))
    reads = []
    for s in rng.integers(0, len(genome) - 150, 2400):
        r = genome[s:s + 150]
        if rng.random() < 0.5:
            r = r[::-1].translate(str.maketrans("ACGT", "TGCA"))
        reads.append(r)
    asm = list(genome)
    for p in sorted(rng.choice(np.arange(200, len(genome) - 200), 60, replace=False), reverse=True):
        kind = rng.integers(0, 3)
        if kind == 0:
            asm[p] = acgt[(acgt.index(asm[p]) + 1 + rng.integers(0, 3)) % 4].lower()
        elif kind == 1:
            del asm[p]
        else:
            asm.insert(p, acgt[rng.integers(0, 4)].lower())
    asm = "".join(asm)
    recs = [("chrA", asm[:14000]), ("chrB", asm[14000:20000] + "NNNN" + asm[20000:])]
    fa, fq = str(tmp_path / "asm.fasta"), str(tmp_path / "reads.fastq")
    with open(fa, "w") as f:
        for h, s in recs:
            f.write(f">{h}\n{s}\n")
    with open(fq, "w") as f:
        for i, r in enumerate(reads):
            f.write(f"@r{i}\n{r}\n+\n{'I' * len(r)}\n")
    got = run(cli, ["validate", "-f", fa, "-r", fq, "-o", "vcf", "--search-depth", "60", "--max-span", "20"])
    while got and got[-1] == "":
        got.pop()
    db = O.OracleDB(21, 128)
    db.count_batch("\n".join(reads).encode(), threads=4)
    want = V.correct_sequences(V.Graph(db.export(), 21), recs, 60, 20)
    assert len(want) > 4 + 30
    assert got == want
    out = str(tmp_path / "calls.vcf")                                  # a named file instead of stdout; stats are printed then ('.' in the name)
    stdout = run(cli, ["validate", "-f", fa, "-r", fq, "-o", out, "--search-depth", "60", "--max-span", "20"])
    assert open(out).read().split("\n")[:-1] == want
    assert stdout[0] == "DBG Summary statistics:"


@pytest.mark.parametrize("scale", [1, 10])
def test_configs4_shape_hifi_k31_vcf(cli, tmp_path, scale):
    """BASELINE configs[4] shape at small scale: 60x HiFi-length (15 kbp) reads with 0.1 % substitutions, k = 31,
    validate + candidate-error VCF -- stdout (summary + QV) and VCF equal the oracle's (C oracle for the table and the QV
    counters, Python restatement for the search).  Read errors are part of the graph (coverage cut-off 0), so the search
    runs at thousands of branching positions.  scale 10: ten times the genome, the reads and the contigs (33 000 searches in
    one batch of lockstep rounds); the Python restatement of the search then checks two of the twenty contigs, record by
    record, and the record count of the others must be in proportion."""
    import numpy as np

    from oracle import oracle as O
    from oracle import variants as V

    k = 31
    rng = np.random.default_rng(31)
    acgt = "ACGT"
    comp = str.maketrans("ACGT", "TGCA")
    genome = "".join(acgt[i] for i in rng.integers(0, 4, 60000 * scale))
    reads = []
    for s in rng.integers(0, len(genome) - 15000, 240 * scale):
        r = list(genome[s:s + 15000])
        for p in np.nonzero(rng.random(15000) < 0.001)[0]:
            r[p] = acgt[(acgt.index(r[p]) + 1 + rng.integers(0, 3)) % 4]
        r = "".join(r)
        reads.append(r[::-1].translate(comp) if rng.random() < 0.5 else r)
    asm = list(genome)
    for p in sorted(rng.choice(np.arange(500, len(genome) - 500), 25 * scale, replace=False), reverse=True):
        kind = rng.integers(0, 3)
        if kind == 0:
            asm[p] = acgt[(acgt.index(asm[p]) + 1 + rng.integers(0, 3)) % 4].lower()
        elif kind == 1:
            del asm[p]
        else:
            asm.insert(p, acgt[rng.integers(0, 4)].lower())
    asm = "".join(asm)
    cuts = [0] + [35000 + 30000 * i for i in range(2 * scale - 1)] + [len(asm)]
    recs = [(f"contig{i + 1}", asm[cuts[i]:cuts[i + 1]]) for i in range(2 * scale)]
    fa, fq = str(tmp_path / "asm.fasta"), str(tmp_path / "hifi.fastq")
    with open(fa, "w") as f:
        for h, s in recs:
            f.write(f">{h}\n{s}\n")
    with open(fq, "w") as f:
        for i, r in enumerate(reads):
            f.write(f"@m{i}\n{r}\n+\n{'~' * len(r)}\n")
    db = O.OracleDB(k, 128)
    db.count_batch("\n".join(reads).encode(), threads=8)
    # validate: summary + QV text
    out = run(cli, ["validate", "-f", fa, "-r", fq, "-k", str(k)])
    ctr = np.zeros(3, dtype=np.uint64)
    for _, s in recs:
        c, _ = db.validate_sequence(s.encode())
        ctr += c
    want = H.stats_block(db.summary()) + H.qv_block(int(ctr[0]), int(ctr[1]), int(ctr[2]), k, O.error_rate, O.qv)
    assert [l for l in out if l] == want
    # candidate errors
    got = run(cli, ["validate", "-f", fa, "-r", fq, "-k", str(k), "-o", "vcf", "--search-depth", "40", "--max-span", "16"])
    while got and got[-1] == "":
        got.pop()
    checked = recs if scale == 1 else [recs[3], recs[-1]]
    want_vcf = V.correct_sequences(V.Graph(db.export(), k), checked, 40, 16)
    assert len(want_vcf) > 4 + 20
    names = {h for h, _ in checked}
    assert got[:4] == want_vcf[:4]
    assert [l for l in got[4:] if l.split("\t")[0] in names] == want_vcf[4:]
    assert len(got) - 4 > (len(want_vcf) - 4) * len(recs) // len(checked) * 7 // 10      # (the other contigs have their records too)


@pytest.mark.parametrize("knobs", [{"KQ_INGEST_PACK": "1"}, {"KQ_INGEST_PACK": "1", "KQ_INGEST_BUFFERS": "2", "KQ_INGEST_CAP_MB": "1"},
                                   {"KQ_INGEST_BUFFERS": "2", "KQ_INGEST_CAP_MB": "1"}, {"KQ_CLI_PENDING_AUTO": "1"}])
def test_cli_ingest_modes_agree(cli, tmp_path, knobs):
    """the ingest pool in its corner configurations -- 2-bit packed submits, two small buffers shared by all parser threads,
    the automatic (doubling) arena -- prints what the default configuration prints"""
    import os

    rng = np.random.default_rng(77)
    genome = rng.integers(0, 4, 400_000)
    fq = str(tmp_path / "r.fastq")
    with open(fq, "wb") as f:
        for i in range(30_000):
            p0 = int(rng.integers(0, len(genome) - 150))
            seq = np.frombuffer(b"ACGT", dtype=np.uint8)[genome[p0:p0 + 150]].copy()
            if i % 97 == 0:
                seq[int(rng.integers(0, 150))] = ord("N")
            sb = seq.tobytes().lower() if i % 5 == 0 else seq.tobytes()
            f.write(b"@r%d\n" % i + sb + b"\n+\n" + b"I" * 150 + b"\n")
    fa = str(tmp_path / "g.fasta")
    with open(fa, "wb") as f:
        f.write(b">c\n" + np.frombuffer(b"ACGT", dtype=np.uint8)[genome].tobytes() + b"\n")
    base = run(cli, ["validate", "-f", fa, "-r", fq, "-j", "7"])
    env = dict(os.environ)
    env.update(knobs)
    got = run(cli, ["validate", "-f", fa, "-r", fq, "-j", "7"], env=env)
    assert got == base and base[0] == "DBG Summary statistics:"



def test_cli_ingest_failure_does_not_hang(cli, tmp_path):
    """a persistent submit failure (out of device memory at scale; injected here) with more parser threads than
    pool buffers used to leave the surviving threads waiting for buffers held by dead ones.  Now: non-zero exit, the
    first error on stderr, no hang."""
    import os

    rng = np.random.default_rng(5)
    fq = str(tmp_path / "r.fastq")
    with open(fq, "wb") as f:
        seq = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, (40_000, 150))]
        for i in range(len(seq)):
            f.write(b"@r\n" + seq[i].tobytes() + b"\n+\n" + b"I" * 150 + b"\n")
    env = dict(os.environ, KQ_INGEST_BUFFERS="2", KQ_INGEST_CAP_MB="1", KQ_TEST_FAIL_SUBMIT="2")
    p = subprocess.run([cli, "validate", "-r", fq, "-j", "8"], capture_output=True, text=True, timeout=60, env=env)
    assert p.returncode != 0
    assert "injected" in p.stderr


def test_cli_sequence_longer_than_a_pool_buffer(cli, tmp_path):
    """a read longer than a pool buffer (a chromosome-scale FASTA record given with -r) travels in a buffer of its
    own instead of being refused; counts equal the oracle's, also with the device side cutting it into slices"""
    import os

    from oracle import oracle as O

    rng = np.random.default_rng(9)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    long_seq = acgt[rng.integers(0, 4, 3_000_000)].tobytes()
    short = [acgt[rng.integers(0, 4, 200)].tobytes() for _ in range(50)]
    fa = str(tmp_path / "chr.fasta")
    with open(fa, "wb") as f:
        for i, s in enumerate(short[:25]):
            f.write(b">s%d\n" % i + s + b"\n")
        f.write(b">chr\n")
        for i in range(0, len(long_seq), 80):
            f.write(long_seq[i:i + 80] + b"\n")
        for i, s in enumerate(short[25:]):
            f.write(b">t%d\n" % i + s + b"\n")
    db = O.OracleDB(21, 128)
    db.count_batch(b"\n".join(short[:25] + [long_seq] + short[25:]), threads=8)
    want = H.stats_block(db.summary())
    env = dict(os.environ, KQ_INGEST_CAP_MB="1", KQ_INGEST_BUFFERS="3")
    got = run(cli, ["validate", "-r", fa, "-j", "4"], env=env)
    assert [l for l in got if l] == want


def test_db_is_streamed_by_map_range(cli, tmp_path, golden_dbs):
    """round-2 VERDICT: `validate -d` and `union` read databases map range by map range (a few maps' entries on the host at a
    time, KQ_DB_CHUNK_ENTRIES forces many chunks) and, under --passes / -m, also evaluate / merge them range by range
    (src/kreeq.cpp:59-74, src/graph-builder.cpp:341-347); every output equals the single-range one"""
    import os

    from tests.golden.make_golden import decode_db

    env = dict(os.environ, KQ_DB_CHUNK_ENTRIES="40")
    _, exp3 = H.parse_tst(os.path.join(H.GOLDEN, "validateFiles", "test.3.tst"))
    _, exp35 = H.parse_tst(os.path.join(H.GOLDEN, "validateFiles", "test.35.tst"))
    d1, d2 = os.path.join(golden_dbs, "test1.kreeq"), os.path.join(golden_dbs, "test2.kreeq")
    u1 = str(tmp_path / "u1.kreeq")
    assert [l for l in run(cli, ["union", "-d", d1, d2, "-o", u1], env=env) if l] == exp35
    for passes in ("3", "128"):
        u = str(tmp_path / f"u{passes}.kreeq")
        assert [l for l in run(cli, ["union", "-d", d1, d2, "-o", u, "--passes", passes], env=env) if l] == exp35
        assert decode_db(u) == decode_db(u1)
        out = run(cli, ["validate", "-f", H.golden_input("random1.fasta"), "-d", u, "--passes", passes], env=env)
        assert [l for l in out if l] == exp3
    um = str(tmp_path / "um.kreeq")
    p = subprocess.run([cli, "union", "-d", d1, d2, "-o", um, "-m", "0.000001", "--verbose"], capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode == 0 and "map ranges" in p.stderr and "Peak host memory" in p.stderr, p.stderr
    assert decode_db(um) == decode_db(u1)
    p = subprocess.run([cli, "validate", "-f", H.golden_input("random1.fasta"), "-d", u1, "-m", "0.000001", "--verbose"], capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode == 0 and "map ranges" in p.stderr and "Pass 2/" in p.stderr, p.stderr
    assert [l for l in p.stdout.split("\n") if l] == exp3
    # per-base output from a database evaluated in ranges: byte-identical to the reference's fixture
    rdb, bk = str(tmp_path / "rep.kreeq"), str(tmp_path / "o.bkwig")
    run(cli, ["validate", "-r", H.golden_input("repeat1.fastq"), "-o", rdb])
    run(cli, ["validate", "-f", H.golden_input("repeat1.fasta"), "-d", rdb, "-o", bk, "--passes", "5"], env=env)
    assert open(bk, "rb").read() == open(H.golden_input("decompressor2.bkwig"), "rb").read()
    # .hist in ranges == .hist in one
    h1, h4 = str(tmp_path / "a.hist"), str(tmp_path / "b.hist")
    run(cli, ["validate", "-d", u1, "-o", h1])
    run(cli, ["validate", "-d", u1, "-o", h4, "--passes", "4"], env=env)
    assert open(h1).read() == open(h4).read() and os.path.getsize(h1) > 0
    h5 = str(tmp_path / "c.hist")
    run(cli, ["validate", "-r", H.golden_input("random1.fastq"), H.golden_input("random2.fastq"), "-o", h5, "--passes", "7"])
    assert open(h5).read() == open(h1).read()


@pytest.mark.parametrize("passes", ["2", "5"])
def test_vcf_under_map_range_passes(cli, tmp_path, passes):
    """round-2 VERDICT: the candidate-error search when the table is resident one map range at a time (the reference's search
    loops over map ranges, src/variants.cpp:78-84): from reads (the ranges go through a temporary database) and from a
    database on disk, the VCF equals the single-range one (= test.50.tst with its one documented deviation)"""
    argv, expected = H.parse_tst(os.path.join(H.GOLDEN, "validateFiles", "test.50.tst"))
    args = remap(argv, None)
    got = run(cli, args + ["--passes", passes], cwd=str(tmp_path))
    while got and got[-1] == "":
        got.pop()
    assert got == H.vcf_expected(expected)
    assert not [f for f in os.listdir(tmp_path) if f.startswith(".kreeq_ranges_")]      # the temporary database is gone
    # the same from a database on disk
    asm, reads = H.parse_validate_cmd(argv)
    db = str(tmp_path / "reads.kreeq")
    run(cli, ["validate", "-r"] + [H.golden_input(r) for r in reads] + ["-o", db])
    rest = [a for a in args[1:] if a not in ("-r",) and not a.endswith((".fastq", ".fastq.gz", ".fq"))]
    got = run(cli, ["validate"] + rest + ["-d", db, "--passes", passes])
    while got and got[-1] == "":
        got.pop()
    assert got == H.vcf_expected(expected)
