"""Pins the CPU oracle (oracle/kreeq_oracle.c) to every golden the reference holds for the hot
path: fixture databases, validate/union stdout, and the .bkwig per-base dumps."""
import os
import struct

import numpy as np
import pytest

from oracle import oracle as O
from tests import helpers as H

VALIDATE_TESTS = list(range(0, 35))

# which reads produced which fixture database (reference src/generate-tests.cpp:54-87 + names)
DB_READS = {"test1": "random1.fastq", "test2": "random2.fastq", "random5": "random5.fastq",
            "random6": "random6.fastq", "random7": "random7.fastq", "random8": "random8.fastq",
            "random9": "random9.fastq", "random10": "random10.fastq", "random11": "random11.fastq",
            "random12": "random12.fastq"}


def test_hash_known_answers():
    # A^21: fw = 0, rv = all T -> key 0, forward.  T^21: canonical is A^21, not forward.
    assert O.hash_kmer([0] * 21, 21) == (0, True)
    assert O.hash_kmer([3] * 21, 21) == (0, False)
    # palindrome ACGT...: isFw must be False (fw == rv)
    pal = [0, 1, 2, 3]
    key, fw = O.hash_kmer(pal, 4)
    assert not fw and key == sum(b << (2 * i) for i, b in enumerate(pal))
    # first base in the low bits
    assert O.hash_kmer([1, 0, 0, 0, 0], 5)[0] == 1


@pytest.mark.parametrize("name", sorted(DB_READS))
def test_fixture_db_reproduced(name):
    db = O.OracleDB(21, 128)
    db.count_batch(H.reads_batch([H.golden_input(DB_READS[name])]))
    assert H.entries_equal(db.export(), H.load_db_table(name))


@pytest.mark.parametrize("idx", VALIDATE_TESTS)
def test_validate_stdout(idx):
    argv, expected = H.parse_tst(os.path.join(H.GOLDEN, "validateFiles", f"test.{idx}.tst"))
    asm, reads = H.parse_validate_cmd(argv)
    db = O.OracleDB(21, 128)
    for r in reads:                                   # one loadKmers per file, src/input.cpp:95-96
        db.count_batch(H.reads_batch([H.golden_input(r)]))
    out = H.stats_block(db.summary())
    ctr = np.zeros(3, dtype=np.uint64)
    for _, seq in H.read_fastx(H.golden_input(asm)):
        c, _ = db.validate_sequence(seq)
        ctr += c
    out += H.qv_block(int(ctr[0]), int(ctr[1]), int(ctr[2]), 21, O.error_rate, O.qv)
    assert out == expected


def test_union_stdout():
    argv, expected = H.parse_tst(os.path.join(H.GOLDEN, "validateFiles", "test.35.tst"))
    a, b = O.OracleDB(), O.OracleDB()
    a.import_entries(H.load_db_table("test1"))
    b.import_entries(H.load_db_table("test2"))
    a.merge(b)
    assert H.stats_block(a.summary()) == expected
    # union == counting both read sets into one database
    c = O.OracleDB()
    c.count_batch(H.reads_batch([H.golden_input("random1.fastq")]))
    c.count_batch(H.reads_batch([H.golden_input("random2.fastq")]))
    assert H.entries_equal(a.export(), c.export())


def read_bkwig(path):
    """reference writer src/kreeq-output.cpp:305-399 (format SURVEY.md §9.5)"""
    d = open(path, "rb").read()
    k = d[0]
    (npaths,) = struct.unpack_from("<I", d, 1)
    off = 5
    paths = []
    for _ in range(npaths):
        (hl,) = struct.unpack_from("<H", d, off); off += 2
        hdr = d[off:off + hl].decode(); off += hl
        (nc,) = struct.unpack_from("<I", d, off); off += 4
        comps = []
        for _ in range(nc):
            pos, ln, step = struct.unpack_from("<QQB", d, off); off += 17
            comps.append((pos, ln))
        paths.append((hdr, comps))
    vals = []
    for hdr, comps in paths:
        for pos, ln in comps:
            a = np.frombuffer(d, dtype="<u4", count=3 * ln, offset=off).reshape(ln, 3)
            off += 12 * ln
            vals.append((hdr, pos, a))
    assert off == len(d)
    return k, vals


def per_base_triplets(pb):
    """(cov, isFw?fw:bw, isFw?bw:fw) as written by printTableCompressedBinary"""
    fw = np.where(pb["isFw"] != 0, pb["fw"], pb["bw"])
    bw = np.where(pb["isFw"] != 0, pb["bw"], pb["fw"])
    return np.stack([pb["cov"], fw, bw], axis=1)


@pytest.mark.parametrize("asm,reads,bkwig", [("repeat1.fasta", "repeat1.fastq", "decompressor2.bkwig"),
                                            ("decompressor1.fasta", "random1.fastq", "decompressor1.bkwig")])
def test_bkwig_per_base(asm, reads, bkwig):
    """decompressor2.bkwig is the only fixture exercising the u8 -> u32 overflow path (cov 480)."""
    k, vals = read_bkwig(H.golden_input(bkwig))
    db = O.OracleDB(k, 128)
    db.count_batch(H.reads_batch([H.golden_input(reads)]))
    seqs = dict(H.read_fastx(H.golden_input(asm)))
    seen = 0
    for hdr, pos, a in vals:
        seq = seqs[hdr]
        _, pb = db.validate_sequence(seq, per_base=True)
        got = per_base_triplets(pb[pos:pos + len(a)])
        assert np.array_equal(got, a), (hdr, pos)
        seen += 1
    assert seen == len(vals) > 0


def test_overflow_invariant():
    """a k-mer is in the 32-bit map iff total cov >= 255 and its u32 values are the
    exact sums, independent of insertion order; the 8-bit entry is then a cov==255 tombstone."""
    rng = np.random.default_rng(7)
    keys = np.concatenate([np.full(300, 5 * 128 + 3, dtype=np.uint64), np.full(254, 9 * 128 + 3, dtype=np.uint64),
                           np.full(255, 11 * 128 + 3, dtype=np.uint64), rng.integers(0, 50, 2000).astype(np.uint64)])
    edges = rng.integers(0, 256, len(keys)).astype(np.uint8)
    exp = {}
    for kk, e in zip(keys.tolist(), edges.tolist()):
        v = exp.setdefault(kk, [0] * 9)
        for w in range(8):
            v[w] += (e >> (7 - w)) & 1
        v[8] += 1
    for trial in range(3):
        p = rng.permutation(len(keys))
        db = O.OracleDB(21, 128)
        db.insert_records(keys[p], edges[p])
        out = db.export()
        assert len(out) == len(exp)
        for r in out:
            v = exp[int(r["key"])]
            assert list(r["fw"]) + list(r["bw"]) + [int(r["cov"])] == v
            assert bool(r["hc"]) == (v[8] >= 255)
        k8, v8 = db.export_raw8(3)
        tomb = {int(k) for k, v in zip(k8, v8) if v["cov"] == 255}
        assert tomb == {kk for kk, v in exp.items() if v[8] >= 255 and kk % 128 == 3}


def test_vcf_golden():
    """candidate-error search (oracle/variants.py, src/variants.cpp) against validateFiles/test.50.tst: 30 of its 31
    records line for line; the 31st is the documented deviation of helpers.VCF_GOLDEN_DEVIATION"""
    from oracle import variants as V

    argv, expected = H.parse_tst(os.path.join(H.GOLDEN, "validateFiles", "test.50.tst"))
    assert argv[-4:] == ["--search-depth", "50", "--max-span", "32"]
    db = O.OracleDB(21, 128)
    db.count_batch(H.reads_batch([H.golden_input("to_correct.fastq")]))
    g = V.Graph(db.export(), 21)
    recs = [(h, s.decode()) for h, s in H.read_fastx(H.golden_input("to_correct.fasta"))]
    got = V.correct_sequences(g, recs, 50, 32)
    assert got == H.vcf_expected(expected)
    assert sum(a != b for a, b in zip(got, expected)) == 1 and len(got) == len(expected) == 35
