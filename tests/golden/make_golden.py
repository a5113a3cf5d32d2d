#!/usr/bin/env python3
"""Regenerates tests/golden/ from the reference's own fixtures (run in the build container only;
/root/reference does not exist on the GPU box and nothing at test time reads it).

What is committed here is DATA, not reference source:
  inputs/            the reference's test inputs (FASTA/FASTQ/.gz/.gfa/.bed/.bkwig), byte for byte
  validateFiles/     the reference's golden stdout files (line 1 = command, line 2 = "embedded",
                     rest = expected stdout; harness: reference src/validate.cpp:52-122)
  kreeq_dbs.tar.gz   the 10 fixture databases testFiles/*.kreeq (phmap binary dumps), tarred
  db_tables/*.tsv    the same databases decoded to logical content by THIS script's independent
                     python reader (format: SURVEY.md §9.4), one line per k-mer:
                     map key fw0 fw1 fw2 fw3 bw0 bw1 bw2 bw3 cov hc  -- sorted by (key)
"""
import io
import os
import shutil
import struct
import sys
import tarfile

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def read_phmap(path, vbytes):
    """phmap::parallel_flat_hash_map binary dump -> [(key, value_bytes)]"""
    data = open(path, "rb").read()
    off = 0
    (nsub,) = struct.unpack_from("<Q", data, off)
    off += 8
    slot = (8 + vbytes + 7) // 8 * 8
    out = []
    for _ in range(nsub):
        ver, size, cap = struct.unpack_from("<QQQ", data, off)
        off += 24
        assert ver == 0xFFFFFFFFFFFFFFF5, hex(ver)
        if size == 0:
            continue
        ctrl = data[off:off + cap + 17]
        off += cap + 17
        slots = data[off:off + cap * slot]
        off += cap * slot
        off += 8  # growth_left
        n = 0
        for i in range(cap):
            if ctrl[i] < 0x80:
                (key,) = struct.unpack_from("<Q", slots, i * slot)
                out.append((key, slots[i * slot + 8:i * slot + 8 + vbytes]))
                n += 1
        assert n == size
    assert off == len(data), (path, off, len(data))
    return out


def decode_db(db):
    k, map_count = [int(x) for x in open(os.path.join(db, ".index")).read().split()]
    rows = []
    hc = {key: struct.unpack("<9I", v) for key, v in read_phmap(os.path.join(db, ".map.hc.bin"), 36)}
    for m in range(map_count):
        for key, v in read_phmap(os.path.join(db, f".map.{m}.bin"), 9):
            assert key % map_count == m
            if v[8] == 255:
                assert key in hc
                continue
            rows.append((key, m) + tuple(v) + (0,))
    for key, v in hc.items():
        rows.append((key, key % map_count) + tuple(v) + (1,))
    rows.sort()
    return k, map_count, rows


def main():
    if not os.path.isdir(REF):
        sys.exit("reference not mounted; golden files are already committed")
    inp = os.path.join(HERE, "inputs")
    os.makedirs(inp, exist_ok=True)
    for f in sorted(os.listdir(os.path.join(REF, "testFiles"))):
        p = os.path.join(REF, "testFiles", f)
        if os.path.isfile(p) and f.split(".")[-1] in ("fasta", "fastq", "gz", "bed", "bkwig", "gfa"):
            shutil.copyfile(p, os.path.join(inp, f))
    vf = os.path.join(HERE, "validateFiles")
    os.makedirs(vf, exist_ok=True)
    # validate (0-34), union (35), bkwig (48,49), vcf (50)
    for i in list(range(0, 36)) + [48, 49, 50]:
        shutil.copyfile(os.path.join(REF, "validateFiles", f"test.{i}.tst"), os.path.join(vf, f"test.{i}.tst"))
    tabs = os.path.join(HERE, "db_tables")
    os.makedirs(tabs, exist_ok=True)
    dbs = sorted(d for d in os.listdir(os.path.join(REF, "testFiles")) if d.endswith(".kreeq"))
    with tarfile.open(os.path.join(HERE, "kreeq_dbs.tar.gz"), "w:gz") as tar:
        for d in dbs:
            src = os.path.join(REF, "testFiles", d)
            for f in sorted(os.listdir(src)):
                ti = tarfile.TarInfo(f"{d}/{f}")
                data = open(os.path.join(src, f), "rb").read()
                ti.size = len(data)
                ti.mtime = 0
                ti.mode = 0o644
                tar.addfile(ti, io.BytesIO(data))
            k, mc, rows = decode_db(src)
            with open(os.path.join(tabs, d.replace(".kreeq", ".tsv")), "w") as o:
                o.write(f"#k={k} map_count={mc} columns: map key fw0 fw1 fw2 fw3 bw0 bw1 bw2 bw3 cov hc\n")
                for r in rows:
                    o.write(" ".join(str(x) for x in (r[1], r[0]) + r[2:]) + "\n")
            print(d, k, mc, len(rows))


if __name__ == "__main__":
    main()
