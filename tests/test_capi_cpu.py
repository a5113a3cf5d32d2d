"""CPU-side checks of the product boundary: the C-ABI library builds for gfx950, loads, exports
every symbol include/kreeq_amd.h declares, and fails loudly (no CPU fallback) without a GPU."""
import os
import re

import pytest

from kreeq_amd import build, capi
from tests.helpers import ROOT


@pytest.fixture(scope="module")
def lib():
    build.build_lib()
    return capi.load()


def test_header_symbols_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "kreeq_amd.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(kq_[a-z_0-9]+)\s*\(", hdr))
    assert declared == set(capi.SYMBOLS)
    for s in declared:
        assert hasattr(lib, s), s


def test_abi_version(lib):
    assert lib.kq_abi_version() == 4       # 4: KQ_OPT_OVERLAP, quad-aligned slot homes and the Feistel table hash (opaque record formats changed)


def test_no_cpu_fallback(lib):
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    assert not capi.device_available()
    with pytest.raises(capi.KqError) as e:
        capi.KreeqDB(21, 128)
    assert e.value.code == -2 and "no CPU path" in str(e.value)


def test_bad_arguments_rejected_before_device(lib):
    for k in (0, 1, 33):
        with pytest.raises(capi.KqError) as e:
            capi.KreeqDB(k, 128)
        assert e.value.code == -1


def test_product_does_not_touch_oracle():
    """the oracle is test infrastructure: nothing under kreeq_amd/ or include/ may reference it"""
    for base in ("kreeq_amd", "include"):
        for dp, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".h", ".hip", ".cpp", ".c")):
                    txt = open(os.path.join(dp, f), errors="replace").read()
                    assert "oracle" not in txt.lower(), os.path.join(dp, f)


def test_pack_bases_host_only(lib):
    """kq_pack_bases needs no GPU: 16 bases -> one u32 of 2-bit codes + one u16 of invalid-base bits"""
    import numpy as np

    rng = np.random.default_rng(11)
    raw = bytes(rng.choice(list(b"ACGTacgtNn\n-"), size=10_007).tolist())
    codes, inv = capi.pack_bases(raw)
    assert len(codes) == len(inv) == (len(raw) + 15) // 16
    lut = np.full(256, 4, dtype=np.uint8)
    for ch, v in zip(b"ACGTacgt", [0, 1, 2, 3, 0, 1, 2, 3]):
        lut[ch] = v
    c = lut[np.frombuffer(raw, dtype=np.uint8)]
    cc = np.concatenate([c, np.full((-len(c)) % 16, 4, dtype=np.uint8)]).reshape(-1, 16)
    want_codes = ((cc & 3).astype(np.uint64) << (2 * np.arange(16, dtype=np.uint64))).sum(axis=1).astype(np.uint32)
    want_inv = (((cc >> 2) & 1).astype(np.uint32) << np.arange(16, dtype=np.uint32)).sum(axis=1).astype(np.uint16)
    assert np.array_equal(codes, want_codes) and np.array_equal(inv, want_inv)
    c0, i0 = capi.pack_bases(b"")
    assert len(c0) == 0 and len(i0) == 0



def test_device_packer_matches_kq_pack_bases(lib):
    """synth.pack_dev (the packer bench.py and the human-scale tests run on the device) writes kq_pack_bases' layout"""
    import numpy as np
    import torch

    from kreeq_amd import synth

    rng = np.random.default_rng(12)
    for n in (1, 15, 16, 17, 4097, 100_003):
        raw = rng.choice(np.frombuffer(b"ACGTacgtNn\n-", dtype=np.uint8), n).astype(np.uint8)
        c, i = synth.pack_dev(torch.from_numpy(raw), chunk_units=1000)
        c2, i2 = capi.pack_bases(raw.tobytes())
        assert np.array_equal(c.numpy().view(np.uint32), c2) and np.array_equal(i.numpy().view(np.uint16), i2), n
