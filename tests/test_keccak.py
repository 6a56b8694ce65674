"""The Keccak-256 oracle (oracle/keccak.py, numpy) is pinned through hashlib: the same sponge with the FIPS-202 domain byte
(0x06 instead of Keccak's 0x01) must reproduce hashlib.sha3_256, byte path and word-vectorised path alike; plus the
well-known Keccak-256 digest of the empty string, and the product's host-side Keccak256Transcript against the oracle's."""
import hashlib

import numpy as np

import era_boojum_amd as E
from oracle import keccak as K


def test_sponge_equals_hashlib_sha3_with_fips_domain_byte():
    for msg in [b"", b"abc", bytes(range(135)), bytes(range(136)), bytes(range(137)), bytes(272), b"x" * 1000]:
        assert K.keccak256_bytes(msg, domain=0x06) == hashlib.sha3_256(msg).digest(), len(msg)
    rng = np.random.default_rng(0)
    for n in (0, 1, 16, 17, 18, 34, 93):
        w = rng.integers(0, 2**63, size=(5, n), dtype=np.uint64)
        d = K.hash_words(w, domain=0x06)
        for i in range(5):
            assert d[i].astype("<u8").tobytes() == hashlib.sha3_256(w[i].astype("<u8").tobytes()).digest(), n


def test_keccak256_known_answers():
    assert K.keccak256_bytes(b"").hex() == "c5d2460186f7233c927e7db2dcc703c0e500b653ca82273b7bfad8045d85a470"
    assert K.keccak256_bytes(b"abc").hex() == "4e03657aea45a94fc7d47ba826c8d667c0d1e6e33a64a036ec44f58fa12d6c45"
    w = np.frombuffer(b"abcdefgh" * 20, dtype="<u8")
    assert K.hash_words(w.reshape(1, -1))[0].astype("<u8").tobytes() == K.keccak256_bytes(b"abcdefgh" * 20)


def test_product_keccak_transcript_matches_oracle():
    L = K.layer()
    rng = np.random.default_rng(3)
    tp, to = E.Transcript(kind=4), L.Transcript()
    seq = [("c", 3), ("a", 5), ("c", 9), ("cap", 8), ("c", 2), ("a", 64), ("c", 5), ("a", 1), ("c", 1), ("a", 40), ("c", 3)]
    for kind, n in seq:
        if kind == "a":
            els = rng.integers(0, 2**64 - 1, size=n, dtype=np.uint64)
            els[0] = np.uint64(2**64 - 1)
            tp.absorb(els); to.absorb(els)
        elif kind == "cap":
            cap = rng.integers(0, 2**64 - 1, size=(n, 4), dtype=np.uint64)
            tp.absorb_cap(cap); to.absorb_cap(cap)
        else:
            assert [tp.challenge() for _ in range(n)] == [to.challenge() for _ in range(n)]
    qi = L.QueryIndexer(10, 3)
    assert [tp.query_index(10, 3) for _ in range(12)] == [qi.next(to) for _ in range(12)]
