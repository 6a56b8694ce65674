"""ORACLE — TEST INFRASTRUCTURE ONLY.

Keccak-256 (the original Keccak padding 0x01 … 0x80 of the `sha3::Keccak256` type the reference uses as a tree hasher,
src/cs/oracle/mod.rs:247-312, and in Keccak256Transcript, transcript.rs:264-372) — python's hashlib only has the FIPS-202
variant (padding 0x06), so the sponge is restated here on numpy uint64 lanes, vectorised over many messages, and PINNED
through hashlib: with the domain byte switched to 0x06 the very same code must reproduce hashlib.sha3_256
(tests/test_keccak.py).  `layer()` gives the merkle_* / Transcript / QueryIndexer / do_fri set of oracle/blake.py over it.
"""
import numpy as np

_RC = [0x0000000000000001, 0x0000000000008082, 0x800000000000808a, 0x8000000080008000, 0x000000000000808b,
       0x0000000080000001, 0x8000000080008081, 0x8000000000008009, 0x000000000000008a, 0x0000000000000088,
       0x0000000080008009, 0x000000008000000a, 0x000000008000808b, 0x800000000000008b, 0x8000000000008089,
       0x8000000000008003, 0x8000000000008002, 0x8000000000000080, 0x000000000000800a, 0x800000008000000a,
       0x8000000080008081, 0x8000000000008080, 0x0000000080000001, 0x8000000080008008]
_ROT = [[0, 36, 3, 41, 18], [1, 44, 10, 45, 2], [62, 6, 43, 15, 61], [28, 55, 25, 21, 56], [27, 20, 39, 8, 14]]   # [x][y]
RATE_LANES = 17


def _rotl(v, n):
    n %= 64
    if n == 0:
        return v.copy()       # never alias a state row: chi below overwrites the rows it still reads through b
    return (v << np.uint64(n)) | (v >> np.uint64(64 - n))


def keccak_f(a):
    """Keccak-f[1600] on a (25, B) uint64 array of lanes a[x + 5y] (FIPS-202 §3.2-3.4), in place."""
    for rnd in range(24):
        c = [a[x] ^ a[x + 5] ^ a[x + 10] ^ a[x + 15] ^ a[x + 20] for x in range(5)]
        d = [c[(x + 4) % 5] ^ _rotl(c[(x + 1) % 5], 1) for x in range(5)]
        for i in range(25):
            a[i] = a[i] ^ d[i % 5]
        b = [None] * 25
        for x in range(5):
            for y in range(5):
                b[y + 5 * ((2 * x + 3 * y) % 5)] = _rotl(a[x + 5 * y], _ROT[x][y])
        for y in range(0, 25, 5):
            for x in range(5):
                a[y + x] = b[y + x] ^ (~b[y + (x + 1) % 5] & b[y + (x + 2) % 5])
        a[0] = a[0] ^ np.uint64(_RC[rnd])
    return a


def hash_words(words, domain=0x01):
    """Digests of B messages of the same length given as (B, n_words) uint64 words (little-endian bytes): (B, 4) words."""
    words = np.ascontiguousarray(words, dtype=np.uint64)
    if words.ndim == 1:
        words = words.reshape(1, -1)
    B, n = words.shape
    a = np.zeros((25, B), dtype=np.uint64)
    pos = 0
    while n - pos >= RATE_LANES:
        for k in range(RATE_LANES):
            a[k] ^= words[:, pos + k]
        keccak_f(a)
        pos += RATE_LANES
    rem = n - pos
    for k in range(rem):
        a[k] ^= words[:, pos + k]
    a[rem] ^= np.uint64(domain)                          # first padding byte right after the message
    a[RATE_LANES - 1] ^= np.uint64(0x8000000000000000)   # final bit of the rate
    keccak_f(a)
    return np.ascontiguousarray(a[:4].T)


def keccak256_bytes(data, domain=0x01):
    """One message given as bytes (any length): 32 digest bytes."""
    rate = 8 * RATE_LANES
    msg = bytearray(data)
    msg.append(domain)
    while len(msg) % rate:
        msg.append(0)
    msg[-1] |= 0x80
    a = np.zeros((25, 1), dtype=np.uint64)
    for off in range(0, len(msg), rate):
        blk = np.frombuffer(bytes(msg[off:off + rate]), dtype="<u8")
        for k in range(RATE_LANES):
            a[k, 0] ^= blk[k]
        keccak_f(a)
    return a[:4, 0].astype("<u8").tobytes()


def layer():
    from oracle import blake
    return blake.ByteHashLayer(keccak256_bytes, lambda words: hash_words(words), kind=4)
