"""ORACLE — TEST INFRASTRUCTURE ONLY.

The Blake2s-256 flavour of the hashing layer (the reference's non-recursive configuration, gadgets/sha256/mod.rs:265-270):
tree hasher `impl TreeHasher<F> for blake2::Blake2s256` (src/cs/oracle/mod.rs:179-245) and `Blake2sTranscript`
(src/cs/implementations/transcript.rs:155-262) + the non-algebraic branch of `BoolsBuffer::get_bits` (:398-411).
The hash itself is python's `hashlib.blake2s` — an implementation nobody here wrote (RFC 7693; the reference links the
`blake2` crate 0.10) — so the GPU kernels and the product's host code are checked against an independent oracle.
Same function names as the Poseidon2 layer of oracle/__init__.py, so oracle/prover.py and oracle/verifier.py switch
between the two by picking a module.  Digests are (…, 4) uint64 arrays: the 32 digest bytes, little-endian packed.
"""
import hashlib

import numpy as np

import oracle as O

P = O.P


def _digest_words(b):
    return np.frombuffer(b, dtype="<u8").copy()


def _canon_words(els):
    a = np.asarray(els, dtype=np.uint64)
    return np.where(a >= np.uint64(P), a - np.uint64(P), a)


class ByteHashLayer:
    """merkle_* / Transcript / QueryIndexer / do_fri over a 32-byte hash of byte strings (`impl TreeHasher for
    Blake2s256 / Keccak256`, oracle/mod.rs:179-312; Blake2sTranscript / Keccak256Transcript, transcript.rs:155-372).
    hash_bytes(bytes) -> 32 bytes; hash_many(words (B, n) uint64) -> (B, 4) uint64 (optional vectorised form)."""

    def __init__(self, hash_bytes, hash_many=None, kind=3):
        self.hash_bytes, self.hash_many, self.kind = hash_bytes, hash_many, kind
        layer = self

        class Transcript:
            """A running hasher (modelled by the bytes fed since the last reset), a byte buffer, unused challenge bytes."""
            kind = layer.kind

            def __init__(self, kind=None):
                self.pending, self.buffer, self.avail = b"", b"", b""

            def absorb(self, els):
                self.buffer += _canon_words(np.asarray(els, dtype=np.uint64).reshape(-1)).astype("<u8").tobytes()

            def absorb_cap(self, cap):
                self.buffer += np.asarray(cap, dtype="<u8").tobytes()      # raw digest bytes

            def _reseed(self):
                out = layer.hash_bytes(self.pending)       # finalize_reset, then the new state starts from the output
                self.pending = out
                self.avail = out

            def _flush(self):
                if self.buffer:
                    self.pending += self.buffer
                    self.buffer = b""
                    self._reseed()

            def challenge_bytes(self, n):
                self._flush()
                while len(self.avail) < n:
                    self._reseed()
                out, self.avail = self.avail[:n], self.avail[n:]
                return out

            def challenge(self):
                self._flush()
                if not self.avail:
                    self._reseed()
                b8, self.avail = self.avail[:8], self.avail[8:]
                return int.from_bytes(b8, "little") % P            # from_u64_with_reduction

            def challenge_ext(self):
                return (self.challenge(), self.challenge())

        self.Transcript = Transcript
        self.QueryIndexer = QueryIndexer

    # ---- tree hasher
    def hash_leaf(self, els):
        return _digest_words(self.hash_bytes(_canon_words(np.asarray(els, dtype=np.uint64).reshape(-1)).astype("<u8").tobytes()))

    def hash_node(self, l, r):
        return _digest_words(self.hash_bytes(np.asarray(l, dtype="<u8").tobytes() + np.asarray(r, dtype="<u8").tobytes()))

    def _leaves(self, rows):
        rows = _canon_words(rows)
        if self.hash_many is not None:
            return self.hash_many(rows)
        return np.stack([_digest_words(self.hash_bytes(rows[i].astype("<u8").tobytes())) for i in range(rows.shape[0])])

    def _nodes(self, leaf_hashes, cap_size):
        layers = [leaf_hashes]
        while layers[-1].shape[0] > cap_size:
            prev = layers[-1]
            pairs = prev.reshape(-1, 8)
            if self.hash_many is not None:
                layers.append(self.hash_many(pairs))
            else:
                layers.append(np.stack([_digest_words(self.hash_bytes(pairs[i].astype("<u8").tobytes())) for i in range(pairs.shape[0])]))
        return np.concatenate(layers, axis=0)            # all layers back to back, like the C oracle / the GPU tree

    def merkle_construct(self, cols, cap_size, threads=1):
        """cols: (n_cols, num_leaves) view; leaf I = hash of column values at I."""
        return self._nodes(self._leaves(np.ascontiguousarray(np.asarray(cols, dtype=np.uint64).T)), cap_size)

    def merkle_construct_chunked(self, srcs, elems_per_leaf, cap_size, threads=1):
        srcs = [np.asarray(s, dtype=np.uint64) for s in srcs]
        E = elems_per_leaf
        rows = np.concatenate([s.reshape(-1, E) for s in srcs], axis=1)
        return self._nodes(self._leaves(rows), cap_size)

    @staticmethod
    def merkle_cap(tree, num_leaves, cap_size):
        return tree[2 * num_leaves - 2 * cap_size: 2 * num_leaves - cap_size]

    @staticmethod
    def merkle_proof(tree, num_leaves, cap_size, idx):
        path, off, ln, i = [], 0, num_leaves, idx
        leaf_hash = tree[idx]
        while ln > cap_size:
            path.append(tree[off + (i ^ 1)])
            off += ln
            ln //= 2
            i //= 2
        return leaf_hash, (np.stack(path) if path else np.zeros((0, 4), dtype=np.uint64))

    def merkle_verify(self, path, cap, leaf_hash, idx):
        cur, i = np.asarray(leaf_hash, dtype=np.uint64), idx
        for sib in np.asarray(path, dtype=np.uint64).reshape(-1, 4):
            cur = self.hash_node(cur, sib) if i % 2 == 0 else self.hash_node(sib, cur)
            i //= 2
        return bool(np.array_equal(cur, np.asarray(cap, dtype=np.uint64).reshape(-1, 4)[i]))

    def do_fri(self, c0, c1, log_lde, schedule, cap_size, transcript, threads=1):
        """do_fri (fri/mod.rs:49-358) with byte-hash oracles: same dict as oracle.do_fri.  Folding / interpolation reuse the
        C oracle (hash independent); the trees and the transcript are the ones of this layer."""
        c0, c1 = np.asarray(c0, dtype=np.uint64), np.asarray(c1, dtype=np.uint64)
        log_full = int(c0.size).bit_length() - 1
        roots = O.twiddles(log_full, inverse=True)
        kappa = O.inv(7)
        out = {"trees": [], "caps": [], "sources": [], "challenges": []}
        cur0, cur1 = c0, c1
        for k in schedule:
            E = 1 << k
            leaves = cur0.size // E
            tree = self.merkle_construct_chunked([cur0, cur1], E, cap_size)
            cap = self.merkle_cap(tree, leaves, cap_size)
            out["trees"].append(tree)
            out["caps"].append(cap)
            out["sources"].append((cur0, cur1))
            transcript.absorb_cap(cap)
            ch = transcript.challenge_ext()
            out["challenges"].append(ch)
            alpha = ch
            for _ in range(k):
                cur0, cur1 = O.fri_fold(cur0, cur1, roots, kappa, alpha)
                kappa = kappa * kappa % P
                alpha = ((alpha[0] * alpha[0] + 7 * alpha[1] * alpha[1]) % P, (2 * alpha[0] * alpha[1]) % P)
        out["last_folded"] = (cur0, cur1)
        coset = O.inv(kappa)
        f0 = O.ifft_natural_to_natural(O.bitreverse(cur0), coset)
        f1 = O.ifft_natural_to_natural(O.bitreverse(cur1), coset)
        out["final_monomials"] = (f0, f1)
        out["final_degree"] = cur0.size >> log_lde
        transcript.absorb(f0[:out["final_degree"]])
        transcript.absorb(f1[:out["final_degree"]])
        return out


class QueryIndexer:
    """BoolsBuffer for a non-algebraic transcript (all 64 bits of 8 challenge bytes) + the index split of prover.rs:2161-2182."""

    def __init__(self, log_n, log_lde):
        self.log_n, self.log_lde, self.bits = log_n, log_lde, []

    def next(self, t):
        need = self.log_n + self.log_lde
        while len(self.bits) < need:
            x = int.from_bytes(t.challenge_bytes(8), "little")
            self.bits += [(x >> i) & 1 for i in range(64)]
        take, self.bits = self.bits[:need], self.bits[need:]
        inner = sum(b << i for i, b in enumerate(take[:self.log_n]))
        coset = sum(b << i for i, b in enumerate(take[self.log_n:]))
        return (coset << self.log_n) + inner


# ---- the Blake2s layer itself (module-level names, so `from oracle import blake as B; B.hash_leaf(...)` keeps working)
_LAYER = ByteHashLayer(lambda data: hashlib.blake2s(data).digest(), None, kind=3)
hash_leaf, hash_node = _LAYER.hash_leaf, _LAYER.hash_node
merkle_construct, merkle_construct_chunked = _LAYER.merkle_construct, _LAYER.merkle_construct_chunked
merkle_cap, merkle_proof, merkle_verify, do_fri = _LAYER.merkle_cap, _LAYER.merkle_proof, _LAYER.merkle_verify, _LAYER.do_fri
Transcript = _LAYER.Transcript
