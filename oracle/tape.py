"""Explicit-randomness tapes -- TEST ORACLE infrastructure.

The reference draws all randomness from `rand::thread_rng()` inside its functions
(ac17/mod.rs:143,201,281; bsw/mod.rs:227; lsw/mod.rs:129,188; aw11/mod.rs:131,249) and
offers no seed parameter.  Parity is therefore defined on explicit
randomness: every restated function takes an `rng` object and calls `rng.fr()` /
`rng.gt()` / `rng.g1()` / `rng.g2()` exactly where -- and in the order -- the reference
calls `rng.gen()`.  The same tape (a plain list of integers) is fed to the HIP engine.
"""
import hashlib

from . import bn254 as bn


class SeededRng:
    """Deterministic stream: SHAKE-256(seed || counter) -> 512 bits -> mod r."""

    def __init__(self, seed):
        self.seed = int(seed)
        self.ctr = 0
        self.log = []          # every Fr drawn, in order (the tape)

    def _u512(self):
        h = hashlib.shake_256(b"rabe-amd-tape" + self.seed.to_bytes(8, "little") + self.ctr.to_bytes(8, "little"))
        self.ctr += 1
        return int.from_bytes(h.digest(64), "little")

    def fr(self):
        v = self._u512() % bn.R
        self.log.append(v)
        return v

    def fr_nonzero(self):
        while True:
            v = self.fr()
            if v:
                return v

    # group elements are drawn as generator^fr (SURVEY.md 8c assumption (iv))
    def g1(self):
        return bn.g1_mul(bn.G1_GEN, self.fr_nonzero())

    def g2(self):
        return bn.g2_mul(bn.G2_GEN, self.fr_nonzero())

    def gt_exponent(self):
        """A Gt sample is represented by its exponent rho w.r.t. a caller-chosen base."""
        return self.fr_nonzero()


class ListRng:
    """Replays a recorded tape of Fr values."""

    def __init__(self, values):
        self.values = list(values)
        self.i = 0

    def fr(self):
        v = self.values[self.i]
        self.i += 1
        return v % bn.R

    fr_nonzero = fr

    def g1(self):
        return bn.g1_mul(bn.G1_GEN, self.fr())

    def g2(self):
        return bn.g2_mul(bn.G2_GEN, self.fr())
