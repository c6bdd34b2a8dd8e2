"""CPU ORACLE for the dropout fields of the HIP path.  TEST INFRASTRUCTURE ONLY (see valle_oracle.py's header).

Philox4x32 (Salmon, Moraes, Dror, Shaw: "Parallel Random Numbers: As Easy as 1, 2, 3", SC'11) with 7 rounds, restated
in numpy from the published algorithm (the Random123 library is not in this image):
    round(ctr, key):  hi0:lo0 = M0 * ctr[0];  hi1:lo1 = M1 * ctr[2]
                      ctr' = (hi1 ^ ctr[1] ^ key[0], lo1, hi0 ^ ctr[3] ^ key[1], lo0)
    key is bumped by (W0, W1) before every round but the first.
Pinned by the known-answer vectors of Random123's kat_vectors for philox4x32 (7 and 10 rounds), in tests/test_dropout_cpu.py.

`keep_field` is include/valle_hip.h's definition of a dropout field: element (row, col) is kept iff word col % 4 of
philox4x32_7(key = seed, counter = (col // 4, row, site lo, site hi)) >= round(p * 2^32).
"""
from __future__ import annotations

import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = 0x9E3779B9, 0xBB67AE85
MASK32 = np.uint64(0xFFFFFFFF)


def philox4x32(ctr, key, rounds=7):
    """ctr: 4 arrays (broadcastable) of uint32 values, key: 2 ints -> 4 uint32 arrays."""
    c = [np.asarray(x, dtype=np.uint64) & MASK32 for x in ctr]
    c = list(np.broadcast_arrays(*c))
    k0, k1 = int(key[0]) & 0xFFFFFFFF, int(key[1]) & 0xFFFFFFFF
    for r in range(rounds):
        p0, p1 = M0 * c[0], M1 * c[2]
        n0 = (p1 >> np.uint64(32)) ^ c[1] ^ np.uint64(k0)
        n2 = (p0 >> np.uint64(32)) ^ c[3] ^ np.uint64(k1)
        c = [n0, p1 & MASK32, n2, p0 & MASK32]
        k0, k1 = (k0 + W0) & 0xFFFFFFFF, (k1 + W1) & 0xFFFFFFFF
    return [x.astype(np.uint32) for x in c]


def threshold(p: float) -> int:
    t = int(np.floor(np.float64(np.float32(p)) * 4294967296.0 + 0.5))
    return max(1, min(t, 0xFFFFFFFF))


def keep_field(seed: int, site: int, p: float, rows: int, cols: int) -> np.ndarray:
    """(rows, cols) uint8 keep mask of the field (seed, site, p)."""
    assert cols % 4 == 0 and 0.0 < p < 1.0
    c4 = np.arange(cols // 4, dtype=np.uint64)[None, :]
    row = np.arange(rows, dtype=np.uint64)[:, None]
    words = philox4x32((c4, row, site & 0xFFFFFFFF, (site >> 32) & 0xFFFFFFFF), (seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF))
    bits = np.stack(words, axis=-1).reshape(rows, cols)            # word j of group c4 is column 4 c4 + j
    return (bits >= np.uint32(threshold(p))).astype(np.uint8)
