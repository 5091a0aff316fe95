"""Bit-exact numpy restatement of the `jax.random` calls `Rodent.reset` makes
[REF Rodent_Env_Brax.py:73-85]: `split`, `randint`, `uniform` on the default threefry2x32 PRNG
(jax itself is not installed here).  Keys are uint32[2]; every function is
vectorised over a leading batch of keys so one call seeds all environments (the reference vmaps
`reset` over `split(key_env, num_envs)`).

Pinned by the known-answer vector in the reference notebook [NB Env_step.ipynb:1612-1630]
(tests/test_known_answers.py).
"""
from __future__ import annotations

import numpy as np

_ROT = ((13, 15, 26, 6), (17, 29, 16, 24))
_M32 = np.uint64(0xFFFFFFFF)


def _rotl(x, r):
    return ((x << np.uint32(r)) | (x >> np.uint32(32 - r))).astype(np.uint32)


def threefry2x32(k0, k1, x0, x1):
    """20-round Threefry-2x32. All arguments uint32 arrays (broadcastable)."""
    k0, k1, x0, x1 = (np.asarray(a, dtype=np.uint32) for a in (k0, k1, x0, x1))
    ks = (k0, k1, (k0 ^ k1 ^ np.uint32(0x1BD11BDA)).astype(np.uint32))
    with np.errstate(over="ignore"):
        x0 = (x0 + ks[0]).astype(np.uint32)
        x1 = (x1 + ks[1]).astype(np.uint32)
        for i in range(5):
            for r in _ROT[i % 2]:
                x0 = (x0 + x1).astype(np.uint32)
                x1 = _rotl(x1, r)
                x1 = (x1 ^ x0).astype(np.uint32)
            x0 = (x0 + ks[(i + 1) % 3]).astype(np.uint32)
            x1 = (x1 + ks[(i + 2) % 3] + np.uint32(i + 1)).astype(np.uint32)
    return x0, x1


def PRNGKey(seed: int) -> np.ndarray:
    seed = int(seed)
    return np.array([(seed >> 32) & 0xFFFFFFFF, seed & 0xFFFFFFFF], dtype=np.uint32)


def random_bits(key, n: int) -> np.ndarray:
    """32-bit random words: key [..., 2] -> [..., n] (jax `_threefry_random_bits`, non-partitionable)."""
    key = np.asarray(key, dtype=np.uint32)
    # jax `threefry_2x32`: an odd-sized counter array is padded with a literal 0 (not with the next counter) before it is
    # split into halves, and the padded output word is dropped
    cnt = np.arange(n, dtype=np.uint32)
    if n % 2:
        cnt = np.concatenate([cnt, np.zeros(1, np.uint32)])
    m = cnt.size
    x0, x1 = cnt[: m // 2], cnt[m // 2:]
    k0, k1 = key[..., 0:1], key[..., 1:2]
    y0, y1 = threefry2x32(k0, k1, x0, x1)
    return np.concatenate([y0, y1], axis=-1)[..., :n]


def split(key, num: int = 2) -> np.ndarray:
    """key [..., 2] -> [..., num, 2]."""
    bits = random_bits(key, 2 * num)
    return bits.reshape(bits.shape[:-1] + (num, 2))


def fold_in(key, data: int) -> np.ndarray:
    key = np.asarray(key, dtype=np.uint32)
    y0, y1 = threefry2x32(key[..., 0], key[..., 1], np.uint32(0), np.uint32(data))
    return np.stack([y0, y1], axis=-1)


def uniform(key, n: int, minval=0.0, maxval=1.0) -> np.ndarray:
    """float32 uniform in [minval, maxval): key [..., 2] -> [..., n]."""
    bits = random_bits(key, n)
    f = ((bits >> np.uint32(9)) | np.uint32(0x3F800000)).view(np.float32) - np.float32(1.0)
    lo, hi = np.float32(minval), np.float32(maxval)
    return np.maximum(lo, f * (hi - lo) + lo).astype(np.float32)


def randint(key, minval: int, maxval: int) -> np.ndarray:
    """Scalar randint per key (shape ()), int32: key [..., 2] -> [...]."""
    ks = split(key, 2)
    hi_bits = random_bits(ks[..., 0, :], 1)[..., 0].astype(np.uint64)
    lo_bits = random_bits(ks[..., 1, :], 1)[..., 0].astype(np.uint64)
    span = np.uint64(maxval - minval)
    mult = (np.uint64(2 ** 32) % span)
    mult = (mult * mult) % span
    off = ((hi_bits % span) * mult + (lo_bits % span)) % span
    return (np.int64(minval) + off.astype(np.int64)).astype(np.int32)


def normal(key, n: int) -> np.ndarray:
    """float32 standard normal via erfinv of a uniform on (-1, 1) (jax.random.normal)."""
    from scipy.special import erfinv
    lo = np.nextafter(np.float32(-1.0), np.float32(0.0))
    u = uniform(key, n, lo, 1.0)
    return (np.float32(np.sqrt(2.0)) * erfinv(u.astype(np.float64))).astype(np.float32)
