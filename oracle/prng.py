"""JAX threefry2x32 PRNG restated in numpy (oracle; test infrastructure only).

The reference draws every random number through ``jax.random`` (jax 0.6.0,
``uv.lock:993``) with the default ``threefry2x32`` implementation and
``jax_threefry_partitionable=True``.  JAX itself is absent from the reference
tree, so this file restates the published algorithm (Random123 Threefry-2x32,
20 rounds) and JAX's derivations of ``split`` / ``random_bits`` / ``uniform`` /
``gumbel`` / ``categorical`` / ``randint`` / ``permutation`` from it.

Reference call sites: rec_magpo.py:135,202,373,439-450,642,660,699;
networks/utils/sable/decode.py:141-142; coordsum/env.py:56-57;
wrappers/auto_reset_wrapper.py:74; wrappers/episode_metrics.py:62.

Pinned by the Random123 known-answer vectors (tests/test_oracle_prng.py);
the JAX-specific derivations are restated from memory of jax/_src/prng.py and
jax/_src/random.py and are PARITY UNPINNED until checked on a machine with JAX.
"""
from __future__ import annotations

import math

import numpy as np

_U32 = np.uint32
_ROT = ((13, 15, 26, 6), (17, 29, 16, 24))
_PARITY = _U32(0x1BD11BDA)


def _rotl(x: np.ndarray, r: int) -> np.ndarray:
    return (x << _U32(r)) | (x >> _U32(32 - r))


def threefry2x32(k0, k1, c0, c1):
    """Threefry-2x32-20 block function; all arguments broadcastable uint32 arrays."""
    with np.errstate(over="ignore"):
        k0 = np.asarray(k0, dtype=_U32)
        k1 = np.asarray(k1, dtype=_U32)
        x0 = np.asarray(c0, dtype=_U32).copy()
        x1 = np.asarray(c1, dtype=_U32).copy()
        k0, k1, x0, x1 = np.broadcast_arrays(k0, k1, x0, x1)
        x0 = x0.copy()
        x1 = x1.copy()
        ks = (k0, k1, k0 ^ k1 ^ _PARITY)
        x0 = x0 + ks[0]
        x1 = x1 + ks[1]
        for i in range(5):
            for r in _ROT[i % 2]:
                x0 = x0 + x1
                x1 = _rotl(x1, r)
                x1 = x1 ^ x0
            x0 = x0 + ks[(i + 1) % 3]
            x1 = x1 + ks[(i + 2) % 3] + _U32(i + 1)
    return x0, x1


def prng_key(seed: int) -> np.ndarray:
    """jax.random.PRNGKey(seed) for 0 <= seed < 2**64: [hi32, lo32]."""
    seed = int(seed)
    return np.array([(seed >> 32) & 0xFFFFFFFF, seed & 0xFFFFFFFF], dtype=_U32)


def _iota_2x32(n: int):
    idx = np.arange(n, dtype=np.uint64)
    return (idx >> np.uint64(32)).astype(_U32), (idx & np.uint64(0xFFFFFFFF)).astype(_U32)


def split(key: np.ndarray, num: int = 2) -> np.ndarray:
    """jax.random.split (partitionable / 'fold-like'): row i = threefry(key, (0, i)).

    ``key`` may be a single key (2,) -> (num, 2), or a batch (..., 2) -> (..., num, 2).
    """
    key = np.asarray(key, dtype=_U32)
    hi, lo = _iota_2x32(num)
    k0 = key[..., 0][..., None]
    k1 = key[..., 1][..., None]
    x0, x1 = threefry2x32(k0, k1, hi, lo)
    return np.stack([x0, x1], axis=-1)


def random_bits(key: np.ndarray, n: int) -> np.ndarray:
    """32-bit ``_random_bits(key, 32, shape)`` flattened: element i = x0 ^ x1 of
    threefry(key, (hi(i), lo(i))).  Batched keys (..., 2) -> (..., n)."""
    key = np.asarray(key, dtype=_U32)
    hi, lo = _iota_2x32(n)
    k0 = key[..., 0][..., None]
    k1 = key[..., 1][..., None]
    x0, x1 = threefry2x32(k0, k1, hi, lo)
    return x0 ^ x1


def bits_to_uniform(bits: np.ndarray, minval: float = 0.0, maxval: float = 1.0) -> np.ndarray:
    """jax.random._uniform for float32 given the raw 32 bits."""
    fbits = (bits >> _U32(9)) | _U32(0x3F800000)
    floats = fbits.view(np.float32) - np.float32(1.0)
    lo = np.float32(minval)
    hi = np.float32(maxval)
    return np.maximum(lo, floats * (hi - lo) + lo).astype(np.float32)


def bits_to_gumbel(bits: np.ndarray) -> np.ndarray:
    """jax.random.gumbel (mode 'low'): -log(-log(uniform(tiny, 1))).
    The two logs are evaluated in float64 and rounded once to float32 so that every
    implementation (this oracle, the HIP kernel) produces the same, correctly rounded value;
    XLA's own float32 log may differ from it in the last ulp (PARITY UNPINNED)."""
    u = bits_to_uniform(bits, np.finfo(np.float32).tiny, 1.0)
    return (-np.log(-np.log(u.astype(np.float64)))).astype(np.float32)


def categorical(key: np.ndarray, logits: np.ndarray) -> np.ndarray:
    """jax.random.categorical(key, logits, axis=-1) with the gumbel tensor laid out
    row-major over ``logits.shape`` (what distrax's ``sample`` with n=1 produces for a
    (B,1,K) logits batch: shape (1,B,1,K); decode.py:141-142)."""
    logits = np.asarray(logits, dtype=np.float32)
    g = bits_to_gumbel(random_bits(key, logits.size)).reshape(logits.shape)
    return np.argmax(g + logits, axis=-1).astype(np.int32)


def randint(key: np.ndarray, n: int, minval: int, maxval: int) -> np.ndarray:
    """jax.random.randint(key, (n,), minval, maxval) for int32. Batched keys allowed."""
    ks = split(key, 2)
    hi_bits = random_bits(ks[..., 0, :], n)
    lo_bits = random_bits(ks[..., 1, :], n)
    span = _U32(maxval - minval) if maxval > minval else _U32(1)
    with np.errstate(over="ignore"):
        mult = _U32(1 << 16) % span
        mult = _U32((np.uint64(mult) * np.uint64(mult)) & np.uint64(0xFFFFFFFF)) % span
        off = (hi_bits % span) * mult + (lo_bits % span)
        off = off % span
    return (np.int64(minval) + off.astype(np.int64)).astype(np.int32)


def permutation(key: np.ndarray, n: int) -> np.ndarray:
    """jax.random.permutation(key, n): repeated stable sort by random 32-bit keys."""
    x = np.arange(n, dtype=np.int32)
    num_rounds = int(math.ceil(3 * math.log(max(1, n)) / math.log(2**32 - 1)))
    key = np.asarray(key, dtype=_U32)
    for _ in range(num_rounds):
        ks = split(key, 2)
        key, sub = ks[0], ks[1]
        sort_keys = random_bits(sub, n)
        order = np.argsort(sort_keys, kind="stable")
        x = x[order]
    return x


def uniform(key: np.ndarray, n: int, minval: float = 0.0, maxval: float = 1.0) -> np.ndarray:
    """jax.random.uniform(key, (n,), float32, minval, maxval) (shape () draws are n = 1)."""
    return bits_to_uniform(random_bits(key, n), minval, maxval)


def gumbel(key: np.ndarray, n: int) -> np.ndarray:
    """jax.random.gumbel(key, (n,), float32)."""
    return bits_to_gumbel(random_bits(key, n))


def choice(key: np.ndarray, n: int, num: int = 1, replace: bool = True, p=None) -> np.ndarray:
    """jax.random.choice(key, n, shape=(num,), replace=replace, p=p) for an integer ``a = n`` (jax/_src/random.py, `choice`;
    jax 0.6.0): all four branches of the published algorithm.

      p is None,  replace      randint(key, shape, 0, n)
      p is None, ~replace      permutation(key, n)[:num]
      p given,    replace      p_cuml = cumsum(p); r = p_cuml[-1] * (1 - uniform(key, shape)); searchsorted(p_cuml, r)   (side 'left')
      p given,   ~replace      Gumbel top-k: g = gumbel(key, (n,)) + log(p); top_k(g, num) indices (descending; ties: lower index first)

    ``p`` is promoted to float32 and NOT normalised by ``choice`` itself (callers that pass a boolean mask get the mask's own cumulative
    counts, which is what makes the draw uniform over the set cells).  Restated from memory of the JAX source: PARITY UNPINNED like
    the other derivations in this file; the float32 arithmetic (one multiply, one subtraction) is IEEE and order-free, the cumulative sum of a
    0 / 1 mask is exact in any summation order, and the gumbel uses this file's correctly rounded double log (see bits_to_gumbel).
    Returns int32 [num]."""
    n, num = int(n), int(num)
    if n <= 0:
        raise ValueError("a must be non-empty")
    if not replace and num > n:
        raise ValueError("Cannot take a larger sample than population when 'replace=False'")
    if p is None:
        if replace:
            return randint(key, num, 0, n)
        return permutation(key, n)[:num].astype(np.int32)
    p_arr = np.asarray(p).astype(np.float32)
    if p_arr.shape != (n,):
        raise ValueError("p must be None or match the shape of a")
    if replace:
        p_cuml = np.cumsum(p_arr, dtype=np.float32)
        r = (p_cuml[-1] * (np.float32(1.0) - uniform(key, num))).astype(np.float32)
        return np.searchsorted(p_cuml, r, side="left").astype(np.int32)
    with np.errstate(divide="ignore"):
        g = (gumbel(key, n) + np.log(p_arr.astype(np.float64)).astype(np.float32)).astype(np.float32)
    order = np.argsort(-g.astype(np.float64), kind="stable")   # descending, equal values keep index order (lax.top_k)
    return order[:num].astype(np.int32)


# ---------------------------------------------------------------------------------------------------------------------------
# Parameter initialisation on the JAX stream (rec_magpo.py:598-604,623: sable_network.init(net_key, ...), actor_network.init(actor_net_key, ...)).
# PARITY UNPINNED, twice over: (a) the samplers below restate jax/_src/random.py and jax/_src/nn/initializers.py from memory (normal =
# sqrt(2) erf_inv(uniform(nextafter(-1, 0), 1)); truncated_normal; variance_scaling; orthogonal = QR of a normal draw with the sign fix)
# and evaluate erf_inv with the single-precision polynomial of M. Giles, "Approximating the erfinv function" (the one XLA uses), in
# numpy float32 -- XLA's own log1p / fused multiply-adds and LAPACK's QR may differ from it in the last bits; (b) the per-parameter keys
# restate flax's scope RNG (flax/core/scope.py: LazyRng / _fold_in_static: sha1 over the module path and the per-scope make_rng counter,
# first four digest bytes folded into the 'params' key), also from memory.  What IS pinned here: distributions (moments, orthogonality,
# truncation bounds) and that the product's initialiser (magpo_amd/params.py) reproduces these arrays bit for bit.


def fold_in(key: np.ndarray, data: int) -> np.ndarray:
    """jax.random.fold_in(key, data) = threefry2x32(key, threefry_seed(uint32 data)) with seed words (0, data)."""
    key = np.asarray(key, dtype=_U32)
    x0, x1 = threefry2x32(key[0], key[1], _U32(0), _U32(int(data) & 0xFFFFFFFF))
    return np.array([x0, x1], dtype=_U32)


def flax_param_key(params_key: np.ndarray, path, counter: int) -> np.ndarray:
    """Key flax hands to the initialiser of the ``counter``-th parameter (1-based, in creation order) of the module at ``path`` (tuple of
    scope names below the root): LazyRng(params_key, path + (counter,)).as_jax_rng() = fold_in(params_key, first 4 bytes (big endian) of
    sha1 over the utf-8 names and the minimal big-endian bytes of the integers)."""
    import hashlib
    m = hashlib.sha1()
    for x in tuple(path) + (int(counter),):
        if isinstance(x, str):
            m.update(x.encode("utf-8"))
        else:
            m.update(int(x).to_bytes((int(x).bit_length() + 7) // 8, byteorder="big"))
    return fold_in(params_key, int.from_bytes(m.digest()[:4], byteorder="big"))


def erf_inv_f32(x: np.ndarray) -> np.ndarray:
    """Single-precision erf_inv (Giles' polynomial, central branch w < 5 and tail branch), every operation rounded to float32."""
    x = np.asarray(x, dtype=np.float32)
    f = np.float32
    w = -np.log1p(-(x * x)).astype(np.float32)
    wc = (w - f(2.5)).astype(np.float32)
    p = f(2.81022636e-08)
    for c in (3.43273939e-07, -3.5233877e-06, -4.39150654e-06, 0.00021858087, -0.00125372503, -0.00417768164, 0.246640727, 1.50140941):
        p = (f(c) + p * wc).astype(np.float32)
    wt = (np.sqrt(np.maximum(w, f(5.0))).astype(np.float32) - f(3.0)).astype(np.float32)
    q = f(-0.000200214257)
    for c in (0.000100950558, 0.00134934322, -0.00367342844, 0.00573950773, -0.0076224613, 0.00943887047, 1.00167406, 2.83297682):
        q = (f(c) + q * wt).astype(np.float32)
    return (np.where(w < f(5.0), p, q) * x).astype(np.float32)


def normal(key: np.ndarray, n: int) -> np.ndarray:
    """jax.random.normal(key, (n,), float32): sqrt(2) * erf_inv(uniform(key, minval=nextafter(-1, 0), maxval=1))."""
    lo = np.nextafter(np.float32(-1.0), np.float32(0.0))
    u = bits_to_uniform(random_bits(key, n), lo, 1.0)
    return (np.float32(np.sqrt(2)) * erf_inv_f32(u)).astype(np.float32)


def truncated_normal(key: np.ndarray, n: int, lower: float = -2.0, upper: float = 2.0) -> np.ndarray:
    """jax.random.truncated_normal(key, lower, upper, (n,), float32)."""
    f = np.float32
    s2 = f(np.sqrt(2))
    a, b = f(math.erf(float(f(lower) / s2))), f(math.erf(float(f(upper) / s2)))
    u = bits_to_uniform(random_bits(key, n), a, b)
    out = (s2 * erf_inv_f32(u)).astype(np.float32)
    return np.clip(out, np.nextafter(f(lower), f(np.inf)), np.nextafter(f(upper), f(-np.inf))).astype(np.float32)


def init_normal(key: np.ndarray, shape, stddev: float) -> np.ndarray:
    """nn.initializers.normal(stddev)(key, shape): random.normal * stddev (retention.py:50-64,237-246: stddev = 1 / embed_dim)."""
    return (normal(key, int(np.prod(shape))).reshape(shape) * np.float32(stddev)).astype(np.float32)


def init_lecun_normal(key: np.ndarray, shape) -> np.ndarray:
    """variance_scaling(1.0, 'fan_in', 'truncated_normal') for a Dense kernel [fan_in, fan_out] (flax GRUCell's input kernels)."""
    std = np.float32(np.sqrt(np.float32(1.0) / np.float32(shape[0]))) / np.float32(0.87962566103423978)
    return (truncated_normal(key, int(np.prod(shape))).reshape(shape) * std).astype(np.float32)


def init_orthogonal(key: np.ndarray, shape, scale: float = 1.0) -> np.ndarray:
    """nn.initializers.orthogonal(scale)(key, (rows, cols)): QR of a normal (max, min) matrix, columns signed by diag(R), transposed
    when rows < cols (sable_network.py:97-107,263-282, torsos.py:42, heads.py:53; flax GRUCell's recurrent kernels)."""
    rows, cols = int(shape[0]), int(shape[1])
    mshape = (cols, rows) if rows < cols else (rows, cols)
    a = normal(key, mshape[0] * mshape[1]).reshape(mshape)
    q, r = np.linalg.qr(a)
    q = (q * np.sign(np.diag(r))[None, :]).astype(np.float32)
    if rows < cols:
        q = q.T
    return (np.float32(scale) * q).astype(np.float32)
