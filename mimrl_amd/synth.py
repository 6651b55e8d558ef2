"""Portable synthetic inputs and parameter initialisers (pure numpy, no torch RNG).

BASELINE.json's configs are defined on *synthetic* MOSI/MOSEI-shaped triples (the datasets and BERT
weights are not shipped), so the generators live in the product package: ``bench.py``,
``Main.py --dataset synthetic`` and the parity tests all draw from here.  Everything is a pure
function of (name/shape, seed) so that the reference-side fixture generator, the CPU oracle and the
HIP path see bit-identical inputs without shipping weights.
"""
from __future__ import annotations

import zlib
from typing import Dict, Iterable, Tuple

import numpy as np


def _rng(tag: str, seed: int) -> np.random.Generator:
    return np.random.Generator(np.random.PCG64([zlib.crc32(tag.encode()), seed & 0xFFFFFFFF]))


def portable_tensor(name: str, shape: Tuple[int, ...], seed: int = 0) -> np.ndarray:
    """Deterministic non-trivial test initialiser: uniform(+-1/sqrt(fan_in)); LayerNorm gains
    ~1 +- 0.1, LayerNorm biases +-0.1 (so gain/bias bugs are visible in parity tests)."""
    g = _rng(name, seed)
    shape = tuple(int(s) for s in shape)
    base = name.rsplit(".", 2)
    is_ln = any(part.startswith("ln_") for part in name.split("."))
    if is_ln:
        u = g.uniform(-0.1, 0.1, size=shape)
        return (1.0 + u if name.endswith("weight") else u).astype(np.float32)
    fan_in = int(np.prod(shape[1:])) if len(shape) >= 2 else shape[0]      # Linear [out,in] / Conv1d [out,in,k]
    del base
    bound = 1.0 / np.sqrt(max(fan_in, 1))
    return g.uniform(-bound, bound, size=shape).astype(np.float32)


def portable_state(named_shapes: Iterable[Tuple[str, Tuple[int, ...]]], seed: int = 0) -> Dict[str, np.ndarray]:
    return {n: portable_tensor(n, s, seed) for n, s in named_shapes}


def default_tensor(name: str, shape: Tuple[int, ...], seed: int = 0, fan_in: int = None) -> np.ndarray:
    """Distribution-faithful default init of the reference modules:
    nn.Linear / nn.GRU uniform(+-1/sqrt(fan_in | H)); LayerNorm 1/0; critic-tower biases 0
    (VMI.py:47-51); every ``weight_hh`` orthogonal (Customization.py:18-21).  ``fan_in``: for a bias, the input width
    of its layer's weight (nn.Linear draws the bias from +-1/sqrt(in_features)); see ``default_state``."""
    g = _rng("default:" + name, seed)
    shape = tuple(int(s) for s in shape)
    parts = name.split(".")
    if any(p.startswith("ln_") for p in parts):
        return (np.ones(shape) if name.endswith("weight") else np.zeros(shape)).astype(np.float32)
    if "critic_model" in name and name.endswith("bias"):
        return np.zeros(shape, np.float32)
    if "weight_hh" in name:
        a = g.standard_normal(size=shape)
        q, r = np.linalg.qr(a)
        q = q * np.sign(np.diag(r))
        return q.astype(np.float32)
    if parts[0].startswith("rnn_"):
        bound = 1.0 / np.sqrt(128.0)
    else:
        if len(shape) >= 2:
            fan_in = int(np.prod(shape[1:]))
        elif fan_in is None:                     # bias without its weight's shape: fall back to its own length
            fan_in = shape[0]
        bound = 1.0 / np.sqrt(max(fan_in, 1))
    return g.uniform(-bound, bound, size=shape).astype(np.float32)


def default_state(named_shapes: Iterable[Tuple[str, Tuple[int, ...]]], seed: int = 0) -> Dict[str, np.ndarray]:
    """``default_tensor`` for a whole model: every ``X.bias`` gets the fan-in of ``X.weight``."""
    named_shapes = [(n, tuple(int(d) for d in s)) for n, s in named_shapes]
    shapes = dict(named_shapes)
    out = {}
    for n, s in named_shapes:
        fi = None
        if n.endswith(".bias") and n[:-5] + ".weight" in shapes:
            w = shapes[n[:-5] + ".weight"]
            fi = int(np.prod(w[1:])) if len(w) >= 2 else None
        out[n] = default_tensor(n, s, seed, fan_in=fi)
    return out


def synthetic_batch(B: int, T: int, d_t: int = 768, d_a: int = 74, d_v: int = 35, seed: int = 0,
                    ragged: bool = False):
    """(t_feat[B,T,d_t], a[B,T,d_a], v[B,T,d_v], y[B]) : N(0,1) features, U(-3,3) labels.
    ``ragged`` zero-fills a per-sample tail of ``a`` and ``v`` (the reference infers lengths from
    all-zero rows, Model.py:425-432)."""
    g = _rng(f"batch:{B}:{T}", seed)
    t = g.standard_normal((B, T, d_t)).astype(np.float32)
    a = g.standard_normal((B, T, d_a)).astype(np.float32)
    v = g.standard_normal((B, T, d_v)).astype(np.float32)
    y = g.uniform(-3.0, 3.0, size=(B,)).astype(np.float32)
    if ragged:
        la = g.integers(1, T + 1, size=B)
        lv = g.integers(1, T + 1, size=B)
        la[0], lv[0] = T, T
        for b in range(B):
            a[b, la[b]:] = 0.0
            v[b, lv[b]:] = 0.0
    return t, a, v, y


def synthetic_banks(N: int, D: int = 128, seed: int = 0):
    """Feature banks as in the reference's own smoke test (Model.py:607): C~U(-3,3)[N,1], F,T,A,V~N(0,1)[N,D]."""
    g = _rng(f"banks:{N}", seed)
    C = g.uniform(-3.0, 3.0, size=(N, 1)).astype(np.float32)
    F, T, A, V = (g.standard_normal((N, D)).astype(np.float32) for _ in range(4))
    return {"C": C, "F": F, "T": T, "A": A, "V": V}


def draw_anchors(N: int, m: int, calls: int = 6):
    """The reference's anchor draw: ``np.random.choice(range(N), size=m, replace=False)`` on the GLOBAL
    numpy RNG, once per prod_knn_sample call (Model.py:81) -> int64 [calls, m]."""
    return np.stack([np.random.choice(N, size=m, replace=False) for _ in range(calls)]).astype(np.int64)
