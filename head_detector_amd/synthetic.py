"""Seeded synthetic stand-ins for the two user-supplied assets (weights: arch.random_state_dict; FLAME
constants: here), used by bench.py / smoke tests because neither the released .trcd nor the licensed
generic_model.pkl can be redistributed."""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np


def synthetic_flame_model(seed: int = 3, V: int = 5023, NB: int = 400, NJ: int = 5, v_template: Optional[np.ndarray] = None) -> Dict[str, np.ndarray]:
    """Same keys / shapes / sparsity pattern as the FLAME pickle: v_template [V,3], shapedirs [V,3,NB] ~ N(0,1e-3)
    with a decaying spectrum, posedirs [V,3,9(NJ-1)], J_regressor [NJ,V] sparse rows summing to 1,
    weights [V,NJ] softmax rows, kintree_table [2,NJ] with parents (-1,0,1,1,1), f [9976,3]."""
    rng = np.random.default_rng(seed)
    if v_template is None:
        u = rng.normal(size=(V, 3))
        u /= np.linalg.norm(u, axis=1, keepdims=True)
        v_template = u * np.array([0.10, 0.16, 0.11]) + np.array([0.0, -0.03, -0.04])
    shapedirs = rng.normal(0, 1e-3, size=(V, 3, NB)) * (1.0 / np.sqrt(1.0 + np.arange(NB) / 10.0))[None, None, :]
    posedirs = rng.normal(0, 1e-3, size=(V, 3, (NJ - 1) * 9))
    J_regressor = np.zeros((NJ, V))
    for j in range(NJ):
        idx = rng.choice(V, size=64, replace=False)
        w = rng.random(64)
        J_regressor[j, idx] = w / w.sum()
    logits = rng.normal(0, 2.0, size=(V, NJ))
    weights = np.exp(logits) / np.exp(logits).sum(1, keepdims=True)
    kintree = np.array([[4294967295, 0, 1, 1, 1][:NJ], list(range(NJ))], dtype=np.int64)
    return dict(v_template=np.asarray(v_template, dtype=np.float64), shapedirs=shapedirs, posedirs=posedirs, J_regressor=J_regressor, kintree_table=kintree,
                weights=weights, f=rng.integers(0, V, size=(9976, 3)))
