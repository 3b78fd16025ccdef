"""Shared test helpers: seeded workloads in the reference's shapes."""
import numpy as np

import oracle


def make_case(V, R, d, L, kind, nb, E, seed=0, train=True, keep=0.8, self_edges=True):
    rng = np.random.RandomState(seed)
    params = oracle.init_params(V, R, d, L, kind, nb, rng=rng)
    # non-zero biases so the test exercises b_emb
    params["b_emb"] = (rng.randn(d) * 0.05).astype(np.float32)
    triples = np.stack([rng.randint(0, V, size=E), rng.randint(0, R, size=E),
                        rng.randint(0, V, size=E)], axis=1).astype(np.int32)
    if not self_edges and E:
        triples[:, 2] = np.where(triples[:, 2] == triples[:, 0], (triples[:, 2] + 1) % V, triples[:, 2])
    masks = [(rng.rand(V, d) < keep).astype(np.uint8) for _ in range(L)] if train else None
    dcodes = (rng.randn(V, d) * 1e-1).astype(np.float32)
    return params, triples, masks, dcodes
