"""The chunked float64 restatement of the decoder oracle (tests/helpers.py: what the BASELINE-scale GPU test compares
with, where N = 330,000 x d = 500 gathers do not fit at once) is the oracle itself on a case both can run."""
import numpy as np

import oracle
from helpers import chunked_distmult_float64


def test_chunked_float64_decoder_is_the_oracle():
    rng = np.random.RandomState(0)
    V, R, d, N = 40, 5, 8, 333
    codes, w_rel = rng.randn(V, d).astype(np.float32), rng.randn(V, d).astype(np.float32)
    X = np.stack([rng.randint(0, V, N), rng.randint(0, R, N), rng.randint(0, V, N)], 1).astype(np.int32)
    Y = (rng.rand(N) < 0.2).astype(np.float32)
    a = chunked_distmult_float64(codes, w_rel, X, Y, 0.01, chunk=50)
    b = oracle.distmult_loss_and_grads(codes, w_rel, X, Y, 0.01)
    assert abs(a[0] - b[0]) <= 1e-6 * abs(b[0])
    assert np.abs(a[1] - b[1]).max() <= 1e-6 * np.abs(b[1]).max() and np.abs(a[2] - b[2]).max() <= 1e-6 * np.abs(b[2]).max()
