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


def test_chunked_float64_block_encoder_is_the_oracle():
    """The edge-chunked float64 restatement of the block encoder (tests/helpers.py: what the 272,115-edge GPU test
    compares with) is oracle.encoder_step on a case both can run -- every norm reading, train and test mode, chunks that
    cut rows and relations anywhere."""
    from helpers import (chunked_block_encoder_forward_float64, chunked_block_encoder_backward_float64, make_case,
                         oracle_float64)
    params, triples, masks, dcodes = make_case(50, 7, 12, 2, "block", 4, 333, seed=4)
    c = {"params": params, "triples": triples, "masks": masks, "dcodes": dcodes, "V": 50, "L": 2}
    for norm in (oracle.NORM_INTENDED, oracle.NORM_TF_AS_EXECUTED, oracle.NORM_NONE):
        for mode in ("train", "test"):
            with oracle_float64():
                p64 = {k: np.asarray(v, dtype=np.float64) for k, v in c["params"].items()}
                acts, grads = oracle.encoder_step(p64, c["triples"], c["V"], c["L"], "block",
                                                  c["dcodes"].astype(np.float64), keep_prob=0.8,
                                                  dropout_masks=c["masks"], norm_mode=norm, mode=mode)
                # (inside the float64 switch: the 1/deg values are then float64 on both sides; outside it the
                # restatement takes them in fp32, as the engine and the reference compute them)
                got, scales = chunked_block_encoder_forward_float64(c["params"], c["triples"], c["V"], c["L"],
                                                                    mode=mode, masks=c["masks"], norm_mode=norm,
                                                                    chunk=37, with_scale=True)
                gg = chunked_block_encoder_backward_float64(c["params"], c["triples"], c["V"], c["L"], acts,
                                                            c["dcodes"], mode=mode, masks=c["masks"], norm_mode=norm,
                                                            chunk=41)
            for a, b, sc in zip(acts, got, scales):
                assert np.abs(a - b).max() <= 1e-12 * max(np.abs(a).max(), 1.0)
                assert (np.abs(b) <= sc * (1 + 1e-12) + 1e-300).all()      # the scale bounds the value it belongs to
            for k, ref in grads.items():
                assert np.abs(gg[k] - ref).max() <= 1e-12 * max(np.abs(ref).max(), 1.0), k


def test_chunked_float64_basis_encoder_is_the_oracle():
    """The edge-chunked float64 restatement of the BASIS encoder (tests/helpers.py: what the 272,115-edge basis GPU test
    compares with; reference dataflow of gcn_basis.py:39-88, per edge) is oracle.encoder_step on a case both can run --
    every norm reading, train and test mode, B = 1 and B = 3, chunks that cut rows and relations anywhere."""
    from helpers import (chunked_basis_encoder_forward_float64, chunked_basis_encoder_backward_float64, make_case,
                         oracle_float64)
    for B, seed in ((3, 6), (1, 7)):
        params, triples, masks, dcodes = make_case(50, 7, 12, 2, "basis", B, 333, seed=seed)
        for norm in (oracle.NORM_INTENDED, oracle.NORM_TF_AS_EXECUTED, oracle.NORM_NONE):
            for mode in ("train", "test"):
                with oracle_float64():
                    p64 = {k: np.asarray(v, dtype=np.float64) for k, v in params.items()}
                    acts, grads = oracle.encoder_step(p64, triples, 50, 2, "basis", dcodes.astype(np.float64),
                                                      keep_prob=0.8, dropout_masks=masks, norm_mode=norm, mode=mode)
                    got, scales = chunked_basis_encoder_forward_float64(params, triples, 50, 2, mode=mode, masks=masks,
                                                                        norm_mode=norm, chunk=37, with_scale=True)
                    gg = chunked_basis_encoder_backward_float64(params, triples, 50, 2, acts, dcodes, mode=mode,
                                                                masks=masks, norm_mode=norm, chunk=41)
                for a, b, sc in zip(acts, got, scales):
                    assert np.abs(a - b).max() <= 1e-12 * max(np.abs(a).max(), 1.0)
                    assert (np.abs(b) <= sc * (1 + 1e-12) + 1e-300).all()
                assert set(gg) == set(grads)
                for k, ref in grads.items():
                    assert np.abs(gg[k] - ref).max() <= 1e-12 * max(np.abs(ref).max(), 1.0), (k, B, norm, mode)
