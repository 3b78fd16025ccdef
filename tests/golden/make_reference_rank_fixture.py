"""Link-prediction ranks at BASELINE scale computed by THE REFERENCE'S OWN Scorer (run where /root/reference exists):

    python -B tests/golden/make_reference_rank_fixture.py     ->  tests/golden/reference_ranks_fullscale.npz

The reference's `common/evaluation.py` Scorer (known-triple dictionaries, raw / filtered rank definition of
MrrScore.append_line, :148-153,349-389) ranks 1,200 query triples against all 14,541 FB15k-237 entities, on scores a
`ScoreTableModel` (the model surface the Scorer needs, model.py:59-81) computes from a code table and a relation table
that a seed regenerates -- sigmoid DistMult scores in fp32, as decoders/bilinear_diag.py:51-61 define them.  Stored:
the triples and the ranks (integers), nothing the test could not regenerate otherwise.  tests/test_gpu_eval.py loads
the same tables into the engine and holds rgcn_rank_device to these ranks.

The code table is made so that a one-layer engine on an EMPTY graph reproduces it exactly as its codes:
codes = [C | 0] with C = P - N, P = max(C, 0), N = max(-C, 0); engine: W_emb = [P | N], b_emb = 0,
W_self = [[I, 0], [-I, 0]]  ->  H1 = relu(W_emb) . W_self = [P - N | 0]  (exact in fp32: one non-zero product per sum).
"""
import os
import sys

sys.dont_write_bytecode = True

import numpy as np  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

V, R, HALF, SEED, NQ = 14541, 237, 32, 77, 1200


def tables(seed=SEED):
    """(codes [V, 2*HALF], rel [R, 2*HALF]) -- the same arrays in the generator and in the test"""
    rng = np.random.RandomState(seed)
    c = (rng.randn(V, HALF) * 0.7).astype(np.float32)
    codes = np.concatenate([c, np.zeros_like(c)], axis=1)
    rel = rng.randn(R, 2 * HALF).astype(np.float32)
    return codes, rel


def splits(seed=SEED):
    """train / valid / test triples: the real FB15k-237 valid+test triples of tests/golden/graphs.npz, re-split"""
    with np.load(os.path.join(HERE, "graphs.npz")) as z:
        pool = z["fb237_valid_test"].astype(np.int64)
    perm = np.random.RandomState(seed + 1).permutation(len(pool))
    pool = pool[perm]
    return pool[NQ + 3000:], pool[NQ:NQ + 3000], pool[:NQ]


def main():
    from make_reference_fixtures import reference_modules, ScoreTableModel
    m = reference_modules()
    codes, rel = tables()
    train, valid, test = splits()
    scorer = m.evaluation.Scorer({"Metric": "MRR"})
    for part in (train, valid, test):
        scorer.register_data(part)
    scorer.register_degrees(train)
    scorer.register_model(ScoreTableModel(codes, rel))
    scorer.finalize_frequency_computation(np.concatenate((train, valid, test), axis=0))
    score = scorer.compute_scores(test, verbose=False)
    summary = score.get_summary()
    raw = np.asarray(score.raw_ranks, dtype=np.int64)
    filt = np.asarray(score.filtered_ranks, dtype=np.int64)
    res = summary.results
    np.savez_compressed(os.path.join(HERE, "reference_ranks_fullscale.npz"), raw_ranks=raw, filtered_ranks=filt,
                        config=np.array([V, R, HALF, SEED, NQ], dtype=np.int64),
                        mrr_raw=np.float64(res['Raw'][summary.mrr_string()]),
                        mrr_filtered=np.float64(res['Filtered'][summary.mrr_string()]))
    print("ranks", raw.shape, filt.shape, "MRR raw %.6f filtered %.6f" % (res['Raw'][summary.mrr_string()],
                                                                       res['Filtered'][summary.mrr_string()]))


if __name__ == "__main__":
    main()
