"""End-to-end through the reference-shaped plugin surface on the GPU: settings -> model_builder chain ->
feed placeholders -> loss / backward / scoring, checked against the oracle (encoder) plus the
oracle's DistMult restatement (decoder loss, regulariser and gradients)."""
import numpy as np
import pytest

import oracle
import helpers
from relationprediction_amd.common import model_builder
from test_plugin_surface import BLOCK_EXP, load_settings

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("concat", ["Yes", "No"])
def test_toy_train_step_through_plugin_chain(tmp_path, concat):
    V, R, d, L, nb = 16, 9, 20, 2, 4
    kind = "block" if concat == "Yes" else "basis"
    triples = helpers.load_graph("toy_train")                       # the reference's data/Toy (43 triples)
    s, enc, dec = load_settings(tmp_path, BLOCK_EXP.replace("Concatenation=Yes", "Concatenation=" + concat),
                                V=V, R=R, E=len(triples))
    model = model_builder.build_decoder(model_builder.build_encoder(enc, triples), dec)
    np.random.seed(3)
    model.preprocess(triples)
    model.register_for_test(triples)
    model.initialize_train()
    weights = model.get_weights()
    params = {n: w.value() for w, n in zip(weights, oracle.weight_names(kind, L))}
    # a training batch shaped like t_func's (train.py:232-245): graph = kept half, X = positives + negatives
    rng = np.random.RandomState(0)
    graph = triples[rng.choice(len(triples), 21, replace=False)]
    neg = triples.copy(); neg[:, 2] = rng.randint(0, V, len(triples))
    X = np.concatenate([triples, neg]).astype(np.int32)
    Y = np.concatenate([np.ones(len(triples)), np.zeros(len(triples))]).astype(np.float32)
    for var, val in zip(model.get_train_input_variables(), (graph, X, Y)):
        var.feed(val)
    loss = model.get_loss('train') + model.get_regularization()
    grads = model.backward()
    assert len(grads) == len(weights)
    # what dropout did the engine draw?  re-materialise it and replay the step on the oracle
    rt = next(c for c in _chain(model) if hasattr(c, 'runtime')).runtime
    masks = [rt.engine.dropout_mask(l) for l in range(1, L + 1)]
    acts = oracle.encoder_forward(params, graph, V, L, kind, mode='train', dropout_masks=masks)
    oloss, dcodes, dWrel = oracle.distmult_loss_and_grads(acts[-1], params["W_relation"], X, Y, 0.01)
    assert abs(loss - oloss) <= 1e-5 * max(1.0, abs(oloss))
    ograds = oracle.encoder_backward(params, graph, V, L, kind, acts, dcodes, mode='train', dropout_masks=masks)
    ograds["W_relation"] = dWrel
    for w, name, g in zip(weights, oracle.weight_names(kind, L), grads):
        assert g.shape == w.shape
        helpers.assert_close(g, ograds[name], rel=5e-4, name=name)
    # scoring = full-graph inference in test mode (model.py:59-81)
    scores = model.score_all_objects(triples[:5])
    tacts = oracle.encoder_forward(params, triples, V, L, kind, mode='test')
    e1r = tacts[-1][triples[:5, 0]] * params["W_relation"][triples[:5, 1]]
    with np.errstate(over='ignore'):
        expect = 1.0 / (1.0 + np.exp(-(e1r @ tacts[-1].T)))
    np.testing.assert_allclose(scores, expect, atol=1e-4)
    assert model.score(triples[:5]).shape == (5,)
    assert model.score_all_subjects(triples[:5]).shape == (5, V)
    # weight assignment goes through to the engine
    weights[0].assign(np.zeros_like(params["W_emb"]))
    np.testing.assert_array_equal(weights[0].value(), 0)


def _chain(model):
    c = model
    while c is not None:
        yield c
        c = c.next_component
