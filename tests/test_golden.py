"""The oracle against the committed fingerprints (tests/golden/expected.npz, made by make_golden.py)."""
import os

import numpy as np
import pytest

import oracle
import helpers


@pytest.fixture(scope="module")
def expected():
    with np.load(os.path.join(helpers.GOLDEN_DIR, "expected.npz")) as z:
        return {k: z[k] for k in z.files}


def probes_for(expected, case, norm):
    pre = "%s/%s/" % (case, norm)
    names = sorted({k[len(pre):].rsplit("/", 1)[0] for k in expected if k.startswith(pre)})
    return {n: {f: expected[pre + n + "/" + f] for f in ("l2", "sum", "idx", "val")} for n in names}


@pytest.mark.parametrize("case", ["toy_basis_L1", "toy_block_L2", "toy_block_small"])
@pytest.mark.parametrize("norm", ["intended", "tf_as_executed"])
def test_oracle_reproduces_golden_small(expected, case, norm):
    c = helpers.golden_inputs(case)
    acts, grads = oracle.encoder_step(c["params"], c["triples"], c["V"], c["L"], c["kind"], c["dcodes"],
                                      keep_prob=0.8, dropout_masks=c["masks"], norm_mode=norm)
    pr = probes_for(expected, case, norm)
    helpers.check_probe(acts[-1], pr["codes"], name="codes")
    helpers.check_probe(acts[0], pr["H0"], name="H0")
    for k, v in grads.items():
        helpers.check_probe(v, pr["grad_" + k], name=k)


def test_oracle_reproduces_golden_fb237_block(expected):
    c = helpers.golden_inputs("fb237_block_L2")
    acts, grads = oracle.encoder_step(c["params"], c["triples"], c["V"], c["L"], c["kind"], c["dcodes"],
                                      keep_prob=0.8, dropout_masks=c["masks"])
    pr = probes_for(expected, "fb237_block_L2", "intended")
    helpers.check_probe(acts[-1], pr["codes"], name="codes")
    for k, v in grads.items():
        helpers.check_probe(v, pr["grad_" + k], name=k)


def test_graph_fixtures_are_the_reference_data():
    toy = helpers.load_graph("toy_train")
    assert toy.shape == (43, 3) and toy[:, [0, 2]].max() == 15 and toy[:, 1].max() == 8
    mb = helpers.load_graph("fb237_minibatch")
    assert mb.shape == (15000, 3) and mb[:, [0, 2]].max() < 14541 and mb[:, 1].max() < 237
    assert len(np.unique(mb, axis=0)) == 15000          # sampled without replacement
    full = helpers.load_graph("fb237_valid_test")
    assert full.shape == (38001, 3)
    # top relation counts of FB15k-237 valid+test
    counts = np.sort(np.bincount(full[:, 1], minlength=237))[::-1]
    assert list(counts[:3]) == [2675, 2437, 1894]
