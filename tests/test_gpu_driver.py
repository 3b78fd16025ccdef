"""End to end through the reference-shaped driver (relationprediction_amd/train.py) on a small synthetic
dataset written in the reference's file formats: the loss falls, validation MRR is computed on the device, the
checkpoint is written in get_weights() order, and the chain's eager numpy surface agrees with the fused paths."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SETTINGS = """[Encoder]
	Name=gcn_basis
	DropoutKeepProbability=0.8
	InternalEncoderDimension=20
	NumberOfBasisFunctions=%(nb)d
	NumberOfLayers=2
	UseInputTransform=Yes
	UseOutputTransform=No
	AddDiagonal=No
	DiagonalCoefficients=No
	SkipConnections=None
	StoreEdgeData=No
	RandomInput=No
	PartiallyRandomInput=No
	Concatenation=%(concat)s

[Decoder]
	Name=bilinear-diag
	RegularizationParameter=0.01

[Shared]
	CodeDimension=20

[Optimizer]
	MaxGradientNorm=1
	ReportTrainLossEvery=10

	[EarlyStopping]
		CheckEvery=20
		BurninPhaseDuration=1000

	[Algorithm]
		Name=Adam
		learning_rate=0.01

[General]
	NegativeSampleRate=5
	GraphSplitSize=0.5
	ExperimentName=%(exp)s
	GraphBatchSize=300

[Evaluation]
	Metric=MRR
"""


def write_dataset(root, V=200, R=6, n_train=700, seed=0):
    """A learnable toy world: relation r maps entity e to (a_r * e + b_r) mod V."""
    rng = np.random.RandomState(seed)
    a, b = [1, 3, 7, 11, 13, 17][:R], rng.randint(0, V, R)
    assert n_train + 120 <= V * R                 # the world only holds V*R distinct triples
    triples = set()
    while len(triples) < n_train + 120:
        s, r = rng.randint(V), rng.randint(R)
        triples.add((s, r, (a[r] * s + b[r]) % V))
    triples = np.array(sorted(triples))
    rng.shuffle(triples)
    os.makedirs(root, exist_ok=True)
    with open(os.path.join(root, "entities.dict"), "w") as f:
        f.writelines("%d\te%d\n" % (i, i) for i in range(V))
    with open(os.path.join(root, "relations.dict"), "w") as f:
        f.writelines("%d\tr%d\n" % (i, i) for i in range(R))
    for name, part in (("train", triples[:n_train]), ("valid", triples[n_train:n_train + 60]),
                       ("test", triples[n_train + 60:n_train + 120])):
        with open(os.path.join(root, name + ".txt"), "w") as f:
            f.writelines("e%d\tr%d\te%d\n" % (s, r, o) for s, r, o in part)
    return triples[:n_train]


# default: all three random draws of an iteration on the device (neighbourhood sampler, edge dropout, negatives);
# --host-sampler: the graph batch from the host port of sample_edge_neighborhood, built ahead by worker threads
@pytest.mark.parametrize("concat,nb,extra", [("Yes", 4, []), ("No", 2, []), ("Yes", 4, ["--host-negatives", "--batch-workers", "0"]),
                                             ("Yes", 4, ["--host-sampler"])])
def test_train_driver_end_to_end(tmp_path, capsys, concat, nb, extra):
    from relationprediction_amd import train
    data = str(tmp_path / "data")
    train_triples = write_dataset(data)
    os.makedirs(str(tmp_path / "models"))
    exp = str(tmp_path / "models" / "Toy")
    settings = tmp_path / "toy.exp"
    settings.write_text(SETTINGS % dict(nb=nb, concat=concat, exp=exp))
    np.random.seed(0)
    model, iterations = train.main(["--settings", str(settings), "--dataset", data, "--max-iterations", "60"] + extra)
    out = capsys.readouterr().out
    assert iterations == 60
    losses = [float(l.split(": ")[-1]) for l in out.splitlines() if l.startswith("Average train loss")]
    initial = float([l for l in out.splitlines() if l.startswith("Initial loss")][0].split(": ")[1])
    assert len(losses) == 5 and losses[-1] < 0.8 * initial, (initial, losses)
    checks = [l for l in out.splitlines() if l.startswith("Tested validation score")]
    assert len(checks) == 3 and all(0.0 < float(c.split("Result: ")[1]) <= 1.0 for c in checks)
    assert "MRR" in out and "H@10" in out                            # test-set summary printed at every check
    assert os.path.exists(exp + "-0.npz") and os.path.exists(exp + "-2.npz")
    with np.load(exp + "-2.npz") as z:
        names = sorted(z.files)
        assert names[0].endswith("W_emb") and names[-1].endswith("W_relation")
        assert z[names[-1]].shape == (200, 20)                        # [EntityCount, d]

    # the eager numpy surface of the chain sees the weights the device optimizer trained
    w = {v.name: v.value() for v in model.get_weights()}
    assert np.isfinite(w["W_relation"]).all() and np.abs(w["W_relation"][:6]).mean() > 0.1
    gvar, xvar = model.get_test_input_variables()
    queries = train_triples[:7]
    gvar.feed(train_triples)
    xvar.feed(queries)
    all_obj = model.predict_all_object_scores()                      # [7, V] sigmoid scores, host numpy
    assert all_obj.shape == (7, 200)
    ptr = np.arange(8, dtype=np.int64)
    raw, filt = model.device_ranks(train_triples, queries, True, ptr, queries[:, 2].astype(np.int32))
    host_raw = np.array([(row >= row[o]).sum() for row, o in zip(all_obj, queries[:, 2])])
    assert np.abs(raw - host_raw).max() <= 1 and (filt == raw).all() # filter = gold only -> filtered == raw


def test_train_driver_with_the_accuracy_metric(tmp_path, capsys):
    """Evaluation.Metric = Accuracy end to end (code/train.py:32-35,116-121; evaluation.py:311-326): validation / test
    data are the *_accuracy.txt pair files, the early-stopping score is the share of pairs whose positive outscores its
    negative (model.score: encoder forward on the device, DistMult on the codes), and a trained model beats a coin."""
    from relationprediction_amd import train
    data = str(tmp_path / "data")
    train_triples = write_dataset(data)
    rng = np.random.RandomState(5)
    known = {tuple(t) for t in train_triples.tolist()}
    for name, lo in (("valid_accuracy", 0), ("test_accuracy", 100)):
        with open(os.path.join(data, name + ".txt"), "w") as f:
            for s, r, o in train_triples[lo:lo + 100]:                  # (positive, corrupted object) on consecutive lines
                bad = int(rng.randint(200))
                while (s, r, bad) in known:
                    bad = int(rng.randint(200))
                f.write("e%d\tr%d\te%d\ne%d\tr%d\te%d\n" % (s, r, o, s, r, bad))
    os.makedirs(str(tmp_path / "models"))
    settings = tmp_path / "toy.exp"
    settings.write_text((SETTINGS % dict(nb=4, concat="Yes", exp=str(tmp_path / "models" / "Toy")))
                        .replace("Metric=MRR", "Metric=Accuracy"))
    np.random.seed(0)
    model, iterations = train.main(["--settings", str(settings), "--dataset", data, "--max-iterations", "60"])
    out = capsys.readouterr().out
    assert iterations == 60
    checks = [float(l.split("Result: ")[1]) for l in out.splitlines() if l.startswith("Tested validation score")]
    # positives seen in training against random objects: a coin at the start, better with every check (0.48 / 0.52 / 0.60)
    assert len(checks) == 3 and checks[-1] >= 0.55 and checks[-1] > checks[0], checks
    printed = [l for l in out.splitlines() if l.startswith("Accuracy\t")]
    assert len(printed) == 3 and "MRR" not in out                       # the test-set summary of every check
    valid = np.array(train.load_dataset(data, "Accuracy")[0]["valid"])
    scores = model.score(valid)
    assert scores.shape == (200,) and abs(checks[-1] - np.mean(scores[::2] > scores[1::2])) <= 0.05   # (<= one step apart)


def test_lookahead_draw_leaves_the_trajectory_unchanged(tmp_path, capsys, monkeypatch):
    """A device-sampled batch is a function of its seed: drawing it one iteration early (HipOptimizer.presample, three
    batch buffers in turn) trains to the same weights, bit for bit, as drawing it when its graph is prepared."""
    from relationprediction_amd import train
    from relationprediction_amd.optimization import optimize
    data = str(tmp_path / "data")
    write_dataset(data)
    weights = []
    for k, lookahead in enumerate((True, False)):
        os.makedirs(str(tmp_path / ("models%d" % k)))
        exp = str(tmp_path / ("models%d" % k) / "Toy")
        settings = tmp_path / ("toy%d.exp" % k)
        settings.write_text(SETTINGS % dict(nb=4, concat="Yes", exp=exp))
        if not lookahead:
            monkeypatch.setattr(optimize.HipOptimizer, "presample", lambda self, batch: None)
        np.random.seed(0)
        model, iterations = train.main(["--settings", str(settings), "--dataset", data, "--max-iterations", "25"])
        assert iterations == 25
        weights.append({v.name: np.array(v.value()) for v in model.get_weights()})
    capsys.readouterr()
    assert weights[0].keys() == weights[1].keys()
    for name in weights[0]:
        assert np.array_equal(weights[0][name], weights[1][name]), name
