"""Host side of the training driver: sampler, negative sampling, scorer bookkeeping,
optimizer stack semantics.  No GPU: the device calls are replaced by numpy stand-ins."""
import numpy as np
import pytest

import oracle
from relationprediction_amd import _native
from relationprediction_amd.common import auxilliaries, evaluation, optimizer_parameter_parser, settings_reader
from relationprediction_amd.optimization import optimize


# ------------------------------------------------------------------ neighbourhood sampler
def small_graph(seed=0, V=14, E=40):
    rng = np.random.RandomState(seed)
    t = np.stack([rng.randint(0, V, E), rng.randint(0, 3, E), rng.randint(0, V, E)], 1).astype(np.int32)
    t[0, 2] = t[0, 0]                     # a self loop
    t[1] = t[2]                           # a duplicate edge
    return t, V


def test_sampler_invariants_and_determinism():
    t, V = small_graph()
    s = _native.NeighborhoodSampler(t, V)
    for size in (0, 1, 7, len(t)):
        ids = s.sample(size, seed=5)
        assert ids.dtype == np.int32 and len(ids) == size
        assert len(np.unique(ids)) == size and (ids >= 0).all() and (ids < len(t)).all()
        assert (s.sample(size, seed=5) == ids).all()
    assert sorted(s.sample(len(t), seed=9)) == list(range(len(t)))     # exhausts the graph, no crash
    with pytest.raises(_native.RgcnError):
        s.sample(len(t) + 1, seed=1)                                   # the reference's NaN crash
    with pytest.raises(_native.RgcnError):
        _native.NeighborhoodSampler(np.array([[0, 0, 99]], np.int32), 5)


def test_sampler_grows_a_connected_patch():
    """Apart from restarts (nothing touched has free edges), every picked edge touches a touched vertex."""
    t, V = small_graph(seed=3, V=30, E=60)
    ids = _native.NeighborhoodSampler(t, V).sample(25, seed=2)
    touched, restarts = set(), 0
    for e in ids:
        s, o = int(t[e, 0]), int(t[e, 2])
        if s not in touched and o not in touched:
            restarts += 1
        touched.update((s, o))
    assert restarts <= 4


def test_sampler_matches_the_reference_process_in_distribution():
    """Same random process as train.py:161-198 (oracle.sample_edge_neighborhood follows it line by line):
    per-edge inclusion frequencies and first-pick frequencies agree within sampling noise."""
    t, V = small_graph(seed=1)
    size, trials = 9, 6000
    s = _native.NeighborhoodSampler(t, V)
    inc_a, inc_b = np.zeros(len(t)), np.zeros(len(t))
    first_a, first_b = np.zeros(len(t)), np.zeros(len(t))
    rng = np.random.RandomState(7)
    for k in range(trials):
        a = s.sample(size, seed=1000 + k)
        b = oracle.sample_edge_neighborhood(t, V, size, rng)
        inc_a[a] += 1; inc_b[b] += 1
        first_a[a[0]] += 1; first_b[b[0]] += 1
    for fa, fb in ((inc_a, inc_b), (first_a, first_b)):
        pa, pb = fa / trials, fb / trials
        sigma = np.sqrt((pa * (1 - pa) + pb * (1 - pb)) / trials) + 1e-9
        assert np.abs(pa - pb).max() < 5 * sigma.max(), (np.abs(pa - pb).max(), sigma.max())
    # second pick given the first: compare the pair distribution on its most frequent first pick
    assert abs(inc_a.sum() - inc_b.sum()) < 1e-9


# ------------------------------------------------------------------ negative sampling
def test_negative_sampler_layout_and_streams():
    t = np.array([[1, 0, 2], [3, 1, 4], [5, 2, 6]], dtype=np.int64)
    ns = auxilliaries.NegativeSampler(4, 50)
    np.random.seed(11)
    idx, lab = ns.transform(t)
    assert idx.dtype == np.int32 and lab.dtype == np.float32 and idx.shape == (15, 3)
    assert (lab[:3] == 1).all() and (lab[3:] == 0).all() and (idx[:3] == t).all()
    # replay the reference's double loop on the same random streams (auxilliaries.py:17-31)
    np.random.seed(11)
    choices = np.random.binomial(1, 0.5, 12)
    values = np.random.randint(50, size=12)
    ref = np.tile(t, (5, 1)).astype(np.int32)
    for i in range(3):
        for j in range(4):
            k = i + j * 3
            ref[k + 3, 2 if choices[k] else 0] = values[k]
    assert (idx == ref).all()
    assert (idx[3:, 1] == np.tile(t[:, 1], 4)).all()                  # relations are never corrupted


def test_negative_sampler_exclusive_avoids_known_positives():
    t = np.array([[0, 0, 1], [0, 0, 2], [1, 0, 0]], dtype=np.int64)
    ns = auxilliaries.NegativeSampler(20, 4)
    ns.set_known_positives(t)
    np.random.seed(0)
    idx, lab = ns.transform_exclusive(t)
    known = {tuple(r) for r in t}
    assert not any(tuple(r) in known for r in idx[3:])


# ------------------------------------------------------------------ scorer
class FakeRankModel(object):
    """device_ranks stand-in: ranks from a fixed random score table through the oracle's definition."""

    def __init__(self, V, R, d, seed=0):
        rng = np.random.RandomState(seed)
        self.codes = rng.randn(V, d).astype(np.float32)
        self.rel = rng.randn(R, d).astype(np.float32)
        self.test_graph = None
        self.calls = []

    def device_ranks(self, graph, triplets, predict_object, ptr, idx):
        self.calls.append((len(triplets), predict_object))
        known = {}
        for i, (s, r, o) in enumerate(triplets):
            known[(s, r) if predict_object else (o, r)] = list(idx[ptr[i]:ptr[i + 1]])
        return oracle.distmult_ranks(self.codes, self.rel, triplets, predict_object, known)


def test_scorer_mrr_against_brute_force():
    V, R, d = 40, 4, 8
    rng = np.random.RandomState(2)
    mk = lambda n: np.stack([rng.randint(0, V, n), rng.randint(0, R, n), rng.randint(0, V, n)], 1)
    train, valid, test = mk(300), mk(60), mk(50)
    model = FakeRankModel(V, R, d)
    model.test_graph = train
    scorer = evaluation.Scorer({'Metric': 'MRR'})
    for part in (train, valid, test):
        scorer.register_data(part)
    scorer.register_degrees(train)
    scorer.register_model(model)
    scorer.finalize_frequency_computation(np.concatenate((train, valid, test)))
    scorer.chunk_size = 16                                             # several chunks
    summary = scorer.compute_scores(test).get_summary()
    # brute force in the reference's order: per chunk subjects then objects
    everything = np.concatenate((train, valid, test))
    raw, filt = [], []
    sig = lambda x: (1.0 / (1.0 + np.exp(-x.astype(np.float64)))).astype(np.float32)
    for c in range(0, len(test), 16):
        chunk = test[c:c + 16]
        for subject_side in (True, False):
            for s, r, o in chunk:
                if subject_side:
                    scores = sig((model.codes @ (model.rel[r] * model.codes[o])).astype(np.float32))
                    gold = s
                    known = {x[0] for x in everything if x[1] == r and x[2] == o}
                else:
                    scores = sig((model.codes @ (model.codes[s] * model.rel[r])).astype(np.float32))
                    gold = o
                    known = {x[2] for x in everything if x[0] == s and x[1] == r}
                n = int((scores >= scores[gold]).sum())
                raw.append(n)
                filt.append(n - int((scores[sorted(known)] >= scores[gold]).sum()) + 1)
    raw, filt = np.array(raw), np.array(filt)
    assert summary.results['Raw']['MRR'] == pytest.approx(np.mean(1.0 / raw))
    assert summary.results['Filtered']['MRR'] == pytest.approx(np.mean(1.0 / filt))
    for h in (1, 3, 10):
        assert summary.results['Filtered']['H@%d' % h] == pytest.approx(np.mean(filt <= h))
    assert model.calls[0] == (16, False) and model.calls[1] == (16, True)   # subjects first (predict_object=False)
    summary.pretty_print()
    with pytest.raises(NotImplementedError):
        evaluation.Scorer({'Metric': 'AUC'}).compute_scores(test)


# ------------------------------------------------------------------ optimizer stack
class FakeTrainModel(object):
    def __init__(self, losses):
        self.losses = list(losses)
        self.steps = []
        self.cfg = None
        self.saved = []

    def configure_device_optimizer(self, *a):
        self.cfg = a

    def device_train_step(self, graph, x, y, seed):
        self.steps.append((graph, x, y))

    def device_loss(self):
        return self.losses[len(self.steps) - 1]

    def save(self, path):
        self.saved.append((path, len(self.steps)))


def reference_settings(tmp_path, extra=""):
    p = tmp_path / "s.exp"
    p.write_text("[Optimizer]\n\tMaxGradientNorm=1\n\tReportTrainLossEvery=3\n%s\n\t[EarlyStopping]\n\t\tCheckEvery=4\n"
                 "\t\tBurninPhaseDuration=8\n\n\t[Algorithm]\n\t\tName=Adam\n\t\tlearning_rate=0.01\n\n"
                 "[General]\n\tExperimentName=models/X\n" % extra)
    s = settings_reader.read(str(p))
    s['Optimizer'].merge(s['General'])
    return s['Optimizer']


def test_stack_order_and_early_stopping_semantics(tmp_path, capsys):
    """Parser output and Converge behaviour: reporter prints the mean of the LAST n losses at iterations
    n+1, 2n+1, ...; the stopper fires at the first non-improving check AFTER the burn-in; the saver runs after both."""
    opp = optimizer_parameter_parser.Parser(reference_settings(tmp_path))
    model = FakeTrainModel(losses=[float(10 - i) for i in range(40)])
    scores = iter([0.1, 0.05, 0.2, 0.15, 0.3])                          # checks at 4, 8, 12, 16
    opp.set_early_stopping_score_function(lambda data: next(scores))
    opp.set_save_function(model.save)
    opp.set_sample_transform_function(lambda x: ("g", "x%d" % len(model.steps), "y"))
    names = [n for n, _ in opp.get_parametrization()]
    assert names == ['SampleTransformer', 'GradientClipping', 'Adam', 'TrainLossReporter', 'EarlyStopper', 'ModelSaver']
    opt = optimize.build_hip(model, opp.get_parametrization())
    assert model.cfg == (0.01, 0.9, 0.999, 1e-8, 1.0)
    iterations = opt.fit(training_data=[1, 2, 3], validation_data=[4])
    out = capsys.readouterr().out
    # check 8 (0.05 < 0.1) is still inside the burn-in; check 16 (0.15 < 0.2) stops
    assert iterations == 16 and "Ignoring criterion while in burn-in phase." in out
    assert out.count("Stopping criterion reached.") == 1
    assert "Initial loss: 10.0" in out
    assert "Average train loss for iteration 1-3: " + str((9.0 + 8.0 + 7.0) / 3) in out
    assert [s[1] for s in model.saved] == [4, 8, 12]                   # saver is outermost: not at the stop
    assert model.steps[3][1] == "x3"                                   # batch i is drawn before step i runs


def test_iteration_counter_and_unknown_component(tmp_path):
    opp = optimizer_parameter_parser.Parser(reference_settings(tmp_path, extra="\tMaxIterations=5"))
    model = FakeTrainModel(losses=[1.0] * 10)
    opp.set_early_stopping_score_function(lambda data: 1.0)
    opp.set_save_function(model.save)
    opp.set_sample_transform_function(lambda x: (0, 1, 2))
    assert optimize.build_hip(model, opp.get_parametrization()).fit([0]) == 5
    with pytest.raises(NotImplementedError):
        optimize.build_stack([('RmsProp', {'learning_rate': 0.1})])


def test_background_batch_producers_are_reproducible(tmp_path):
    """Batches built ahead by worker threads come out in request order and depend only on the seeds drawn in
    that order: two runs from the same numpy seed see the same batches; a graph-free encoder stub keeps it CPU-only."""
    from relationprediction_amd import train

    class NoGraphEncoder(object):
        def needs_graph(self):
            return False

    rng = np.random.RandomState(0)
    triples = np.stack([rng.randint(0, 30, 200), rng.randint(0, 3, 200), rng.randint(0, 30, 200)], 1).astype(np.int32)
    general = {'NegativeSampleRate': '2', 'EntityCount': 30}
    runs = []
    for workers in (3, 3, 0, 0):
        opp = optimizer_parameter_parser.Parser(reference_settings(tmp_path, extra="\tMaxIterations=12"))
        model = FakeTrainModel(losses=[1.0] * 40)
        opp.set_early_stopping_score_function(lambda data: 1.0)
        opp.set_save_function(model.save)
        t_func = train.make_transform(triples, general, NoGraphEncoder())
        opp.set_sample_transform_function(lambda x, f=t_func: (None,) + tuple(f(x)))
        wrapped = opp.sample_transform_function
        wrapped.seeded = lambda x, seed, f=t_func: (None,) + tuple(f.seeded(x, seed))
        np.random.seed(5)
        assert optimize.build_hip(model, opp.get_parametrization(), batch_workers=workers).fit(triples) == 12
        runs.append([(s[1].copy(), s[2].copy()) for s in model.steps])
    for a, b in ((runs[0], runs[1]), (runs[2], runs[3])):
        assert all((x[0] == y[0]).all() and (x[1] == y[1]).all() for x, y in zip(a, b))
    first = runs[0]
    assert first[0][0].shape == (600, 3) and (first[0][1][:200] == 1).all()
    assert any((first[0][0] != first[k][0]).any() for k in range(1, 12))      # batches differ from one another


def test_minibatch_transform_semantics():
    """SURVEY 9 H8 / H9 (reference train.py:227-245): the message graph is an EXACT-k random subset of the sampled
    batch (k = int(GraphSplitSize * batch), host side, without replacement); the decoder's positives are ALL
    batch triples, the dropped ones included; negatives follow in NegativeSampleRate blocks with label 0.
    Without GraphBatchSize the batch is the whole training set (H7: how the Toy configuration has to run)."""
    from relationprediction_amd import train

    class GraphEncoder(object):
        def needs_graph(self):
            return True

    rng = np.random.RandomState(1)
    V, R, n = 60, 4, 500
    triples = np.unique(np.stack([rng.randint(0, V, n), rng.randint(0, R, n), rng.randint(0, V, n)], 1), axis=0)
    triples = triples.astype(np.int32)
    as_set = {tuple(t) for t in triples}
    for batch_size in (120, None):
        general = {'NegativeSampleRate': '3', 'EntityCount': V, 'GraphSplitSize': '0.5'}
        if batch_size:
            general['GraphBatchSize'] = str(batch_size)
        t_func = train.make_transform(triples, general, GraphEncoder())
        graph_split, X, Y = t_func.seeded(triples, 7)
        nb = batch_size or len(triples)
        assert graph_split.shape == (int(0.5 * nb), 3)                       # exact-k
        split_set = {tuple(t) for t in graph_split}
        assert len(split_set) == len(graph_split) and split_set <= as_set    # without replacement, real edges
        assert X.shape == (nb * 4, 3) and Y.shape == (nb * 4,)
        positives = {tuple(t) for t in X[:nb]}
        assert (Y[:nb] == 1).all() and (Y[nb:] == 0).all()
        assert len(positives) == nb and split_set <= positives               # H9: dropped edges stay positives
        assert len(positives - split_set) == nb - len(graph_split)
        neg = X[nb:].reshape(3, nb, 3)
        for block in neg:                                                    # each block corrupts one end of each positive
            same_s, same_o = block[:, 0] == X[:nb, 0], block[:, 2] == X[:nb, 2]
            assert (block[:, 1] == X[:nb, 1]).all() and (same_s | same_o).all()
        again = t_func.seeded(triples, 7)
        assert np.array_equal(again[0], graph_split) and np.array_equal(again[1], X)
    # H7: more edges asked for than the graph has (Toy under the shipped gcn_block.exp) is refused at start-up,
    # by name, instead of dying in the sampler at the first batch
    with pytest.raises(ValueError, match="GraphBatchSize"):
        train.make_transform(triples, {'NegativeSampleRate': '3', 'EntityCount': V, 'GraphSplitSize': '0.5',
                                       'GraphBatchSize': str(len(triples) + 1)}, GraphEncoder())


def test_unsupported_metric_and_checkpoint_directory(tmp_path):
    """An Evaluation.Metric the reference does not have is refused when the dataset is loaded; 'Accuracy' switches the
    validation / test splits to the *_accuracy.txt files (code/train.py:32-35); Model.save creates the checkpoint's
    directory and load restores by stored position"""
    from relationprediction_amd import train
    with pytest.raises(NotImplementedError, match="MRR"):
        train.load_dataset(str(tmp_path), metric='AUC')
    (tmp_path / "entities.dict").write_text("0\ta\n1\tb\n2\tc\n")
    (tmp_path / "relations.dict").write_text("0\tr\n1\tq\n")
    (tmp_path / "train.txt").write_text("a\tr\tb\nb\tq\tc\n")
    (tmp_path / "valid.txt").write_text("a\tq\tc\n")
    (tmp_path / "test.txt").write_text("c\tq\ta\n")
    (tmp_path / "valid_accuracy.txt").write_text("a\tr\tb\na\tr\tc\n")
    (tmp_path / "test_accuracy.txt").write_text("b\tq\tc\nb\tq\ta\nc\tr\ta\nb\tr\ta\n")
    splits, entities, relations = train.load_dataset(str(tmp_path), metric='Accuracy')
    assert splits['train'].tolist() == [[0, 0, 1], [1, 1, 2]]
    assert splits['valid'].tolist() == [[0, 0, 1], [0, 0, 2]]                          # (positive, negative) pairs
    assert splits['test'].tolist() == [[1, 1, 2], [1, 1, 0], [2, 0, 0], [1, 0, 0]]
    assert train.load_dataset(str(tmp_path), metric='MRR')[0]['valid'].tolist() == [[0, 1, 2]]
    from relationprediction_amd.model import Model

    class W(object):
        def __init__(self, name, v):
            self.name, self.v = name, v

        def value(self):
            return self.v

        def assign(self, v):
            self.v = v

    class M(Model):
        def __init__(self, ws):
            self.ws, self.save_iter = ws, 0

        def get_weights(self):
            return self.ws
    ws = [W("w%d" % i, np.full(2, float(i), np.float32)) for i in range(120)]       # more than 100 tensors
    m = M(ws)
    m.save(str(tmp_path / "models" / "Run"))
    m2 = M([W("w%d" % i, np.zeros(2, np.float32)) for i in range(120)])
    m2.load(str(tmp_path / "models" / "Run-0.npz"))
    assert all(np.array_equal(a.v, b.v) for a, b in zip(m.ws, m2.ws))


def test_reference_layout_aliases_the_reference_module_names():
    """A driver written against the reference's top-level module names (it runs from inside code/) keeps them:
    `import relationprediction_amd.reference_layout` registers `model`, `common.*`, `encoders.*`, `decoders.*`,
    `extras.*`, `optimization.optimize` as this package's mirrors; what is not mirrored stays an ImportError; a name
    already imported from another tree is never replaced.  (In a subprocess: the aliases are process-wide.)"""
    import os
    import subprocess
    import sys as _sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import sys, types
sys.path.insert(0, %r)
import relationprediction_amd.reference_layout as rl
from common import settings_reader, io, model_builder, optimizer_parameter_parser, auxilliaries, evaluation
from encoders.message_gcns.gcn_basis import BasisGcn
from encoders.message_gcns.gcn_basis_concat import ConcatGcn
from encoders.relation_embedding import RelationEmbedding
from encoders.affine_transform import AffineTransform
from decoders.bilinear_diag import BilinearDiag
from extras.graph_representations import Representation
from model import Model
import relationprediction_amd.model, relationprediction_amd.common.model_builder as mine
assert Model is relationprediction_amd.model.Model and model_builder is mine
try:
    import encoders.bipartite_gcn          # a TensorFlow-only encoder of the reference: not mirrored
    raise SystemExit("unmirrored module imported")
except ImportError:
    pass
rl.uninstall()
assert "common" not in sys.modules and "model" not in sys.modules
sys.modules["model"] = types.ModuleType("model")      # somebody else's `model`
try:
    rl.install()
    raise SystemExit("install() replaced a foreign module")
except ImportError as e:
    assert "refusing" in str(e)
print("aliases ok")
''' % root
    r = subprocess.run([_sys.executable, "-B", "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "aliases ok" in r.stdout, (r.stdout, r.stderr)


def test_device_sampled_batches_are_drawn_one_iteration_ahead(tmp_path):
    """The loop's order of calls for device-sampled minibatches (optimize.HipOptimizer.fit): step i is enqueued, then
    batch i + 1 is staged (its graph prepared beside the step), then batch i + 2 is drawn (presample), then the loss of
    step i is read -- every batch staged once, drawn ahead at most once, stepped once, in request order; the last
    iterations stage / draw nothing that does not exist."""
    class Recorder(FakeTrainModel):
        def __init__(self, losses):
            FakeTrainModel.__init__(self, losses)
            self.log = []

        def device_train_step_minibatch(self, mb, seed):
            self.steps.append(mb)
            self.log.append(("step", mb.sample[2]))

        def device_stage(self, *a):
            raise AssertionError("minibatches are staged through device_stage_minibatch")

        def device_stage_minibatch(self, mb):
            self.log.append(("stage", mb.sample[2]))

        def device_presample_minibatch(self, mb):
            self.log.append(("draw", mb.sample[2]))

        def device_loss(self):
            self.log.append(("loss", self.steps[-1].sample[2]))
            return FakeTrainModel.device_loss(self)

    opp = optimizer_parameter_parser.Parser(reference_settings(tmp_path, extra="\tMaxIterations=5"))
    model = Recorder(losses=[1.0] * 10)
    opp.set_early_stopping_score_function(lambda data: 1.0)
    opp.set_save_function(model.save)
    counter = iter(range(100))
    train = object()
    opp.set_sample_transform_function(
        lambda x: optimize.DeviceMinibatch(None, 10, 0, 2, sample=(train, 20, next(counter))))
    assert optimize.build_hip(model, opp.get_parametrization()).fit([0]) == 5
    assert model.log == [("step", 0), ("stage", 1), ("draw", 2), ("loss", 0),
                         ("step", 1), ("stage", 2), ("draw", 3), ("loss", 1),
                         ("step", 2), ("stage", 3), ("draw", 4), ("loss", 2),
                         ("step", 3), ("stage", 4), ("loss", 3),
                         ("step", 4), ("loss", 4)]
