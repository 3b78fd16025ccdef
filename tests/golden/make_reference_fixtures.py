"""Golden vectors produced by the REFERENCE'S OWN CODE (run in the build container, where /root/reference exists):

    python -B tests/golden/make_reference_fixtures.py        ->  tests/golden/reference_host_logic.json

The encoder arithmetic lives in TensorFlow 1.4 and cannot be run here ("parity unpinned"), but
the reference's host logic around it is plain Python + numpy and imports as it stands: settings_reader,
optimizer_parameter_parser, the Converge stack (optimization/abstract.py + shared/algorithms.py + the fit loop and
the stack constructor of optimization/optimize.py), NegativeSampler, the MRR Scorer, and (extracted from train.py's source) sample_edge_neighborhood.  This script drives those
modules on seeded inputs and records what they return / print; tests/test_reference_fixtures.py replays the same
inputs through this repository's counterparts and demands identical results.  Only theano and tensorflow are
stubbed (empty modules: the code paths exercised never touch them).  Nothing is copied from the reference: the
fixture holds inputs written here and the outputs the reference computed for them.

The reference tree is read-only: bytecode writing is switched off before anything is imported from it.
"""
import contextlib
import importlib.util
import io
import json
import os
import sys
import types

sys.dont_write_bytecode = True

import numpy as np  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/code"

# settings written for this fixture (the reference's format; small periods so that a short script exercises
# reporting, burn-in, early stopping and saving); tests/test_reference_fixtures.py parses the same text
SETTINGS_TEXT = """[Optimizer]
\tMaxGradientNorm=1
\tReportTrainLossEvery=3
%s
\t[EarlyStopping]
\t\tCheckEvery=4
\t\tBurninPhaseDuration=8

\t[Algorithm]
\t\tName=Adam
\t\tlearning_rate=0.01

[General]
\tNegativeSampleRate=2
\tGraphSplitSize=0.5
\tExperimentName=models/X
"""

SCENARIOS = {
    # name: (extra Optimizer lines, scripted train losses, scripted validation scores)
    "early_stop_after_burnin": ("", [1.0 / (i + 1) for i in range(40)], [0.10, 0.20, 0.15, 0.30, 0.25, 0.40]),
    "max_iterations": ("\tMaxIterations=10", [0.5 + 0.01 * i for i in range(40)], [0.1, 0.2, 0.3, 0.4]),
    "improving_forever": ("\tMaxIterations=22", [2.0 - 0.05 * i for i in range(40)], [0.1, 0.2, 0.3, 0.4, 0.5, 0.6]),
    "stop_ignored_in_burnin_then_taken": ("", [0.3] * 40, [0.5, 0.4, 0.45, 0.2, 0.1]),
}


def load(name, path, **preset):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    for k, v in preset.items():
        setattr(mod, k, v)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def reference_modules():
    for stub in ("theano", "theano.tensor", "tensorflow"):
        sys.modules.setdefault(stub, types.ModuleType(stub))
    sys.modules["theano"].tensor = sys.modules["theano.tensor"]
    m = types.SimpleNamespace()
    m.settings_reader = load("ref_settings_reader", REF + "/common/settings_reader.py")
    m.parser = load("ref_optimizer_parameter_parser", REF + "/common/optimizer_parameter_parser.py")
    m.aux = load("ref_auxilliaries", REF + "/common/auxilliaries.py")
    m.evaluation = load("ref_evaluation", REF + "/common/evaluation.py")
    # the optimizer package imports its siblings by bare name (Python-2 style implicit relative imports)
    m.abstract = load("abstract", REF + "/optimization/abstract.py")
    shared = types.ModuleType("shared")
    sys.modules["shared"] = shared
    m.shared = load("shared.algorithms", REF + "/optimization/shared/algorithms.py")
    shared.algorithms = m.shared
    for backend in ("tensorflow_backend", "theano_backend"):
        pkg = types.ModuleType(backend)
        sys.modules[backend] = pkg
        pkg.algorithms = load(backend + ".algorithms", REF + "/optimization/%s/algorithms.py" % backend)
    m.optimize = load("ref_optimize", REF + "/optimization/optimize.py")
    return m


def jsonable(x):
    if isinstance(x, dict):
        return {str(k): jsonable(v) for k, v in x.items()}
    if isinstance(x, (list, tuple, zip)):
        return [jsonable(v) for v in x]
    if isinstance(x, np.ndarray):
        return x.tolist()
    if isinstance(x, (np.integer,)):
        return int(x)
    if isinstance(x, (np.floating,)):
        return float(x)
    if callable(x):
        return "<callable>"
    return x


def settings_for(m, extra, tmpdir):
    path = os.path.join(tmpdir, "fixture.exp")
    with open(path, "w") as f:
        f.write(SETTINGS_TEXT % extra)
    s = m.settings_reader.read(path)
    s["Optimizer"].merge(s["General"])
    return s


def run_stack_scenario(m, extra, losses, scores, tmpdir):
    """The reference's parser -> its stack constructor (tensorflow backend: GradientClipping and Adam are built, their
    TF methods are never called) -> its fit loop, with update_from_batch returning the scripted losses."""
    s = settings_for(m, extra, tmpdir)
    opp = m.parser.Parser(s["Optimizer"])
    events = {"saves": [], "transforms": 0, "scored": 0}
    score_iter = iter(scores)

    def save(path):
        events["saves"].append([path, state["i"]])

    def score(validation_data):
        events["scored"] += 1
        return next(score_iter)

    def transform(x):
        events["transforms"] += 1
        return x

    opp.set_save_function(save)
    opp.set_early_stopping_score_function(score)
    opp.set_sample_transform_function(transform)
    parametrization = opp.get_parametrization()
    construct = getattr(m.optimize, "__construct_optimizer")
    state = {"i": 0}

    out = io.StringIO()
    with contextlib.redirect_stdout(out):
        optimizer = construct(parametrization, backend="tensorflow")

        def update_from_batch(processed):
            state["i"] += 1
            return losses[state["i"] - 1]

        optimizer.initialize_for_fitting = lambda: None
        optimizer.update_from_batch = update_from_batch
        optimizer.fit([0, 1, 2], validation_data=[3, 4])
    return {"extra": extra, "losses": losses, "scores": scores,
            "parametrization": jsonable(parametrization), "stdout": out.getvalue(),
            "iterations": state["i"], "saves": events["saves"], "transforms": events["transforms"],
            "validations": events["scored"]}


def negative_sampler_cases(m):
    cases = []
    for seed, n, rate, V in ((11, 3, 4, 50), (5, 40, 10, 14541), (0, 1, 1, 2), (9, 17, 3, 5)):
        rng = np.random.RandomState(100 + seed)
        t = np.stack([rng.randint(0, V, n), rng.randint(0, 7, n), rng.randint(0, V, n)], 1).astype(np.int64)
        ns = m.aux.NegativeSampler(rate, V)
        np.random.seed(seed)
        idx, lab = ns.transform(t)
        cases.append({"seed": seed, "rate": rate, "entities": V, "triples": t.tolist(),
                      "indexes": idx.tolist(), "labels": lab.tolist(),
                      "index_dtype": str(idx.dtype), "label_dtype": str(lab.dtype)})
    return cases


class ScoreTableModel(object):
    """What the reference's Scorer asks of a model (model.py:59-81), from fixed tables: sigmoid DistMult scores of
    every entity as subject / object of each triple, fp32."""

    def __init__(self, codes, rel):
        self.codes, self.rel = codes, rel

    @staticmethod
    def _sig(x):
        return (1.0 / (1.0 + np.exp(-x.astype(np.float64)))).astype(np.float32)

    def score_all_subjects(self, triples):
        return [self._sig((self.codes @ (self.rel[r] * self.codes[o])).astype(np.float32)) for s, r, o in triples]

    def score_all_objects(self, triples):
        return [self._sig((self.codes @ (self.codes[s] * self.rel[r])).astype(np.float32)) for s, r, o in triples]


    def score(self, triples):
        """model.score (model.py:59-62): the decoder's prediction for the listed triples"""
        t = np.asarray(triples)
        return self._sig(np.sum(self.codes[t[:, 0]] * self.rel[t[:, 1]] * self.codes[t[:, 2]], axis=1).astype(np.float32))


def accuracy_cases(m):
    """Evaluation.Metric = Accuracy (evaluation.py:178-209, 311-331): the *_accuracy.txt files hold (positive, negative)
    pairs on consecutive lines; the score is the share of pairs whose positive outscores its negative."""
    cases = []
    for seed, V, R, d, pairs, scale in ((11, 30, 3, 6, 40, 1.0), (12, 20, 2, 4, 7, 1.0), (13, 25, 3, 5, 60, 40.0)):
        rng = np.random.RandomState(seed)
        triples = np.stack([rng.randint(0, V, 2 * pairs), rng.randint(0, R, 2 * pairs), rng.randint(0, V, 2 * pairs)], 1)
        if seed == 12:
            triples[5] = triples[4]           # a pair with equal scores: `>` counts it as wrong
        table = np.random.RandomState(seed + 50)
        codes = (table.randn(V, d) * scale).astype(np.float32)      # scale 40: saturated sigmoids, exact ties
        rel = table.randn(R, d).astype(np.float32)
        m.evaluation.AccuracySummary.results = {'Filtered': {}, 'Raw': {}}    # (a class attribute there: start clean)
        scorer = m.evaluation.Scorer({"Metric": "Accuracy"})
        scorer.register_model(ScoreTableModel(codes, rel))
        score = scorer.compute_scores(triples, verbose=False)
        summary = score.get_summary()
        out = io.StringIO()
        with contextlib.redirect_stdout(out):
            summary.pretty_print()
        cases.append({"seed": seed, "entities": V, "relations": R, "dim": d, "pairs": pairs, "scale": scale,
                      "triples": jsonable(triples), "predictions": [bool(x) for x in score.predictions],
                      "results": jsonable(summary.results), "accuracy_string": summary.accuracy_string(),
                      "pretty_print": out.getvalue()})
    return cases


def scorer_cases(m):
    cases = []
    for seed, V, R, d, sizes, scale in ((2, 40, 4, 8, (300, 60, 50), 1.0), (7, 25, 3, 6, (120, 30, 1100), 1.0),
                                        (3, 30, 3, 6, (150, 20, 40), 30.0)):
        rng = np.random.RandomState(seed)
        mk = lambda n: np.stack([rng.randint(0, V, n), rng.randint(0, R, n), rng.randint(0, V, n)], 1)  # noqa: E731
        train, valid, test = mk(sizes[0]), mk(sizes[1]), mk(sizes[2])
        table = np.random.RandomState(seed + 50)
        codes = (table.randn(V, d) * scale).astype(np.float32)      # scale 30: saturated sigmoids, many exact ties
        rel = table.randn(R, d).astype(np.float32)
        scorer = m.evaluation.Scorer({"Metric": "MRR"})
        for part in (train, valid, test):
            scorer.register_data(part)
        scorer.register_degrees(train)
        scorer.register_model(ScoreTableModel(codes, rel))
        scorer.finalize_frequency_computation(np.concatenate((train, valid, test), axis=0))
        score = scorer.compute_scores(test, verbose=False)
        summary = score.get_summary()
        out = io.StringIO()
        with contextlib.redirect_stdout(out):
            summary.pretty_print()
        results = jsonable(summary.results)
        if len(test) > 60:          # keep the fixture small: per-triple degree / frequency tables only for the small cases
            results = {k: {kk: vv for kk, vv in v.items() if kk not in ("Degree", "Frequency")}
                       for k, v in results.items()}
        cases.append({"seed": seed, "entities": V, "relations": R, "dim": d, "sizes": list(sizes), "scale": scale,
                      "raw_ranks": jsonable(score.raw_ranks), "filtered_ranks": jsonable(score.filtered_ranks),
                      "results": results, "mrr_string": summary.mrr_string(),
                      "pretty_print": out.getvalue()})
    return cases


def neighborhood_sampler_cases():
    """sample_edge_neighborhood lives in the reference's train.py, a script whose top level needs TensorFlow: the
    function definition and the module-level statements that build its globals (`adj_list`, `degrees`,
    train.py:133-139) are pulled out of the parsed source and executed on their own -- the reference's code, run
    as it stands, on numpy's global stream."""
    import ast
    tree = ast.parse(open(REF + "/train.py").read())
    wanted = []
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name == "sample_edge_neighborhood":
            wanted.append(node)
        elif isinstance(node, ast.Assign):
            names = {n.id for n in ast.walk(node) if isinstance(n, ast.Name) and isinstance(n.ctx, ast.Store)}
            if names & {"adj_list", "degrees"}:
                wanted.append(node)
        elif isinstance(node, ast.For):       # the loop that fills adj_list (it only calls .append on it)
            if any(isinstance(n, ast.Name) and n.id == "adj_list" for n in ast.walk(node)):
                wanted.append(node)
    assert [type(n).__name__ for n in wanted] == ["Assign", "For", "Assign", "Assign", "FunctionDef"], wanted
    code = compile(ast.Module(body=wanted, type_ignores=[]), REF + "/train.py", "exec")
    cases = []
    for seed, V, n, size in ((1, 12, 30, 30), (2, 40, 120, 60), (3, 25, 60, 10), (4, 300, 500, 200)):
        rng = np.random.RandomState(seed)
        t = np.unique(np.stack([rng.randint(0, V, n), rng.randint(0, 5, n), rng.randint(0, V, n)], 1), axis=0)
        ns = {"np": np, "entities": list(range(V)), "train_triplets": t}
        exec(code, ns)
        np.random.seed(1000 + seed)
        ids = ns["sample_edge_neighborhood"](t, min(size, len(t)))
        cases.append({"seed": seed, "entities": V, "triples": t.tolist(), "sample_size": int(min(size, len(t))),
                      "numpy_seed": 1000 + seed, "edge_ids": [int(i) for i in ids]})
    return cases


def io_cases(m_io, tmpdir):
    """The reference's reader (common/io.py) on the shipped data files -- on COPIES: it opens files 'r+', and the
    reference tree is not to be opened for writing.  Recorded: shapes and SHA-256 of the int32 triple arrays."""
    import hashlib
    import shutil
    data = "/root/reference/data"
    out = {}
    for dataset, parts in (("Toy", ("train", "valid", "test")), ("FB-Toutanova", ("valid", "test")),
                           ("wn18", ("valid", "test")), ("FB15k", ("valid",))):
        dst = os.path.join(tmpdir, dataset)
        os.makedirs(dst, exist_ok=True)
        for f in ["entities.dict", "relations.dict"] + [p_ + ".txt" for p_ in parts]:
            shutil.copyfile(os.path.join(data, dataset, f), os.path.join(dst, f))
        ent = m_io.read_dictionary(os.path.join(dst, "entities.dict"))
        rel = m_io.read_dictionary(os.path.join(dst, "relations.dict"))
        rec = {"entities": len(ent), "relations": len(rel), "parts": {}}
        for p_ in parts:
            t = np.array(m_io.read_triplets_as_list(os.path.join(dst, p_ + ".txt"), os.path.join(dst, "entities.dict"),
                                                    os.path.join(dst, "relations.dict")), dtype=np.int32)
            rec["parts"][p_] = {"shape": list(t.shape), "sha256": hashlib.sha256(np.ascontiguousarray(t).tobytes()).hexdigest()}
        out[dataset] = rec
    return out


def main():
    import tempfile
    m = reference_modules()
    tmp = tempfile.mkdtemp()
    fixture = {"settings_text": SETTINGS_TEXT, "stack": {}, "negative_sampler": negative_sampler_cases(m),
               "scorer": scorer_cases(m), "accuracy": accuracy_cases(m), "neighborhood_sampler": neighborhood_sampler_cases()}
    for name, (extra, losses, scores) in SCENARIOS.items():
        fixture["stack"][name] = run_stack_scenario(m, extra, losses, scores, tmp)
    fixture["io"] = io_cases(load("ref_io", REF + "/common/io.py"), tmp)
    # the reference's own settings file, as its reader parses it (values only; the file itself is not copied)
    shipped = m.settings_reader.read("/root/reference/settings/gcn_block.exp")
    import ast
    fixture["gcn_block_exp_parsed"] = ast.literal_eval(str(shipped))      # Settings.__str__ is the dict's repr
    with open(os.path.join(HERE, "reference_host_logic.json"), "w") as f:
        json.dump(fixture, f, sort_keys=True, separators=(",", ":"))
    print("wrote reference_host_logic.json:", {k: (len(v) if hasattr(v, "__len__") else v) for k, v in fixture.items()})


if __name__ == "__main__":
    main()
