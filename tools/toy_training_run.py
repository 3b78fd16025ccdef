#!/usr/bin/env python
"""A complete small training run through the reference-shaped driver: a synthetic relational world with learnable
cluster structure, written in the reference's file formats, gcn_block-style settings, validation MRR every
250 iterations.  Evidence that the whole loop (sampler, device negatives, fused train step, device ranking,
early stopping, checkpoints) learns.   python tools/toy_training_run.py [iterations]"""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import test_gpu_driver as t  # noqa: E402
from relationprediction_amd import train  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
tmp = tempfile.mkdtemp()
data = os.path.join(tmp, "data")


def write_cluster_world(root, V=400, C=20, R=6, n_train=6000, seed=1):
    """Entities fall into C clusters; relation r sends cluster i to cluster perm_r(i); a triple is (s, r, any o of the
    image cluster).  Unseen (s, r, o) are predictable from the cluster structure: a model that learns it ranks the
    ~V/C members of the right cluster first (filtered MRR about H(V/C)/(V/C) = 0.18 here against 0.015 at chance)."""
    rng = np.random.RandomState(seed)
    cluster = np.arange(V) % C
    members = [np.flatnonzero(cluster == c) for c in range(C)]
    perms = [rng.permutation(C) for _ in range(R)]
    triples = set()
    while len(triples) < n_train + 400:
        s, r = rng.randint(V), rng.randint(R)
        triples.add((s, r, int(rng.choice(members[perms[r][cluster[s]]]))))
    triples = np.array(sorted(triples))
    rng.shuffle(triples)
    os.makedirs(root, exist_ok=True)
    with open(os.path.join(root, "entities.dict"), "w") as f:
        f.writelines("%d\te%d\n" % (i, i) for i in range(V))
    with open(os.path.join(root, "relations.dict"), "w") as f:
        f.writelines("%d\tr%d\n" % (i, i) for i in range(R))
    for name, part in (("train", triples[:n_train]), ("valid", triples[n_train:n_train + 200]),
                       ("test", triples[n_train + 200:n_train + 400])):
        with open(os.path.join(root, name + ".txt"), "w") as f:
            f.writelines("e%d\tr%d\te%d\n" % (s, r, o) for s, r, o in part)


write_cluster_world(data)
os.makedirs(os.path.join(tmp, "models"))
settings = os.path.join(tmp, "toy.exp")
with open(settings, "w") as f:
    f.write((t.SETTINGS % dict(nb=4, concat="Yes", exp=os.path.join(tmp, "models", "Toy")))
            .replace("CheckEvery=20", "CheckEvery=250").replace("ReportTrainLossEvery=10", "ReportTrainLossEvery=250")
            .replace("GraphBatchSize=300", "GraphBatchSize=2000").replace("BurninPhaseDuration=1000", "BurninPhaseDuration=100000"))
np.random.seed(0)
t0 = time.time()
model, n = train.main(["--settings", settings, "--dataset", data, "--max-iterations", str(iters)])
print("%d iterations in %.1f s (%.2f ms / iteration including %d validation + test evaluations)"
      % (n, time.time() - t0, (time.time() - t0) * 1e3 / n, n // 250))
