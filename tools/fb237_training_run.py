#!/usr/bin/env python
"""A training run of the reference-shaped driver at the reference's own FB15k-237 gcn_block configuration
(settings/gcn_block.exp: d = 500, 100 blocks of 5x5, 2 layers, GraphBatchSize 30000, GraphSplitSize 0.5,
NegativeSampleRate 10, Adam 0.01, clip 1, early stopping every 2000 after 6000) on REAL FB15k-237 graph structure:
the 38,001 valid+test triples committed as a fixture (tests/golden/graphs.npz), over the real 14,541-entity /
237-relation space, re-split 34,001 / 2,000 / 2,000 (seed 0).  The train split of FB15k-237 is not shipped with the
reference, so the absolute MRR is not comparable with the paper's; the run shows the whole loop (neighbourhood
sampler, device negatives, fused train step, device ranking, early stopping, checkpoints) working and learning at
full size, and what one iteration costs under the driver.

    python tools/fb237_training_run.py [iterations] [driver flags, e.g. --host-sampler]        (default 4000)
"""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from relationprediction_amd import train  # noqa: E402

SETTINGS = """[Encoder]
	Name=gcn_basis
	DropoutKeepProbability=0.8
	InternalEncoderDimension=500
	NumberOfBasisFunctions=100
	NumberOfLayers=2
	UseInputTransform=Yes
	UseOutputTransform=No
	AddDiagonal=No
	DiagonalCoefficients=No
	SkipConnections=None
	StoreEdgeData=No
	RandomInput=No
	PartiallyRandomInput=No
	Concatenation=Yes

[Decoder]
	Name=bilinear-diag
	RegularizationParameter=0.01

[Shared]
	CodeDimension=500

[Optimizer]
	MaxGradientNorm=1
	ReportTrainLossEvery=100

	[EarlyStopping]
		CheckEvery=2000
		BurninPhaseDuration=6000

	[Algorithm]
		Name=Adam
		learning_rate=0.01

[General]
	NegativeSampleRate=10
	GraphSplitSize=0.5
	ExperimentName=%s
	GraphBatchSize=30000

[Evaluation]
	Metric=MRR
"""


def write_dataset(root):
    with np.load(os.path.join(ROOT, "tests", "golden", "graphs.npz")) as z:
        triples = z["fb237_valid_test"].astype(np.int64)
    V, R = 14541, 237
    triples = triples[np.random.RandomState(0).permutation(len(triples))]
    os.makedirs(root, exist_ok=True)
    with open(os.path.join(root, "entities.dict"), "w") as f:
        f.writelines("%d\te%d\n" % (i, i) for i in range(V))
    with open(os.path.join(root, "relations.dict"), "w") as f:
        f.writelines("%d\tr%d\n" % (i, i) for i in range(R))
    for name, part in (("train", triples[:34001]), ("valid", triples[34001:36001]), ("test", triples[36001:])):
        with open(os.path.join(root, name + ".txt"), "w") as f:
            f.writelines("e%d\tr%d\te%d\n" % (s, r, o) for s, r, o in part)


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
    tmp = tempfile.mkdtemp()
    data = os.path.join(tmp, "data")
    write_dataset(data)
    os.makedirs(os.path.join(tmp, "models"))
    settings = os.path.join(tmp, "gcn_block.exp")
    with open(settings, "w") as f:
        f.write(SETTINGS % os.path.join(tmp, "models", "GcnBlock"))
    np.random.seed(0)
    t0 = time.time()
    model, n = train.main(["--settings", settings, "--dataset", data, "--max-iterations", str(iters)] + sys.argv[2:])
    wall = time.time() - t0
    print("%d iterations in %.1f s = %.2f ms / iteration, everything included (dataset load, engine creation, "
          "%d validation + test evaluations of 2,000 triples each, both sides, filtered)"
          % (n, wall, wall * 1e3 / n, n // 2000))


if __name__ == "__main__":
    main()
