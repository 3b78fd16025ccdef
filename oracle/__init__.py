"""CPU oracle for the R-GCN encoder hot path.  TEST INFRASTRUCTURE ONLY.

Nothing under ``oracle/`` is part of the product.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it, and only as the checker / the timed CPU baseline -- never as a fallback
for the HIP path (``relationprediction_amd`` fails loudly without its HIP
library).

PARITY STATUS.  The reference (MichSchli/RelationPrediction) ships no tests, golden vectors or checkpoints, and
TensorFlow 1.4 is not installable here, so nothing can be checked against outputs of a TF session: at the TF-kernel
boundary parity stays UNPINNED (which reading of tf.sparse_softmax is executed -- SURVEY 9 H1 --, TF's accumulation
order).  Everything above that boundary is pinned against the reference's OWN CODE, imported in the build container
(generating scripts committed under tests/golden/, fixtures travel):
  * tests/golden/make_reference_model_fixtures.py runs the reference's model_builder / Representation /
    AffineTransform / ConcatGcn / BasisGcn / RelationEmbedding / BilinearDiag as they stand over an eager numpy
    stand-in for the ~25 TF primitives they call; tests/test_reference_model.py holds this oracle (and, -m gpu, the
    HIP path) to the initial weights (bitwise), codes in test and train mode, the loss and the score matrices;
  * tests/golden/make_reference_fixtures.py drives the reference's plain-Python host logic (settings, optimizer
    parser, Converge stack and fit loop, negative sampler, neighbourhood sampler, MRR scorer);
    tests/test_reference_fixtures.py replays it.
Independently (tests/test_oracle.py): a closed-form dense restatement, torch-CPU autograd of the same dataflow
(the gradients), and hand-computed cases.
"""
from .rgcn_oracle import *  # noqa: F401,F403
