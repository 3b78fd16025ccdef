"""CPU oracle for the R-GCN encoder hot path.  TEST INFRASTRUCTURE ONLY.

Nothing under ``oracle/`` is part of the product.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it, and only as the checker / the timed CPU baseline -- never as a fallback
for the HIP path (``relationprediction_amd`` fails loudly without its HIP
library).

PARITY UNPINNED: the reference (MichSchli/RelationPrediction) ships no tests,
golden vectors or checkpoints, and TensorFlow 1.4 is not installable here, so
this restatement cannot be pinned against reference outputs.  It is pinned
three independent ways instead (see tests/test_oracle.py): a closed-form dense
restatement, torch-CPU autograd of the same dataflow, and finite differences.
"""
from .rgcn_oracle import *  # noqa: F401,F403
