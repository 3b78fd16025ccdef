"""db_emb from the row-gradient kernel's column partials (block_rows.hip, CombineArgs::colsum): the same bits from one
engine to the next?  (The first version walked the long rows in their REGISTRATION order -- atomics in k_ptrs -- and
differed by an ulp or two between engines; they are now taken from the tail of row_order, by vertex id.)
Prints per (engine, repetition): sum of db_emb, max |difference to the first engine's|, max |db_emb - colsum(dW_emb)|."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers, oracle
from relationprediction_amd import _native
V, R, d, L, nb = 14951, 1345, 500, 2, 100
triples = helpers.load_graph("fb15k_minibatch")
params = oracle.init_params(V, R, d, L, "block", nb, rng=np.random.RandomState(4))
dcodes = np.random.RandomState(1).randn(V, d).astype(np.float32)
res = []
for rep in range(3):
    e = _native.Engine(V, R, d, L, "block", nb, keep_prob=0.8, max_edges=len(triples))
    e.set_params(params); e.set_graph(triples)
    for it in range(2):
        e.forward(train=True, seed=5); e.backward(dcodes)
        g = e.get_grads()
        res.append(g["b_emb"].copy())
        print(rep, it, float(g["b_emb"].sum()), float(np.abs(g["b_emb"] - res[0]).max()), float(np.abs(g["b_emb"] - g["W_emb"].astype(np.float64).sum(0)).max()))
    e.close()
