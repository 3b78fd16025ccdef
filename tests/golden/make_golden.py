"""Generate the committed fixtures under tests/golden/.

    python tests/golden/make_golden.py          (needs /root/reference for the real graphs)

graphs.npz   real graph structure read from the reference's data files with the reference's own
             file formats (code/common/io.py:5-39): Toy train (43 triples, complete) and the
             FB15k-237 minibatch of SURVEY.md 8d config 2 "graph A" (30,000 of the 38,001 valid+test
             triples, seed 0, no replacement; then a random 15,000 of them = GraphSplitSize 0.5); the 10,000 WN18
             valid+test triples (config 4's entity / relation space: 40,943 / 18); 15,000 of the 50,000 real FB15k
             valid triples (config 5's space: 14,951 / 1,345; seed 0, no replacement).
expected.npz fingerprints (l2 norm, sum, 256 sampled entries) of the ORACLE's outputs on the cases
             in tests/helpers.py:GOLDEN_CASES.  The reference ships no golden vectors and TF 1.4
             cannot run here ("parity unpinned"), so these pin the oracle against regressions and
             give the GPU tests a reference-file-free target; they are NOT reference outputs.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))

import oracle  # noqa: E402
import helpers  # noqa: E402

REF = "/root/reference/data"


def read_dictionary(path):
    d = {}
    for line in open(path):
        k, v = line.rstrip("\n").split("\t")
        d[v] = int(k)
    return d


def read_triples(path, ent, rel):
    out = []
    for line in open(path):
        s, r, o = line.rstrip("\n").split("\t")
        out.append([ent[s], rel[r], ent[o]])
    return np.array(out, dtype=np.int32)


def make_graphs():
    ent = read_dictionary(REF + "/Toy/entities.dict")
    rel = read_dictionary(REF + "/Toy/relations.dict")
    toy = read_triples(REF + "/Toy/train.txt", ent, rel)
    assert toy.shape == (43, 3) and len(ent) == 16 and len(rel) == 9
    ent = read_dictionary(REF + "/FB-Toutanova/entities.dict")
    rel = read_dictionary(REF + "/FB-Toutanova/relations.dict")
    fb = np.concatenate([read_triples(REF + "/FB-Toutanova/valid.txt", ent, rel),
                         read_triples(REF + "/FB-Toutanova/test.txt", ent, rel)], axis=0)
    assert fb.shape == (38001, 3) and len(ent) == 14541 and len(rel) == 237
    rng = np.random.default_rng(0)
    batch = rng.choice(fb.shape[0], size=30000, replace=False)
    split = rng.choice(batch, size=15000, replace=False)
    ent = read_dictionary(REF + "/wn18/entities.dict")
    rel = read_dictionary(REF + "/wn18/relations.dict")
    wn = np.concatenate([read_triples(REF + "/wn18/valid.txt", ent, rel),
                         read_triples(REF + "/wn18/test.txt", ent, rel)], axis=0)
    assert wn.shape == (10000, 3) and len(ent) == 40943 and len(rel) == 18
    ent = read_dictionary(REF + "/FB15k/entities.dict")
    rel = read_dictionary(REF + "/FB15k/relations.dict")
    fb15k = read_triples(REF + "/FB15k/valid.txt", ent, rel)
    assert fb15k.shape == (50000, 3) and len(ent) == 14951 and len(rel) == 1345
    fb15k_batch = fb15k[np.random.default_rng(0).choice(fb15k.shape[0], size=15000, replace=False)]
    np.savez_compressed(os.path.join(HERE, "graphs.npz"), toy_train=toy, fb237_minibatch=fb[split],
                        fb237_valid_test=fb, wn18_valid_test=wn, fb15k_minibatch=fb15k_batch)


def make_expected():
    out = {}
    for name in helpers.GOLDEN_CASES:
        c = helpers.golden_inputs(name)
        for norm in ("intended", "tf_as_executed"):
            acts, grads = oracle.encoder_step(c["params"], c["triples"], c["V"], c["L"], c["kind"],
                                              c["dcodes"], keep_prob=0.8, dropout_masks=c["masks"],
                                              norm_mode=norm)
            tensors = {"codes": acts[-1], "H0": acts[0]}
            tensors.update({"grad_" + k: v for k, v in grads.items()})
            for tname, t in tensors.items():
                pr = helpers.probe(t)
                for field, v in pr.items():
                    out["%s/%s/%s/%s" % (name, norm, tname, field)] = v
        print("done", name)
    np.savez_compressed(os.path.join(HERE, "expected.npz"), **out)


if __name__ == "__main__":
    if os.path.isdir(REF):
        make_graphs()
    if "--graphs-only" not in sys.argv:
        make_expected()
