"""Time THE REFERENCE'S OWN MODEL CODE on BASELINE config 2's minibatch (run where /root/reference exists):

    python -B tests/golden/time_reference_code.py [steps]     ->  tests/golden/reference_code_timing.json

This is the closest thing to "the reference's TF1 CPU path" that can run here (TF 1.4 cannot be installed): the
reference's `common/model_builder.py` chain (Representation -> AffineTransform -> ConcatGcn x 2 -> RelationEmbedding ->
BilinearDiag, exactly as `train.py` assembles it) executed over `tests/golden/tf_torch_shim.py`, which maps the ~25 TF
symbols the path calls onto torch-CPU tensors (torch's intra-op thread pool = TF's), so that one step is
`loss = get_loss('train') + get_regularization(); loss.backward()` = the forward pass plus what
`tf.gradients(loss, weights)` computes (optimization/abstract.py:117-118).  Same graph (the 15,000-edge FB15k-237
minibatch of tests/golden/graphs.npz), same shapes (V 14,541, R 237, d 500, 100 blocks, 2 layers), a 3,000-triple
decoder batch (small beside the encoder: the metric is the encoder's forward + backward).

/root/reference does not exist on the GPU box, so bench.py cannot run this there: it reports the committed JSON
(`cpu_baseline_reference_code`, with this host's description and `measured_in_this_run: false`).  The tree is read-only:
run with python -B.
"""
import json
import os
import sys
import time
import types

sys.dont_write_bytecode = True

import numpy as np  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/code"


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    sys.path.insert(0, HERE)
    import torch
    import tf_torch_shim as tft
    from make_reference_model_fixtures import CASES, settings_for
    sys.modules['tensorflow'] = tft
    for stub in ("theano", "theano.tensor"):
        sys.modules.setdefault(stub, types.ModuleType(stub))
    sys.modules["theano"].tensor = sys.modules["theano.tensor"]
    sys.path.insert(0, REF)
    from common import model_builder                       # the reference's
    from encoders.message_gcns.message_gcn import MessageGcn
    from decoders.bilinear_diag import BilinearDiag

    kind, V, R, d, nb, L, E, N, seed, mode = CASES["fb237_block_full"]
    with np.load(os.path.join(HERE, "graphs.npz")) as z:
        triples = z["fb237_minibatch"].astype(np.int32)
    rng = np.random.RandomState(100 + seed)
    X = np.stack([rng.randint(0, V, N), rng.randint(0, R, N), rng.randint(0, V, N)], 1).astype(np.int32)
    X[:N // 3] = triples[:N // 3]
    Y = (np.arange(N) < N // 3).astype(np.float32)
    masks = [(np.random.RandomState(3 + l).rand(V, d) < 0.8).astype(np.float32) for l in range(L)]
    enc, dec = settings_for(kind, V, R, d, nb, L, E)
    np.random.seed(seed)
    tft.reset({'graph_edges': triples, 'X': X, 'Y': Y}, masks, sparse_softmax_mode=mode)
    MessageGcn.vertex_embedding_function = {'train': None, 'test': None}
    BilinearDiag.encoder_cache = {'train': None, 'test': None}
    encoder = model_builder.build_encoder(enc, triples)
    model = model_builder.build_decoder(encoder, dec)
    model.preprocess(triples)
    model.register_for_test(triples)
    model.initialize_train()
    weights = model.get_weights()

    def step():
        # the eager shim executes where TF would build the graph: one evaluation of the train graph = one session.run
        tft.reset({'graph_edges': triples, 'X': X, 'Y': Y}, masks, sparse_softmax_mode=mode)
        MessageGcn.vertex_embedding_function = {'train': None, 'test': None}
        BilinearDiag.encoder_cache = {'train': None, 'test': None}
        for w in weights:
            w.grad = None
        loss = model.get_loss(mode='train') + model.get_regularization()
        loss.backward()
        return float(loss)

    step()
    ts = []
    for _ in range(steps):
        t0 = time.perf_counter()
        loss = step()
        ts.append(time.perf_counter() - t0)
    med = float(np.median(ts))
    model_name = ""
    try:
        with open("/proc/cpuinfo") as f:
            model_name = next((ln.split(":", 1)[1].strip() for ln in f if ln.startswith("model name")), "")
    except OSError:
        pass
    out = {"value": round(E / med, 1), "unit": "edges/s", "cores": torch.get_num_threads(),
           "kind": "reference-code-over-torch-shim", "ms_per_step": round(med * 1e3, 1),
           "sample": "%d steps (median) of the reference's own model_builder chain, forward + loss + autograd "
                     "(= tf.gradients), fb237_block minibatch E_g=%d, decoder batch %d" % (steps, E, N),
           "host": "build container: %s, %d cores" % (model_name, os.cpu_count()),
           "measured_in_this_run": False, "loss": loss, "all_ms": [round(t * 1e3, 1) for t in ts]}
    with open(os.path.join(HERE, "reference_code_timing.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
