"""Forward-pass golden vectors computed by THE REFERENCE'S OWN MODEL CODE (run where /root/reference exists):

    python -B tests/golden/make_reference_model_fixtures.py      ->  tests/golden/reference_model.npz

The reference's `common/model_builder.py` assembles Representation -> AffineTransform -> ConcatGcn | BasisGcn x L ->
RelationEmbedding -> BilinearDiag exactly as `train.py` does; `initialize_train()` draws the weights from numpy's
global stream; `get_loss('train') + get_regularization()`, `get_all_codes(mode)` and the two score-everything
graphs are evaluated.  The only thing that is not the reference is TensorFlow itself: `tests/golden/tf_numpy_shim.py`
is registered as `tensorflow` and evaluates each primitive eagerly in numpy (TF 1.4 cannot be installed here).  What
this pins: the initial weights (distributions, shapes, creation order = numpy stream order), and the whole
composition of the forward pass and the loss -- gathers, reshapes, which weight index is the output index, which
incidence matrix multiplies which messages, where dropout and relu sit -- and, through a second run of the same model
code on torch tensors (tf_torch_shim.py), the gradient of that loss w.r.t. every weight: what tf.gradients(loss,
weights) differentiates (optimization/abstract.py:117-118), by autograd over the reference's own dataflow.  tests/test_reference_model.py checks the
oracle (CPU) and the HIP path (`-m gpu`, 1e-4 absolute: north_star's tolerance) against these arrays.

The reference tree is read-only: no bytecode is written (sys.dont_write_bytecode, run with python -B).
"""
import os
import sys
import types

sys.dont_write_bytecode = True

import numpy as np  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/code"

CASES = {
    # name: (kind, V, R, d, nb|B, L, E, N, seed, sparse_softmax_mode)
    "block_2layer": ("block", 20, 4, 6, 3, 2, 50, 30, 1, "intended"),
    "block_sd5": ("block", 30, 7, 10, 2, 2, 120, 40, 2, "intended"),
    "basis_b2": ("basis", 20, 4, 6, 2, 2, 50, 30, 3, "intended"),
    "basis_b5_1layer": ("basis", 16, 9, 10, 5, 1, 43, 20, 4, "intended"),      # BASELINE config 1's shape family
    "block_h1_sorted_rows": ("block", 20, 4, 6, 3, 2, 50, 30, 5, "sorted_rows"),
    # corners where tf.squeeze (drops EVERY size-1 dimension) and the reshapes could bite
    "block_one_edge": ("block", 12, 3, 6, 3, 2, 1, 8, 7, "intended"),
    "block_sd1": ("block", 15, 3, 6, 6, 2, 40, 12, 8, "intended"),            # 1x1 blocks
    "block_nb1": ("block", 15, 3, 4, 1, 2, 40, 12, 9, "intended"),            # one 4x4 block = a dense W per relation
    "basis_b1": ("basis", 15, 3, 6, 1, 2, 40, 12, 10, "intended"),
    "basis_h1_sorted_rows": ("basis", 20, 4, 6, 2, 2, 50, 30, 11, "sorted_rows"),
    # BASELINE config 2 at full size: the real FB15k-237 minibatch graph of tests/golden/graphs.npz, d = 500, 100
    # blocks of 5x5, 2 layers; stored as fingerprints (weights and masks are regenerated from the seeds by the test)
    "fb237_block_full": ("block", 14541, 237, 500, 100, 2, 15000, 3000, 6, "intended"),
    # BASELINE config 3 at full size: the same minibatch graph under the basis decomposition, B = 2
    # (reference gcn_basis.py:39-88)
    "fb237_basis_b2_full": ("basis", 14541, 237, 500, 2, 2, 15000, 3000, 12, "intended"),
}
FULL_SIZE = {"fb237_block_full": "fb237_minibatch", "fb237_basis_b2_full": "fb237_minibatch"}


def fingerprint(arr, n=256, seed=0):
    """tests/helpers.py:probe -- l2 norm, sum and n sampled entries at fixed positions"""
    a = np.asarray(arr, dtype=np.float32).ravel()
    idx = np.random.RandomState(seed).randint(0, a.size, size=min(n, a.size))
    return {"l2": np.float64(np.sqrt(np.sum(a.astype(np.float64) ** 2))), "sum": np.float64(a.astype(np.float64).sum()),
            "idx": idx.astype(np.int64), "val": a[idx].copy()}


def settings_for(kind, V, R, d, nb, L, E):
    enc = {'Name': 'gcn_basis', 'DropoutKeepProbability': '0.8', 'InternalEncoderDimension': str(d),
           'NumberOfBasisFunctions': str(nb), 'NumberOfLayers': str(L), 'UseInputTransform': 'Yes',
           'UseOutputTransform': 'No', 'AddDiagonal': 'No', 'DiagonalCoefficients': 'No', 'SkipConnections': 'None',
           'StoreEdgeData': 'No', 'RandomInput': 'No', 'PartiallyRandomInput': 'No',
           'Concatenation': 'Yes' if kind == 'block' else 'No', 'CodeDimension': str(d),
           'EntityCount': V, 'RelationCount': R, 'EdgeCount': E, 'NegativeSampleRate': '10', 'GraphSplitSize': '0.5'}
    dec = {'Name': 'bilinear-diag', 'RegularizationParameter': '0.01', 'CodeDimension': str(d),
           'EntityCount': V, 'RelationCount': R, 'EdgeCount': E, 'NegativeSampleRate': '10'}
    return enc, dec


def main():
    sys.path.insert(0, HERE)
    import tf_numpy_shim as tf
    sys.modules['tensorflow'] = tf
    for stub in ("theano", "theano.tensor"):
        sys.modules.setdefault(stub, types.ModuleType(stub))
    sys.modules["theano"].tensor = sys.modules["theano.tensor"]
    sys.path.insert(0, REF)
    from common import model_builder                       # the reference's
    from encoders.message_gcns.message_gcn import MessageGcn
    from decoders.bilinear_diag import BilinearDiag

    out = {}
    for name, (kind, V, R, d, nb, L, E, N, seed, mode) in CASES.items():
        rng = np.random.RandomState(100 + seed)
        triples = np.stack([rng.randint(0, V, E), rng.randint(0, R, E), rng.randint(0, V, E)], 1).astype(np.int32)
        full = name in FULL_SIZE
        if full:
            with np.load(os.path.join(HERE, "graphs.npz")) as z:
                triples = z[FULL_SIZE[name]].astype(np.int32)
            assert triples.shape == (E, 3)
        X = np.stack([rng.randint(0, V, N), rng.randint(0, R, N), rng.randint(0, V, N)], 1).astype(np.int32)
        X[:N // 3] = triples[:N // 3]
        Y = (np.arange(N) < N // 3).astype(np.float32)
        tf.reset({'graph_edges': triples, 'X': X, 'Y': Y}, dropout_seed=seed, sparse_softmax_mode=mode)
        # the reference keeps its per-mode caches in CLASS attributes: fresh ones for every model
        MessageGcn.vertex_embedding_function = {'train': None, 'test': None}
        BilinearDiag.encoder_cache = {'train': None, 'test': None}
        enc, dec = settings_for(kind, V, R, d, nb, L, E)
        np.random.seed(seed)
        encoder = model_builder.build_encoder(enc, triples)
        model = model_builder.build_decoder(encoder, dec)
        model.preprocess(triples)
        model.register_for_test(triples)
        model.initialize_train()
        weights = [np.array(w) for w in model.get_weights()]
        loss = model.get_loss(mode='train') + model.get_regularization()      # train.py's loss
        codes_train = np.array(encoder.get_all_codes(mode='train')[0])
        masks = [np.array(m) for m in tf.DROPOUT_MASKS]                        # call order: bottom GCN layer first
        assert len(masks) == L, (name, len(masks))
        codes_test = np.array(encoder.get_all_codes(mode='test')[0])
        assert len(tf.DROPOUT_MASKS) == L                                      # test mode draws nothing
        subj = obj = None
        if name not in FULL_SIZE:
            subj = np.array(model.predict_all_subject_scores())
            obj = np.array(model.predict_all_object_scores())
        out[name + "/config"] = np.array([{'block': 0, 'basis': 1}[kind], V, R, d, nb, L, E, N, seed,
                                          {'intended': 0, 'sorted_rows': 1}[mode]], dtype=np.int64)
        out[name + "/X"], out[name + "/Y"] = X, Y
        if not full:
            out[name + "/triples"] = triples            # (the full-size graph is already in graphs.npz)
        def store(key, arr):
            if full:
                for field, v in fingerprint(arr).items():
                    out["%s/%s/%s" % (name, key, field)] = v
            else:
                out["%s/%s" % (name, key)] = arr
        if not full:
            for i, w in enumerate(weights):
                out["%s/weight%02d" % (name, i)] = w
            for i, m in enumerate(masks):
                out["%s/mask%d" % (name, i + 1)] = m
            out[name + "/subject_scores"], out[name + "/object_scores"] = subj, obj
        else:
            out[name + "/n_weights"] = np.int64(len(weights))
            for i, w in enumerate(weights):
                store("weight%02d" % i, w)
        out[name + "/loss_train"] = np.float64(loss)
        store("codes_train", codes_train)
        store("codes_test", codes_test)
        # ---- the same graph once more on torch tensors: tf.gradients(loss, weights) = autograd over the
        # reference's own dataflow (same weights: same seed; same dropout masks: replayed)
        import tf_torch_shim as tft
        ref_modules = [m for m in list(sys.modules.values())
                       if getattr(m, '__file__', None) and str(m.__file__).startswith(REF) and hasattr(m, 'tf')]
        for m in ref_modules:
            m.tf = tft
        try:
            tft.reset({'graph_edges': triples, 'X': X, 'Y': Y}, masks, sparse_softmax_mode=mode)
            MessageGcn.vertex_embedding_function = {'train': None, 'test': None}
            BilinearDiag.encoder_cache = {'train': None, 'test': None}
            np.random.seed(seed)
            encoder_t = model_builder.build_encoder(enc, triples)
            model_t = model_builder.build_decoder(encoder_t, dec)
            model_t.preprocess(triples)
            model_t.register_for_test(triples)
            model_t.initialize_train()
            weights_t = model_t.get_weights()
            for w_np, w_t in zip(weights, weights_t):
                assert np.array_equal(w_np, w_t.detach().numpy())
            loss_t = model_t.get_loss(mode='train') + model_t.get_regularization()
            assert abs(float(loss_t) - float(loss)) <= 1e-5 * max(1.0, abs(float(loss))), (float(loss_t), float(loss))
            loss_t.backward()
            for i, w_t in enumerate(weights_t):
                connected = w_t.grad is not None
                store("grad%02d" % i, w_t.grad.numpy() if connected else np.zeros_like(weights[i]))
                out["%s/grad%02d_connected" % (name, i)] = np.array(connected)
        finally:
            for m in ref_modules:
                m.tf = tf
        print(name, "weights", [w.shape for w in weights], "loss %.6f" % loss, "codes", codes_test.shape,
              "unconnected", [i for i, w_t in enumerate(weights_t) if w_t.grad is None])
    np.savez_compressed(os.path.join(HERE, "reference_model.npz"), **out)


if __name__ == "__main__":
    main()
