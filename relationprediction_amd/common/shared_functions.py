"""Host-side helpers mirroring code/common/shared_functions.py of the reference.

The reference's helpers build TF variables / ops; here variables are plain float32 numpy arrays
that the plugin chain pushes into the HIP engine (rgcn_set_param).  `dot_or_lookup` has no host
counterpart: the matmul branch is the engine's fp32-MFMA GEMM, the lookup branch its fused gathers.
"""
import numpy as np


def glorot_variance(shape):
    """shared_functions.py:12-13.  NOTE: the reference passes this value as the *scale* (std-dev)
    of np.random.normal (:17), not as a variance; kept bit-for-bit."""
    return 3 / np.sqrt(shape[0] + shape[1])


def make_variable(mean, variance, shape, init="normal", rng=None):
    """make_tf_variable (shared_functions.py:16-22) without the tf.Variable wrapper."""
    rng = np.random if rng is None else rng
    if init == "normal":
        return rng.normal(mean, variance, size=shape).astype(np.float32)
    elif init == "uniform":
        return rng.uniform(mean, variance, size=shape).astype(np.float32)
    raise ValueError(init)


def make_bias(shape, init=0):
    """make_tf_bias (shared_functions.py:25-29)."""
    if init == 0:
        return np.zeros(shape).astype(np.float32)
    elif init == 1:
        return np.ones(shape).astype(np.float32)
    raise ValueError(init)


def init_encoder_params(V, R, d, num_layers, kind, num_bases, rng=None):
    """All weights of the encoder/decoder pair with the reference's distributions, in the reference's
    creation order (outermost component first: code/model.py:156-164): RelationEmbedding
    (relation_embedding.py:15-18, randn [EntityCount, d]), top GCN layer ... bottom GCN layer
    (gcn_basis_concat.py:17-27 | gcn_basis.py:15-30), then AffineTransform (affine_transform.py:24-28).
    Returns {name: array} keyed like rgcn_param_info names."""
    rng = np.random if rng is None else rng
    p = {}
    p["W_relation"] = rng.randn(V, d).astype(np.float32)
    for l in range(num_layers, 0, -1):
        if kind == "block":
            sd = int(d / num_bases)
            shape = (R, num_bases, sd, sd)
            var = glorot_variance([shape[0], shape[2]])
            p["W_f%d" % l] = make_variable(0, var, shape, rng=rng)
            p["W_b%d" % l] = make_variable(0, var, shape, rng=rng)
            p["W_self%d" % l] = make_variable(0, var, (d, d), rng=rng)
        else:
            shape = (d, num_bases, d)
            var = glorot_variance([shape[0], shape[2]])
            p["W_f%d" % l] = make_variable(0, var, shape, rng=rng)
            p["W_b%d" % l] = make_variable(0, var, shape, rng=rng)
            p["W_self%d" % l] = make_variable(0, var, (d, d), rng=rng)
            p["C_f%d" % l] = make_variable(0, 1, (R, num_bases), rng=rng)
            p["C_b%d" % l] = make_variable(0, 1, (R, num_bases), rng=rng)
        p["b%d" % l] = make_bias(d)
    p["W_emb"] = make_variable(0, glorot_variance([V, d]), (V, d), rng=rng)
    p["b_emb"] = make_bias(d)
    return p
