"""An eager numpy stand-in for the ~25 TensorFlow-1.4 symbols the reference's R-GCN path calls (fixture generation
only: tests/golden/make_reference_model_fixtures.py registers this module as `tensorflow` and then runs THE
REFERENCE'S OWN model code -- model_builder, Representation / MessageGraph, AffineTransform, ConcatGcn / BasisGcn,
RelationEmbedding, BilinearDiag -- on small seeded inputs).

TensorFlow 1.4 itself cannot be installed here, so the arithmetic of each primitive below is this file's statement
of the documented TF semantics; what the fixture pins is everything above the primitives: which tensors the
reference gathers, reshapes, transposes, multiplies and sums, in which order and with which index conventions --
the part a restatement can get wrong.  Every op evaluates immediately, so placeholders are bound to their feed values
BEFORE the model is built (FEED), and tf.nn.dropout records the mask it drew (DROPOUT_MASKS) so that the same mask
can be injected into the oracle / the device path.

sparse_softmax has two modes: "intended" normalises each entry by the entries of its own row;
"sorted_rows" reproduces the suspected behaviour of the TF kernel on non-canonical indices (results computed in
row-sorted order, attached to the original index list).
"""
import numpy as np

float32 = np.float32
float64 = np.float64
int32 = np.int32
int64 = np.int64

FEED = {}                 # 'graph_edges' -> int [E,3], 'X' -> int [N,3], 'Y' -> float [N]
DROPOUT_MASKS = []        # masks drawn by nn.dropout, in call order
DROPOUT_RNG = np.random.RandomState(0)
SPARSE_SOFTMAX_MODE = "intended"


def reset(feed, dropout_seed=0, sparse_softmax_mode="intended"):
    global DROPOUT_RNG, SPARSE_SOFTMAX_MODE
    FEED.clear()
    FEED.update(feed)
    del DROPOUT_MASKS[:]
    DROPOUT_RNG = np.random.RandomState(dropout_seed)
    SPARSE_SOFTMAX_MODE = sparse_softmax_mode


def placeholder(dtype, shape=None, name=None):
    if name == 'graph_edges':
        return np.asarray(FEED['graph_edges'], dtype=np.int32)
    if dtype is np.float32:
        return np.asarray(FEED['Y'], dtype=np.float32)
    return np.asarray(FEED['X'], dtype=np.int32)


class _Tensor(np.ndarray):
    """`x += y` on a TF tensor builds a new tensor; the reference relies on that (affine_transform.py:72-76 adds the
    bias to `hidden = self.W`), so in-place addition must not write into the variable"""

    def __iadd__(self, other):
        return np.add(self, other)


def Variable(initial_value, *args, **kwargs):
    return np.array(initial_value).view(_Tensor)


def to_float(x):
    return np.asarray(x).astype(np.float32)


def to_int32(x):
    return np.asarray(x).astype(np.int32)


def to_int64(x):
    return np.asarray(x).astype(np.int64)


def stack(values, axis=0):
    return np.stack([np.asarray(v) for v in values], axis=axis)


def transpose(a, perm=None):
    return np.transpose(a, perm)


def reshape(tensor, shape):
    return np.reshape(tensor, [int(s) for s in shape])


def shape(x):
    return np.array(np.shape(x), dtype=np.int32)


def range(limit):                                   # noqa: A001  (tf.range)
    return np.arange(int(limit), dtype=np.int32)


def ones_like(x):
    return np.ones_like(x)


def expand_dims(x, axis):
    return np.expand_dims(x, axis)


def squeeze(x):
    return np.squeeze(x)                            # like tf.squeeze: EVERY size-1 dimension goes


def square(x):
    return x * x


def reduce_sum(x, axis=None):
    return np.sum(x, axis=axis, dtype=np.asarray(x).dtype)


def reduce_mean(x, axis=None):
    return np.mean(x, axis=axis, dtype=np.asarray(x).dtype)


def matmul(a, b):
    return np.matmul(a, b)                          # batched over leading dimensions, like tf.matmul


class SparseTensor(object):
    def __init__(self, indices, values, dense_shape):
        self.indices = np.asarray(indices, dtype=np.int64)
        self.values = np.asarray(values)
        self.dense_shape = np.asarray(dense_shape, dtype=np.int64)


def sparse_softmax(sp):
    """softmax over the non-zero entries of each row (2-D) of sp"""
    rows = sp.indices[:, 0]
    e = np.exp(sp.values.astype(np.float32) - np.max(sp.values)) if len(sp.values) else sp.values.astype(np.float32)
    denom = np.zeros(int(sp.dense_shape[0]), dtype=np.float32)
    np.add.at(denom, rows, e)
    per_entry = (e / denom[rows]).astype(np.float32)
    if SPARSE_SOFTMAX_MODE == "sorted_rows":
        order = np.lexsort((sp.indices[:, 1], rows))          # canonical row-major order
        per_entry = per_entry[order]                           # results in sorted order, original index list kept
    return SparseTensor(sp.indices, per_entry, sp.dense_shape)


def sparse_tensor_dense_matmul(sp, dense):
    dense = np.asarray(dense)
    out = np.zeros((int(sp.dense_shape[0]), dense.shape[1]), dtype=dense.dtype)
    np.add.at(out, sp.indices[:, 0], sp.values[:, None].astype(dense.dtype) * dense[sp.indices[:, 1]])
    return out


class _NN(object):
    @staticmethod
    def embedding_lookup(params, ids):
        return np.asarray(params)[np.asarray(ids)]

    @staticmethod
    def relu(x):
        return np.maximum(x, 0)

    @staticmethod
    def sigmoid(x):
        x = np.asarray(x)
        return (1.0 / (1.0 + np.exp(-x.astype(np.float64)))).astype(x.dtype)

    @staticmethod
    def dropout(x, keep_prob):
        x = np.asarray(x)
        mask = np.floor(keep_prob + DROPOUT_RNG.uniform(size=x.shape)).astype(np.uint8)
        DROPOUT_MASKS.append(mask)
        return (x / np.float32(keep_prob) * mask).astype(x.dtype)

    @staticmethod
    def weighted_cross_entropy_with_logits(targets, logits, pos_weight):
        z, x = np.asarray(targets, dtype=np.float32), np.asarray(logits, dtype=np.float32)
        log_weight = 1 + (pos_weight - 1) * z
        return ((1 - z) * x + log_weight * (np.log1p(np.exp(-np.abs(x))) + np.maximum(-x, 0))).astype(np.float32)


nn = _NN()


class _Train(object):
    class Saver(object):
        def __init__(self, *a, **k):
            pass


train = _Train()


class _Missing(object):
    """any other tf symbol: importable (the reference's model_builder imports every encoder / decoder variant), not usable"""

    def __init__(self, name):
        self._name = name

    def __getattr__(self, item):
        return _Missing(self._name + "." + item)

    def __call__(self, *a, **k):
        raise NotImplementedError("tensorflow.%s is outside the shimmed R-GCN path" % self._name)


def __getattr__(name):
    return _Missing(name)
