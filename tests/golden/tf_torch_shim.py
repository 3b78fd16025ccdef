"""The same stand-in as tf_numpy_shim.py, on torch CPU tensors, so that the REFERENCE'S OWN forward dataflow can be
differentiated: `tf.gradients(loss, weights)` (optimization/abstract.py:117-118) becomes torch autograd over the graph
the reference's model code builds.  Fixture generation only (tests/golden/make_reference_model_fixtures.py).

Variables are float32 leaf tensors with requires_grad; integer tensors stay int64 torch tensors; dropout replays the
masks the numpy run recorded (REPLAY_MASKS), so both runs see the same forward pass."""
import numpy as np
import torch

float32 = torch.float32
float64 = torch.float64
int32 = torch.int32
int64 = torch.int64

FEED = {}
VARIABLES = []            # in creation order
REPLAY_MASKS = []
SPARSE_SOFTMAX_MODE = "intended"


def reset(feed, masks, sparse_softmax_mode="intended"):
    global SPARSE_SOFTMAX_MODE
    FEED.clear()
    FEED.update(feed)
    del VARIABLES[:]
    del REPLAY_MASKS[:]
    REPLAY_MASKS.extend(masks)
    SPARSE_SOFTMAX_MODE = sparse_softmax_mode


def _t(x):
    if isinstance(x, torch.Tensor):
        return x
    a = np.asarray(x)
    if a.dtype.kind in "iu":
        return torch.from_numpy(a.astype(np.int64))
    return torch.from_numpy(a.astype(np.float32))


def placeholder(dtype, shape=None, name=None):
    if name == 'graph_edges':
        return _t(np.asarray(FEED['graph_edges'], dtype=np.int64))
    if dtype is torch.float32:
        return _t(np.asarray(FEED['Y'], dtype=np.float32))
    return _t(np.asarray(FEED['X'], dtype=np.int64))


class _Tensor(torch.Tensor):
    """`x += y` on a TF tensor builds a new tensor; the reference relies on that (affine_transform.py:72-76 adds the
    bias to `hidden = self.W`), so in-place addition is rerouted to the out-of-place one"""

    def __iadd__(self, other):
        return torch.Tensor.add(self, other)


def Variable(initial_value, *args, **kwargs):
    v = torch.Tensor._make_subclass(_Tensor, torch.tensor(np.asarray(initial_value, dtype=np.float32)), True)
    VARIABLES.append(v)
    return v


def to_float(x):
    return _t(x).to(torch.float32)


def to_int32(x):
    return _t(x).to(torch.int64)


def to_int64(x):
    return _t(x).to(torch.int64)


def stack(values, axis=0):
    return torch.stack([_t(v) if not isinstance(v, int) else torch.tensor(v) for v in values], dim=axis)


def transpose(a, perm=None):
    a = _t(a)
    if perm is None:
        perm = list(reversed(range(a.dim())))
    return a.permute(*perm)


def reshape(tensor, shape):
    return _t(tensor).reshape([int(s) for s in shape])


def shape(x):
    return torch.tensor(list(_t(x).shape), dtype=torch.int64)


def range(limit):                                   # noqa: A001
    return torch.arange(int(limit), dtype=torch.int64)


def ones_like(x):
    return torch.ones_like(_t(x))


def expand_dims(x, axis):
    return _t(x).unsqueeze(axis)


def squeeze(x):
    return _t(x).squeeze()


def square(x):
    return x * x


def reduce_sum(x, axis=None):
    return _t(x).sum() if axis is None else _t(x).sum(dim=axis)


def reduce_mean(x, axis=None):
    return _t(x).mean() if axis is None else _t(x).mean(dim=axis)


def matmul(a, b):
    return torch.matmul(_t(a), _t(b))


class SparseTensor(object):
    def __init__(self, indices, values, dense_shape):
        self.indices, self.values, self.dense_shape = _t(indices).to(torch.int64), _t(values), _t(dense_shape)


def sparse_softmax(sp):
    rows = sp.indices[:, 0]
    e = torch.exp(sp.values - sp.values.max()) if sp.values.numel() else sp.values
    denom = torch.zeros(int(sp.dense_shape[0]), dtype=torch.float32).index_add(0, rows, e)
    per_entry = e / denom[rows]
    if SPARSE_SOFTMAX_MODE == "sorted_rows":
        order = np.lexsort((sp.indices[:, 1].numpy(), rows.numpy()))
        per_entry = per_entry[torch.from_numpy(order)]
    return SparseTensor(sp.indices, per_entry, sp.dense_shape)


def sparse_tensor_dense_matmul(sp, dense):
    dense = _t(dense)
    out = torch.zeros((int(sp.dense_shape[0]), dense.shape[1]), dtype=dense.dtype)
    return out.index_add(0, sp.indices[:, 0], sp.values[:, None] * dense[sp.indices[:, 1]])


class _NN(object):
    @staticmethod
    def embedding_lookup(params, ids):
        return _t(params)[_t(ids)]

    @staticmethod
    def relu(x):
        return torch.relu(x)

    @staticmethod
    def sigmoid(x):
        return torch.sigmoid(x)

    @staticmethod
    def dropout(x, keep_prob):
        mask = torch.from_numpy(np.asarray(REPLAY_MASKS.pop(0), dtype=np.float32))
        return x / float(keep_prob) * mask

    @staticmethod
    def weighted_cross_entropy_with_logits(targets, logits, pos_weight):
        z, x = _t(targets), _t(logits)
        log_weight = 1 + (pos_weight - 1) * z
        return (1 - z) * x + log_weight * (torch.log1p(torch.exp(-torch.abs(x))) + torch.clamp(-x, min=0))


nn = _NN()


class _Train(object):
    class Saver(object):
        def __init__(self, *a, **k):
            pass


train = _Train()


class _Missing(object):
    def __init__(self, name):
        self._name = name

    def __getattr__(self, item):
        return _Missing(self._name + "." + item)

    def __call__(self, *a, **k):
        raise NotImplementedError("tensorflow.%s is outside the shimmed R-GCN path" % self._name)


def __getattr__(name):
    return _Missing(name)
