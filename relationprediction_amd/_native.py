"""ctypes binding of librgcn.so (C ABI: include/rgcn.h).

This is the only place Python crosses into native code -- the analogue of
``session.run`` in the reference (code/optimization/optimize.py:81-88,
code/model.py:56,69,81).  There is NO fallback: if the HIP library is missing
or no GPU is visible, construction fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "librgcn.so")

ABI_VERSION = 1
KIND_BLOCK, KIND_BASIS = 0, 1
NORM_INTENDED, NORM_TF_AS_EXECUTED, NORM_NONE = 0, 1, 2
BUF_EXCHANGE, BUF_SELF, BUF_DSELF_EXCHANGE, BUF_INDEG, BUF_OUTDEG, BUF_ROWPTR, BUF_NORM_EXCHANGE, \
    BUF_DBASIS_EXCHANGE, BUF_PERM_VERTEX, BUF_PERM_RELATION, BUF_RANK_ENERGIES = range(11)

KINDS = {"block": KIND_BLOCK, "basis": KIND_BASIS}
NORMS = {"intended": NORM_INTENDED, "tf_as_executed": NORM_TF_AS_EXECUTED, "none": NORM_NONE}


class RgcnError(RuntimeError):
    def __init__(self, status, message):
        super().__init__("librgcn status %d: %s" % (status, message))
        self.status = status


class RgcnConfig(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32), ("device", C.c_int32), ("num_entities", C.c_int32),
        ("num_relations", C.c_int32), ("dim", C.c_int32), ("num_layers", C.c_int32),
        ("kind", C.c_int32), ("num_bases", C.c_int32), ("keep_prob", C.c_float),
        ("norm_mode", C.c_int32), ("max_edges", C.c_int64), ("rank", C.c_int32),
        ("world", C.c_int32), ("reserved", C.c_int32),
    ]


_lib = None

_P = C.c_void_p
_SIGS = {
    "rgcn_abi_version": (C.c_int32, []),
    "rgcn_create": (C.c_int32, [C.POINTER(RgcnConfig), C.POINTER(_P)]),
    "rgcn_destroy": (C.c_int32, [_P]),
    "rgcn_last_error": (C.c_char_p, [_P]),
    "rgcn_sync": (C.c_int32, [_P]),
    "rgcn_param_count": (C.c_int32, [_P]),
    "rgcn_param_info": (C.c_int32, [_P, C.c_int32, C.c_char_p, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int32)]),
    "rgcn_set_param": (C.c_int32, [_P, C.c_int32, _P, C.c_int64]),
    "rgcn_get_param": (C.c_int32, [_P, C.c_int32, _P, C.c_int64]),
    "rgcn_get_grad": (C.c_int32, [_P, C.c_int32, _P, C.c_int64]),
    "rgcn_set_graph": (C.c_int32, [_P, _P, C.c_int64]),
    "rgcn_set_graph_device": (C.c_int32, [_P, _P, C.c_int64]),
    "rgcn_forward": (C.c_int32, [_P, C.c_int32, C.c_uint64, _P]),
    "rgcn_get_codes": (C.c_int32, [_P, _P, C.c_int64]),
    "rgcn_get_activation": (C.c_int32, [_P, C.c_int32, _P, C.c_int64]),
    "rgcn_get_dropout_mask": (C.c_int32, [_P, C.c_int32, _P, C.c_int64]),
    "rgcn_codes_device": (_P, [_P]),
    "rgcn_backward": (C.c_int32, [_P, _P, C.c_int64]),
    "rgcn_backward_device": (C.c_int32, [_P, _P]),
    "rgcn_step_device": (C.c_int32, [_P, _P, C.c_int64, C.c_int32, C.c_uint64, _P]),
    "rgcn_decoder_reserve": (C.c_int32, [_P, C.c_int64]),
    "rgcn_decoder_loss_backward_device": (C.c_int32, [_P, _P, _P, C.c_int64, C.c_float]),
    "rgcn_dcodes_device": (_P, [_P]),
    "rgcn_get_loss": (C.c_int32, [_P, C.POINTER(C.c_double)]),
    "rgcn_optimizer_config": (C.c_int32, [_P, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float]),
    "rgcn_sampler_create": (C.c_int32, [_P, C.c_int64, C.c_int32, C.POINTER(_P)]),
    "rgcn_sampler_destroy": (None, [_P]),
    "rgcn_sampler_edge_neighborhood": (C.c_int32, [_P, C.c_int64, C.c_uint64, _P]),
    "rgcn_neighborhood_reserve": (C.c_int32, [_P, _P, C.c_int64]),
    "rgcn_sample_neighborhood_device": (C.c_int32, [_P, C.c_int64, C.c_uint64, _P, C.c_int32]),
    "rgcn_capture_begin": (C.c_int32, [_P]),
    "rgcn_capture_end": (C.c_int32, [_P, C.POINTER(C.c_int32)]),
    "rgcn_graph_launch": (C.c_int32, [_P, C.c_int32]),
    "rgcn_graph_destroy": (C.c_int32, [_P, C.c_int32]),
    "rgcn_negative_sample_device": (C.c_int32, [_P, _P, C.c_int64, C.c_int32, C.c_uint64, _P, _P]),
    "rgcn_rank_reserve": (C.c_int32, [_P, C.c_int64]),
    "rgcn_rank_device": (C.c_int32, [_P, _P, C.c_int64, C.c_int32, _P, _P, _P, _P]),
    "rgcn_optimizer_step": (C.c_int32, [_P]),
    "rgcn_optimizer_norm_partial": (C.c_int32, [_P]),
    "rgcn_optimizer_apply": (C.c_int32, [_P]),
    "rgcn_train_step_device": (C.c_int32, [_P, _P, C.c_int64, _P, _P, C.c_int64, C.c_uint64, C.c_float]),
    "rgcn_prefetch_graph_device": (C.c_int32, [_P, _P, C.c_int64]),
    "rgcn_prefetch_graph_dropout_device": (C.c_int32, [_P, _P, C.c_int64, C.c_int64, C.c_uint64]),
    "rgcn_set_graph_dropout_device": (C.c_int32, [_P, _P, C.c_int64, C.c_int64, C.c_uint64, _P]),
    "rgcn_get_graph_edges": (C.c_int32, [_P, _P, C.c_int64]),
    "rgcn_train_step_minibatch_device": (C.c_int32, [_P, _P, C.c_int64, C.c_int64, C.c_uint64, C.c_int32, C.c_uint64,
                                                     _P, _P, C.c_uint64, C.c_float]),
    "rgcn_set_relation_owner": (C.c_int32, [_P, _P, C.c_int32]),
    "rgcn_comm_unique_id": (C.c_int32, [_P]),
    "rgcn_comm_init": (C.c_int32, [_P, _P]),
    "rgcn_comm_allreduce_sum": (C.c_int32, [_P, _P, C.c_int64]),
    "rgcn_forward_begin": (C.c_int32, [_P, C.c_int32, C.c_uint64, _P]),
    "rgcn_forward_layer_partial": (C.c_int32, [_P, C.c_int32]),
    "rgcn_forward_layer_finish": (C.c_int32, [_P, C.c_int32]),
    "rgcn_backward_begin": (C.c_int32, [_P, _P]),
    "rgcn_backward_layer_partial": (C.c_int32, [_P, C.c_int32]),
    "rgcn_backward_layer_finish": (C.c_int32, [_P, C.c_int32]),
    "rgcn_backward_end": (C.c_int32, [_P]),
    "rgcn_read_buffer": (C.c_int32, [_P, C.c_int32, _P, C.c_int64]),
    "rgcn_write_buffer": (C.c_int32, [_P, C.c_int32, _P, C.c_int64]),
    "rgcn_device_alloc": (C.c_int32, [_P, C.c_int64, C.POINTER(_P)]),
    "rgcn_device_free": (C.c_int32, [_P, _P]),
    "rgcn_copy_to_device": (C.c_int32, [_P, _P, _P, C.c_int64]),
    "rgcn_copy_to_device_async": (C.c_int32, [_P, _P, _P, C.c_int64, C.c_int32]),
    "rgcn_copy_to_host": (C.c_int32, [_P, _P, _P, C.c_int64]),
    "rgcn_timer_start": (C.c_int32, [_P]),
    "rgcn_timer_stop": (C.c_int32, [_P, C.POINTER(C.c_float)]),
    "rgcn_set_overlap": (C.c_int32, [_P, C.c_int32]),
    "rgcn_set_gemm_mode": (C.c_int32, [_P, C.c_int32]),
    "rgcn_set_fusion": (C.c_int32, [_P, C.c_int32]),
    "rgcn_comm_info": (C.c_int32, [_P, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "rgcn_device_info": (C.c_int32, [C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int64)]),
    "rgcn_profile_enable": (C.c_int32, [_P, C.c_int32]),
    "rgcn_profile_reset": (C.c_int32, [_P]),
    "rgcn_profile_count": (C.c_int32, [_P]),
    "rgcn_profile_get_compulsory": (C.c_int32, [_P, C.c_int32, C.POINTER(C.c_double)]),
    "rgcn_profile_get": (C.c_int32, [_P, C.c_int32, C.c_char_p, C.c_int32, C.POINTER(C.c_int64),
                                    C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
}
# librgcn_devtools.so only (include/rgcn_devtools.h)
_DEVTOOLS_SIGS = {
    "rgcn_debug_gemm": (C.c_int32, [_P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P]),
    "rgcn_debug_xcd_map": (C.c_int32, [_P, C.c_int32, _P]),
    "rgcn_debug_gemm_presplit": (C.c_int32, [_P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P,
                                             C.POINTER(C.c_float)]),
    "rgcn_debug_gemm_time": (C.c_int32, [_P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                         C.c_int32, _P, _P, C.POINTER(C.c_float)]),
}
_lib_devtools = None


def exported_symbols():
    """Names this binding expects librgcn.so to export (checked against include/rgcn.h in tests)."""
    return sorted(_SIGS)


def exported_devtools_symbols():
    return sorted(_DEVTOOLS_SIGS)


def load_library(path=None, devtools=False):
    """dlopen librgcn.so and attach prototypes.  Raises ImportError (never falls back) if missing.
    devtools=True: librgcn_devtools.so, the same sources plus the stand-alone GEMM entry points."""
    global _lib, _lib_devtools
    if devtools:
        if _lib_devtools is None or path is not None:
            p = path or os.path.join(os.path.dirname(LIB_PATH), "librgcn_devtools.so")
            if not os.path.exists(p):
                raise ImportError("%s not found: python -m relationprediction_amd.build makes it" % p)
            lib = C.CDLL(p, mode=C.RTLD_LOCAL)
            for name, (res, args) in list(_SIGS.items()) + list(_DEVTOOLS_SIGS.items()):
                fn = getattr(lib, name)
                fn.restype = res
                fn.argtypes = args
            if path is not None:
                return lib
            _lib_devtools = lib
        return _lib_devtools
    if _lib is not None and path is None:
        return _lib
    if path is None and os.environ.get("RGCN_LIBRARY") == "devtools":
        # the multi-process tests run their ranks on the devtools build: only that one honours RGCN_RCCL_LIBRARY
        _lib = load_library(devtools=True)
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise ImportError(
            "%s not found: build the HIP library first (python -m relationprediction_amd.build, or "
            "__graft_entry__.build()).  There is no CPU fallback." % p)
    lib = C.CDLL(p, mode=C.RTLD_GLOBAL)
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)   # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.rgcn_abi_version() != ABI_VERSION:
        raise ImportError("librgcn.so ABI version %d != binding %d" % (lib.rgcn_abi_version(), ABI_VERSION))
    if path is None:
        _lib = lib
    return lib


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class DeviceBuffer:
    """A raw device allocation owned through the engine (no torch needed)."""

    def __init__(self, engine, nbytes):
        self.engine = engine
        self.nbytes = int(nbytes)
        p = C.c_void_p()
        engine._check(engine.lib.rgcn_device_alloc(engine.ctx, self.nbytes, C.byref(p)))
        self.ptr = p

    def upload(self, arr):
        arr = np.ascontiguousarray(arr)
        assert arr.nbytes == self.nbytes, (arr.nbytes, self.nbytes)
        self.engine._check(self.engine.lib.rgcn_copy_to_device(self.engine.ctx, self.ptr, _ptr(arr), arr.nbytes))
        return self

    def download(self, dtype, shape):
        out = np.empty(shape, dtype=dtype)
        assert out.nbytes <= self.nbytes
        self.engine._check(self.engine.lib.rgcn_copy_to_host(self.engine.ctx, _ptr(out), self.ptr, out.nbytes))
        return out

    def free(self):
        if self.ptr is not None and self.engine.ctx is not None:
            self.engine.lib.rgcn_device_free(self.engine.ctx, self.ptr)
        self.ptr = None


class NeighborhoodSampler:
    """sample_edge_neighborhood of the reference's train loop (code/train.py:161-198) in O(log V) per pick
    (include/rgcn.h rgcn_sampler_*).  Host only: needs the library, not a GPU."""

    def __init__(self, triples, num_entities):
        self.lib = load_library()
        t = np.ascontiguousarray(triples, dtype=np.int32).reshape(-1, 3)
        self.n = len(t)
        h = C.c_void_p()
        st = self.lib.rgcn_sampler_create(_ptr(t), self.n, int(num_entities), C.byref(h))
        if st != 0:
            raise RgcnError(st, "rgcn_sampler_create (ids out of range or empty graph)")
        self.handle = h

    def sample(self, sample_size, seed):
        out = np.empty(int(sample_size), dtype=np.int32)
        st = self.lib.rgcn_sampler_edge_neighborhood(self.handle, int(sample_size), C.c_uint64(int(seed)), _ptr(out))
        if st != 0:
            raise RgcnError(st, "rgcn_sampler_edge_neighborhood (sample_size %d of %d edges)" % (sample_size, self.n))
        return out

    def close(self):
        if getattr(self, "handle", None) is not None:
            self.lib.rgcn_sampler_destroy(self.handle)
            self.handle = None

    __del__ = close


class Engine:
    """One rgcn_ctx: the encoder (input layer + L relational graph-convolution layers) on one GPU."""

    def __init__(self, num_entities, num_relations, dim, num_layers, kind, num_bases, keep_prob=0.8,
                 norm_mode="intended", max_edges=0, device=0, rank=0, world=1, devtools=False):
        self.lib = load_library(devtools=devtools)
        self.ctx = None
        cfg = RgcnConfig()
        cfg.abi_version = ABI_VERSION
        cfg.device = int(device)
        cfg.num_entities = int(num_entities)
        cfg.num_relations = int(num_relations)
        cfg.dim = int(dim)
        cfg.num_layers = int(num_layers)
        cfg.kind = KINDS[kind] if isinstance(kind, str) else int(kind)
        cfg.num_bases = int(num_bases)
        cfg.keep_prob = float(keep_prob)
        cfg.norm_mode = NORMS[norm_mode] if isinstance(norm_mode, str) else int(norm_mode)
        cfg.max_edges = int(max_edges)
        cfg.rank = int(rank)
        cfg.world = int(world)
        cfg.reserved = 0
        self.cfg = cfg
        ctx = C.c_void_p()
        st = self.lib.rgcn_create(C.byref(cfg), C.byref(ctx))
        if st != 0:
            raise RgcnError(st, (self.lib.rgcn_last_error(None) or b"").decode())
        self.ctx = ctx
        self.V, self.R, self.d, self.L = cfg.num_entities, cfg.num_relations, cfg.dim, cfg.num_layers
        self.max_edges = int(cfg.max_edges)
        self.param_names, self.param_shapes = [], []
        for i in range(self.lib.rgcn_param_count(self.ctx)):
            name = C.create_string_buffer(64)
            shape = (C.c_int64 * 4)()
            nd = C.c_int32()
            self._check(self.lib.rgcn_param_info(self.ctx, i, name, 64, shape, C.byref(nd)))
            self.param_names.append(name.value.decode())
            self.param_shapes.append(tuple(int(shape[k]) for k in range(nd.value)))

    # -- plumbing
    def _check(self, st):
        if st != 0:
            raise RgcnError(st, (self.lib.rgcn_last_error(self.ctx) or b"").decode())

    def close(self):
        if self.ctx is not None:
            if getattr(self, "_rank_buf", None) is not None:
                self._rank_buf.free()
                self._rank_buf = None
            self.lib.rgcn_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def sync(self):
        self._check(self.lib.rgcn_sync(self.ctx))

    def _index(self, key):
        return key if isinstance(key, int) else self.param_names.index(key)

    # -- parameters
    def set_param(self, key, value):
        i = self._index(key)
        a = np.ascontiguousarray(value, dtype=np.float32)
        if tuple(a.shape) != self.param_shapes[i]:
            raise ValueError("%s: shape %s != %s" % (self.param_names[i], a.shape, self.param_shapes[i]))
        self._check(self.lib.rgcn_set_param(self.ctx, i, _ptr(a), a.size))

    def set_params(self, params):
        for n in self.param_names:
            self.set_param(n, params[n])

    def get_param(self, key):
        i = self._index(key)
        out = np.empty(self.param_shapes[i], dtype=np.float32)
        self._check(self.lib.rgcn_get_param(self.ctx, i, _ptr(out), out.size))
        return out

    def get_grad(self, key):
        i = self._index(key)
        out = np.empty(self.param_shapes[i], dtype=np.float32)
        self._check(self.lib.rgcn_get_grad(self.ctx, i, _ptr(out), out.size))
        return out

    def get_grads(self):
        return {n: self.get_grad(n) for n in self.param_names}

    def get_params(self):
        return {n: self.get_param(n) for n in self.param_names}

    # -- graph
    def set_graph(self, triples):
        t = np.asarray(triples)
        if t.size == 0:
            t = np.zeros((0, 3), dtype=np.int32)
        if t.ndim != 2 or t.shape[1] != 3:
            raise ValueError("graph_edges must be [E,3]")
        if t.dtype != np.int32:
            # the reference feeds int64 numpy arrays into an int32 placeholder
            if t.size and (t.min() < -2 ** 31 or t.max() >= 2 ** 31):
                raise ValueError("ids do not fit int32")
            t = t.astype(np.int32)
        t = np.ascontiguousarray(t)
        self._check(self.lib.rgcn_set_graph(self.ctx, _ptr(t), t.shape[0]))
        self.num_edges = int(t.shape[0])

    def set_graph_device(self, dev_buffer, num_edges):
        self._check(self.lib.rgcn_set_graph_device(self.ctx, dev_buffer.ptr, int(num_edges)))
        self.num_edges = int(num_edges)

    def set_graph_dropout_device(self, batch_buffer, num_edges, keep, seed=0, keep_mask=None):
        """graph = exact-`keep` random subset of the batch's edges, drawn on the device (rgcn_set_graph_dropout_device);
        keep_mask: a DeviceBuffer with uint8 [num_edges] holding the caller's choice instead"""
        self._check(self.lib.rgcn_set_graph_dropout_device(self.ctx, batch_buffer.ptr, int(num_edges), int(keep),
                                                           C.c_uint64(seed), keep_mask.ptr if keep_mask else None))
        self.num_edges = int(keep)

    def graph_edges(self):
        """the [E,3] rows of the graph currently set (after edge dropout, if any)"""
        out = np.empty((self.num_edges, 3), dtype=np.int32)
        self._check(self.lib.rgcn_get_graph_edges(self.ctx, _ptr(out), self.num_edges))
        return out

    def prefetch_graph_dropout_device(self, batch_buffer, num_edges, keep, seed):
        self._check(self.lib.rgcn_prefetch_graph_dropout_device(self.ctx, batch_buffer.ptr, int(num_edges), int(keep),
                                                                C.c_uint64(seed)))

    def train_step_minibatch_device(self, batch_buffer, num_edges, keep, edge_seed, rate, negative_seed, x_dev, y_dev,
                                    seed=0, reg_param=0.01):
        self._check(self.lib.rgcn_train_step_minibatch_device(
            self.ctx, batch_buffer.ptr, int(num_edges), int(keep), C.c_uint64(edge_seed), int(rate),
            C.c_uint64(negative_seed), x_dev.ptr, y_dev.ptr, C.c_uint64(seed), C.c_float(reg_param)))
        self.num_edges = int(keep)

    def set_relation_owner(self, owner):
        o = np.ascontiguousarray(owner, dtype=np.int32)
        self._check(self.lib.rgcn_set_relation_owner(self.ctx, _ptr(o), o.size))

    # -- forward / backward
    def _masks(self, masks):
        if masks is None:
            return None, None
        m = np.ascontiguousarray(np.stack([np.asarray(x) for x in masks]).astype(np.uint8))
        if m.shape != (self.L, self.V, self.d):
            raise ValueError("dropout masks must be [L,V,d]")
        return m, _ptr(m)

    def forward(self, train=True, seed=0, masks=None):
        m, p = self._masks(masks)
        self._check(self.lib.rgcn_forward(self.ctx, 1 if train else 0, C.c_uint64(seed), p))

    def codes(self):
        return self.activation(self.L)

    def activation(self, layer):
        out = np.empty((self.V, self.d), dtype=np.float32)
        self._check(self.lib.rgcn_get_activation(self.ctx, int(layer), _ptr(out), out.size))
        return out

    def dropout_mask(self, layer):
        out = np.empty((self.V, self.d), dtype=np.uint8)
        self._check(self.lib.rgcn_get_dropout_mask(self.ctx, int(layer), _ptr(out), out.size))
        return out

    def backward(self, dcodes):
        g = np.ascontiguousarray(dcodes, dtype=np.float32)
        if g.shape != (self.V, self.d):
            raise ValueError("dcodes must be [V,d]")
        self._check(self.lib.rgcn_backward(self.ctx, _ptr(g), g.size))

    def backward_device(self, dev_buffer):
        self._check(self.lib.rgcn_backward_device(self.ctx, dev_buffer.ptr))

    def step_device(self, triples_dev, num_edges, dcodes_dev, train=True, seed=0):
        self._check(self.lib.rgcn_step_device(self.ctx, triples_dev.ptr, int(num_edges), 1 if train else 0,
                                              C.c_uint64(seed), dcodes_dev.ptr))

    # -- decoder / optimizer / whole train step on the device
    def decoder_reserve(self, max_triples):
        self._check(self.lib.rgcn_decoder_reserve(self.ctx, int(max_triples)))

    def decoder_loss_backward_device(self, x_dev, y_dev, num_triples, reg_param):
        self._check(self.lib.rgcn_decoder_loss_backward_device(self.ctx, x_dev.ptr, y_dev.ptr, int(num_triples),
                                                               C.c_float(reg_param)))

    def backward_from_decoder(self):
        """encoder backward fed with the decoder's dL/dcodes (stays on the device)"""
        self._check(self.lib.rgcn_backward_device(self.ctx, self.lib.rgcn_dcodes_device(self.ctx)))

    def dcodes(self):
        out = np.empty((self.V, self.d), dtype=np.float32)
        self._check(self.lib.rgcn_copy_to_host(self.ctx, _ptr(out), self.lib.rgcn_dcodes_device(self.ctx), out.nbytes))
        return out

    def loss(self):
        v = C.c_double()
        self._check(self.lib.rgcn_get_loss(self.ctx, C.byref(v)))
        return float(v.value)

    def capture_begin(self):
        """Start recording device calls into a hipGraph (include/rgcn.h rgcn_capture_begin)."""
        self._check(self.lib.rgcn_capture_begin(self.ctx))

    def capture_end(self):
        gid = C.c_int32(-1)
        self._check(self.lib.rgcn_capture_end(self.ctx, C.byref(gid)))
        return int(gid.value)

    def graph_launch(self, graph_id):
        self._check(self.lib.rgcn_graph_launch(self.ctx, int(graph_id)))

    def graph_destroy(self, graph_id):
        self._check(self.lib.rgcn_graph_destroy(self.ctx, int(graph_id)))

    def neighborhood_reserve(self, train_triples):
        """the training graph to the device, once: what rgcn_sample_neighborhood_device draws graph batches from"""
        t = np.ascontiguousarray(train_triples, dtype=np.int32).reshape(-1, 3)
        self._check(self.lib.rgcn_neighborhood_reserve(self.ctx, _ptr(t), len(t)))
        self._nbr_edges = len(t)

    def sample_neighborhood_device(self, sample_size, seed, batch_dev, on_prefetch_stream=False):
        """sample_edge_neighborhood (code/train.py:161-198) on the device: [sample_size,3] rows into batch_dev"""
        self._check(self.lib.rgcn_sample_neighborhood_device(self.ctx, int(sample_size), C.c_uint64(int(seed)),
                                                             batch_dev.ptr, 1 if on_prefetch_stream else 0))

    def negative_sample_device(self, batch_dev, n, rate, seed, x_dev, y_dev):
        """X [n*(rate+1),3] and Y [n*(rate+1)] from the batch of n triples, all on the device."""
        self._check(self.lib.rgcn_negative_sample_device(self.ctx, batch_dev.ptr, int(n), int(rate), C.c_uint64(int(seed)),
                                                         x_dev.ptr, y_dev.ptr))

    def rank_reserve(self, max_queries):
        self._check(self.lib.rgcn_rank_reserve(self.ctx, int(max_queries)))
        self._rank_reserved = max(getattr(self, "_rank_reserved", 0), int(max_queries))

    def ranks(self, triples, predict_object, filter_ptr, filter_idx):
        """Raw and filtered ranks (include/rgcn.h rgcn_rank_device) of the gold subject / object of every
        query triple on the codes of the last forward.  filter_ptr int64 [N+1], filter_idx int32 [nnz]."""
        x = np.ascontiguousarray(triples, dtype=np.int32).reshape(-1, 3)
        n = len(x)
        if n == 0:
            return np.zeros(0, np.int32), np.zeros(0, np.int32)
        fp = np.ascontiguousarray(filter_ptr, dtype=np.int64)
        fi = np.ascontiguousarray(filter_idx, dtype=np.int32)
        assert fp.shape == (n + 1,) and fp[-1] == len(fi)
        # ONE upload (filter pointers | queries | filter list, packed) into a buffer the engine keeps, ONE download (raw |
        # filtered ranks): a call used to cost five device allocations, three blocking uploads and two downloads --
        # more host time than the scoring itself
        nnz = max(len(fi), 1)
        off_x, off_fi = 8 * (n + 1), 8 * (n + 1) + 12 * n
        off_out = (off_fi + 4 * nnz + 7) // 8 * 8
        total = off_out + 8 * n
        if getattr(self, "_rank_buf", None) is None or self._rank_buf.nbytes < total:
            if getattr(self, "_rank_buf", None) is not None:
                self._rank_buf.free()
            self._rank_buf = DeviceBuffer(self, max(total, 1 << 20))
        base = self._rank_buf.ptr.value if hasattr(self._rank_buf.ptr, "value") else int(self._rank_buf.ptr)
        # staged through the library's pinned slots (<= 1 MB each), ordered on the main stream, no host wait: the filter
        # list (megabytes for the subject side of FB15k-237) travels while the queries are being scored
        head = np.empty(off_fi, dtype=np.uint8)
        head[:off_x] = fp.view(np.uint8)
        head[off_x:] = x.reshape(-1).view(np.uint8)
        self._check(self.lib.rgcn_copy_to_device_async(self.ctx, C.c_void_p(base), _ptr(head), off_fi, 0))
        fb = fi.view(np.uint8)
        for lo in range(0, len(fb), 1 << 20):
            part = fb[lo:lo + (1 << 20)]
            self._check(self.lib.rgcn_copy_to_device_async(self.ctx, C.c_void_p(base + off_fi + lo), _ptr(part), len(part), 0))
        self._check(self.lib.rgcn_rank_device(self.ctx, C.c_void_p(base + off_x), n, 1 if predict_object else 0,
                                              C.c_void_p(base), C.c_void_p(base + off_fi), C.c_void_p(base + off_out),
                                              C.c_void_p(base + off_out + 4 * n)))
        out = np.empty(2 * n, dtype=np.int32)
        self._check(self.lib.rgcn_copy_to_host(self.ctx, _ptr(out), C.c_void_p(base + off_out), 8 * n))
        return out[:n].copy(), out[n:].copy()

    def optimizer_config(self, lr=0.01, beta1=0.9, beta2=0.999, eps=1e-8, max_grad_norm=1.0):
        self._check(self.lib.rgcn_optimizer_config(self.ctx, lr, beta1, beta2, eps, max_grad_norm))

    def optimizer_step(self):
        self._check(self.lib.rgcn_optimizer_step(self.ctx))

    def optimizer_norm_partial(self):
        self._check(self.lib.rgcn_optimizer_norm_partial(self.ctx))

    def optimizer_apply(self):
        self._check(self.lib.rgcn_optimizer_apply(self.ctx))

    def train_step_device(self, triples_dev, num_edges, x_dev, y_dev, num_triples, seed=0, reg_param=0.01):
        self._check(self.lib.rgcn_train_step_device(self.ctx, triples_dev.ptr, int(num_edges), x_dev.ptr, y_dev.ptr,
                                                    int(num_triples), C.c_uint64(seed), C.c_float(reg_param)))

    def prefetch_graph_device(self, triples_dev, num_edges):
        self._check(self.lib.rgcn_prefetch_graph_device(self.ctx, triples_dev.ptr, int(num_edges)))

    # -- phase API (sharding exchange points)
    def forward_begin(self, train=True, seed=0, masks=None):
        m, p = self._masks(masks)
        self._check(self.lib.rgcn_forward_begin(self.ctx, 1 if train else 0, C.c_uint64(seed), p))

    def forward_layer_partial(self, l):
        self._check(self.lib.rgcn_forward_layer_partial(self.ctx, l))

    def forward_layer_finish(self, l):
        self._check(self.lib.rgcn_forward_layer_finish(self.ctx, l))

    def backward_begin(self, dcodes_dev=None):
        """dcodes_dev None: the decoder's own dL/dcodes (rgcn_dcodes_device), which never leaves the device"""
        ptr = self.lib.rgcn_dcodes_device(self.ctx) if dcodes_dev is None else dcodes_dev.ptr
        self._check(self.lib.rgcn_backward_begin(self.ctx, ptr))

    def backward_layer_partial(self, l):
        self._check(self.lib.rgcn_backward_layer_partial(self.ctx, l))

    def backward_layer_finish(self, l):
        self._check(self.lib.rgcn_backward_layer_finish(self.ctx, l))

    def backward_end(self):
        self._check(self.lib.rgcn_backward_end(self.ctx))

    def read_buffer(self, which):
        if which in (BUF_INDEG, BUF_OUTDEG):
            out = np.empty(self.V, dtype=np.int32)
        elif which == BUF_ROWPTR:
            out = np.empty(self.V + 1, dtype=np.int32)
        elif which in (BUF_PERM_VERTEX, BUF_PERM_RELATION):
            out = np.empty(2 * self.num_edges, dtype=np.int32)
        elif which == BUF_RANK_ENERGIES:
            out = np.empty((getattr(self, "_rank_reserved", 0), self.V), dtype=np.float32)
        elif which == BUF_DSELF_EXCHANGE:
            out = np.empty((self.d, self.d), dtype=np.float32)
        elif which == BUF_NORM_EXCHANGE:
            out = np.empty(1, dtype=np.float32)
        elif which == BUF_DBASIS_EXCHANGE:
            out = np.empty((2, int(self.cfg.num_bases), self.d, self.d), dtype=np.float32)
        else:
            out = np.empty((self.V, self.d), dtype=np.float32)
        self._check(self.lib.rgcn_read_buffer(self.ctx, which, _ptr(out), out.nbytes))
        return out

    def write_buffer(self, which, arr):
        a = np.ascontiguousarray(arr, dtype=np.float32)
        self._check(self.lib.rgcn_write_buffer(self.ctx, which, _ptr(a), a.nbytes))

    # -- communicator
    @staticmethod
    def comm_unique_id():
        lib = load_library()
        buf = (C.c_uint8 * 128)()
        st = lib.rgcn_comm_unique_id(buf)
        if st != 0:
            raise RgcnError(st, (lib.rgcn_last_error(None) or b"").decode())
        return bytes(buf)

    @staticmethod
    def device_info(device=0):
        """(HIP devices this process sees, PCI address of `device` as domain << 16 | bus << 8 | device, or -1)"""
        lib = load_library()
        n, pci = C.c_int32(0), C.c_int64(-1)
        st = lib.rgcn_device_info(int(device), C.byref(n), C.byref(pci))
        if st != 0:
            raise RgcnError(st, (lib.rgcn_last_error(None) or b"").decode())
        return n.value, pci.value

    def comm_init(self, unique_id):
        buf = (C.c_uint8 * 128).from_buffer_copy(bytes(unique_id))
        self._check(self.lib.rgcn_comm_init(self.ctx, buf))

    def comm_info(self):
        """(ranks the collective library's communicator sees, this rank's index in it, its device); -1: unknown / none"""
        n, r, dev = C.c_int32(-1), C.c_int32(-1), C.c_int32(-1)
        self._check(self.lib.rgcn_comm_info(self.ctx, C.byref(n), C.byref(r), C.byref(dev)))
        return n.value, r.value, dev.value

    def comm_allreduce_sum(self, dev_buffer, count):
        self._check(self.lib.rgcn_comm_allreduce_sum(self.ctx, dev_buffer.ptr, int(count)))

    # -- device memory / timing / profile
    def alloc(self, nbytes):
        return DeviceBuffer(self, nbytes)

    def copy_to_device(self, buf, arr):
        """Upload into the front of an existing (possibly larger) device buffer."""
        arr = np.ascontiguousarray(arr)
        assert arr.nbytes <= buf.nbytes
        self._check(self.lib.rgcn_copy_to_device(self.ctx, buf.ptr, _ptr(arr), arr.nbytes))

    def copy_to_device_async(self, buf, arr, on_prefetch_stream=False):
        """As copy_to_device without waiting for the transfer (rgcn_copy_to_device_async): staged through pinned
        memory, ordered on the main stream or on the prefetch stream."""
        arr = np.ascontiguousarray(arr)
        assert arr.nbytes <= buf.nbytes
        self._check(self.lib.rgcn_copy_to_device_async(self.ctx, buf.ptr, _ptr(arr), arr.nbytes,
                                                       1 if on_prefetch_stream else 0))

    def to_device(self, arr):
        arr = np.ascontiguousarray(arr)
        return DeviceBuffer(self, arr.nbytes).upload(arr)

    def timer_start(self):
        self._check(self.lib.rgcn_timer_start(self.ctx))

    def timer_stop(self):
        ms = C.c_float()
        self._check(self.lib.rgcn_timer_stop(self.ctx, C.byref(ms)))
        return float(ms.value)

    def set_overlap(self, on=True):
        self._check(self.lib.rgcn_set_overlap(self.ctx, 1 if on else 0))

    def set_fusion(self, mode):
        """form of the block layer: 0 two kernels + message buffer, 1 (default) destination-major banded single pass
        (block_rows.hip)"""
        self._check(self.lib.rgcn_set_fusion(self.ctx, int(mode)))

    def set_gemm_mode(self, mode):
        """0 = fp32 MFMA, 9 / 6 = exact bf16 operand split with 9 / 6 partial products (include/rgcn.h)."""
        self._check(self.lib.rgcn_set_gemm_mode(self.ctx, int(mode)))

    def profile_enable(self, on=True):
        self._check(self.lib.rgcn_profile_enable(self.ctx, 1 if on else 0))

    def profile_reset(self):
        self._check(self.lib.rgcn_profile_reset(self.ctx))

    def profile(self):
        """[{name, calls, total_ms, alg_bytes (design), alg_flops, compulsory_bytes}] aggregated per kernel name."""
        out = []
        n = self.lib.rgcn_profile_count(self.ctx)
        for i in range(n):
            name = C.create_string_buffer(64)
            calls = C.c_int64()
            ms, by, fl = C.c_double(), C.c_double(), C.c_double()
            self._check(self.lib.rgcn_profile_get(self.ctx, i, name, 64, C.byref(calls), C.byref(ms),
                                                  C.byref(by), C.byref(fl)))
            cb = C.c_double()
            self._check(self.lib.rgcn_profile_get_compulsory(self.ctx, i, C.byref(cb)))
            out.append({"name": name.value.decode(), "calls": calls.value, "total_ms": ms.value,
                        "alg_bytes": by.value, "alg_flops": fl.value, "compulsory_bytes": cb.value})
        return out

    def debug_gemm_presplit(self, a, b, trans_b=False, iters=0):
        """A . op(B) with B pre-split into MFMA fragments (devtools build); returns C, or (C, ms per product) if iters"""
        a = np.ascontiguousarray(a, dtype=np.float32)
        b = np.ascontiguousarray(b, dtype=np.float32)
        M, K = a.shape
        N = b.shape[0] if trans_b else b.shape[1]
        out = np.empty((M, N), dtype=np.float32)
        ms = C.c_float()
        self._check(self.lib.rgcn_debug_gemm_presplit(self.ctx, int(trans_b), M, N, K, int(iters), _ptr(a), _ptr(b),
                                                      _ptr(out), C.byref(ms)))
        return (out, float(ms.value)) if iters else out

    def debug_xcd_map(self, n_blocks):
        """XCD of every workgroup of a plain 1-D launch (devtools build)"""
        out = np.zeros(int(n_blocks), dtype=np.int32)
        self._check(self.lib.rgcn_debug_xcd_map(self.ctx, int(n_blocks), _ptr(out)))
        return out

    def debug_gemm_time(self, a, b, trans_a=False, trans_b=False, split_k=0, iters=20):
        a = np.ascontiguousarray(a, dtype=np.float32)
        b = np.ascontiguousarray(b, dtype=np.float32)
        K, M = (a.shape if trans_a else a.shape[::-1])
        N = b.shape[0] if trans_b else b.shape[1]
        ms = C.c_float()
        self._check(self.lib.rgcn_debug_gemm_time(self.ctx, int(trans_a), int(trans_b), M, N, K, int(split_k),
                                                  int(iters), _ptr(a), _ptr(b), C.byref(ms)))
        return float(ms.value)

    def debug_gemm(self, a, b, trans_a=False, trans_b=False, split_k=0):
        a = np.ascontiguousarray(a, dtype=np.float32)
        b = np.ascontiguousarray(b, dtype=np.float32)
        K, M = (a.shape if trans_a else a.shape[::-1])
        N = b.shape[0] if trans_b else b.shape[1]
        assert (b.shape[1] if trans_b else b.shape[0]) == K
        out = np.empty((M, N), dtype=np.float32)
        self._check(self.lib.rgcn_debug_gemm(self.ctx, int(trans_a), int(trans_b), M, N, K, int(split_k),
                                             _ptr(a), _ptr(b), _ptr(out)))
        return out
