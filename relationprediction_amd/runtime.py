"""EncoderRuntime: binds a chain of reference-shaped plugin components to ONE HIP engine context.

The reference builds a TF graph by recursing through `get_all_codes` (message_gcn.py:39,58) and runs
it in `session.run`.  Here the same recursion is collapsed: the first `get_all_codes` of a step makes
the engine run the whole encoder stack (input layer + L graph-convolution layers) and every component
then reads "its" activation.  The runtime lives on the graph `Representation` at the bottom of the chain.
"""
import numpy as np

from . import _native


class EncoderRuntime(object):
    def __init__(self, layers, affine, representation):
        """layers: MessageGcn components bottom -> top."""
        self.layers = layers
        self.affine = affine
        self.representation = representation
        kinds = {type(l).KIND for l in layers}
        if len(kinds) != 1:
            raise NotImplementedError("mixed graph-convolution layer types in one encoder")
        self.kind = kinds.pop()
        top = layers[-1]
        for i, l in enumerate(layers):
            if l.use_nonlinearity != (i < len(layers) - 1):
                raise NotImplementedError("only 'relu on all but the last layer' stacks are supported "
                                          "(model_builder.py:275)")
            if l.onehot_input:
                raise NotImplementedError("onehot_input graph-convolution layers (UseInputTransform=No)")
        s = top.settings
        self.V, self.R = top.entity_count, top.relation_count
        self.d = int(top.shape[1])
        norm = s['IncidenceNormalization'] if 'IncidenceNormalization' in s else 'intended'
        device = int(s['Device']) if 'Device' in s else 0
        max_edges = max(int(top.edge_count), 1)
        if 'GraphBatchSize' in s:
            max_edges = max(max_edges, int(s['GraphBatchSize']))
        self.engine = _native.Engine(self.V, self.R, self.d, len(layers), self.kind, top.n_coefficients,
                                     keep_prob=top.dropout_keep_probability, norm_mode=norm,
                                     max_edges=max_edges, device=device)
        self._state = None        # (graph version, mode) of the activations held by the engine
        self._dev = {}            # persistent device buffers of the fused train step (name -> DeviceBuffer)
        self._dec_reserved = 0
        self._rank_reserved = 0
        self._graph_version = None
        self.weights_version = 0
        self._fwd_weights_version = -1
        # move the weights into the engine (names follow rgcn_param_info)
        affine.W.bind(*self._accessors("W_emb"))
        affine.b.bind(*self._accessors("b_emb"))
        for i, l in enumerate(layers, start=1):
            l.layer_index = i
            for var, base in l.engine_variables():
                var.bind(*self._accessors("%s%d" % (base, i)))

    def _accessors(self, name):
        eng = self.engine

        def getter():
            return eng.get_param(name)

        def setter(value):
            eng.set_param(name, value)
            self.weights_version += 1

        return getter, setter

    def forward(self, mode):
        ph = self.representation.X
        if ph.value is None:
            raise RuntimeError("graph_edges was not fed")
        if self._graph_version != ph.version:
            self.engine.set_graph(ph.value)
            self._graph_version = ph.version
            self._state = None
        key = (ph.version, mode, self.weights_version)
        if self._state != key:
            train = mode == 'train'
            seed = int(np.random.randint(0, 2 ** 31 - 1)) if train else 0
            self.engine.forward(train=train, seed=seed)
            self._state = key
        return self

    def activation(self, layer):
        return self.engine.activation(layer)

    # ---- fused device paths (include/rgcn.h: rgcn_train_step_device, rgcn_rank_device)
    def _upload(self, name, arr):
        """Device copy of `arr` in a persistent buffer that only ever grows."""
        arr = np.ascontiguousarray(arr)
        buf = self._dev.get(name)
        if buf is None or buf.nbytes < arr.nbytes:
            if buf is not None:
                buf.free()
            buf = _native.DeviceBuffer(self.engine, max(arr.nbytes, 16))
            self._dev[name] = buf
        if arr.nbytes:
            self.engine.copy_to_device(buf, arr)
        return buf

    def _buffer(self, name, nbytes):
        buf = self._dev.get(name)
        if buf is None or buf.nbytes < nbytes:
            if buf is not None:
                buf.free()
            buf = _native.DeviceBuffer(self.engine, max(nbytes, 16))
            self._dev[name] = buf
        return buf

    def stage(self, graph_edges, batch=None):
        """Feed the NEXT train step while the current one runs: its triples go to the inactive one of two device
        buffer pairs without the host waiting (pinned staging), the message graph on the prefetch stream where
        rgcn_prefetch_graph_device then builds its structures beside the running step, the decoder batch (if the
        negatives are drawn on the device) on the main stream behind that step.  The step that is then asked to
        train on these very arrays (identity) finds everything in place."""
        g = np.ascontiguousarray(graph_edges, dtype=np.int32).reshape(-1, 3)
        if len(g) > self.engine.max_edges:
            return                                   # the step itself reports the error
        slot = getattr(self, "_slot", 0) ^ 1
        gd = self._buffer("graph%d" % slot, g.nbytes)
        self.engine.copy_to_device_async(gd, g, on_prefetch_stream=True)
        self.engine.prefetch_graph_device(gd, len(g))
        bd, nb = None, 0
        if batch is not None:
            b = np.ascontiguousarray(batch, dtype=np.int32).reshape(-1, 3)
            bd, nb = self._buffer("batch%d" % slot, b.nbytes), len(b)
            self.engine.copy_to_device_async(bd, b)
        self._staged = (graph_edges, batch, slot, gd, len(g), bd, nb)

    def _sampled_batch(self, mb, slot_name, on_prefetch_stream):
        """the graph batch of a device-sampled minibatch: drawn into a device buffer by the device sampler (or
        already drawn there by presample_minibatch)"""
        train, size, seed = mb.sample
        pre = getattr(self, "_presampled", None)
        if pre is not None and pre[0] is mb:
            self._presampled = None
            return pre[1], int(size)
        if getattr(self, "_nbr_graph", None) is not train:
            self.engine.neighborhood_reserve(train)        # the training graph moves to the device once
            self._nbr_graph = train
        bd = self._buffer(slot_name, 12 * int(size))
        self.engine.sample_neighborhood_device(int(size), int(seed), bd, on_prefetch_stream=on_prefetch_stream)
        return bd, int(size)

    def presample_minibatch(self, mb):
        """Draw the graph batch of the minibatch AFTER the staged one, on the prefetch stream behind the staged one's
        graph preparation.  Three buffers in turn: the step that runs reads one, the staged preparation read the
        second, this draw writes the third -- and it is ordered, on the prefetch stream, behind a preparation that
        waited for the step which last read that third buffer."""
        if getattr(mb, "sample", None) is None or int(mb.sample[1]) > self.engine.max_edges:
            return
        ring = getattr(self, "_sample_ring", 0)
        self._sample_ring = (ring + 1) % 3
        self._presampled = None
        bd, _ = self._sampled_batch(mb, "sampled%d" % ring, True)
        self._presampled = (mb, bd)

    def stage_minibatch(self, mb):
        """The device-dropout flavour of stage(): ONE upload (the graph batch; it is the edge-dropout input and the
        decoder's positives at once) on the prefetch stream, then the draw of the kept edges and the preparation of
        their graph there, beside the running step (rgcn_prefetch_graph_dropout_device).  A device-sampled minibatch
        has no upload at all: the batch itself is drawn there first (rgcn_sample_neighborhood_device)."""
        slot = getattr(self, "_slot", 0) ^ 1
        if getattr(mb, "sample", None) is not None:
            if int(mb.sample[1]) > self.engine.max_edges:
                return
            bd, nb = self._sampled_batch(mb, "mbatch%d" % slot, True)
        else:
            b = np.ascontiguousarray(mb.batch, dtype=np.int32).reshape(-1, 3)
            if len(b) > self.engine.max_edges:
                return
            bd, nb = self._buffer("mbatch%d" % slot, b.nbytes), len(b)
            self.engine.copy_to_device_async(bd, b, on_prefetch_stream=True)
        self.engine.prefetch_graph_dropout_device(bd, nb, mb.keep, mb.edge_seed)
        self._staged = (mb, None, slot, bd, nb, None, 0)

    def train_step_minibatch(self, mb, reg_param, seed):
        """One iteration from a graph batch resident on the device: edge dropout, negative sampling and the train
        step in one asynchronous call (rgcn_train_step_minibatch_device)."""
        sampled = getattr(mb, "sample", None) is not None
        if sampled:
            nb = int(mb.sample[1])
        else:
            b = np.ascontiguousarray(mb.batch, dtype=np.int32).reshape(-1, 3)
            nb = len(b)
        n = nb * (int(mb.rate) + 1)
        if n == 0:
            raise ValueError("empty decoder batch")
        if nb > self.engine.max_edges:
            raise ValueError("graph batch of %d edges exceeds the context's max_edges %d"
                             % (nb, self.engine.max_edges))
        if n > self._dec_reserved:
            self.engine.decoder_reserve(n)
            self._dec_reserved = n
        st = self._take_staged(mb, None)
        if st is not None:
            bd = st[3]
        elif sampled:
            bd, _ = self._sampled_batch(mb, "mbatch", False)
        else:
            bd = self._upload("mbatch", b)
        xd, yd = self._buffer("X", 12 * n), self._buffer("Y", 4 * n)
        self.engine.train_step_minibatch_device(bd, nb, mb.keep, mb.edge_seed, mb.rate, seed ^ 0x5bd1e995, xd, yd,
                                                seed=seed, reg_param=reg_param)
        self._state = None
        self._graph_version = None
        self.weights_version += 1

    def _take_staged(self, graph_edges, batch):
        st = getattr(self, "_staged", None)
        self._staged = None
        if st is not None and st[0] is graph_edges and st[1] is batch:
            self._slot = st[2]
            return st
        return None

    def configure_optimizer(self, learning_rate, beta1=0.9, beta2=0.999, epsilon=1e-8, max_grad_norm=0.0):
        self.engine.optimizer_config(learning_rate, beta1, beta2, epsilon, max_grad_norm)

    def train_step(self, graph_edges, x, y, reg_param, seed):
        """One update_from_batch (optimize.py:81-88) on the device: graph prep, encoder forward, DistMult
        loss + gradients, encoder backward, clip + Adam.  Asynchronous; `loss()` synchronises."""
        g = np.ascontiguousarray(graph_edges, dtype=np.int32).reshape(-1, 3)
        x = np.ascontiguousarray(x, dtype=np.int32).reshape(-1, 3)
        y = np.ascontiguousarray(y, dtype=np.float32).ravel()
        if len(x) != len(y) or len(x) == 0:
            raise ValueError("decoder batch: X [N,3] and Y [N] must have the same non-zero length")
        if len(g) > self.engine.max_edges:
            raise ValueError("graph batch of %d edges exceeds the context's max_edges %d"
                             % (len(g), self.engine.max_edges))
        if len(x) > self._dec_reserved:
            self.engine.decoder_reserve(len(x))
            self._dec_reserved = len(x)
        st = self._take_staged(graph_edges, None)
        gd = st[3] if st is not None else self._upload("graph", g)
        xd, yd = self._upload("X", x), self._upload("Y", y)
        self.engine.train_step_device(gd, len(g), xd, yd, len(x), seed=seed, reg_param=reg_param)
        self._state = None            # activations now belong to this train step's graph
        self._graph_version = None
        self.weights_version += 1

    def train_step_device_negatives(self, graph_edges, batch, rate, reg_param, seed):
        """As train_step, with the decoder batch (positives + `rate` corruptions each) drawn on the device from
        `batch` (rgcn_negative_sample_device): only the two triple lists cross the PCIe bus."""
        g = np.ascontiguousarray(graph_edges, dtype=np.int32).reshape(-1, 3)
        b = np.ascontiguousarray(batch, dtype=np.int32).reshape(-1, 3)
        n = len(b) * (int(rate) + 1)
        if n == 0:
            raise ValueError("empty decoder batch")
        if len(g) > self.engine.max_edges:
            raise ValueError("graph batch of %d edges exceeds the context's max_edges %d"
                             % (len(g), self.engine.max_edges))
        if n > self._dec_reserved:
            self.engine.decoder_reserve(n)
            self._dec_reserved = n
        st = self._take_staged(graph_edges, batch)
        if st is not None:
            gd, bd = st[3], st[5]
        else:
            gd, bd = self._upload("graph", g), self._upload("batch", b)
        for name, nbytes in (("X", 12 * n), ("Y", 4 * n)):
            buf = self._dev.get(name)
            if buf is None or buf.nbytes < nbytes:
                if buf is not None:
                    buf.free()
                self._dev[name] = _native.DeviceBuffer(self.engine, nbytes)
        xd, yd = self._dev["X"], self._dev["Y"]
        self.engine.negative_sample_device(bd, len(b), rate, seed ^ 0x5bd1e995, xd, yd)
        self.engine.train_step_device(gd, len(g), xd, yd, n, seed=seed, reg_param=reg_param)
        self._state = None
        self._graph_version = None
        self.weights_version += 1

    def loss(self):
        return self.engine.loss()

    def ranks(self, triples, predict_object, filter_ptr, filter_idx, chunk=2048):
        """Raw / filtered ranks on the codes of a test-mode forward over the fed graph."""
        self.forward('test')
        if self._rank_reserved < chunk:
            self.engine.rank_reserve(chunk)
            self._rank_reserved = chunk
        return self.engine.ranks(triples, predict_object, filter_ptr, filter_idx)

    def backward(self, dcodes):
        if self._state is None or self._state[1] != 'train':
            raise RuntimeError("backward() needs a train-mode forward pass on the current graph")
        self.engine.backward(dcodes)
        return self

    def grad(self, name):
        return self.engine.get_grad(name)
