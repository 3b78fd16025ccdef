"""Message-passing graph-convolution layer template (reference:
code/encoders/message_gcns/message_gcn.py).

Reference template (compute_vertex_embeddings, :49-79): gather sender / receiver features, compute
forward / backward messages, compute the self-loop term, drop out ONLY the self-loop term in train
mode (:60-64), combine.  All of it runs inside the engine (rgcn_forward); a layer object here
contributes its hyper-parameters and weights and reads back its own activation.
"""
from ...model import Model
from ...runtime import EncoderRuntime


class MessageGcn(Model):
    KIND = None
    onehot_input = True
    use_nonlinearity = True

    def __init__(self, shape, settings, next_component=None, onehot_input=False, use_nonlinearity=True):
        self.onehot_input = onehot_input
        self.use_nonlinearity = use_nonlinearity
        self.shape = shape
        self.layer_index = None
        Model.__init__(self, next_component, settings)

    def needs_graph(self):
        return True

    # ---- engine binding
    def engine_variables(self):
        """[(Variable, base name in rgcn_param_info)] of this layer."""
        raise NotImplementedError

    def _runtime(self):
        """The runtime shared by the whole stack; built on first use from the TOP layer's view."""
        below, comp = [self], self.next_component
        while isinstance(comp, MessageGcn):
            below.append(comp)
            comp = comp.next_component
        affine = comp
        rep = affine.next_component
        if rep.runtime is None:
            # `self` may be an inner layer asked directly; the stack always starts at the top layer,
            # which registered itself on the representation when the chain was initialised
            top = getattr(rep, '_top_gcn', self)
            layers, c = [], top
            while isinstance(c, MessageGcn):
                layers.append(c)
                c = c.next_component
            rep.runtime = EncoderRuntime(list(reversed(layers)), affine, rep)
        return rep.runtime

    def local_initialize_train(self):
        self.create_variables()
        # the first layer initialised is the outermost one (model.py:156-164): remember it
        comp = self.next_component
        while isinstance(comp, MessageGcn):
            comp = comp.next_component
        rep = comp.next_component
        if not hasattr(rep, '_top_gcn'):
            rep._top_gcn = self

    def get_runtime(self):
        return self._runtime()

    # ---- reference surface
    def get_all_codes(self, mode='train'):
        collected_messages = self.compute_vertex_embeddings(mode=mode)
        return collected_messages, None, collected_messages

    def compute_vertex_embeddings(self, mode='train'):
        rt = self._runtime().forward(mode)
        return rt.activation(self.layer_index)

    def get_all_subject_codes(self, mode='train'):
        return self.compute_vertex_embeddings(mode=mode)

    def get_all_object_codes(self, mode='train'):
        return self.compute_vertex_embeddings(mode=mode)

    def backward(self, upstream):
        """Top layer: `upstream` is dL/dcodes [V,d] and triggers the engine's backward pass over the
        whole stack; inner layers receive the runtime and only report their own gradients."""
        if isinstance(upstream, EncoderRuntime):
            rt = upstream
        else:
            rt = self._runtime().backward(upstream)
        return self.next_component.backward(rt) + [rt.grad("%s%d" % (base, self.layer_index))
                                                   for _, base in self.engine_variables()]
