"""Input embedding layer (reference: code/encoders/affine_transform.py).

With `onehot_input=True` the layer is a lookup table plus bias plus relu: `H0 = relu(W + b)`
(:63-83), `W ~ N(0, glorot_variance(shape))`, `b = 0` (:24-28).  That is the only configuration the
BASELINE settings build (UseInputTransform=Yes, model_builder.py:141-146) and the only one wired to
the engine (kernel k_input_fwd).  The dense branch (`matmul(code, W)`, used only by
UseOutputTransform=Yes) is out of scope.
"""
from ..common.shared_functions import glorot_variance, make_variable, make_bias
from ..model import Model, Variable


class AffineTransform(Model):
    W = None
    b = None
    use_nonlinearity = False
    onehot_input = False
    use_bias = True
    shape = None

    def __init__(self, shape, settings, next_component=None, use_nonlinearity=False, onehot_input=False,
                 use_bias=True):
        Model.__init__(self, next_component, settings)
        self.shape = shape
        self.use_nonlinearity = use_nonlinearity
        self.use_bias = use_bias
        self.onehot_input = onehot_input
        if not (onehot_input and use_bias and use_nonlinearity):
            raise NotImplementedError(
                "AffineTransform is implemented as the encoder's input layer only "
                "(onehot_input=True, use_bias=True, use_nonlinearity=True); the dense / bias-free "
                "variants belong to UseOutputTransform=Yes and the 'embedding' encoder (out of scope)")

    def local_initialize_train(self):
        variance = glorot_variance(self.shape)
        self.W = Variable("W_emb", self.shape, make_variable(0, variance, tuple(self.shape)))
        self.b = Variable("b_emb", (self.shape[1],), make_bias(self.shape[1]))

    def local_get_weights(self):
        return [self.W, self.b]

    def _runtime(self):
        rep = self.next_component
        if rep.runtime is None:
            raise RuntimeError("the encoder has no graph-convolution layer above the input layer")
        return rep.runtime

    def get_all_codes(self, mode='train'):
        h = self._runtime().forward(mode).activation(0)
        return h, None, h

    def get_all_subject_codes(self, mode='train'):
        return self.get_all_codes(mode)[0]

    def get_all_object_codes(self, mode='train'):
        return self.get_all_codes(mode)[2]

    def backward(self, runtime):
        return self.next_component.backward(runtime) + [runtime.grad("W_emb"), runtime.grad("b_emb")]
