"""Basis-decomposition relational layer `BasisGcn` (reference:
code/encoders/message_gcns/gcn_basis.py; the default of the `gcn_basis` encoder,
model_builder.py:293-294).

W_r = sum_b C[r, b] W_b: message = sum_b C[r,b] (x . W[:, b, :]) (:39-68).  Weights and their creation
order (:15-30): W_forward, W_backward `[d, B, d]` (in, basis, out), W_self `[d, d]`, all
N(0, glorot_variance([d, d])); C_forward, C_backward `[R, B]` ~ N(0, 1); b = 0, never added.
`get_weights()` order (:33-37): W_forward, W_backward, C_forward, C_backward, W_self, b.
"""
from ...common.shared_functions import glorot_variance, make_variable, make_bias
from ...model import Variable
from .message_gcn import MessageGcn


class BasisGcn(MessageGcn):
    KIND = "basis"

    def parse_settings(self):
        self.dropout_keep_probability = float(self.settings['DropoutKeepProbability'])
        self.n_coefficients = int(self.settings['NumberOfBasisFunctions'])

    def create_variables(self):
        d_in, d_out = self.shape[0], self.shape[1]
        type_matrix_shape = (self.relation_count, self.n_coefficients)
        vertex_matrix_shape = (d_in, self.n_coefficients, d_out)
        self_matrix_shape = (d_in, d_out)
        var = glorot_variance([vertex_matrix_shape[0], vertex_matrix_shape[2]])
        self.W_forward = Variable("W_forward", vertex_matrix_shape, make_variable(0, var, vertex_matrix_shape))
        self.W_backward = Variable("W_backward", vertex_matrix_shape, make_variable(0, var, vertex_matrix_shape))
        self.W_self = Variable("W_self", self_matrix_shape, make_variable(0, var, self_matrix_shape))
        self.C_forward = Variable("C_forward", type_matrix_shape, make_variable(0, 1, type_matrix_shape))
        self.C_backward = Variable("C_backward", type_matrix_shape, make_variable(0, 1, type_matrix_shape))
        self.b = Variable("b", (d_out,), make_bias(d_out))

    def engine_variables(self):
        return [(self.W_forward, "W_f"), (self.W_backward, "W_b"), (self.C_forward, "C_f"),
                (self.C_backward, "C_b"), (self.W_self, "W_self"), (self.b, "b")]

    def local_get_weights(self):
        return [self.W_forward, self.W_backward, self.C_forward, self.C_backward, self.W_self, self.b]

    def local_get_regularization(self):
        return 0.0      # the reference multiplies its L2 term by 0.0 (:90-95)
