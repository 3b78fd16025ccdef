"""Block-diagonal relational layer `ConcatGcn` (reference:
code/encoders/message_gcns/gcn_basis_concat.py; selected by Concatenation=Yes,
model_builder.py:291-292).

Per relation and direction a block-diagonal transform of `NumberOfBasisFunctions` blocks of size
`sd = int(d / nb)` (:15): message_i = sum_j W[r, b, i, j] x[b*sd + j] (:46-47).  Weights and their
creation order (:17-27): W_forward, W_backward `[R, nb, sd, sd]`, W_self `[d, d]`, all
N(0, glorot_variance([R, sd])), and b = 0 which the layer never adds.
"""
from ...common.shared_functions import glorot_variance, make_variable, make_bias
from ...model import Variable
from .message_gcn import MessageGcn


class ConcatGcn(MessageGcn):
    KIND = "block"

    def parse_settings(self):
        self.dropout_keep_probability = float(self.settings['DropoutKeepProbability'])
        self.n_coefficients = int(self.settings['NumberOfBasisFunctions'])
        self.submatrix_d = int(self.shape[1] / self.n_coefficients)
        if self.submatrix_d * self.n_coefficients != self.shape[1]:
            raise ValueError("InternalEncoderDimension must be divisible by NumberOfBasisFunctions "
                             "(the reference's reshape at gcn_basis_concat.py:42 silently mis-groups)")

    def create_variables(self):
        vertex_matrix_shape = (self.relation_count, self.n_coefficients, self.submatrix_d, self.submatrix_d)
        self_matrix_shape = tuple(self.shape)
        var = glorot_variance([vertex_matrix_shape[0], vertex_matrix_shape[2]])
        self.W_forward = Variable("W_forward", vertex_matrix_shape, make_variable(0, var, vertex_matrix_shape))
        self.W_backward = Variable("W_backward", vertex_matrix_shape, make_variable(0, var, vertex_matrix_shape))
        self.W_self = Variable("W_self", self_matrix_shape, make_variable(0, var, self_matrix_shape))
        self.b = Variable("b", (self.shape[1],), make_bias(self.shape[1]))

    def engine_variables(self):
        return [(self.W_forward, "W_f"), (self.W_backward, "W_b"), (self.W_self, "W_self"), (self.b, "b")]

    def local_get_weights(self):
        return [self.W_forward, self.W_backward, self.W_self, self.b]

    def local_get_regularization(self):
        return 0.0      # the reference multiplies its L2 term by 0.0 (:85-90)
