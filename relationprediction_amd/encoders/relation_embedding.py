"""Relation codes of the encoder chain (the reference's `encoders/relation_embedding.py`).

The component hands the vertex codes of the layer below through unchanged and contributes the third element of
the `(subject codes, relation codes, object codes)` triple: one trainable vector per relation, drawn from N(0,1)
(reference: relation_embedding.py:15-25).  The reference sizes the table `[EntityCount, CodeDimension]`
(model_builder.py:134-135,180-182) although only the first RelationCount rows are ever gathered; the
shape is kept so that weight lists and checkpoints stay interchangeable.

The table is the engine's last parameter (`W_relation` in rgcn_param_info): the device decoder reads it, the
device optimizer updates it, and this component's `Variable` is a view of it once the stack's runtime exists.
"""
import numpy as np

from ..model import Model, Variable


class RelationEmbedding(Model):
    shape = None
    W_relation = None
    _engine_bound = False

    def __init__(self, shape, settings, next_component=None):
        self.shape = shape
        Model.__init__(self, next_component, settings)

    def parse_settings(self):
        self.embedding_width = int(self.settings['CodeDimension'])

    # ---- weights
    def local_initialize_train(self):
        rows, width = int(self.shape[0]), int(self.shape[1])
        initial = np.random.randn(rows, width).astype(np.float32)      # one numpy draw, as the reference makes
        self.W_relation = Variable("W_relation", (rows, width), initial)

    def local_get_weights(self):
        return [self.W_relation]

    def _attach_to_engine(self):
        if self._engine_bound:
            return
        runtime = self.next_component.get_runtime()
        self.W_relation.bind(*runtime._accessors("W_relation"))         # pushes the initial value once
        self._engine_bound = True

    def get_runtime(self):
        self._attach_to_engine()
        return self.next_component.get_runtime()

    # ---- chain surface
    def get_all_codes(self, mode='train'):
        self._attach_to_engine()
        subject_codes, _, object_codes = self.next_component.get_all_codes(mode=mode)
        return subject_codes, self.W_relation.value(), object_codes

    def backward(self, upstream):
        """`upstream` = (dL/dcodes, dL/dW_relation) from the decoder; the table's gradient is appended last,
        matching its position in get_weights()."""
        self._attach_to_engine()
        d_codes, d_relation = upstream
        return self.next_component.backward(d_codes) + [d_relation]
