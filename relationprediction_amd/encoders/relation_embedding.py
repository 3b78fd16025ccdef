"""Relation codes (reference: code/encoders/relation_embedding.py): passes the vertex codes through
and supplies `W_relation ~ N(0,1)` as the relation codes (:15-25).  The reference allocates it with
shape `[EntityCount, CodeDimension]` (model_builder.py:134-135,180-182) and only rows
< RelationCount are ever read; the shape is kept so weight lists stay interchangeable."""
import numpy as np

from ..model import Model, Variable


class RelationEmbedding(Model):
    shape = None

    def __init__(self, shape, settings, next_component=None):
        Model.__init__(self, next_component, settings)
        self.shape = shape

    def parse_settings(self):
        self.embedding_width = int(self.settings['CodeDimension'])

    def local_initialize_train(self):
        relation_initial = np.random.randn(self.shape[0], self.shape[1]).astype(np.float32)
        self.W_relation = Variable("W_relation", tuple(self.shape), relation_initial)

    def local_get_weights(self):
        return [self.W_relation]

    def _bind(self):
        """W_relation lives in the engine (last parameter of rgcn_param_info) so that the device decoder and
        the device optimizer see it; bound on first use, when the stack's runtime exists."""
        if not getattr(self, '_bound', False):
            rt = self.next_component.get_runtime()
            self.W_relation.bind(*rt._accessors("W_relation"))
            self._bound = True

    def get_runtime(self):
        self._bind()
        return self.next_component.get_runtime()

    def get_all_codes(self, mode='train'):
        self._bind()
        codes = self.next_component.get_all_codes(mode=mode)
        return codes[0], self.W_relation.value(), codes[2]

    def backward(self, upstream):
        self._bind()
        dcodes, d_relation = upstream
        return self.next_component.backward(dcodes) + [d_relation]
