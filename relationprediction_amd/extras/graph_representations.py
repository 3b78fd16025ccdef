"""Graph input of the encoder chain (reference: code/extras/graph_representations.py).

`Representation` keeps the reference's constructor and placeholder surface: `graph_edges` is an
int32 `[None,3]` placeholder of (subject, relation, object) rows (:173-174), listed first among the
train / test input variables (:176-180).  The reference's `MessageGraph` (edge split :21-27, the
`[V,E]` incidence matrices with 'global' normalisation :69-147) has no host counterpart any more: the
engine derives the same quantities on the device in rgcn_set_graph (csrc/graph_prep.hip).
"""
import numpy as np

from ..model import Model, Placeholder


class MessageGraph(object):
    """Light view of the fed graph for code that wants the reference's accessors."""

    def __init__(self, edges, vertex_count, label_count):
        self.edges = edges
        self.vertex_count = vertex_count
        self.label_count = label_count

    def _col(self, c):
        v = self.edges.value
        return None if v is None else v[:, c]

    def get_sender_indices(self):
        return self._col(0)

    def get_type_indices(self):
        return self._col(1)

    def get_receiver_indices(self):
        return self._col(2)

    @property
    def edge_count(self):
        v = self.edges.value
        return 0 if v is None else v.shape[0]


class Representation(Model):
    normalization = "global"
    graph = None
    X = None

    def __init__(self, triples, settings, bipartite=False):
        if bipartite:
            raise NotImplementedError("bipartite graph representation (dead code in the reference)")
        self.triples = np.array(triples)
        self.settings = settings
        self.next_component = None
        self.entity_count = settings['EntityCount']
        self.relation_count = settings['RelationCount']
        self.edge_count = self.triples.shape[0] * 2
        self.runtime = None

    def get_graph(self):
        if self.graph is None:
            self.graph = MessageGraph(self.X, self.entity_count, self.relation_count)
        return self.graph

    def local_initialize_train(self):
        self.X = Placeholder('graph_edges', np.int32, ncols=3)

    def local_get_train_input_variables(self):
        return [self.X]

    def local_get_test_input_variables(self):
        return [self.X]

    def backward(self, upstream=None):
        return []
