"""DistMult decoder (reference: code/decoders/bilinear_diag.py).

energy = sum_k e1_k r_k e2_k over the gathered codes (:18-21,30); loss = mean sigmoid cross-entropy
with pos_weight forced to 1 (:32-34); regulariser = RegularizationParameter x (mean(e1^2) + mean(r^2) +
mean(e2^2)) (:63-69); scoring against every entity as `sigmoid(codes . (r*e2)^T)` (:46-61).

BASELINE.json: "the DistMult decoder and negative-sampling loss stay as-is".  The call-by-call surface of
the reference (get_loss, predict_*, backward) is evaluated eagerly in numpy on codes the engine computed;
the training driver and the scorer use the fused device paths below (csrc/decoder.hip,
csrc/optimizer.hip, csrc/ranking.hip).
"""
import numpy as np

from ..model import Model, Placeholder


def _sigmoid(x):
    x = np.asarray(x, dtype=np.float32)
    e = np.exp(-np.abs(x))                       # never overflows
    return np.where(x >= 0, 1.0 / (1.0 + e), e / (1.0 + e)).astype(np.float32)


class BilinearDiag(Model):
    X = None
    Y = None

    def parse_settings(self):
        self.regularization_parameter = float(self.settings['RegularizationParameter'])

    def local_initialize_train(self):
        self.Y = Placeholder('Y', np.float32)
        self.X = Placeholder('X', np.int32, ncols=3)

    def local_get_train_input_variables(self):
        return [self.X, self.Y]

    def local_get_test_input_variables(self):
        return [self.X]

    def compute_codes(self, mode='train'):
        subject_codes, relation_codes, object_codes = self.next_component.get_all_codes(mode=mode)
        x = self.X.value
        return subject_codes[x[:, 0]], relation_codes[x[:, 1]], object_codes[x[:, 2]]

    def get_loss(self, mode='train'):
        e1s, rs, e2s = self.compute_codes(mode=mode)
        energies = np.sum(e1s * rs * e2s, axis=1)
        z = self.Y.value
        # tf.nn.weighted_cross_entropy_with_logits(targets=z, logits=x, pos_weight=1)
        per = (1 - z) * energies + np.log1p(np.exp(-np.abs(energies))) + np.maximum(-energies, 0)
        return float(np.mean(per, dtype=np.float64))

    def local_get_regularization(self):
        e1s, rs, e2s = self.compute_codes(mode='train')
        reg = np.mean(np.square(e1s), dtype=np.float64) + np.mean(np.square(rs), dtype=np.float64) \
            + np.mean(np.square(e2s), dtype=np.float64)
        return self.regularization_parameter * float(reg)

    def predict(self):
        e1s, rs, e2s = self.compute_codes(mode='test')
        return _sigmoid(np.sum(e1s * rs * e2s, axis=1))

    def predict_all_subject_scores(self):
        e1s, rs, e2s = self.compute_codes(mode='test')
        all_subject_codes = self.next_component.get_all_subject_codes(mode='test')
        return _sigmoid(np.matmul(all_subject_codes, (rs * e2s).T).T)

    def predict_all_object_scores(self):
        e1s, rs, e2s = self.compute_codes(mode='test')
        all_object_codes = self.next_component.get_all_object_codes(mode='test')
        return _sigmoid(np.matmul(e1s * rs, all_object_codes.T))

    # ---- fused device paths: what the training driver and the scorer use (the eager numpy methods above
    # keep the reference's call-by-call surface and serve as their small-size cross-check)
    def configure_device_optimizer(self, learning_rate, beta1=0.9, beta2=0.999, epsilon=1e-8, max_grad_norm=0.0):
        self.next_component.get_runtime().configure_optimizer(learning_rate, beta1, beta2, epsilon, max_grad_norm)

    def device_train_step(self, graph_edges, x, y, seed):
        """loss + regularisation, all gradients, clip and Adam in one asynchronous device step."""
        self.next_component.get_runtime().train_step(graph_edges, x, y, self.regularization_parameter, seed)

    def device_train_step_negatives(self, graph_edges, batch, rate, seed):
        """device_train_step with the negatives drawn on the device from `batch`."""
        self.next_component.get_runtime().train_step_device_negatives(graph_edges, batch, rate,
                                                                      self.regularization_parameter, seed)

    def device_stage(self, graph_edges, batch=None):
        """feed the next train step beside the running one (runtime.stage)"""
        self.next_component.get_runtime().stage(graph_edges, batch)

    def device_train_step_minibatch(self, minibatch, seed):
        """the whole iteration from the graph batch on: edge dropout, negatives, train step, all on the device"""
        self.next_component.get_runtime().train_step_minibatch(minibatch, self.regularization_parameter, seed)

    def device_stage_minibatch(self, minibatch):
        self.next_component.get_runtime().stage_minibatch(minibatch)

    def device_presample_minibatch(self, minibatch):
        self.next_component.get_runtime().presample_minibatch(minibatch)

    def device_loss(self):
        return self.next_component.get_runtime().loss()

    def device_ranks(self, graph, triplets, predict_object, filter_ptr, filter_idx):
        """Ranks of the gold subjects / objects of `triplets` against every entity, codes from a test-mode
        pass over `graph` (what score_all_subjects / score_all_objects + MrrScore.append_line compute)."""
        variables = self.get_test_input_variables()
        if getattr(self, '_ranks_graph', None) is not graph or variables[0].value is None:
            variables[0].feed(graph)          # re-encoded only when the graph or the weights changed
            self._ranks_graph = graph
        return self.next_component.get_runtime().ranks(triplets, predict_object, filter_ptr, filter_idx)

    def backward(self, upstream=None):
        """d(loss + regularisation)/d(codes, W_relation), then down the chain."""
        subject_codes, relation_codes, object_codes = self.next_component.get_all_codes(mode='train')
        x, z = self.X.value, self.Y.value
        e1s, rs, e2s = subject_codes[x[:, 0]], relation_codes[x[:, 1]], object_codes[x[:, 2]]
        n, d = e1s.shape
        energies = np.sum(e1s * rs * e2s, axis=1)
        dx = ((_sigmoid(energies) - z) / n).astype(np.float32)[:, None]
        k = np.float32(self.regularization_parameter * 2.0 / (n * d))
        g_e1, g_r, g_e2 = dx * (rs * e2s) + k * e1s, dx * (e1s * e2s) + k * rs, dx * (e1s * rs) + k * e2s
        dcodes = np.zeros_like(subject_codes)
        np.add.at(dcodes, x[:, 0], g_e1)
        np.add.at(dcodes, x[:, 2], g_e2)
        d_rel = np.zeros_like(relation_codes)
        np.add.at(d_rel, x[:, 1], g_r)
        return self.next_component.backward((dcodes, d_rel))
