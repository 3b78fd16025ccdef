"""Plugin base class: the reference's chain-of-responsibility `Model` (code/model.py) without TensorFlow.

Components are chained through `next_component`; a call is answered by the outermost component that
defines it and otherwise delegated inward (`__delegate__`), or run locally AND delegated
(`__local_run_delegate__`), or concatenated along the chain, innermost first
(`__local_expand_delegate__`).  Differences from the reference, all forced by dropping the TF graph:

  * methods that returned symbolic tensors return numpy arrays computed eagerly by the HIP engine
    (the `session.run` boundary of code/model.py:56,69,81 is the C ABI of librgcn.so);
  * inputs are `Placeholder` objects: `get_train_input_variables()` keeps its meaning and order
    `[graph_edges, X, Y]`; the driver does `placeholder.feed(value)` where the reference zipped them
    into a feed_dict (code/optimization/optimize.py:81-88);
  * `backward()` replaces `tf.gradients(loss, weights)` (code/optimization/abstract.py:117-118): it
    returns the gradient list aligned with `get_weights()`.
"""
import os

import numpy as np


class Placeholder(object):
    """Stand-in for tf.placeholder: holds the value fed for the current step."""

    def __init__(self, name, dtype, ncols=None):
        self.name = name
        self.dtype = dtype
        self.ncols = ncols
        self.value = None
        self.version = 0

    def feed(self, value):
        v = np.asarray(value)
        if self.ncols is not None:
            if v.size == 0:
                v = np.zeros((0, self.ncols), dtype=self.dtype)
            if v.ndim != 2 or v.shape[1] != self.ncols:
                raise ValueError("%s expects shape [None,%d], got %s" % (self.name, self.ncols, v.shape))
        self.value = np.ascontiguousarray(v.astype(self.dtype))
        self.version += 1

    def __repr__(self):
        return "<Placeholder %s>" % self.name


class Variable(object):
    """Stand-in for tf.Variable: a named weight.  Engine-resident weights read/write through the
    engine; host weights (W_relation) keep a numpy array."""

    def __init__(self, name, shape, initial=None, getter=None, setter=None):
        self.name = name
        self.shape = tuple(shape)
        self._value = None if initial is None else np.ascontiguousarray(initial, dtype=np.float32)
        self._getter = getter
        self._setter = setter

    def bind(self, getter, setter):
        """Move the weight into the engine; the initial value is pushed once."""
        self._getter, self._setter = getter, setter
        if self._value is not None:
            setter(self._value)
            self._value = None

    def value(self):
        return self._getter() if self._getter is not None else self._value

    def assign(self, new_value):
        new_value = np.ascontiguousarray(new_value, dtype=np.float32)
        if new_value.shape != self.shape:
            raise ValueError("%s: shape %s != %s" % (self.name, new_value.shape, self.shape))
        if self._setter is not None:
            self._setter(new_value)
        else:
            self._value = new_value

    def __repr__(self):
        return "<Variable %s %s>" % (self.name, self.shape)


class Model(object):
    next_component = None
    save_iter = 0

    def __init__(self, next_component, settings):
        self.next_component = next_component
        self.settings = settings
        self.entity_count = int(self.settings['EntityCount'])
        self.relation_count = int(self.settings['RelationCount'])
        self.edge_count = int(self.settings['EdgeCount'])
        self.parse_settings()

    def parse_settings(self):
        pass

    # ---- checkpoint: weight list in get_weights() order (reference: tf.train.Saver, model.py:30-39)
    def save(self, save_path):
        """`.npz` of the weight list (NOT interchangeable with the reference's tf.train.Saver checkpoints):
        key = "<position in get_weights(), 4 digits>_<name>"."""
        weights = self.get_weights()
        print("saving...")
        parent = os.path.dirname(save_path)
        if parent:
            os.makedirs(parent, exist_ok=True)      # e.g. ExperimentName = models/GcnBlock
        np.savez(save_path + "-" + str(self.save_iter) + ".npz",
                 **{"%04d_%s" % (i, w.name): w.value() for i, w in enumerate(weights)})
        self.save_iter += 1

    def load(self, npz_path):
        with np.load(npz_path) as z:
            stored = [z[k] for k in sorted(z.files, key=lambda k: int(k.split("_", 1)[0]))]
        weights = self.get_weights()
        if len(stored) != len(weights):
            raise ValueError("checkpoint has %d tensors, model has %d" % (len(stored), len(weights)))
        for w, v in zip(weights, stored):
            w.assign(v)

    # ---- high-level scoring (model.py:46-81): encode the graph, then the decoder's predict_*
    def _feed_test(self, graph, triplets):
        variables = self.get_test_input_variables()
        if self.needs_graph():
            variables[0].feed(graph)
            variables[1].feed(triplets)
        else:
            variables[0].feed(triplets)

    def score(self, triplets):
        self._feed_test(self.train_triplets, triplets)
        return self.predict()

    def score_all_subjects(self, triplets):
        self._feed_test(self.test_graph, triplets)
        return self.predict_all_subject_scores()

    def score_all_objects(self, triplets):
        self._feed_test(self.test_graph, triplets)
        return self.predict_all_object_scores()

    def register_for_test(self, triplets):
        self.test_graph = triplets

    def preprocess(self, triplets):
        self.train_triplets = triplets

    # ---- chain-wide operations
    def initialize_train(self):
        return self.__local_run_delegate__('initialize_train')

    def get_weights(self):
        return self.__local_expand_delegate__('get_weights')

    def get_train_input_variables(self):
        return self.__local_expand_delegate__('get_train_input_variables')

    def get_test_input_variables(self):
        return self.__local_expand_delegate__('get_test_input_variables')

    def get_regularization(self):
        return self.__local_expand_delegate__('get_regularization', base=0)

    def get_additional_ops(self):
        return self.__local_expand_delegate__('get_additional_ops')

    def get_loss(self, mode='train'):
        return self.__delegate__('get_loss', mode)

    def get_all_subject_codes(self, mode='train'):
        return self.__delegate__('get_all_subject_codes', mode)

    def get_all_object_codes(self, mode='train'):
        return self.__delegate__('get_all_object_codes', mode)

    def get_all_codes(self, mode='train'):
        return self.__delegate__('get_all_codes', mode)

    def predict(self):
        return self.__delegate__('predict')

    def predict_all_subject_scores(self):
        return self.__delegate__('predict_all_subject_scores')

    def predict_all_object_scores(self):
        return self.__delegate__('predict_all_object_scores')

    def get_graph(self):
        return self.__delegate__('get_graph')

    def configure_device_optimizer(self, *args, **kwargs):
        return self.__delegate__('configure_device_optimizer', *args, **kwargs)

    def device_train_step(self, *args):
        return self.__delegate__('device_train_step', *args)

    def device_train_step_negatives(self, *args):
        return self.__delegate__('device_train_step_negatives', *args)

    def device_stage(self, *args):
        return self.__delegate__('device_stage', *args)

    def device_train_step_minibatch(self, *args):
        return self.__delegate__('device_train_step_minibatch', *args)

    def device_stage_minibatch(self, *args):
        return self.__delegate__('device_stage_minibatch', *args)

    def device_presample_minibatch(self, *args):
        return self.__delegate__('device_presample_minibatch', *args)

    def device_loss(self):
        return self.__delegate__('device_loss')

    def device_ranks(self, *args):
        return self.__delegate__('device_ranks', *args)

    def get_runtime(self):
        """The EncoderRuntime (one HIP engine context) the chain's graph-convolution stack runs on."""
        return self.__delegate__('get_runtime')

    def needs_graph(self):
        return False if self.next_component is None else self.next_component.needs_graph()

    def backward(self, upstream=None):
        """Gradients of (loss + regularisation) w.r.t. get_weights(), same order.  Each component
        consumes the gradient its consumer hands down (`upstream`), contributes the gradients of
        its own weights and passes the rest inward."""
        return self.__delegate__('backward', upstream)

    # ---- delegation primitives (same semantics as model.py:147-182)
    def __delegate__(self, name, *args, **kwargs):
        if self.next_component is None:
            return None
        return getattr(self.next_component, name)(*args, **kwargs)

    def __local_run_delegate__(self, name, *args):
        local = getattr(self, 'local_' + name, None)
        if local is not None:
            local(*args)
        if self.next_component is not None:
            getattr(self.next_component, name)(*args)

    def __local_expand_delegate__(self, name, *args, base=None):
        local = getattr(self, 'local_' + name, None)
        mine = local(*args) if local is not None else ([] if base is None else base)
        if self.next_component is None:
            return mine
        return getattr(self.next_component, name)(*args) + mine
