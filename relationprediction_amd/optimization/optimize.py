"""Training loop ("Converge", reference: code/optimization/optimize.py + shared/algorithms.py +
tensorflow_backend/algorithms.py) on top of the fused device train step.

The reference composes the loop from a stack of components, each wrapping the next: data flows down
(`next_batch`, `process_data`), the loss flows back up (`postprocess`), and two of them (GradientClipping, Adam)
rewrite the TF gradient / update ops.  The same components exist here with the same parameters and the same
reporting / stopping behaviour; clipping and Adam become the configuration of the device optimizer
(rgcn_optimizer_config) and `update_from_batch` is ONE asynchronous device call (rgcn_train_step_device).

One deliberate reordering: the host work of the NEXT batch (neighbourhood sampling, negative sampling) is
done while the GPU runs the current step, and the loss is fetched afterwards; per iteration the reference's order
`process_data -> update -> postprocess` is kept, batch i+1 is merely drawn before postprocess(i) runs."""
import random

import numpy as np


class IOptimizer(object):
    next_component = None
    iteration = 0
    validation_data = None
    training_data = None

    def __init__(self, next_component, parameters):
        self.next_component = next_component
        for k, v in parameters.items():
            setattr(self, k, v)

    def valid(self):
        return True

    def verify(self):
        return self.valid() and (self.next_component is None or self.next_component.verify())

    def process_data(self, data):
        return self.next_component.process_data(data)

    def postprocess(self, loss):
        return self.next_component.postprocess(loss) if self.next_component is not None else 'continue'

    def set_iteration(self, iteration):
        self.iteration = iteration
        if self.next_component is not None:
            self.next_component.set_iteration(iteration)

    def next_batch(self):
        return self.next_component.next_batch()

    def set_validation_data(self, validation_data):
        self.validation_data = validation_data
        if self.next_component is not None:
            self.next_component.set_validation_data(validation_data)

    def set_training_data(self, training_data):
        self.training_data = training_data
        if self.next_component is not None:
            self.next_component.set_training_data(training_data)

    def configure_device(self, cfg):
        """Collect what the device optimizer needs (replaces process_gradient/update_function)."""
        if self.next_component is not None:
            self.next_component.configure_device(cfg)


class BaseOptimizer(IOptimizer):
    def __init__(self):
        self.next_component = None

    def next_batch(self):
        return self.training_data

    def process_data(self, data):
        return data

    def postprocess(self, loss):
        return 'continue'


class IterationCounter(IOptimizer):            # shared/algorithms.py:5-19
    max_iterations = None
    iterations = 0

    def valid(self):
        return self.max_iterations is not None

    def next_batch(self):
        if self.iterations < self.max_iterations:
            self.iterations += 1
            return self.next_component.next_batch()
        return None


class Minibatches(IOptimizer):                 # shared/algorithms.py:21-48 (random sampling branch)
    batch_size = None
    contiguous_sampling = None

    def valid(self):
        return self.batch_size is not None and self.contiguous_sampling is False

    def next_batch(self):
        data = self.next_component.next_batch()
        sample = random.sample(range(len(data)), self.batch_size)
        return [data[i] for i in sample]


class SampleTransformer(IOptimizer):           # shared/algorithms.py:51-60
    transform_function = None

    def valid(self):
        return self.transform_function is not None

    def process_data(self, training_data):
        return self.transform_function(self.next_component.process_data(training_data))

    def deferred(self, training_data):
        """(callable, seed) building the same batch later, possibly on another thread — available when the
        transform function is a pure function of (data, seed) (`transform_function.seeded`)."""
        seeded = getattr(self.transform_function, 'seeded', None)
        if seeded is None or not isinstance(self.next_component, BaseOptimizer):
            return None
        seed = int(np.random.randint(0, 2 ** 31 - 1))       # drawn in request order: reproducible
        data = self.next_component.process_data(training_data)
        return lambda: seeded(data, seed)


class GradientClipping(IOptimizer):            # tensorflow_backend/algorithms.py:58-68 (clip_by_global_norm)
    max_norm = None

    def valid(self):
        return self.max_norm is not None

    def configure_device(self, cfg):
        cfg['max_grad_norm'] = float(self.max_norm)
        IOptimizer.configure_device(self, cfg)


class Adam(IOptimizer):                        # tensorflow_backend/algorithms.py:27-42
    learning_rate = None
    historical_moment_weight = 0.9             # beta1
    historical_second_moment_weight = 0.999    # beta2
    epsilon = 1e-8                             # tf.train.AdamOptimizer default

    def valid(self):
        return self.learning_rate is not None

    def configure_device(self, cfg):
        cfg.update(learning_rate=float(self.learning_rate), beta1=float(self.historical_moment_weight),
                   beta2=float(self.historical_second_moment_weight), epsilon=float(self.epsilon))
        IOptimizer.configure_device(self, cfg)


class AdditionalOp(IOptimizer):                # TF update ops of variational layers: none in scope
    op = None


class PostStepHook(IOptimizer):
    """A component that only acts AFTER a step.  The reference's reporting / saving / stopping components all share
    one control flow (shared/algorithms.py:62-159): let the wrapped component see the loss first, give way to its
    'stop', then do the own thing.  Here that flow exists once; a hook states a period (`period_attr` names the
    parameter the settings provide, `phase` = the iteration residue it fires on) and a `fire(loss)` that may
    return 'stop'."""
    period_attr = None
    phase = 0

    def observe(self, loss):
        """every iteration, before the periodic part"""

    def fire(self, loss):
        raise NotImplementedError

    def postprocess(self, loss):
        verdict = self.next_component.postprocess(loss)
        if verdict == 'stop':
            return verdict
        self.observe(loss)
        if self.iteration % getattr(self, self.period_attr) == self.phase and self.fire(loss) == 'stop':
            return 'stop'
        return verdict


class ModelSaver(PostStepHook):                # shared/algorithms.py:62-80
    model_path = None
    save_function = None
    save_every_n = 1
    period_attr = 'save_every_n'

    def valid(self):
        return None not in (self.model_path, self.save_function)

    def fire(self, loss):
        self.save_function(self.model_path)


class TrainLossReporter(PostStepHook):         # shared/algorithms.py:83-115
    evaluate_every_n = 1
    period_attr = 'evaluate_every_n'
    phase = 1                                   # iteration k*n + 1 closes the window (k-1)*n+1 .. k*n (n = 1: never,
                                                # as in the reference: x % 1 is never 1)

    def __init__(self, next_component, parameters):
        PostStepHook.__init__(self, next_component, parameters)
        self._window = 0.0

    def observe(self, loss):
        self._window += loss

    def postprocess(self, loss):
        if self.iteration == 1:                 # the first loss is reported on its own and opens no window
            verdict = self.next_component.postprocess(loss)
            if verdict != 'stop':
                print("Initial loss: " + str(loss))
            return verdict
        return PostStepHook.postprocess(self, loss)

    def fire(self, loss):
        n = self.evaluate_every_n
        mean, self._window = self._window / float(n), 0.0
        print("Average train loss for iteration %s-%s: %s" % (self.iteration - n, self.iteration - 1, mean))


class EarlyStopper(PostStepHook):              # shared/algorithms.py:118-159
    criteria = None
    evaluate_every_n = 1
    burnin = 0
    scoring_function = None
    comparator = None
    period_attr = 'evaluate_every_n'

    def __init__(self, next_component, parameters):
        PostStepHook.__init__(self, next_component, parameters)
        self._last_score = None

    def valid(self):
        return self.criteria == 'score_validation_data' and \
            None not in (self.scoring_function, self.comparator, self.evaluate_every_n)

    def fire(self, loss):
        score = self.scoring_function(self.validation_data)
        print("Tested validation score at iteration %s. Result: %s" % (self.iteration, score))
        worse = self._last_score is not None and not self.comparator(score, self._last_score)
        if worse and self.iteration > self.burnin:
            print("Stopping criterion reached.")
            return 'stop'
        if worse:
            print("Ignoring criterion while in burn-in phase.")
        self._last_score = score


COMPONENTS = {c.__name__: c for c in (IterationCounter, Minibatches, SampleTransformer, GradientClipping, Adam,
                                      AdditionalOp, ModelSaver, TrainLossReporter, EarlyStopper)}


def build_stack(parameters):
    """The reference's __construct_optimizer (optimize.py:203-215): every pair wraps what was built so far, so
    the LAST pair of the list is the outermost component and, because each `postprocess` runs its inner
    neighbour first, the reporting order is TrainLossReporter, EarlyStopper, ModelSaver."""
    stack = BaseOptimizer()
    for name, params in parameters:
        if name not in COMPONENTS:
            raise NotImplementedError("optimizer component '%s' (only Adam is built as Algorithm.Name)" % name)
        stack = COMPONENTS[name](stack, params)
    if not stack.verify():
        raise ValueError("optimizer parameters do not describe a valid stack")
    return stack


class DeviceNegatives(object):
    """A processed batch whose decoder triples are still to be drawn — on the device, at step time — from
    `batch` (NegativeSampler.transform's distribution: rgcn_negative_sample_device)."""

    def __init__(self, graph_edges, batch, rate):
        self.graph_edges, self.batch, self.rate = graph_edges, batch, int(rate)


class DeviceMinibatch(object):
    """A processed batch that leaves BOTH random draws of the reference's t_func to the device step: which `keep` of
    the batch's edges the encoder sees (edge dropout, train.py:233-238) and the corruptions (auxilliaries.py:13-33).
    Only `batch` crosses PCIe (rgcn_train_step_minibatch_device).

    With `sample = (train_triplets, sample_size, sampler_seed)` and `batch = None` the THIRD draw happens on the device
    too -- the neighbourhood edge sampler (train.py:161-198; rgcn_sample_neighborhood_device): nothing is built on the
    host and nothing is uploaded, an iteration is three seeds."""

    def __init__(self, batch, keep, edge_seed, rate, sample=None):
        self.batch, self.keep, self.edge_seed, self.rate = batch, int(keep), int(edge_seed), int(rate)
        self.sample = sample


class HipOptimizer(object):
    """TensorflowOptimizer's role (optimize.py:42-90) on the device train step."""

    def __init__(self, stack, model, batch_workers=0):
        self.stack = stack
        self.model = model
        self.batch_workers = int(batch_workers)
        cfg = {}
        stack.configure_device(cfg)
        if 'learning_rate' not in cfg:
            raise ValueError("the optimizer stack has no Algorithm")
        model.configure_device_optimizer(cfg['learning_rate'], cfg['beta1'], cfg['beta2'], cfg['epsilon'],
                                         cfg.get('max_grad_norm', 0.0))

    def update_from_batch(self, processed_batch, seed):
        """Enqueue one train step; processed_batch = (graph_edges, X, Y) as the transform function returns."""
        if isinstance(processed_batch, DeviceMinibatch):
            self.model.device_train_step_minibatch(processed_batch, seed)
            return
        if isinstance(processed_batch, DeviceNegatives):
            self.model.device_train_step_negatives(processed_batch.graph_edges, processed_batch.batch,
                                                   processed_batch.rate, seed)
            return
        graph_edges, x, y = processed_batch
        self.model.device_train_step(graph_edges, x, y, seed)

    def stage(self, processed_batch):
        """Hand the NEXT batch's triples to the device while the step just enqueued runs (runtime.stage)."""
        stage = getattr(self.model, 'device_stage', None)
        if stage is None:
            return
        if isinstance(processed_batch, DeviceMinibatch):
            self.model.device_stage_minibatch(processed_batch)
        elif isinstance(processed_batch, DeviceNegatives):
            stage(processed_batch.graph_edges, processed_batch.batch)
        else:
            stage(processed_batch[0], None)

    def presample(self, processed_batch):
        """A device-sampled minibatch is a function of its seed alone: draw it a whole iteration before its graph is
        prepared (runtime.presample_minibatch), so that the sampler's chain of short kernels has the length of a step
        to finish in beside the step, instead of standing between the step and the next one."""
        pre = getattr(self.model, 'device_presample_minibatch', None)
        if pre is not None and isinstance(processed_batch, DeviceMinibatch) and processed_batch.sample is not None:
            pre(processed_batch)

    def _sample_transformer(self):
        """The SampleTransformer if it sits directly on the data source (the only place the reference puts it
        when no Minibatches component is configured)."""
        comp = self.stack
        while comp is not None:
            if isinstance(comp, SampleTransformer):
                return comp
            comp = comp.next_component
        return None

    def _batches(self):
        """Generator of processed batches in request order.  With batch_workers > 0 and a seedable transform the
        batches are built by a thread pool up to batch_workers + 1 ahead of the consumer (the neighbourhood
        sampler and most numpy work release the GIL); otherwise each batch is built when asked for."""
        st = self._sample_transformer()
        pool = None
        if self.batch_workers > 0 and st is not None:
            from concurrent.futures import ThreadPoolExecutor
            pool = ThreadPoolExecutor(max_workers=self.batch_workers)
        try:
            pending = []
            exhausted = False

            def request():
                nonlocal exhausted
                if exhausted:
                    return
                nb = self.stack.next_batch()
                if nb is None:
                    exhausted = True
                    return
                job = st.deferred(nb) if pool is not None else None
                pending.append(pool.submit(job) if job is not None else ('inline', nb))
            depth = self.batch_workers + 1 if pool is not None else 1
            for _ in range(depth):
                request()
            while pending:
                head = pending.pop(0)
                if isinstance(head, tuple):
                    yield self.stack.process_data(head[1])
                else:
                    yield head.result()
                request()
        finally:
            if pool is not None:
                pool.shutdown(wait=False)

    def fit(self, training_data, validation_data=None):
        self.stack.set_training_data(training_data)
        if validation_data is not None:
            self.stack.set_validation_data(validation_data)
        i = 0
        batches = self._batches()
        processed = next(batches, None)
        ahead = None                                         # a device-sampled batch fetched one iteration early
        while processed is not None:
            i += 1
            self.stack.set_iteration(i)
            self.update_from_batch(processed, seed=int(np.random.randint(0, 2 ** 31 - 1)))
            # host work of the next iteration (or the wait for a background-built batch) while the device runs
            processed = ahead if ahead is not None else next(batches, None)
            ahead = None
            if processed is not None:
                self.stage(processed)                        # upload + graph prep of the next step, beside this one
                if isinstance(processed, DeviceMinibatch) and processed.sample is not None:
                    ahead = next(batches, None)              # ... and the draw of the batch after it, behind that
                    if ahead is not None:
                        self.presample(ahead)
            train_loss = self.model.device_loss()            # synchronises with the step
            if self.stack.postprocess(train_loss) == 'stop':
                print("Stopping training.")
                break
        batches.close()
        return i


def build_hip(model, parameters, batch_workers=0):
    return HipOptimizer(build_stack(parameters), model, batch_workers=batch_workers)
