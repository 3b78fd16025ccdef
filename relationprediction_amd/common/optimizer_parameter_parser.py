"""Optimizer settings -> ordered description of the Converge stack.

Takes the place of the reference's `common/optimizer_parameter_parser.py`: `train.py` builds a `Parser` from the
merged [Optimizer] settings, registers three callbacks (minibatch transform, checkpoint function, early-stopping
score) and asks for `get_parametrization()`, a list of `(component name, parameter dict)` pairs that
`optimization.optimize.build_hip` turns into the stack.  Which settings key switches which component on, the
parameter names and the ORDER of the list (it decides who wraps whom, hence the order of the per-iteration
reports) are the reference's; the implementation is a table of small rule functions.
"""


def _rule_minibatches(p):
    if 'BatchSize' not in p.settings:
        return None
    return {'batch_size': int(p.settings['BatchSize']), 'contiguous_sampling': False}


def _rule_sample_transformer(p):
    fn = p.sample_transform_function
    return None if fn is None else {'transform_function': fn}


def _rule_iteration_counter(p):
    if 'MaxIterations' not in p.settings:
        return None
    return {'max_iterations': int(p.settings['MaxIterations'])}


def _rule_gradient_clipping(p):
    if 'MaxGradientNorm' not in p.settings:
        return None
    return {'max_norm': float(p.settings['MaxGradientNorm'])}


def _rule_loss_reporter(p):
    if 'ReportTrainLossEvery' not in p.settings:
        return None
    return {'evaluate_every_n': int(p.settings['ReportTrainLossEvery'])}


def _rule_early_stopper(p):
    if 'EarlyStopping' not in p.settings:
        return None
    section = p.settings['EarlyStopping']
    burnin = int(section['BurninPhaseDuration']) if 'BurninPhaseDuration' in section else 0
    return {'criteria': 'score_validation_data',
            'evaluate_every_n': int(section['CheckEvery']),
            'scoring_function': p.early_stopping_score_function,
            'comparator': lambda current, previous: current > previous,     # a score: higher is better
            'burnin': burnin}


def _rule_model_saver(p):
    if 'SaveEveryN' in p.settings:
        every = int(p.settings['SaveEveryN'])
    elif 'EarlyStopping' in p.settings:
        every = int(p.settings['EarlyStopping']['CheckEvery'])      # a checkpoint per validation check
    else:
        every = 1
    return {'save_function': p.save_function, 'model_path': p.settings['ExperimentName'], 'save_every_n': every}


# position in this table = position in the parametrization (additional ops and the algorithm are spliced in
# between gradient clipping and the loss reporter, as in the reference)
_BEFORE_ALGORITHM = (('Minibatches', _rule_minibatches), ('SampleTransformer', _rule_sample_transformer),
                     ('IterationCounter', _rule_iteration_counter), ('GradientClipping', _rule_gradient_clipping))
_AFTER_ALGORITHM = (('TrainLossReporter', _rule_loss_reporter), ('EarlyStopper', _rule_early_stopper),
                    ('ModelSaver', _rule_model_saver))
_RULES = dict(_BEFORE_ALGORITHM + _AFTER_ALGORITHM)


class Parser(object):
    def __init__(self, optimizer_settings):
        self.settings = optimizer_settings
        self.sample_transform_function = None
        self.save_function = None
        self.early_stopping_score_function = None
        self.additional_ops = []

    # ---- callbacks registered by the driver
    def set_sample_transform_function(self, function):
        self.sample_transform_function = function

    def set_save_function(self, function):
        self.save_function = function

    def set_early_stopping_score_function(self, function):
        self.early_stopping_score_function = function

    def set_additional_ops(self, ops):
        self.additional_ops = list(ops)

    # ---- one pair (or None) per component; the reference's method names are kept as entry points
    def _pair(self, name):
        params = _RULES[name](self)
        return None if params is None else (name, params)

    def minibatches(self):
        return self._pair('Minibatches')

    def sample_transform(self):
        return self._pair('SampleTransformer')

    def iteration_counter(self):
        return self._pair('IterationCounter')

    def gradient_clipping(self):
        return self._pair('GradientClipping')

    def train_loss_reporter(self):
        return self._pair('TrainLossReporter')

    def early_stopping(self):
        return self._pair('EarlyStopper')

    def model_saving(self):
        return self._pair('ModelSaver')

    def optimization_algorithm(self):
        """('Adam', {'learning_rate': 0.01, ...}): every key of the [Algorithm] section but Name, as floats."""
        section = self.settings['Algorithm']
        return (section['Name'], {key: float(section[key]) for key in section if key != 'Name'})

    def get_additional_ops(self):
        return [('AdditionalOp', {'op': op}) for op in self.additional_ops]

    def get_parametrization(self):
        pairs = [self._pair(name) for name, _ in _BEFORE_ALGORITHM]
        pairs += self.get_additional_ops()
        pairs.append(self.optimization_algorithm())
        pairs += [self._pair(name) for name, _ in _AFTER_ALGORITHM]
        return [pair for pair in pairs if pair is not None]
