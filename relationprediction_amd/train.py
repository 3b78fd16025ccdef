"""Training driver with the reference's command line (code/train.py):

    python -m relationprediction_amd.train --settings settings/gcn_block.exp --dataset data/FB-Toutanova

Same stages in the same order: read the dataset (:21-47), merge the settings sections (:66-86), build the
encoder / decoder chain (:92-93), the scorer with validation MRR as early-stopping score (:99-128), the
minibatch transform (neighbourhood graph batch -> edge dropout split -> negative sampling, :133-247), and the
Converge loop (:255-284).  What runs where: the chain's components configure ONE HIP engine context; every
iteration is one asynchronous device step (graph prep, R-GCN forward, DistMult loss, backward, clip, Adam);
the host draws the next minibatch meanwhile (neighbourhood sampler in librgcn.so, O(log V) per pick); validation
encodes the training graph once and ranks on the device."""
import argparse

import numpy as np

from . import _native
from .common import settings_reader, io, model_builder, optimizer_parameter_parser, evaluation, auxilliaries
from .optimization.optimize import build_hip


def load_dataset(dataset, metric='MRR'):
    """code/train.py:21-47.  Metric 'Accuracy' reads valid_accuracy.txt / test_accuracy.txt instead of valid.txt /
    test.txt (:32-35).  A metric the reference does not have is refused HERE, at start-up, not at the first
    early-stopping check thousands of iterations in."""
    if metric not in ('MRR', 'Accuracy'):
        raise NotImplementedError("Evaluation.Metric = %r: the reference has 'MRR' and 'Accuracy'" % (metric,))
    relations_path = dataset + '/relations.dict'
    entities_path = dataset + '/entities.dict'
    splits = {}
    for name in ('train', 'valid', 'test'):
        path = dataset + '/' + name + ('_accuracy' if metric == 'Accuracy' and name != 'train' else '') + '.txt'
        splits[name] = np.array(io.read_triplets_as_list(path, entities_path, relations_path), dtype=np.int32)
    return splits, io.read_dictionary(entities_path), io.read_dictionary(relations_path)


def make_transform(train_triplets, general_settings, encoder, device_negatives=False, device_dropout=False,
                   device_sampler=False):
    """The reference's t_func (train.py:201-247): minibatch -> (graph_split, X, Y).

    Every batch is a function of ONE seed drawn from numpy's global generator when the batch is requested, so
    batches can be built ahead of time by background threads (`t_func.seeded(data, seed)`, used by
    optimization.optimize.HipOptimizer) and still come out in a reproducible order; `t_func(data)` itself
    draws the seed and builds the batch in place, like the reference's function."""
    import threading
    ns = auxilliaries.NegativeSampler(int(general_settings['NegativeSampleRate']), general_settings['EntityCount'])
    ns.set_known_positives(train_triplets)
    use_sampler = 'GraphBatchSize' in general_settings
    if use_sampler and int(general_settings['GraphBatchSize']) > len(train_triplets):
        # the reference's sampler runs out of edges and dies on NaN probabilities (train.py:173-178)
        raise ValueError("General.GraphBatchSize = %d exceeds the %d training edges (EdgeCount): remove the key to "
                         "train on the whole graph, or lower it" % (int(general_settings['GraphBatchSize']),
                                                                    len(train_triplets)))
    local = threading.local()            # the native sampler keeps per-sample state: one per thread
    train_arr = np.ascontiguousarray(train_triplets, dtype=np.int32)     # (what the device sampler keeps resident)

    def seeded(x, seed):
        rng = np.random.RandomState(seed)
        arr = np.asarray(x)
        if not encoder.needs_graph():
            return ns.transform(arr, rng)
        if use_sampler and device_sampler and device_dropout:
            # all three draws on the device: the iteration is three seeds, nothing is built on the host or uploaded
            from .optimization.optimize import DeviceMinibatch
            size = int(general_settings['GraphBatchSize'])
            split_size = int(float(general_settings['GraphSplitSize']) * size)
            sample = (train_arr, size, rng.randint(0, 2 ** 31 - 1))
            return DeviceMinibatch(None, split_size, rng.randint(0, 2 ** 31 - 1), ns.negative_sample_rate, sample=sample)
        if use_sampler:
            if not hasattr(local, 'sampler'):
                local.sampler = _native.NeighborhoodSampler(train_triplets, int(general_settings['EntityCount']))
            graph_batch_ids = local.sampler.sample(int(general_settings['GraphBatchSize']), rng.randint(0, 2 ** 31 - 1))
        else:
            graph_batch_ids = np.arange(arr.shape[0])
        graph_batch = train_triplets[graph_batch_ids]
        # edge dropout: the encoder sees a random GraphSplitSize fraction of the batch (exact-k, :235-238)
        split_size = int(float(general_settings['GraphSplitSize']) * graph_batch.shape[0])
        if device_dropout:            # both draws (kept edges, corruptions) are made by the device step
            from .optimization.optimize import DeviceMinibatch
            return DeviceMinibatch(graph_batch, split_size, rng.randint(0, 2 ** 31 - 1), ns.negative_sample_rate)
        graph_split_ids = rng.choice(graph_batch_ids, size=split_size, replace=False)
        graph_split = train_triplets[graph_split_ids]
        if device_negatives:          # corruptions are drawn by the device step (optimize.DeviceNegatives)
            from .optimization.optimize import DeviceNegatives
            return DeviceNegatives(graph_split, graph_batch, ns.negative_sample_rate)
        t = ns.transform(graph_batch, rng)
        return (graph_split, t[0], t[1])

    def t_func(x):
        return seeded(x, np.random.randint(0, 2 ** 31 - 1))

    t_func.seeded = seeded
    return t_func


def main(argv=None):
    parser = argparse.ArgumentParser(description="Train a model on a given dataset.")
    parser.add_argument("--settings", help="Filepath for settings file.", required=True)
    parser.add_argument("--dataset", help="Filepath for dataset.", required=True)
    parser.add_argument("--max-iterations", type=int, default=None,
                        help="stop after this many iterations (sets Optimizer.MaxIterations)")
    parser.add_argument("--host-negatives", action="store_true",
                        help="draw the negative samples with the reference's numpy code on the host instead of on the device")
    parser.add_argument("--host-edge-dropout", action="store_true",
                        help="choose the GraphSplitSize subset of each graph batch with numpy on the host (the "
                             "reference's np.random.choice) instead of on the device")
    parser.add_argument("--host-sampler", action="store_true",
                        help="draw the neighbourhood graph batch with the host sampler (librgcn.so's O(log V) port of "
                             "sample_edge_neighborhood, built ahead by --batch-workers threads) instead of on the device")
    parser.add_argument("--batch-workers", type=int, default=8,
                        help="background threads that build minibatches ahead of the device (0: build in line)")
    args = parser.parse_args(argv)

    settings = settings_reader.read(args.settings)
    print(settings)

    encoder_settings = settings['Encoder']
    decoder_settings = settings['Decoder']
    shared_settings = settings['Shared']
    general_settings = settings['General']
    optimizer_settings = settings['Optimizer']
    evaluation_settings = settings['Evaluation']

    splits, entities, relations = load_dataset(args.dataset, evaluation_settings['Metric'])
    train_triplets, valid_triplets, test_triplets = splits['train'], splits['valid'], splits['test']

    general_settings.put('EntityCount', len(entities))
    general_settings.put('RelationCount', len(relations))
    general_settings.put('EdgeCount', len(train_triplets))
    encoder_settings.merge(shared_settings)
    encoder_settings.merge(general_settings)
    decoder_settings.merge(shared_settings)
    decoder_settings.merge(general_settings)
    optimizer_settings.merge(general_settings)
    evaluation_settings.merge(general_settings)
    if args.max_iterations is not None:
        optimizer_settings.put('MaxIterations', args.max_iterations)

    encoder = model_builder.build_encoder(encoder_settings, train_triplets)
    model = model_builder.build_decoder(encoder, decoder_settings)

    opp = optimizer_parameter_parser.Parser(optimizer_settings)
    opp.set_save_function(model.save)

    scorer = evaluation.Scorer(evaluation_settings)
    scorer.register_data(train_triplets)
    scorer.register_data(valid_triplets)
    scorer.register_data(test_triplets)
    scorer.register_degrees(train_triplets)
    scorer.register_model(model)
    scorer.finalize_frequency_computation(np.concatenate((train_triplets, valid_triplets, test_triplets), axis=0))

    def score_validation_data(validation_data):
        score_summary = scorer.compute_scores(validation_data, verbose=False).get_summary()
        lookup_string = (score_summary.mrr_string() if evaluation_settings['Metric'] == 'MRR'
                         else score_summary.accuracy_string())               # code/train.py:116-121
        early_stopping = score_summary.results['Filtered'][lookup_string]
        score_summary = scorer.compute_scores(test_triplets, verbose=False).get_summary()
        score_summary.pretty_print()
        return early_stopping

    opp.set_early_stopping_score_function(score_validation_data)
    print(len(train_triplets))

    if 'NegativeSampleRate' in general_settings:
        opp.set_sample_transform_function(make_transform(train_triplets, general_settings, encoder,
                                                         device_negatives=not args.host_negatives,
                                                         device_dropout=not (args.host_negatives or
                                                                             args.host_edge_dropout),
                                                         device_sampler=not args.host_sampler))

    model.preprocess(train_triplets)
    model.register_for_test(train_triplets)
    model.initialize_train()
    print(model.get_train_input_variables())

    optimizer = build_hip(model, opp.get_parametrization(), batch_workers=args.batch_workers)
    iterations = optimizer.fit(train_triplets, validation_data=valid_triplets)
    return model, iterations


if __name__ == '__main__':
    main()
