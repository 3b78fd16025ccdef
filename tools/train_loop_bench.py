#!/usr/bin/env python
"""Whole training iterations of the reference-shaped driver at FB15k-237 gcn_block size (settings/gcn_block.exp:
GraphBatchSize 30000, GraphSplitSize 0.5, NegativeSampleRate 10 -> E_g = 15000, N = 330000).  The training graph
is synthetic (272,115 triples drawn from the valid+test histograms would need the reference's files; here:
oracle.synthetic_graph at the same size).  Prints ms per iteration and where the host time goes.

    python tools/train_loop_bench.py [iterations]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402  (only its synthetic graph generator)
from relationprediction_amd import train  # noqa: E402
from relationprediction_amd.common import model_builder, optimizer_parameter_parser, settings_reader  # noqa: E402
from relationprediction_amd.optimization.optimize import build_hip  # noqa: E402

SETTINGS = """[Encoder]
	Name=gcn_basis
	DropoutKeepProbability=0.8
	InternalEncoderDimension=500
	NumberOfBasisFunctions=100
	NumberOfLayers=2
	UseInputTransform=Yes
	UseOutputTransform=No
	Concatenation=Yes
[Decoder]
	Name=bilinear-diag
	RegularizationParameter=0.01
[Shared]
	CodeDimension=500
[Optimizer]
	MaxGradientNorm=1
	ReportTrainLossEvery=100
	MaxIterations=%d
	[Algorithm]
		Name=Adam
		learning_rate=0.01
[General]
	NegativeSampleRate=10
	GraphSplitSize=0.5
	ExperimentName=/tmp/rgcn_train_loop_bench
	GraphBatchSize=30000
[Evaluation]
	Metric=MRR
"""


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 40
    V, R, E = 14541, 237, 272115
    triples = oracle.synthetic_graph(V, R, E, np.random.RandomState(0)).astype(np.int32)
    path = "/tmp/rgcn_train_loop_bench.exp"
    with open(path, "w") as f:
        f.write(SETTINGS % (iters + 5))
    s = settings_reader.read(path)
    general = s['General']
    general.put('EntityCount', V); general.put('RelationCount', R); general.put('EdgeCount', E)
    for part in ('Encoder', 'Decoder'):
        s[part].merge(s['Shared']); s[part].merge(general)
    s['Optimizer'].merge(general)
    encoder = model_builder.build_encoder(s['Encoder'], triples)
    model = model_builder.build_decoder(encoder, s['Decoder'])
    opp = optimizer_parameter_parser.Parser(s['Optimizer'])
    opp.set_save_function(lambda p: None)
    t_func = train.make_transform(triples, general, encoder, device_negatives="--host-negatives" not in sys.argv)
    host_times = []

    def timed(x):
        t0 = time.perf_counter()
        out = t_func(x)
        host_times.append(time.perf_counter() - t0)
        return out

    opp.set_sample_transform_function(timed)
    model.preprocess(triples); model.register_for_test(triples); model.initialize_train()
    opt = build_hip(model, [p for p in opp.get_parametrization() if p[0] != 'ModelSaver'])
    np.random.seed(0)
    # warm-up iterations, then the timed ones
    opt.stack.set_training_data(triples)
    eng = model.get_runtime().engine
    dev_ms = []
    for i in range(iters + 5):
        batch = opt.stack.process_data(opt.stack.next_batch())
        if i == 5:
            t0 = time.perf_counter()
        eng.timer_start()
        opt.update_from_batch(batch, seed=i)
        ms = eng.timer_stop()
        loss = model.device_loss()
        if i >= 5:
            dev_ms.append(ms)
    wall = time.perf_counter() - t0
    print("serial loop: %.2f ms / iteration  (host minibatch construction %.2f ms, device step %.3f ms, loss %.4f)"
          % (wall * 1e3 / iters, np.mean(host_times[5:]) * 1e3, np.mean(dev_ms), loss))
    # per-kernel durations of the device step (side-stream overlap off: exclusive times)
    eng.set_overlap(False)
    eng.profile_reset(); eng.profile_enable(True)
    for i in range(10):
        opt.update_from_batch(batch, seed=100 + i)
    eng.sync()
    prof = sorted(eng.profile(), key=lambda p: -p["total_ms"])
    eng.profile_enable(False); eng.set_overlap(True)
    tot = sum(p["total_ms"] for p in prof) / 10
    print("device step kernels (exclusive, ms per step, total %.3f):" % tot)
    for p_ in prof[:14]:
        print("   %-24s %5.1f launches  %.4f ms" % (p_["name"], p_["calls"] / 10, p_["total_ms"] / 10))
    for workers in (0, 8):
        opp2 = optimizer_parameter_parser.Parser(s['Optimizer'])
        opp2.set_save_function(lambda p: None)
        opp2.set_sample_transform_function(t_func)
        opt2 = build_hip(model, [p for p in opp2.get_parametrization() if p[0] != 'ModelSaver'], batch_workers=workers)
        opt2.stack.next_component  # noqa: B018
        t0 = time.perf_counter()
        n = opt2.fit(triples)
        print("driver loop, %d background batch builders: %.2f ms / iteration over %d iterations"
              % (workers, (time.perf_counter() - t0) * 1e3 / n, n))


if __name__ == "__main__":
    main()
