"""One rank of the captured-sharded-step check (N ranks on ONE GPU, the collective provided by RGCN_RCCL_LIBRARY =
tests/collective_double's DEVICE-SIDE stand-in, ipc_collective.hip, whose collectives are kernels a stream capture can
record; RGCN_CAPTURE_SHARDED=1).  BASELINE.json configs[4] as written: a hipGraph-captured train step on a
relation-sharded context -- reduce-scatter / all-gather exchanges of both passes, the decoder divided by triples with its
three all-reduces, the replicated-gradient all-reduce and the squared-norm exchange all inside the graph.  A replayed
graph must do exactly what the same call does when issued directly with (captured seed + replay number): loss and every
weight bitwise.  Prints CAPTURED-SHARDED-OK on success."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
from helpers import make_case  # noqa: E402
from relationprediction_amd import _native  # noqa: E402
from relationprediction_amd.sharding import lpt_partition, share_unique_id  # noqa: E402
from sharded_worker import decoder_batch  # noqa: E402


def main():
    kind, nb = sys.argv[1], int(sys.argv[2])
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    V, R, d, L, E = 300, 12, 40, 2, 2000
    params, triples, _, _ = make_case(V, R, d, L, kind, nb, E, seed=5)
    X, Y = decoder_batch(np.random.RandomState(6), triples[:500], V)
    owner = lpt_partition(np.bincount(triples[:, 1], minlength=R), world)
    base = "/tmp/rgcn_captured_id_%s_%d" % (os.environ.get("MASTER_PORT", "0"), os.getppid())
    engines = []
    for tag in ("graph", "direct"):                      # two sharded contexts per rank, a communicator each
        e = _native.Engine(V, R, d, L, kind, nb, max_edges=E, device=0, rank=rank, world=world)
        e.set_params(params)
        e.set_relation_owner(owner)
        e.comm_init(share_unique_id(rank, base + "." + tag, _native.Engine.comm_unique_id, timeout=120))
        e.decoder_reserve(len(X))
        e.optimizer_config(lr=0.01, max_grad_norm=1.0)
        e.t, e.x, e.y = e.to_device(triples), e.to_device(X), e.to_device(Y)
        e.train_step_device(e.t, E, e.x, e.y, len(X), seed=5, reg_param=0.01)     # warm-up: lazy allocations
        e.sync()
        engines.append(e)
    eng, ref = engines
    eng.capture_begin()
    eng.train_step_device(eng.t, E, eng.x, eng.y, len(X), seed=50, reg_param=0.01)
    gid = eng.capture_end()
    for launch in (1, 2, 3):
        eng.graph_launch(gid)
        ref.train_step_device(ref.t, E, ref.x, ref.y, len(X), seed=50 + launch, reg_param=0.01)
        a, b = eng.loss(), ref.loss()
        assert np.isfinite(a) and a == b, (launch, a, b)
    got, want = eng.get_params(), ref.get_params()
    for k in want:
        assert np.array_equal(got[k], want[k]), k
    assert max(float(np.abs(want[k] - params[k]).max()) for k in want) > 0.02          # it trained
    eng.graph_destroy(gid)
    for e in engines:
        for b in (e.t, e.x, e.y):
            b.free()
        e.close()
    if rank == 0:
        for tag in ("graph", "direct"):
            if os.path.exists(base + "." + tag):
                os.remove(base + "." + tag)
        print("CAPTURED-SHARDED-OK world=%d kind=%s" % (world, kind), flush=True)


if __name__ == "__main__":
    main()
