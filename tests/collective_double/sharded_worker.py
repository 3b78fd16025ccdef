"""One rank of the multi-process sharding check (launched by torch.distributed.run, N ranks on ONE GPU, the
collective provided by RGCN_RCCL_LIBRARY = tests/collective_double's shared-memory stand-in).  Every rank runs
the library's own sharded paths with a communicator — rgcn_step_device and rgcn_train_step_device on a world > 1
context, i.e. the comm_allreduce call sites the in-process phase-API tests cannot reach — and rank 0 compares
with an unsharded context in its own process.  Prints SHARDED-OK on success."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
from helpers import make_case  # noqa: E402
from relationprediction_amd import _native  # noqa: E402
from relationprediction_amd.sharding import lpt_partition, share_unique_id  # noqa: E402


def decoder_batch(rng, triples, V, neg_rate=3):
    neg = np.tile(triples, (neg_rate, 1))
    side = rng.rand(len(neg)) < 0.5
    rnd = rng.randint(0, V, len(neg))
    neg[side, 2] = rnd[side]
    neg[~side, 0] = rnd[~side]
    X = np.concatenate([triples, neg]).astype(np.int32)
    Y = np.concatenate([np.ones(len(triples)), np.zeros(len(neg))]).astype(np.float32)
    return X, Y


def main():
    kind, nb = sys.argv[1], int(sys.argv[2])
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    V, R, d, L, E = 300, 12, 40, 2, 2000
    params, triples, _, dcodes = make_case(V, R, d, L, kind, nb, E, seed=5)
    X, Y = decoder_batch(np.random.RandomState(6), triples[:500], V)
    owner = lpt_partition(np.bincount(triples[:, 1], minlength=R), world)
    eng = _native.Engine(V, R, d, L, kind, nb, max_edges=E, device=0, rank=rank, world=world)
    eng.set_params(params)
    eng.set_relation_owner(owner)
    path = "/tmp/rgcn_worker_id_%s_%d" % (os.environ.get("MASTER_PORT", "0"), os.getppid())
    eng.comm_init(share_unique_id(rank, path, _native.Engine.comm_unique_id, timeout=120))
    ref = None
    if rank == 0:
        ref = _native.Engine(V, R, d, L, kind, nb, max_edges=E, device=0)
        ref.set_params(params)
    sharded = ("W_f", "W_b") if kind == "block" else ("C_f", "C_b")
    mine = owner == rank

    def near(a, b, tol, what):
        scale = max(float(np.abs(b).max()), 1e-6)
        err = float(np.abs(a - b).max())
        assert err <= tol * scale + 1e-7, (what, err, scale)

    # ---- encoder step (the bench's step): forward + backward with the library's own exchanges
    for e in [eng] + ([ref] if ref else []):
        e.t, e.dc = e.to_device(triples), e.to_device(dcodes)
        e.step_device(e.t, E, e.dc, train=True, seed=11)
    codes, grads = eng.codes(), eng.get_grads()
    if ref:
        near(codes, ref.codes(), 1e-5, "codes")
        rg = ref.get_grads()
        for k, g in rg.items():
            if k.startswith(sharded):
                near(grads[k][mine], g[mine], 2e-4, k)
                assert not grads[k][~mine].any(), k
            else:
                near(grads[k], g, 2e-4, k)

    # ---- whole train steps: decoder replicated, sharded squared-norm exchange, Adam
    for e in [eng] + ([ref] if ref else []):
        e.decoder_reserve(len(X))
        e.optimizer_config(lr=0.01, max_grad_norm=1.0)
        e.x, e.y = e.to_device(X), e.to_device(Y)
        for step in range(3):
            e.train_step_device(e.t, E, e.x, e.y, len(X), seed=30 + step, reg_param=0.01)
    loss = eng.loss()
    got = eng.get_params()
    if ref:
        assert abs(loss - ref.loss()) <= 1e-5 * max(1.0, abs(ref.loss())), (loss, ref.loss())
        want = ref.get_params()
        for k, w in want.items():
            a, b = (got[k][mine], w[mine]) if k.startswith(sharded) else (got[k], w)
            diff = np.abs(a - b)
            assert float(diff.max()) <= 2 * 0.01 * 3 + 1e-6 and float((diff > 2e-4).mean()) <= 0.002, k
        assert max(float(np.abs(want[k] - params[k]).max()) for k in want) > 0.02
    # every rank holds the same replicated weights: compare through one more all-reduce (sum == world * mine)
    w = got["W_self1"].astype(np.float32)
    buf = eng.to_device(w)
    eng.comm_allreduce_sum(buf, w.size)
    total = buf.download(np.float32, w.shape)
    assert float(np.abs(total - world * w).max()) <= 1e-5 * world, "replicated weights diverged across ranks"
    buf.free()
    for e in [eng] + ([ref] if ref else []):
        for b in (e.t, e.dc, e.x, e.y):
            b.free()
        e.close()
    if rank == 0:
        if os.path.exists(path):
            os.remove(path)
        print("SHARDED-OK world=%d kind=%s" % (world, kind), flush=True)


if __name__ == "__main__":
    main()
