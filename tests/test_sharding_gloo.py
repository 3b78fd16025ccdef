"""The N>1 path on CPU: relation sharding + sum-all-reduce reproduces the unsharded encoder.
world_size-2 `gloo` process group (torch.distributed), per-rank partials computed by the oracle
restricted to the rank's relations and self-loop rows -- exactly what each GPU rank computes
(rgcn_forward_layer_partial / rgcn_backward_layer_partial)."""
import os
import socket
import sys

import numpy as np
import pytest

import oracle
from helpers import make_case
from relationprediction_amd.sharding import lpt_partition, row_shard, shard_imbalance

torch = pytest.importorskip("torch")
import torch.distributed as dist  # noqa: E402
import torch.multiprocessing as mp  # noqa: E402


def test_lpt_partition_properties():
    counts = np.array([2675, 2437, 1894, 1604, 1551, 1381, 1254, 1140, 900, 813] + [50] * 200)
    for world in (1, 2, 4, 8):
        owner = lpt_partition(counts, world)
        assert owner.shape == counts.shape and owner.min() >= 0 and owner.max() < world
        assert shard_imbalance(counts, owner, world) < 1.05
        np.testing.assert_array_equal(owner, lpt_partition(counts, world))   # deterministic
    # WN18-like: 18 relations dominated by three
    wn = np.array([2425, 2389, 2152, 546, 526, 343, 326, 229, 223, 219, 150, 120, 100, 90, 80, 50, 20, 12])
    assert shard_imbalance(wn, lpt_partition(wn, 4), 4) < 1.08
    assert row_shard(10, 0, 3) == (0, 3) and row_shard(10, 2, 3) == (6, 10)


def partial_forward(params, triples, owner, rank, world, V, l, L, kind, Hin, masks, norm_mode="intended"):
    """What rank `rank` contributes to layer l's pre-activation."""
    s, r, o = oracle.split_graph(triples)
    mine = owner[r] == rank
    n_f = oracle.incidence_values(o, V, norm_mode)       # degrees are GLOBAL
    n_b = oracle.incidence_values(s, V, norm_mode)
    W_f, W_b, W_self = params[f"W_f{l}"], params[f"W_b{l}"], params[f"W_self{l}"]
    if kind == "block":
        F, K = oracle.concat_messages(Hin, s[mine], r[mine], o[mine], W_f, W_b)
    else:
        F, K = oracle.basis_messages(Hin, s[mine], r[mine], o[mine], W_f, W_b, params[f"C_f{l}"], params[f"C_b{l}"])
    part = np.zeros_like(Hin)
    np.add.at(part, o[mine], F * n_f[mine][:, None])
    np.add.at(part, s[mine], K * n_b[mine][:, None])
    lo, hi = row_shard(V, rank, world)
    S = oracle.dropout(oracle.self_loop(Hin[lo:hi], W_self), 0.8, masks[l - 1][lo:hi])
    part[lo:hi] += S
    return part


def _worker(rank, world, port, kind, nb, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        V, R, d, L, E = 40, 7, 12, 2, 150
        params, triples, masks, dcodes = make_case(V, R, d, L, kind, nb, E, seed=5)
        owner = lpt_partition(np.bincount(triples[:, 1], minlength=R), world)
        H = oracle.affine_onehot_forward(params["W_emb"], params["b_emb"])
        for l in range(1, L + 1):
            part = torch.from_numpy(partial_forward(params, triples, owner, rank, world, V, l, L, kind, H, masks))
            dist.all_reduce(part)                                   # the RCCL all-reduce's stand-in
            pre = part.numpy()
            H = np.maximum(pre, 0) if l < L else pre                # relu AFTER the reduce
        ref = oracle.encoder_forward(params, triples, V, L, kind, mode="train", dropout_masks=masks)[-1]
        ret[rank] = float(np.abs(H - ref).max())
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("kind,nb", [("block", 3), ("basis", 2)])
def test_sharded_forward_equals_unsharded_gloo(kind, nb):
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), kind, nb, ret), nprocs=world, join=True)
    assert len(ret) == world
    for rank in range(world):
        assert ret[rank] < 1e-5, ret


def _decoder_worker(rank, world, port, ret):
    """The decoder of a sharded train step (csrc/rgcn_api.hip train_step_tail + decoder_allreduce): rank g scores the
    slice [g ceil(N / world), ...) of the batch, its loss terms and gradients normalised by the WHOLE batch's N, and the
    partial loss, dL/dcodes and dL/dW_relation are summed over the ranks."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        V, R, d, N, reg = 50, 6, 16, 203, 0.01            # N not divisible by the world size
        rng = np.random.RandomState(9)
        codes = rng.randn(V, d).astype(np.float32)
        w_rel = rng.randn(V, d).astype(np.float32)
        X = np.stack([rng.randint(0, V, N), rng.randint(0, R, N), rng.randint(0, V, N)], 1).astype(np.int32)
        Y = (rng.rand(N) < 0.1).astype(np.float32)
        per = (N + world - 1) // world
        lo, hi = min(N, rank * per), min(N, rank * per + per)
        n = hi - lo
        # the oracle normalises by the size of the batch it is given: rescale the slice's means to the whole batch's N
        loss, dcodes, drel = oracle.distmult_loss_and_grads(codes, w_rel, X[lo:hi], Y[lo:hi], reg)
        part = [torch.tensor([loss * n / N], dtype=torch.float64), torch.from_numpy(dcodes.astype(np.float64) * n / N),
                torch.from_numpy(drel.astype(np.float64) * n / N)]
        for t in part:
            dist.all_reduce(t)
        floss, fdcodes, fdrel = oracle.distmult_loss_and_grads(codes, w_rel, X, Y, reg)
        ret[rank] = (abs(float(part[0]) - floss) / abs(floss), float(np.abs(part[1].numpy() - fdcodes).max()),
                     float(np.abs(part[2].numpy() - fdrel).max()), float(np.abs(fdcodes).max()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_decoder_divided_by_triples_equals_the_whole_batch_gloo(world):
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_decoder_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert len(ret) == world
    for rank in range(world):
        rel_loss, e_codes, e_rel, scale = ret[rank]
        assert rel_loss < 1e-6 and e_codes < 2e-6 * max(scale, 1.0) and e_rel < 2e-6 * max(scale, 1.0), ret[rank]
