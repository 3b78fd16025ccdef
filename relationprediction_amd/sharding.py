"""Relation sharding for multi-GPU runs (new: the reference is single-device).

Aggregation is linear in the messages and messages are per-relation, so the graph and the
per-relation weights shard BY RELATION; degrees stay global; the self-loop GEMM is row-sharded.
One [V,d] sum-all-reduce per layer per direction recombines the partial results.
"""
from __future__ import annotations

import numpy as np


def lpt_partition(relation_counts, world):
    """Greedy longest-processing-time bin packing of relations onto `world` ranks by edge count.
    Returns int32 owner[R].  Deterministic (ties broken by relation id) so every rank computes the
    same assignment without communication."""
    counts = np.asarray(relation_counts, dtype=np.int64)
    owner = np.zeros(counts.shape[0], dtype=np.int32)
    if world <= 1:
        return owner
    load = np.zeros(world, dtype=np.int64)
    nrel = np.zeros(world, dtype=np.int64)
    order = np.lexsort((np.arange(counts.shape[0]), -counts))
    for r in order:
        # lightest rank; among equals the one holding fewer relations, then the lowest rank
        k = int(np.lexsort((np.arange(world), nrel, load))[0])
        owner[r] = k
        load[k] += counts[r]
        nrel[k] += 1
    return owner


def row_shard(num_rows, rank, world):
    """[lo, hi) rows of the self-loop product computed by `rank` (matches rgcn_create)."""
    return (rank * num_rows) // world, ((rank + 1) * num_rows) // world


def shard_imbalance(relation_counts, owner, world):
    """max load / mean load of an assignment (1.0 = perfect)."""
    counts = np.asarray(relation_counts, dtype=np.float64)
    load = np.bincount(owner, weights=counts, minlength=world)
    return float(load.max() / max(load.mean(), 1e-12))


def share_unique_id(rank, path, make_id, timeout=300.0):
    """Hand rank 0's 128-byte communicator id to the other ranks of this node through a file (single-node
    contract): rank 0 calls `make_id()` and publishes it atomically at `path`, the others wait for the file.
    No process group, no torch: the launcher only has to provide RANK / WORLD_SIZE and a path every rank agrees on."""
    import os
    import time
    if rank == 0:
        uid = make_id()
        with open(path + ".tmp", "wb") as f:
            f.write(uid)
        os.replace(path + ".tmp", path)
        return uid
    deadline = time.time() + timeout
    while not os.path.exists(path):
        if time.time() > deadline:
            raise TimeoutError("rank %d: no communicator id at %s after %.0f s" % (rank, path, timeout))
        time.sleep(0.05)
    with open(path, "rb") as f:
        return f.read()
