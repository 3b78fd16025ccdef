#!/usr/bin/env python
"""bench.py -- processed edges/sec of the R-GCN encoder forward+backward (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME] [--no-extra-workloads]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one minibatch graph already resident in HBM:
device-side graph preparation of the fed [E,3] triples + 2-layer encoder forward (train mode, generated
self-loop dropout) + backward for every encoder weight (rgcn_step_device).  value = E_g * steps / s.

Rank 0 prints ONE COMPACT JSON line LAST on stdout (a few KB: tests hold it under 8 KB).  Besides the contract keys:
  "roofline":     the dominant kernel against its gfx950 roofline (live HIP-event durations; fractions on COMPULSORY
                  bytes / ALGORITHMIC flops, design and PMC bytes beside them),
  "step_roofline": the whole step against the chip,
  "cpu_baseline": the CPU oracle (TF-dataflow-shaped numpy/scipy port of the reference) timed on this box's host cores
                  on the same minibatch; "cpu_baseline_reference_code": the reference's OWN model code over the numpy /
                  torch shims, timed where /root/reference exists (the build container; committed, not re-measured here),
  "top_kernels":  [name, launches/step, exclusive avg us, frac] of the six largest kernels,
  "workloads":    (N = 1, default workload) one-line summaries -- value, ms/step, dominant kernel, frac, fraction of the
                  serial ceiling -- of the other BASELINE.json configurations on one GPU: basis B = 2 / B = 5 (config 3), the
                  WN18 and FB15k shapes (configs 4 and 5), the 38,001-edge and the 272,115-edge full graphs,
  "train_steps":  (same run) whole training iterations on the device at the FB15k-237 and FB15k sizes: the minibatch
                  step train.py launches and configs[4]'s hipGraph-captured train step, replayed against stream-launched,
  "evaluation":   (same run) test-mode encoding of the 272,115-edge graph, raw + filtered ranks of 2,000 triples,
  "train_loop":   (same run) the reference's whole iteration through the training driver: host neighbourhood sampler
                  (k builder threads) + upload + device step + loss read-back: ms / iteration, device idle fraction.
Everything else -- the per-kernel tables of every workload and train step, notes -- goes to bench_details.json next to
this file (and to gpurun_out/ when that directory exists), never to the result line.

`python bench.py --gpus N` with no WORLD_SIZE in the environment spawns its own N ranks (one per GPU) and relays rank 0's
line; under torch.distributed.run it is one of the ranks.
"""
from __future__ import annotations

import argparse
import glob
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_HBM_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured achievable)
PEAK_F32_MFMA_TFS = 157.3  # MI355X_MICROARCH.md: fp32-input MFMA dense peak
PEAK_BF16_MFMA_TFS = 2500.0  # MI355X_MICROARCH.md: bf16 MFMA dense peak (no sparsity)

WORKLOADS = {
    # name: (graph fixture | "synth:..." , V, R, d, L, kind, nb, E_g)   -- BASELINE.json configs[1] is the headline
    "fb237_block": ("fb237_minibatch", 14541, 237, 500, 2, "block", 100, 15000),
    "fb237_block_fullgraph": ("fb237_valid_test", 14541, 237, 500, 2, "block", 100, 38001),
    # SURVEY 8d graph B: the FB15k-237 training graph's size (272,115 unique triples), relations and endpoints drawn
    # from the empirical histograms of the real valid+test triples (the train split is not shipped with the reference)
    "fb237_block_traingraph": ("synth:fb237_valid_test:272115", 14541, 237, 500, 2, "block", 100, 272115),
    "toy_block": ("toy_train", 16, 9, 500, 2, "block", 100, 43),
    # BASELINE.json configs[3]'s shape on one GPU at SURVEY 8d's size: WN18 (V 40,943 / R 18), a uniform 15,000-edge
    # minibatch of the 141,442-edge training-graph shape (drawn from the real valid+test histograms)
    "wn18_block": ("sample:wn18_valid_test:141442:15000", 40943, 18, 500, 2, "block", 100, 15000),
    # BASELINE.json configs[4]'s space on one GPU: FB15k (V 14,951 / R 1,345), 15,000 real valid triples as graph
    "fb15k_block": ("fb15k_minibatch", 14951, 1345, 500, 2, "block", 100, 15000),
    # BASELINE.json configs[2]: basis decomposition, B = 2 (and settings/gcn_basis.exp's own B = 5)
    "fb237_basis_b2": ("fb237_minibatch", 14541, 237, 500, 2, "basis", 2, 15000),
    "fb237_basis_b5": ("fb237_minibatch", 14541, 237, 500, 2, "basis", 5, 15000),
}
# measured after the headline workload in the default single-GPU run (one driver line carries all of them)
EXTRA_WORKLOADS = ["fb237_basis_b2", "fb237_basis_b5", "wn18_block", "fb15k_block", "fb237_block_fullgraph",
                   "fb237_block_traingraph"]


def synthetic_from_histograms(pool, n, seed=0):
    """n unique triples: r ~ the pool's relation histogram, s | r and o | r ~ the pool's conditional endpoint
    histograms (SURVEY 8d 'graph B').  Relations whose empirical support cannot hold their share (|S_r| x |O_r|
    distinct pairs) saturate; the remainder is redrawn until n unique triples exist."""
    import numpy as np
    rng = np.random.RandomState(seed)
    pool = np.asarray(pool, dtype=np.int64)
    order = np.argsort(pool[:, 1], kind="stable")
    by_rel = pool[order]
    rels, starts, counts = np.unique(by_rel[:, 1], return_index=True, return_counts=True)
    p_rel = counts / counts.sum()
    V = int(pool[:, [0, 2]].max()) + 1
    R = int(pool[:, 1].max()) + 1
    seen = np.empty(0, dtype=np.int64)
    for _ in range(64):
        need = n - seen.size
        if need <= 0:
            break
        m = int(need * 1.5) + 1024
        ri = rng.choice(len(rels), size=m, p=p_rel)
        # endpoint | relation: a uniformly chosen occurrence of that relation in the pool (= its histogram)
        si = starts[ri] + (rng.random_sample(m) * counts[ri]).astype(np.int64)
        oi = starts[ri] + (rng.random_sample(m) * counts[ri]).astype(np.int64)
        key = (by_rel[si, 0] * R + rels[ri]) * V + by_rel[oi, 2]
        seen = np.unique(np.concatenate([seen, key]))
    if seen.size < n:
        raise RuntimeError("the empirical histograms support only %d unique triples" % seen.size)
    seen = seen[rng.permutation(seen.size)[:n]]
    out = np.stack([seen // (R * V), (seen // V) % R, seen % V], axis=1)
    return np.ascontiguousarray(out.astype(np.int32))


def load_graph(name):
    """graph of a workload: a fixture of tests/golden/graphs.npz (made by tests/golden/make_golden.py from the
    reference's data files; /root/reference itself is not needed at run time) or 'synth:<pool fixture>:<n>'"""
    import numpy as np
    with np.load(os.path.join(ROOT, "tests", "golden", "graphs.npz")) as z:
        if name.startswith("synth:"):
            _, pool, n = name.split(":")
            return synthetic_from_histograms(z[pool], int(n), seed=0)
        if name.startswith("sample:"):      # 'sample:<pool fixture>:<n_graph>:<n>': a uniform n-edge minibatch of that synthetic graph
            _, pool, n_graph, n = name.split(":")
            g = synthetic_from_histograms(z[pool], int(n_graph), seed=0)
            pick = np.random.RandomState(3).choice(g.shape[0], size=int(n), replace=False)
            return np.ascontiguousarray(g[np.sort(pick)])
        return np.ascontiguousarray(z[name].astype(np.int32))


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="fb237_block", choices=sorted(WORKLOADS))
    ap.add_argument("--no-extra-workloads", action="store_true",
                    help="only the --workload line (default: the single-GPU run of the headline workload also "
                         "measures the other BASELINE configurations into \"workloads\")")
    ap.add_argument("--extra-steps", type=int, default=20, help="timed steps of each extra workload")
    ap.add_argument("--cpu-steps", type=int, default=12, help="timed CPU-oracle steps (0 disables)")
    ap.add_argument("--no-kernel-profile", action="store_true")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="skip the two rocprofv3 --pmc passes that measure the headline workload's HBM bytes per launch in "
                         "this run (the committed profiles/*_traffic.json figures are reported instead)")
    ap.add_argument("--no-fp32-reference", action="store_true",
                    help="skip the extra timed run with the dense contractions on the fp32 MFMA (rocprofv3 passes)")
    ap.add_argument("--hipgraph", action="store_true",
                    help="replay the steps from a captured hipGraph (two steps + their prefetches per launch; "
                         "single GPU only)")
    ap.add_argument("--gemm-mode", type=int, default=int(os.environ.get("RGCN_GEMM_MODE", "6")), choices=[0, 6, 9],
                    help="arithmetic of the dense contractions (include/rgcn.h rgcn_set_gemm_mode): 6 / 9 = exact "
                         "bf16 operand split with 6 / 9 partial products on the bf16 matrix cores, 0 = fp32 MFMA")
    return ap.parse_args()


class Ranks:
    """rank / world plumbing of one bench process (one rank per GPU under torch.distributed.run)"""

    def __init__(self, args):
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        if self.world != args.gpus:       # under a launcher the environment decides (main() self-spawns otherwise)
            args.gpus = self.world
        self.rdv_path = "/tmp/rgcn_rccl_id_%s_%d" % (os.environ.get("MASTER_PORT", "0"), os.getppid())


# profile tag (csrc ProfScope name) -> HIP kernel name prefix in the rocprofv3 summaries
# (the pre-split-weight instantiation <true, true, true, TERMS, true> serves both the NN and the NT products)
# (a list: the first kernel name found in the table -- the forward products run on k_gemm_w8 where the launch is one round of
# 128 x 256 tiles, csrc/gemm_bf16x3.hip, and on the 128 x 128 kernel otherwise)
KERNEL_OF_TAG = {"gemm_self_fwd": ["k_gemm_w8<", "k_gemm_bf16x3<true, "], "gemm_self_dh": "k_gemm_bf16x3<true, ",
                 "gemm_self_dw": "k_gemm_bf16x3<false, false", "block_msg_fwd": "k_block_msg_fwd",
                 "block_msg_bwd": "k_block_msg_bwd", "combine_fwd": "k_combine", "combine_bwd": "k_combine",
                 "input_fwd": "k_input_fwd", "top_grad_dropout": "k_scale_dropout", "block_dw_reduce": "k_block_dw_reduce",
                 "splitk_reduce": "k_splitk_reduce", "prep_sort": "k_sort_scatter",
                 "prep_keys": "k_keys", "prep_ptrs": "k_ptrs", "prep_build_msgs": "k_build_msgs",
                 "gemm_basis_fwd": ["k_gemm_w8<", "k_gemm_bf16x3<true, "], "gemm_basis_dz": "k_gemm_bf16x3<true, ",
                 "gemm_basis_dw": "k_gemm_bf16x3<false, false", "basis_agg": "k_basis_agg",
                 "basis_bwd_gather": "k_basis_bwd_gather", "basis_dcoef": "k_basis_dcoef",
                 # (prefix, substring): template arguments in the middle of the name tell the instantiations apart
                 "block_rows_fwd": ("k_block_rows<", ", false,"), "block_rows_bwd": ("k_block_rows<", ", true,"),
                 "block_dw_msgs": ("k_block_msg_bwd<", ", false>"), "prep_row_order": "k_sort_hist",
                 "block_wtile_build": "k_wtile_build", "basis_aggregate": "k_basis_agg",
                 "basis_gather_units": "k_gather_units", "gemm_presplit_b": "k_presplit_b", "prep_scan": "k_exscan",
                 "bias_grad_colsum": "k_colsum_final", "basis_dcoef_reduce": "k_basis_dcoef_reduce"}


def kernel_traffic(traffic, tag, args):
    """bytes per launch of the kernel behind a profile tag in a {HIP kernel name: bytes} table (None: not there)"""
    pref = KERNEL_OF_TAG.get(tag)
    if not pref:
        return None
    if isinstance(pref, list):
        pref = next((p for p in pref if any(n.startswith(p) for n in traffic)), pref[-1])
    pref, sub = pref if isinstance(pref, tuple) else (pref, "")
    if tag == "block_msg_bwd":
        sub = ", true>"
    if pref.startswith("k_gemm") and args.gemm_mode == 0:
        return None                   # the PMC passes ran the default arithmetic
    return next((round(v) for n, v in traffic.items() if n.startswith(pref) and sub in n), None)


def traffic_table(workload):
    """HBM bytes per launch: measured in this run for the headline workload (measure_live_traffic: counters cannot be
    collected from inside the process, so the script runs itself twice under rocprofv3 --pmc), else from the committed
    rocprofv3 PMC passes of this same command (profiles/*traffic*.json, FETCH_SIZE doubled per MI355X_MICROARCH.md)"""
    if workload in LIVE_TRAFFIC:
        return LIVE_TRAFFIC[workload]
    try:
        pick = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_rocprof_%s_serial_traffic.json" % workload)))
        if pick:
            return json.load(open(pick[-1]))["kernels"], os.path.relpath(pick[-1], ROOT)
    except Exception:  # noqa: BLE001
        pass
    return {}, None


LIVE_TRAFFIC = {}      # workload -> (kernel name -> bytes per launch, source text): filled by measure_live_traffic


def measure_live_traffic(workload, args):
    """HBM bytes per launch of every kernel of `workload`, measured in THIS run: two extra passes of this script under
    `rocprofv3 --pmc` (FETCH_SIZE, WRITE_SIZE: counters only, one counter per pass, as MI355X_MICROARCH.md prescribes;
    no trace domains), a few steps each on one stream; bytes = (2 x FETCH_SIZE + WRITE_SIZE) KB x 1024 (gfx950 FETCH_SIZE
    half-count correction).  Any failure (no rocprofv3, a pass that times out) leaves the committed figures in force."""
    import re
    import shutil
    import sqlite3
    import subprocess
    import tempfile
    prof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(prof):
        return False
    # already running under a profiler (someone wrapped this script in rocprofv3 / set a tools library): do not nest
    if any(k.startswith(("ROCP_", "ROCPROFILER_", "ROCPROF_")) or k == "HSA_TOOLS_LIB" for k in os.environ) or \
            "rocprof" in os.environ.get("LD_PRELOAD", ""):
        return False
    cmd = [sys.executable, os.path.abspath(__file__), "--workload", workload, "--steps", "4", "--warmup", "2", "--cpu-steps", "0",
           "--no-kernel-profile", "--no-extra-workloads", "--no-fp32-reference", "--no-live-traffic", "--gemm-mode",
           str(args.gemm_mode)]
    env = dict(os.environ, TMPDIR="/tmp", RGCN_STREAMS="0", RGCN_BENCH_PREFETCH="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)

    def short(name):
        return re.sub(r"\(.*", "", name.replace("rgcn::(anonymous namespace)::", "").replace("void ", ""))
    per = {}
    try:
        with tempfile.TemporaryDirectory(dir="/tmp") as tmp:
            for counter in ("FETCH_SIZE", "WRITE_SIZE"):
                out = os.path.join(tmp, counter)
                r = subprocess.run([prof, "--pmc", counter, "-d", out, "-o", "p", "--"] + cmd, cwd="/tmp", env=env,
                                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=75)
                db = os.path.join(out, "p_results.db")
                if r.returncode != 0 or not os.path.exists(db):
                    hits = glob.glob(os.path.join(out, "**", "*_results.db"), recursive=True)
                    if not hits:
                        return False
                    db = hits[0]
                rows = sqlite3.connect(db).execute("select kernel_name, count(*), sum(value) from counters_collection "
                                                   "where counter_name=? group by kernel_name", (counter,)).fetchall()
                acc = {}
                for name, n, tot in rows:
                    a = acc.setdefault(short(name), [0, 0.0])
                    a[0] += n
                    a[1] += tot
                per[counter] = {k: v[1] / max(v[0], 1) for k, v in acc.items()}
    except Exception:  # noqa: BLE001
        return False
    fetch, write = per.get("FETCH_SIZE", {}), per.get("WRITE_SIZE", {})
    kernels = {k: (2 * fetch[k] + write[k]) * 1024 for k in fetch if k in write}
    if not kernels:
        return False
    LIVE_TRAFFIC[workload] = (kernels, "measured in this run: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, one pass each of `bench.py "
                                       "--workload %s --steps 4 --warmup 2` on one stream; bytes = (2 x FETCH_SIZE + WRITE_SIZE) KB x 1024"
                              % workload)
    return True


def apply_live_traffic(out, args):
    """AFTER every timed region of the run (two profiler passes just before one cost the headline 3 % on the same box):
    measure the headline workload's HBM bytes per launch and put them where the committed figures stood."""
    if not out.get("kernels") or not measure_live_traffic(out["config"]["workload"], args):
        return
    traffic, src = LIVE_TRAFFIC[out["config"]["workload"]]
    for k in out["kernels"]:
        k["traffic"] = kernel_traffic(traffic, k["kernel"], args)
    if out.get("roofline"):
        top = next((k for k in out["kernels"] if k["kernel"] == out["roofline"]["kernel"]), None)
        if top is not None:
            out["roofline"]["traffic"] = top["traffic"]
            out["roofline"]["traffic_source"] = src
    sr = out.get("step_roofline")
    if sr:
        sr["pmc_bytes_per_step"] = sum(k["traffic"] * k["launches_per_step"] for k in out["kernels"] if k["traffic"]) or None
        sr["pmc_missing_kernels"] = [k["kernel"] for k in out["kernels"] if k["traffic"] is None
                                     and not k["kernel"].startswith("rccl_") and k["ms_per_step"] >= 0.002]


def kernel_table(prof, steps, args, in_pipeline=None):
    """rgcn_profile_get rows -> per-kernel accounting: exclusive average duration and, per launch, three byte counts --
    compulsory (every distinct input byte once + every output byte once), design (what the launch site asks
    of the memory system: gathers per use, staging slabs), pmc (filled in by the caller) -- and the algorithmic flops.
    `frac` is computed on the COMPULSORY bytes (HBM-bound rows) or the ALGORITHMIC flops (MFMA-bound rows)."""
    in_pipeline = in_pipeline or {}
    terms = args.gemm_mode if args.gemm_mode else 1
    # roofline of an fp32 contraction in this arithmetic: every fp32 product costs `terms` bf16 MFMA products
    mfma_peak = PEAK_BF16_MFMA_TFS / terms if args.gemm_mode else PEAK_F32_MFMA_TFS
    kernels = []
    for p in prof:
        if p["calls"] == 0:
            continue
        avg_ms = p["total_ms"] / p["calls"]
        design, fl = p["alg_bytes"] / p["calls"], p["alg_flops"] / p["calls"]
        comp = p.get("compulsory_bytes", p["alg_bytes"]) / p["calls"]
        t_hbm = comp / (PEAK_HBM_GBS * 1e9)
        t_mfma = fl / (mfma_peak * 1e12)
        bound = "mfma" if t_mfma > t_hbm else "hbm"
        sec = max(avg_ms, 1e-9) * 1e-3
        if bound == "mfma":
            ach, peak, unit = fl / sec / 1e12, mfma_peak, "TFLOP/s"
        else:
            ach, peak, unit = comp / sec / 1e9, PEAK_HBM_GBS, "GB/s"
        k = {"kernel": p["name"], "launches_per_step": p["calls"] / steps,
             "avg_us": round(avg_ms * 1e3, 2),
             "avg_us_in_pipeline": round(in_pipeline.get(p["name"], 0.0) * 1e3, 2),
             "ms_per_step": round(p["total_ms"] / steps, 4),
             "bound": bound, "achieved": round(ach, 2), "peak": round(peak, 1), "unit": unit,
             "frac": round(ach / peak, 4), "compulsory_bytes": comp, "design_bytes": design, "alg_flops": fl,
             "design_gbs": round(design / sec / 1e9, 1), "traffic": None}
        if bound == "mfma":
            if args.gemm_mode:
                k["peak_is"] = ("bf16 MFMA dense peak %.0f TF / %d partial products per fp32 product (exact 3-way bf16 "
                                "operand split, fp32 accumulate)" % (PEAK_BF16_MFMA_TFS, terms))
                k["frac_of_fp32_mfma_peak"] = round(fl / sec / 1e12 / PEAK_F32_MFMA_TFS, 4)
                k["frac_of_bf16_peak_on_alg_flops"] = round(fl / sec / 1e12 / PEAK_BF16_MFMA_TFS, 4)
                k["executed_bf16_tflops"] = round(terms * fl / sec / 1e12, 1)
            else:
                k["peak_is"] = "fp32 MFMA dense peak"
        kernels.append(k)
    return kernels


class EncoderBench:
    """One encoder workload on this rank's engine: the engine, two alternating minibatches resident in HBM, the step loop
    and the legs that are measured on it.  `measure` below strings the legs together."""

    def __init__(self, workload, args, rk):
        import numpy as np
        from relationprediction_amd import _native
        from relationprediction_amd.sharding import lpt_partition
        from relationprediction_amd.common.shared_functions import init_encoder_params

        self.np, self.native = np, _native
        self.workload, self.args, self.rk = workload, args, rk
        self.rank, self.world = rank, world = rk.rank, rk.world
        self.graph_name, self.V, self.R, self.d, self.L, self.kind, self.nb, self.E_g = WORKLOADS[workload]
        graph_name, V, R, d, L, kind, nb, E_g = WORKLOADS[workload]
        self.triples = triples = load_graph(graph_name)
        assert triples.shape[0] == E_g
        # a second minibatch of the same size so that consecutive steps see different graphs (as in training, where
        # t_func samples a new one every step): steps alternate A, B, A, B ...
        if graph_name == "fb237_minibatch":
            pool = load_graph("fb237_valid_test")
            triples_b = np.ascontiguousarray(pool[np.random.RandomState(7).choice(pool.shape[0], size=E_g, replace=False)])
        else:
            triples_b = np.ascontiguousarray(triples[np.random.RandomState(7).permutation(E_g)])
        self.params = init_encoder_params(V, R, d, L, kind, nb, rng=np.random.RandomState(1))
        self.dcodes = (np.random.RandomState(2).randn(V, d) * 1e-3).astype(np.float32)

        # RGCN_BENCH_SHARE_GPU=1 (tests only, with RGCN_RCCL_LIBRARY pointing at the shared-memory collective of
        # tests/collective_double): every rank on device 0, to run the multi-process path on a one-GPU box.
        # A launcher that masks the visible devices per rank leaves every rank with ONE device, index 0: take LOCAL_RANK only
        # while the process sees that many devices (which GPU it is gets checked by PCI address in self_check).
        n_visible, _ = _native.Engine.device_info(0)
        self.device = 0 if os.environ.get("RGCN_BENCH_SHARE_GPU") == "1" or (world > 1 and n_visible == 1) else rk.local_rank
        self.eng = eng = _native.Engine(V, R, d, L, kind, nb, keep_prob=0.8, norm_mode="intended", max_edges=E_g,
                                        device=self.device, rank=rank, world=world)
        if os.environ.get("RGCN_STREAMS", "1") == "0":      # bench.py's own switch (the PMC passes run on one stream)
            eng.set_overlap(False)
        eng.set_params(self.params)
        if world > 1:
            # One rank per GPU (launched by torch.distributed.run, which only provides the environment).
            # The process keeps ONE ROCm stack: no torch import here; the 128-byte RCCL id travels through a
            # file on this node (single-node contract), everything else (barriers, max over ranks) rides on
            # the library's own RCCL communicator.
            owner = lpt_partition(np.bincount(triples[:, 1], minlength=R), world)
            eng.set_relation_owner(owner)
            from relationprediction_amd.sharding import share_unique_id
            uid = share_unique_id(rank, rk.rdv_path, _native.Engine.comm_unique_id)
            eng.comm_init(uid)
        self.rccl_ranks = eng.comm_info()[0] if world > 1 else None
        self.comm_buf = eng.to_device(np.zeros(max(world, 1), dtype=np.float32)) if world > 1 else None
        if world > 1:
            self.self_check()
        self.tri_dev = eng.to_device(triples)
        self.tri_dev_b = eng.to_device(triples_b)
        self.dc_dev = eng.to_device(self.dcodes)
        self.graphs = [self.tri_dev, self.tri_dev_b]
        self.pipeline = os.environ.get("RGCN_BENCH_PREFETCH", "1") != "0"
        self.graph_id = None             # --hipgraph: the captured pair of steps
        eng.set_gemm_mode(args.gemm_mode)

    def allgather_scalar(self, x):
        """every rank's value, through a sum-all-reduce of a one-hot vector"""
        v = self.np.zeros(self.world, dtype=self.np.float32)
        v[self.rank] = x
        self.comm_buf.upload(v)
        self.eng.comm_allreduce_sum(self.comm_buf, self.world)
        return self.comm_buf.download(self.np.float32, (self.world,))

    def self_check(self):
        """First contact with a multi-GPU node must diagnose itself: the collective library has to SEE `world` ranks, and no
        two ranks may sit on one device (a mis-set LOCAL_RANK / HIP_VISIBLE_DEVICES gives N processes on GPU 0 that "scale"
        at 1/N).  What RCCL reports (ncclCommCount, ncclCommCuDevice) and the PCI address of that HIP device are gathered
        over the communicator itself; any mismatch ends the run with a non-zero exit instead of a number."""
        rank, world, device = self.rank, self.world, self.device
        n_seen, r_seen, dev_seen = self.eng.comm_info()
        pci = self.native.Engine.device_info(dev_seen if dev_seen >= 0 else device)[1]
        # (domain, bus << 8 | device) as two exact floats; an unknown address falls back to the device index
        devs = list(zip(self.allgather_scalar(float(pci >> 16) if pci >= 0 else -1.0),
                        self.allgather_scalar(float(pci & 0xffff) if pci >= 0 else float(device))))
        ranks_seen = self.allgather_scalar(float(n_seen))
        problems = []
        if n_seen >= 0 and n_seen != world:      # (-1: the bound library has no ncclCommCount -- nothing to compare)
            problems.append("the collective library reports %d ranks, the launcher started %d" % (n_seen, world))
        if r_seen >= 0 and r_seen != rank:
            problems.append("rank %d is rank %d of the communicator" % (rank, r_seen))
        share = os.environ.get("RGCN_BENCH_SHARE_GPU") == "1"      # tests: N ranks on ONE GPU over a stand-in collective
        if not share and len(set(devs)) != world:
            problems.append("ranks share a device: PCI (domain, bus:device) by rank = %s"
                            % [(int(a), "%02x:%02x" % (int(b) >> 8, int(b) & 0xff)) for a, b in devs])
        if n_seen >= 0 and any(int(x) != world for x in ranks_seen):
            problems.append("ranks disagree about the communicator's size: %s" % [int(x) for x in ranks_seen])
        if problems:
            sys.stderr.write("bench.py --gpus %d: multi-GPU self-check FAILED on rank %d: %s\n" % (world, rank, "; ".join(problems)))
            sys.exit(3)

    def barrier(self):
        self.eng.sync()
        if self.world > 1:
            self.allgather_scalar(1.0)      # returns only after every rank has joined the collective
            self.eng.sync()

    def run(self, n, seed0, pipeline=None):
        """n steps.  Step i works on graph i % 2; while it runs, the next graph's CSR / message list is prepared on a side
        stream (what a training loop does with the next sampled minibatch).  After capture_two_steps: one launch of the
        captured graph per two steps."""
        pipeline = self.pipeline if pipeline is None else pipeline
        if self.graph_id is not None and pipeline:
            for _ in range(n // 2):
                self.eng.graph_launch(self.graph_id)
            return
        for i in range(n):
            self.eng.step_device(self.graphs[i % 2], self.E_g, self.dc_dev, train=True, seed=seed0 + i)
            if pipeline:
                self.eng.prefetch_graph_device(self.graphs[(i + 1) % 2], self.E_g)

    def capture_two_steps(self):
        """--hipgraph: the steady state of the pipelined loop as ONE graph: step on A beside the preparation of B, then the
        reverse (RGCN_BENCH_GRAPH_NOPF=1, an A/B knob: no prefetches in the graph, every step prepares its own graph in line)"""
        eng, graphs, E_g = self.eng, self.graphs, self.E_g
        in_graph_prefetch = os.environ.get("RGCN_BENCH_GRAPH_NOPF") != "1"
        if in_graph_prefetch:
            eng.prefetch_graph_device(graphs[0], E_g)
        eng.sync()
        eng.capture_begin()
        for i in (0, 1):
            eng.step_device(graphs[i], E_g, self.dc_dev, train=True, seed=2000 + i)
            if in_graph_prefetch:
                eng.prefetch_graph_device(graphs[1 - i], E_g)
        self.graph_id = eng.capture_end()
        self.run(4, 0)

    def timed(self, n, seed0, gpu_timer=False):
        """n steps between barrier + device synchronisation on both sides: (wall seconds, max over ranks; HIP-event ms)"""
        self.barrier()
        t0 = time.perf_counter()
        if gpu_timer:
            self.eng.timer_start()
        self.run(n, seed0)
        gpu_ms = self.eng.timer_stop() if gpu_timer else None
        self.eng.sync()
        self.barrier()
        wall = time.perf_counter() - t0
        if self.world > 1:
            wall = float(self.allgather_scalar(wall).max())
        return wall, gpu_ms

    def steady_state(self, steps_before):
        """The same loop once the device has settled (informative, never `value`): a fresh process's first ~30 steps run
        6-13 % slower than the steady state (profiles/r05_first_steps_series.txt: every kernel alike, the device's power
        management after idle), and the driver's `--steps 20 --warmup 5` measures exactly those."""
        n_s = 40
        self.run(15, 2500)
        wall_s, _ = self.timed(n_s, 2600)
        return {"ms_per_step": round(wall_s * 1e3 / n_s, 4), "value": round(self.E_g * n_s / wall_s, 1), "unit": "edges/s",
                "steps": n_s, "after_steps": steps_before + 15,
                "note": "the same loop continued after the timed region; informative only -- `value` is the timed region's"}

    def fp32_reference(self, steps):
        """the same step with the dense contractions on the fp32 MFMA (mode 0), for reference"""
        n2 = max(10, steps // 2)
        self.eng.set_gemm_mode(0)
        self.run(3, 3000)
        wall2, _ = self.timed(n2, 4000)
        self.eng.set_gemm_mode(self.args.gemm_mode)
        return {"value": round(self.E_g * n2 / wall2, 1), "unit": "edges/s", "ms_per_step": round(wall2 * 1e3 / n2, 4),
                "steps": n2, "arithmetic": "v_mfma_f32_32x32x2_f32 (rgcn_set_gemm_mode 0)"}

    def kernel_accounting(self, steps):
        """Per-kernel accounting: the same steps again with HIP events around every launch.  Pass 1 as timed (side streams
        + pipelined prep: durations of co-running kernels overlap and stretch); pass 2 with overlap off, every kernel alone
        on the main stream: EXCLUSIVE durations, which the per-kernel roofline numbers use.  -> (kernels, roofline)"""
        eng, args = self.eng, self.args
        eng.profile_reset()
        eng.profile_enable(True)
        self.run(steps, 2000)
        in_pipeline = {p["name"]: p["total_ms"] / max(p["calls"], 1) for p in eng.profile()}
        eng.profile_enable(False)
        eng.set_overlap(False)
        eng.profile_reset()
        eng.profile_enable(True)
        self.run(steps, 2000, pipeline=False)
        prof = eng.profile()
        eng.profile_enable(False)
        eng.set_overlap(os.environ.get("RGCN_STREAMS", "1") != "0")
        kernels = kernel_table(prof, steps, args, in_pipeline)
        traffic, traffic_src = traffic_table(self.workload)
        for k in kernels:
            k["traffic"] = kernel_traffic(traffic, k["kernel"], args)
        kernels.sort(key=lambda k: -k["ms_per_step"])
        roofline = None
        if kernels:
            k = kernels[0]
            roofline = {"kernel": k["kernel"], "bound": k["bound"], "achieved": k["achieved"], "peak": k["peak"],
                        "unit": k["unit"], "frac": k["frac"], "traffic": k["traffic"], "traffic_source": traffic_src,
                        "avg_us": k["avg_us"], "avg_us_in_pipeline": k["avg_us_in_pipeline"],
                        "launches_per_step": k["launches_per_step"],
                        "compulsory_bytes": k["compulsory_bytes"], "design_bytes": k["design_bytes"],
                        "alg_flops": k["alg_flops"],
                        "basis": "frac = algorithmic flops (MFMA) or compulsory bytes (HBM) per launch / exclusive "
                                 "HIP-event duration / peak"}
            for extra in ("peak_is", "frac_of_fp32_mfma_peak", "frac_of_bf16_peak_on_alg_flops", "executed_bf16_tflops"):
                if extra in k:
                    roofline[extra] = k[extra]
        return kernels, roofline

    def step_roofline(self, kernels, ms_per_step):
        """The whole step against the chip (north_star: throughput "as achieved fraction of HBM roofline"): compulsory /
        design / PMC bytes and algorithmic flops of one step (this rank's share) over the measured step time."""
        V, R, d, L, kind, nb, E_g = self.V, self.R, self.d, self.L, self.kind, self.nb, self.E_g
        P, I = 4.0 * V * d, 12.0 * E_g
        if kind == "block":
            sd = d // nb
            Wl = 4.0 * (2 * R * nb * sd * sd + d * d)
        else:
            Wl = 4.0 * (2 * d * nb * d + 2 * R * nb + d * d)
        compulsory = L * (2 * I + 5 * P + 3 * Wl) + 4 * P          # B_alg of one encoder step
        staged = sum(k["design_bytes"] * k["launches_per_step"] for k in kernels)
        missing = [k["kernel"] for k in kernels if k["traffic"] is None and not k["kernel"].startswith("rccl_")
                   and k["ms_per_step"] >= 0.002]
        measured = sum(k["traffic"] * k["launches_per_step"] for k in kernels if k["traffic"])
        sf = sum(k["alg_flops"] * k["launches_per_step"] for k in kernels)
        sec = ms_per_step * 1e-3
        ceiling = compulsory / (PEAK_HBM_GBS * 1e9) + sf / (PEAK_F32_MFMA_TFS * 1e12)
        return {"compulsory_bytes_per_step": compulsory, "design_bytes_per_step": staged,
                "pmc_bytes_per_step": measured or None, "pmc_missing_kernels": missing,
                "alg_flops_per_step": sf,
                "hbm_gbs": round(compulsory / sec / 1e9, 1),
                "hbm_frac": round(compulsory / sec / 1e9 / PEAK_HBM_GBS, 4),
                "hbm_frac_design": round(staged / sec / 1e9 / PEAK_HBM_GBS, 4),
                "fp32_tflops": round(sf / sec / 1e12, 2),
                "fp32_mfma_frac": round(sf / sec / 1e12 / PEAK_F32_MFMA_TFS, 4),
                "serial_ceiling_ms": round(ceiling * 1e3, 4),
                "frac_of_serial_ceiling": round(ceiling / sec, 4),
                "sum_exclusive_kernel_ms": round(sum(k["ms_per_step"] for k in kernels), 4),
                "note": "whole step on this rank over the measured step time.  compulsory = SURVEY 8d's "
                        "L*(2I+5P+3W)+4P (each distinct input read once, each output written once); "
                        "design = what the kernels move by construction (messages staged through "
                        "HBM / L2 between the relation-major and the row-major stage); pmc = rocprofv3 "
                        "FETCH/WRITE_SIZE bytes of the committed PMC passes x launches; ceiling = "
                        "compulsory / 8 TB/s + fp32 flops / 157.3 TF with nothing overlapped"}

    def cpu_baseline(self):
        """CPU baseline: the oracle (port of the reference's TF dataflow) on this host, rank 0, N=1"""
        import oracle  # test infrastructure; used here ONLY as the timed CPU baseline
        np, V, d, L, kind, E_g = self.np, self.V, self.d, self.L, self.kind, self.E_g
        masks = [(np.random.RandomState(3 + l).rand(V, d) < 0.8).astype(np.uint8) for l in range(L)]
        step = lambda: oracle.encoder_step(self.params, self.triples, V, L, kind, self.dcodes, keep_prob=0.8,  # noqa: E731
                                           dropout_masks=masks)
        step()  # warm-up
        ts = []
        for _ in range(self.args.cpu_steps):
            c0 = time.perf_counter()
            step()
            ts.append(time.perf_counter() - c0)
        med = float(np.median(ts))
        model_name = ""
        try:
            with open("/proc/cpuinfo") as f:
                model_name = next((ln.split(":", 1)[1].strip() for ln in f if ln.startswith("model name")), "")
        except OSError:
            pass
        return {"value": round(E_g / med, 1), "unit": "edges/s", "cores": os.cpu_count(), "cpu_model": model_name,
                "kind": "port",
                "ms_per_step": round(med * 1e3, 1),
                "threads": "BLAS threads = host cores for the dense products (self-loop, basis); the sparse incidence "
                           "products, gathers and batched 5x5 products of the TF-shaped dataflow run on one thread",
                "sample": "%d steps of the same %s minibatch (E_g=%d) through oracle.encoder_step "
                          "(numpy/scipy fp32, TF-dataflow-shaped), median" % (self.args.cpu_steps, self.workload, E_g)}

    def describe_data(self):
        """-> (dataset name, the `data` sentence, config.graph)"""
        workload, graph_name, E_g = self.workload, self.graph_name, self.E_g
        dataset = {"wn18": "WN18", "toy_": "Toy", "fb15": "FB15k"}.get(workload[:4], "FB15k-237")
        graph_spec = None
        if graph_name.startswith("synth:"):
            _, pool, n_graph = graph_name.split(":")
            graph_spec = {"kind": "synthetic", "pool_fixture": pool, "edges": int(n_graph)}
            data = ("synthetic graph: %d unique triples drawn from the relation and endpoint-given-relation histograms of the "
                    "real %s triples in fixture '%s' (SURVEY 8d graph B; the train split is not shipped), "
                    % (E_g, dataset, pool))
        elif graph_name.startswith("sample:"):
            _, pool, n_graph, n_pick = graph_name.split(":")
            graph_spec = {"kind": "uniform sample of a synthetic graph", "pool_fixture": pool, "graph_edges": int(n_graph),
                          "edges": int(n_pick), "since_round": 5}
            data = ("a uniform %d-edge minibatch (seed 3) of a synthetic %s-edge graph drawn from the relation and "
                    "endpoint-given-relation histograms of the real %s triples in fixture '%s' (since round 5; earlier rounds "
                    "timed this workload on the fixture's own 10,000 triples: not comparable), "
                    % (E_g, n_graph, dataset, pool))
        else:
            data = "real graph structure (fixture %s of tests/golden/graphs.npz, %d edges; SURVEY 8d), " % (graph_name, E_g)
        return dataset, data, graph_spec or {"kind": "fixture", "name": graph_name}

    def close(self):
        self.tri_dev.free()
        self.tri_dev_b.free()
        self.dc_dev.free()
        if self.comm_buf is not None:
            self.comm_buf.free()
        self.eng.close()


def measure(workload, args, rk, steps, warmup, full):
    """One workload on this rank's engine: W untimed steps, then EXACTLY `steps` timed ones (barrier + sync on both sides,
    max over ranks) -- `value` comes from that region and nothing else --, then the informative legs and the per-kernel
    passes.  full = the headline extras (steady state, fp32-MFMA reference run, CPU baseline)."""
    b = EncoderBench(workload, args, rk)
    rank, world = b.rank, b.world
    b.run(warmup, 1000)
    use_graph = args.hipgraph and world == 1 and b.pipeline
    if use_graph:
        if steps % 2:
            sys.exit("--hipgraph: --steps must be even (one launch = two steps)")
        b.capture_two_steps()
    wall, gpu_ms = b.timed(steps, 2000, gpu_timer=True)
    ms_per_step = wall * 1e3 / steps
    value = b.E_g * steps / wall

    steady = b.steady_state(warmup + steps) if full and world == 1 else None
    fp32_ref = b.fp32_reference(steps) if full and args.gemm_mode != 0 and not args.no_fp32_reference else None
    kernels, roofline = ([], None) if args.no_kernel_profile else b.kernel_accounting(steps)
    step_roofline = b.step_roofline(kernels, ms_per_step) if kernels else None
    cpu = b.cpu_baseline() if full and rank == 0 and world == 1 and args.cpu_steps > 0 else None

    out = None
    if rank == 0:
        dataset, data, graph_spec = b.describe_data()
        out = {
            "metric": "processed edges/sec (R-GCN forward+backward), %s gcn_%s" % (dataset, b.kind),
            "value": round(value, 1), "unit": "edges/s", "n_gpus": world, "steps": steps,
            "warmup": warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None,
            "dtype": "f32" if args.gemm_mode == 0 else
                     "f32 (dense contractions: fp32 operands split exactly into 3 bf16, %d of 9 partial products on the "
                     "bf16 matrix cores, fp32 accumulation; error vs float64 equal to the fp32 MFMA's)" % args.gemm_mode,
            "data": data + "reference-distribution random-init weights, synthetic upstream gradient",
            "config": {"workload": workload, "graph": graph_spec,
                       "entities": b.V, "relations": b.R, "dim": b.d, "layers": b.L,
                       "kind": b.kind, "num_blocks_or_bases": b.nb, "graph_edges": b.E_g,
                       "norm_mode": "intended (1/deg of the edge's own row; SURVEY H1, DESIGN section 10)",
                       "step": ("device graph prep + encoder fwd (train, dropout) + bwd (all encoder grads); "
                                "two alternating minibatches, next graph's prep pipelined on a side stream"
                                + ("; replayed from a captured hipGraph (2 steps per launch)" if use_graph else ""))
                               if b.pipeline else
                               "device graph prep + encoder fwd (train, dropout) + bwd (all encoder grads)",
                       "parallelism": "relation-sharded x%d + RCCL reduce-scatter / all-gather per layer, one all-reduce "
                                      "of the replicated weight gradients" % world if world > 1 else "single GPU"},
            "gpu_event_ms_per_step": round(gpu_ms / steps, 4),
            # collectives of one step on this rank (exclusive durations of the rccl_* launches; N = 1: none)
            "comm_ms_per_step": round(sum(k["ms_per_step"] for k in kernels if k["kernel"].startswith("rccl_")), 4),
            # what the collective library itself says about the communicator (ncclCommCount): N ranks seen, or -1
            "rccl_ranks": b.rccl_ranks,
            "collectives": [[k["kernel"], k["launches_per_step"], k["avg_us"]] for k in kernels
                            if k["kernel"].startswith("rccl_")],
            "message_edges_per_s": round(2 * b.L * value, 1),
            "roofline": roofline, "step_roofline": step_roofline, "kernels": kernels, "cpu_baseline": cpu,
            "fp32_mfma_reference": fp32_ref,
            "steady_state": steady,
        }
        if cpu:
            out["speedup_vs_cpu"] = round(value / cpu["value"], 1)
    b.close()
    return out


TRAIN_STEP_WORKLOADS = {
    # name: (pool of real triples the 30,000-triple graph batch comes from, V, R)  -- d 500, 2 layers, block kind, 100
    # blocks; settings/gcn_block.exp: GraphBatchSize 30000, GraphSplitSize 0.5, NegativeSampleRate 10
    "fb237_block_train_step": ("fb237_valid_test", 14541, 237),
    "fb15k_block_train_step": ("fb15k_minibatch", 14951, 1345),      # BASELINE.json configs[4]'s space on one GPU
}


def measure_train_step(name, args, steps, warmup):
    """One whole training iteration on the device (the "next" rows f1, f2, f4 of SURVEY 8 around the hot path), one GPU:
    (a) rgcn_train_step_minibatch_device, what train.py launches per iteration: exact-k edge dropout of the resident
        30,000-triple graph batch (15,000 edges kept), 10 negatives per positive (N = 330,000), graph preparation,
        encoder forward, DistMult loss + gradients, encoder backward, clip, Adam;
    (b) BASELINE.json configs[4]'s "hipGraph-captured train step": rgcn_train_step_device on a fixed 15,000-edge graph
        and a fixed batch, captured once and replayed."""
    import numpy as np
    from relationprediction_amd import _native
    from relationprediction_amd.common.shared_functions import init_encoder_params
    pool_name, V, R = TRAIN_STEP_WORKLOADS[name]
    d, L, kind, nb, n_batch, keep, rate = 500, 2, "block", 100, 30000, 15000, 10
    pool = load_graph(pool_name)
    if pool.shape[0] >= n_batch:
        batch = np.ascontiguousarray(pool[np.random.RandomState(11).choice(pool.shape[0], n_batch, replace=False)])
    else:      # fewer real triples than a graph batch: the rest is drawn from their histograms (as graph B)
        extra = synthetic_from_histograms(pool, n_batch - pool.shape[0], seed=11)
        batch = np.ascontiguousarray(np.concatenate([pool, extra]).astype(np.int32))
    N = n_batch * (rate + 1)
    eng = _native.Engine(V, R, d, L, kind, nb, keep_prob=0.8, norm_mode="intended", max_edges=n_batch)
    out = {"workload": name, "graph_batch": n_batch, "graph_edges": keep, "decoder_triples": N, "steps": steps,
           "entities": V, "relations": R}
    held = []
    try:
        eng.set_gemm_mode(args.gemm_mode)
        eng.set_params(init_encoder_params(V, R, d, L, kind, nb, rng=np.random.RandomState(1)))
        eng.decoder_reserve(N)
        eng.optimizer_config(lr=0.01, max_grad_norm=1.0)
        batch_dev = eng.to_device(batch)
        X = _native.DeviceBuffer(eng, 12 * N)
        Y = _native.DeviceBuffer(eng, 4 * N)
        held += [batch_dev, X, Y]

        def run(n, seed0):
            for i in range(n):
                eng.train_step_minibatch_device(batch_dev, n_batch, keep, seed0 + i, rate, seed0 + i, X, Y,
                                                seed=seed0 + i, reg_param=0.01)
        run(warmup, 100)
        eng.sync()
        t0 = time.perf_counter()
        run(steps, 1000)
        eng.sync()
        wall = time.perf_counter() - t0
        loss = eng.loss()
        out["minibatch_step"] = {"ms_per_step": round(wall * 1e3 / steps, 4), "edges_per_s": round(keep * steps / wall, 1),
                                 "triples_per_s": round(N * steps / wall, 1), "loss_after": round(loss, 6),
                                 "entry_point": "rgcn_train_step_minibatch_device"}
        # (b) captured: the graph the last step kept, the batch the last step drew
        graph = eng.to_device(eng.graph_edges())
        held.append(graph)
        eng.train_step_device(graph, keep, X, Y, N, seed=1, reg_param=0.01)
        eng.sync()
        eng.capture_begin()
        eng.train_step_device(graph, keep, X, Y, N, seed=2, reg_param=0.01)
        gid = eng.capture_end()
        for _ in range(warmup):
            eng.graph_launch(gid)
        eng.sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            eng.graph_launch(gid)
        eng.sync()
        wall_g = time.perf_counter() - t0
        for i in range(warmup):
            eng.train_step_device(graph, keep, X, Y, N, seed=50 + i, reg_param=0.01)
        eng.sync()
        t0 = time.perf_counter()
        for i in range(steps):
            eng.train_step_device(graph, keep, X, Y, N, seed=100 + i, reg_param=0.01)
        eng.sync()
        wall_s = time.perf_counter() - t0
        if not args.no_kernel_profile:
            # exclusive durations (side streams off) of every kernel of the stream-launched step, with the algorithmic
            # bytes / flops its launch site declares: the decoder, optimizer and edge-dropout kernels' own rooflines
            eng.set_overlap(False)
            eng.profile_reset()
            eng.profile_enable(True)
            run(steps, 3000)
            eng.sync()
            prof = eng.profile()
            eng.profile_enable(False)
            eng.set_overlap(os.environ.get("RGCN_STREAMS", "1") != "0")
            ks = kernel_table(prof, steps, args)
            ks.sort(key=lambda k: -k["ms_per_step"])
            out["minibatch_step"]["kernels"] = [
                {f: k[f] for f in ("kernel", "launches_per_step", "avg_us", "ms_per_step", "bound", "achieved", "unit",
                                   "frac", "compulsory_bytes", "design_bytes", "design_gbs")} for k in ks]
        out["captured_step"] = {"ms_per_step_hipgraph_replay": round(wall_g * 1e3 / steps, 4),
                                "ms_per_step_stream_launched": round(wall_s * 1e3 / steps, 4),
                                "entry_point": "rgcn_train_step_device inside rgcn_capture_begin / _end",
                                "loss_after": round(eng.loss(), 6)}
        eng.graph_destroy(gid)
    finally:
        for b in held:
            b.free()
        eng.close()
    return out


TRAIN_LOOP_SETTINGS = """[Encoder]
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
	ReportTrainLossEvery=100000
	MaxIterations=%d
	[Algorithm]
		Name=Adam
		learning_rate=0.01
[General]
	NegativeSampleRate=10
	GraphSplitSize=0.5
	ExperimentName=/tmp/rgcn_bench_train_loop
	GraphBatchSize=30000
[Evaluation]
	Metric=MRR
"""


def measure_train_loop(args, device_step_ms, iterations=150, workers=(0, 8)):
    """The reference's WHOLE training iteration (code/train.py:161-247 + optimization/optimize.py:81-88) through this
    repository's driver, at settings/gcn_block.exp's values on the 272,115-edge synthetic FB15k-237 training graph: the
    neighbourhood edge sampler (30,000 picks), then ONE device call (edge dropout + negatives + prep + encoder + DistMult
    + clip + Adam) and the loss read back every iteration, as the reference's loop does -- with the sampler on the DEVICE
    (csrc/neighborhood.hip, the driver's default: no batch built on the host, nothing uploaded) and with the host
    sampler (librgcn.so's O(log V) port of the reference's process + upload of the graph batch) building batches in line
    and on k background threads.  Reported: ms per iteration of each form, the host's cost of building one batch, and
    the fraction of the iteration the device is idle -- row f4 of SURVEY 8 measured instead of asserted."""
    import contextlib
    import tempfile
    import numpy as np
    from relationprediction_amd import train
    from relationprediction_amd.common import model_builder, optimizer_parameter_parser, settings_reader
    from relationprediction_amd.optimization.optimize import build_hip
    V, R, E = 14541, 237, 272115
    triples = load_graph("synth:fb237_valid_test:%d" % E)
    out = {"workload": "fb237_block_train_loop", "train_graph_edges": E, "graph_batch": 30000, "graph_edges": 15000,
           "decoder_triples": 330000, "iterations": iterations,
           "sampler": "host, the reference's process (train.py:161-198) in O(log V) per pick"}
    with tempfile.TemporaryDirectory() as tmp, contextlib.redirect_stdout(sys.stderr):
        def build(max_iterations):
            path = os.path.join(tmp, "loop.exp")
            with open(path, "w") as f:
                f.write(TRAIN_LOOP_SETTINGS % max_iterations)
            s = settings_reader.read(path)
            general = s['General']
            general.put('EntityCount', V)
            general.put('RelationCount', R)
            general.put('EdgeCount', E)
            for part in ('Encoder', 'Decoder'):
                s[part].merge(s['Shared'])
                s[part].merge(general)
            s['Optimizer'].merge(general)
            return s, general
        s, general = build(iterations)
        encoder = model_builder.build_encoder(s['Encoder'], triples)
        model = model_builder.build_decoder(encoder, s['Decoder'])
        t_func = train.make_transform(triples, general, encoder, device_negatives=True, device_dropout=True)
        t_func_dev = train.make_transform(triples, general, encoder, device_negatives=True, device_dropout=True,
                                          device_sampler=True)
        model.preprocess(triples)
        model.register_for_test(triples)
        model.initialize_train()
        try:
            eng = model.get_runtime().engine
            eng.set_gemm_mode(args.gemm_mode)
            # host cost of one batch (sampler + bookkeeping), single thread
            ts = []
            for i in range(6):
                t0 = time.perf_counter()
                t_func.seeded(triples, 1000 + i)
                ts.append(time.perf_counter() - t0)
            out["host_batch_build_ms"] = round(float(np.median(ts[1:])) * 1e3, 3)

            def fit(n_iter, nworkers, transform):
                s2, _ = build(n_iter)
                opp = optimizer_parameter_parser.Parser(s2['Optimizer'])
                opp.set_save_function(lambda p: None)
                opp.set_sample_transform_function(transform)
                opt = build_hip(model, [p for p in opp.get_parametrization() if p[0] != 'ModelSaver'],
                                batch_workers=nworkers)
                np.random.seed(0)
                t0 = time.perf_counter()
                n = opt.fit(triples)
                return (time.perf_counter() - t0) * 1e3 / max(n, 1), n
            fit(12, 0, t_func)                            # warm-up: lazy allocations, first sampler state
            per_worker = {}
            for w in workers:
                ms, n = fit(iterations, w, t_func)
                per_worker[str(w)] = round(ms, 4)
            out["ms_per_iteration_by_builder_threads"] = per_worker
            # all three draws on the device (neighbourhood sampler = parallel first-passage percolation, edge dropout,
            # negatives): no batch is built on the host, nothing is uploaded
            fit(12, 0, t_func_dev)
            ms_dev, _ = fit(iterations, 0, t_func_dev)
            out["ms_per_iteration_device_sampler"] = round(ms_dev, 4)
            out["sampler"] = ("device (csrc/neighborhood.hip: the reference's process as parallel first-passage "
                              "percolation) for ms_per_iteration_device_sampler; host O(log V) port with k builder "
                              "threads for ms_per_iteration_by_builder_threads")
            dev_ms = float(device_step_ms)      # the same device call, launched back to back without host waits (train_steps)
            out["device_step_ms"] = round(dev_ms, 4)
            best = min(min(per_worker.values()), out["ms_per_iteration_device_sampler"])
            out["ms_per_iteration"] = best
            out["iterations_per_s"] = round(1e3 / best, 1)
            out["device_idle_frac"] = round(max(0.0, 1.0 - dev_ms / best), 4)
            out["loss_after"] = round(float(model.device_loss()), 6)
        finally:
            rt = model.get_runtime()
            if rt is not None and getattr(rt, "engine", None) is not None:
                rt.engine.close()
    return out


def measure_evaluation(args, queries=2000):
    """The evaluation half of the reference's loop ("next" row f3 of SURVEY 8; code/common/evaluation.py:148-153,349-389,
    model.py:59-81): one test-mode encoding of the full training graph (272,115 edges, SURVEY 8d graph B) and raw +
    filtered ranks of `queries` test triples on both sides (subject and object), filter lists from the graph itself."""
    import numpy as np
    from relationprediction_amd import _native
    from relationprediction_amd.common.shared_functions import init_encoder_params
    graph_name, V, R, d, L, kind, nb, E = WORKLOADS["fb237_block_traingraph"]
    triples = load_graph(graph_name)
    q = np.ascontiguousarray(load_graph("fb237_valid_test")[:queries])
    eng = _native.Engine(V, R, d, L, kind, nb, keep_prob=0.8, norm_mode="intended", max_edges=E)
    out = {"workload": "fb237_block_evaluation", "graph_edges": E, "queries": int(len(q)), "entities": V}
    try:
        eng.set_gemm_mode(args.gemm_mode)
        eng.set_params(init_encoder_params(V, R, d, L, kind, nb, rng=np.random.RandomState(1)))
        tri_dev = eng.to_device(triples)
        eng.set_graph_device(tri_dev, E)
        eng.forward(train=False)
        eng.sync()
        t0 = time.perf_counter()
        for _ in range(5):
            eng.set_graph_device(tri_dev, E)
            eng.forward(train=False)
        eng.sync()
        out["encode_full_graph_ms"] = round((time.perf_counter() - t0) * 1e3 / 5, 3)
        # filter lists (Scorer.known_*_triples): every completion of (entity, relation) seen in graph + queries
        allt = np.concatenate([triples, q]).astype(np.int64)
        eng.rank_reserve(1000)                                  # the reference scores chunks of 1000 triples
        ranks_ms, mrr = 0.0, {}
        for object_side in (True, False):
            ent, other = (allt[:, 0], allt[:, 2]) if object_side else (allt[:, 2], allt[:, 0])
            key = ent * R + allt[:, 1]
            order = np.argsort(key, kind="stable")
            ks, vs = key[order], other[order]
            qkey = (q[:, 0].astype(np.int64) if object_side else q[:, 2].astype(np.int64)) * R + q[:, 1]
            lo, hi = np.searchsorted(ks, qkey, "left"), np.searchsorted(ks, qkey, "right")
            idx = np.concatenate([np.unique(vs[a:b]) for a, b in zip(lo, hi)]).astype(np.int32)
            ptr = np.concatenate([[0], np.cumsum([len(np.unique(vs[a:b])) for a, b in zip(lo, hi)])]).astype(np.int64)
            # steady state (a test set is tens of such calls): the first call of an engine and side allocates / grows the
            # staging buffer and loads the kernels
            eng.ranks(q, object_side, ptr, idx)
            eng.sync()
            t0 = time.perf_counter()
            for _ in range(3):
                raw, filt = eng.ranks(q, object_side, ptr, idx)
            ranks_ms += (time.perf_counter() - t0) * 1e3 / 3
            mrr["object" if object_side else "subject"] = [round(float(np.mean(1.0 / raw)), 5),
                                                           round(float(np.mean(1.0 / filt)), 5)]
        out["rank_both_sides_ms"] = round(ranks_ms, 3)
        out["ranked_triples_per_s"] = round(len(q) / (ranks_ms * 1e-3), 1)
        out["mrr_raw_filtered_random_weights"] = mrr
        out["note"] = ("ranks: host call to host call (uploads of the queries and filter lists and the download of the "
                       "ranks included), mean of 3 calls per side after one warm-up call; scores of every query against "
                       "all %d entities, chunks of 1000" % V)
        tri_dev.free()
    finally:
        eng.close()
    return out


EVALUATION_ENCODES = {
    # name: (training-graph shape, V, R, kind, nb)  -- the graph model.py:59-81 feeds when it scores: the FULL train graph
    "fb237_basis_b2": ("synth:fb237_valid_test:272115", 14541, 237, "basis", 2),
    "wn18_block": ("synth:wn18_valid_test:141442", 40943, 18, "block", 100),
    "fb15k_block": ("synth:fb15k_minibatch:483142", 14951, 1345, "block", 100),
}


def measure_evaluation_encodes(args):
    """The evaluation encode (test-mode forward pass over the whole training graph, code/model.py:59-81) of BASELINE
    configs 3, 4 and 5 on one GPU: graph preparation + two layers, mean of 5 after a warm-up.  Element-wise parity of
    exactly these graphs: tests/test_gpu_parity.py (test_training_graph_elementwise_parity_basis,
    test_wn18_training_graph_elementwise_parity, test_fb15k_training_graph_evaluation_encode)."""
    import numpy as np
    from relationprediction_amd import _native
    from relationprediction_amd.common.shared_functions import init_encoder_params
    out = []
    for name, (graph_name, V, R, kind, nb) in EVALUATION_ENCODES.items():
        triples = load_graph(graph_name)
        E, d, L = int(triples.shape[0]), 500, 2
        with _native.Engine(V, R, d, L, kind, nb, keep_prob=0.8, norm_mode="intended", max_edges=E) as eng:
            eng.set_gemm_mode(args.gemm_mode)
            eng.set_params(init_encoder_params(V, R, d, L, kind, nb, rng=np.random.RandomState(1)))
            tri_dev = eng.to_device(triples)
            eng.set_graph_device(tri_dev, E)
            eng.forward(train=False)
            eng.sync()
            t0 = time.perf_counter()
            for _ in range(5):
                eng.set_graph_device(tri_dev, E)
                eng.forward(train=False)
            eng.sync()
            ms = (time.perf_counter() - t0) * 1e3 / 5
            tri_dev.free()
        out.append({"workload": name, "graph_edges": E, "entities": V, "relations": R, "kind": kind,
                    "encode_full_graph_ms": round(ms, 3), "edges_per_s": round(E / (ms * 1e-3), 1)})
    return out


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves (one process per GPU; plain
    subprocesses with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT in their environment -- the same
    contract torch.distributed.run provides), relay rank 0's stdout (its LAST line is the result line), send the
    other ranks' output to stderr, and exit non-zero if any rank fails."""
    import socket
    import subprocess
    n = args.gpus
    port = os.environ.get("MASTER_PORT")
    if not port:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = str(sk.getsockname()[1])
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RGCN_BENCH_SPAWNED="1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, stderr=sys.stderr, text=True))
    # a rank that never comes back (a collective one peer never joined) must not hold the caller for ever: after
    # RGCN_BENCH_TIMEOUT seconds (default 420) the ranks are killed -- these exact processes -- and the run fails
    limit = float(os.environ.get("RGCN_BENCH_TIMEOUT", "420"))
    try:
        out0, _ = procs[0].communicate(timeout=limit)
        rcs = [procs[0].returncode] + [q.wait(timeout=60) for q in procs[1:]]
    except subprocess.TimeoutExpired:
        for q in procs:
            if q.poll() is None:
                q.kill()
        sys.exit("bench.py --gpus %d: no result after %.0f s, ranks killed" % (n, limit))
    sys.stdout.write(out0)
    sys.stdout.flush()
    bad = [(r, rc) for r, rc in enumerate(rcs) if rc != 0]
    if bad:
        sys.exit("bench.py --gpus %d: rank(s) failed: %s" % (n, ", ".join("rank %d rc %d" % b for b in bad)))


def compact_line(out):
    """The one line the driver parses: contract keys + roofline + cpu_baseline(s) + step_roofline + summaries.
    Kept to a few KB (tests/test_gpu_multiprocess.py holds it under 8 KB); the tables live in bench_details.json."""
    line = {k: out[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                "scaling", "vs_baseline", "dtype", "data", "config")}
    for k in ("gpu_event_ms_per_step", "comm_ms_per_step", "rccl_ranks", "collectives", "message_edges_per_s",
              "speedup_vs_cpu"):
        if k in out:
            line[k] = out[k]
    rf = out.get("roofline")
    line["roofline"] = ({k: v for k, v in rf.items() if k != "basis"} if rf else None)
    sr = out.get("step_roofline")
    line["step_roofline"] = ({k: sr[k] for k in ("compulsory_bytes_per_step", "design_bytes_per_step", "pmc_bytes_per_step",
                                                  "alg_flops_per_step", "hbm_frac", "fp32_mfma_frac", "serial_ceiling_ms",
                                                  "frac_of_serial_ceiling", "sum_exclusive_kernel_ms")} if sr else None)
    cpu = out.get("cpu_baseline")
    line["cpu_baseline"] = ({k: cpu[k] for k in ("value", "unit", "cores", "kind", "sample", "ms_per_step")} if cpu else None)
    if out.get("cpu_baseline_reference_code"):
        line["cpu_baseline_reference_code"] = out["cpu_baseline_reference_code"]
    if out.get("fp32_mfma_reference"):
        line["fp32_mfma_reference_ms_per_step"] = out["fp32_mfma_reference"]["ms_per_step"]
    if out.get("steady_state"):
        line["steady_state"] = out["steady_state"]
    line["top_kernels"] = [[k["kernel"], k["launches_per_step"], k["avg_us"], k["frac"]] for k in out.get("kernels", [])[:6]]
    line["workloads"] = [
        {"workload": w["config"]["workload"], "value": w["value"], "ms_per_step": w["ms_per_step"],
         "kernel": (w.get("roofline") or {}).get("kernel"), "bound": (w.get("roofline") or {}).get("bound"),
         "frac": (w.get("roofline") or {}).get("frac"), "traffic": (w.get("roofline") or {}).get("traffic"),
         "ceil": (w.get("step_roofline") or {}).get("frac_of_serial_ceiling")} for w in out.get("workloads", [])]
    line["train_steps"] = [
        {"workload": t["workload"], "ms_per_step": t["minibatch_step"]["ms_per_step"],
         "edges_per_s": t["minibatch_step"]["edges_per_s"],
         "hipgraph_ms": t["captured_step"]["ms_per_step_hipgraph_replay"],
         "stream_ms": t["captured_step"]["ms_per_step_stream_launched"],
         # which mode the row's numbers are: ms_per_step = the whole iteration (device edge dropout + negatives + step),
         # stream-launched -- the documented default for every configuration, config 5 included; hipgraph_ms / stream_ms =
         # the SAME fixed step (rgcn_train_step_device) replayed from a captured hipGraph / launched on the four streams
         "step": ("ms_per_step: stream-launched rgcn_train_step_minibatch_device (default mode); hipgraph_ms: "
                  "rgcn_train_step_device replayed from a captured hipGraph; stream_ms: the same call stream-launched. "
                  "Replay %s stream launch here (%.3f vs %.3f ms); DESIGN.md section 6"
                  % ("beats" if t["captured_step"]["ms_per_step_hipgraph_replay"] < t["captured_step"]["ms_per_step_stream_launched"]
                     else "does not beat", t["captured_step"]["ms_per_step_hipgraph_replay"],
                     t["captured_step"]["ms_per_step_stream_launched"])),
         "top": [[k["kernel"], k["avg_us"], k["frac"]] for k in t["minibatch_step"].get("kernels", [])[:3]]}
        for t in out.get("train_steps", [])]
    ev = out.get("evaluation")
    line["evaluation"] = ({k: ev[k] for k in ("graph_edges", "queries", "encode_full_graph_ms", "rank_both_sides_ms",
                                               "ranked_triples_per_s")} if ev else None)
    line["evaluation_encodes"] = [[e["workload"], e["graph_edges"], e["encode_full_graph_ms"]]
                                  for e in out.get("evaluation_encodes", [])]
    tl = out.get("train_loop")
    line["train_loop"] = ({k: tl[k] for k in ("ms_per_iteration", "iterations_per_s", "device_step_ms", "device_idle_frac",
                                               "host_batch_build_ms", "ms_per_iteration_by_builder_threads",
                                               "ms_per_iteration_device_sampler")} if tl else None)
    line["details"] = out.get("details")
    return line


def reference_code_baseline():
    """The reference's OWN model code (code/common/model_builder.py chain over tests/golden/tf_numpy_shim.py forward +
    tests/golden/tf_torch_shim.py autograd) timed on BASELINE config 2's minibatch.  /root/reference does not exist on
    the GPU box, so this is measured where it does (tests/golden/time_reference_code.py, in the build container) and
    committed; reported with its own host description, never mixed into `cpu_baseline`."""
    try:
        with open(os.path.join(ROOT, "tests", "golden", "reference_code_timing.json")) as f:
            t = json.load(f)
        return {k: t[k] for k in ("value", "unit", "cores", "kind", "sample", "ms_per_step", "host", "measured_in_this_run")}
    except Exception:  # noqa: BLE001
        return None


def write_details(out):
    """per-kernel tables of every workload / train step: a side file, not the result line"""
    paths = [os.path.join(ROOT, "bench_details.json")]
    if os.path.isdir(os.path.join(ROOT, "gpurun_out")):
        paths.append(os.path.join(ROOT, "gpurun_out", "bench_details.json"))
    written = None
    for p in paths:
        try:
            with open(p, "w") as f:
                json.dump(out, f, indent=1)
            written = written or os.path.relpath(p, ROOT)
        except OSError:
            pass
    return written


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn_ranks(args)
    rk = Ranks(args)
    if rk.world > 1:
        # launched by torch.distributed.run (or by spawn_ranks above): a watchdog of its own, for the same reason
        import threading

        def give_up():
            sys.stderr.write("bench.py rank %d: no result after %s s, giving up\n" % (rk.rank, os.environ.get("RGCN_BENCH_TIMEOUT", "420")))
            sys.stderr.flush()
            os._exit(124)
        dog = threading.Timer(float(os.environ.get("RGCN_BENCH_TIMEOUT", "420")), give_up)
        dog.daemon = True
        dog.start()
    if os.environ.get("RGCN_BENCH_IMPORT_TORCH") == "1":
        import torch  # noqa: F401  (test knob: exercise a torch-first load order on one GPU)

    out = measure(args.workload, args, rk, args.steps, args.warmup, full=True)
    extras = []
    widen = rk.world == 1 and args.workload == "fb237_block" and not args.no_extra_workloads and not args.hipgraph
    if widen:
        for w in EXTRA_WORKLOADS:
            o = measure(w, args, rk, args.extra_steps, min(args.warmup, 5), full=False)
            for drop in ("cpu_baseline", "fp32_mfma_reference", "higher_is_better", "vs_baseline"):
                o.pop(drop, None)
            extras.append(o)
    train_steps = []
    if widen:
        for w in TRAIN_STEP_WORKLOADS:
            train_steps.append(measure_train_step(w, args, args.extra_steps, min(args.warmup, 5)))
    evaluation = measure_evaluation(args) if widen else None
    evaluation_encodes = measure_evaluation_encodes(args) if widen else []
    train_loop = (measure_train_loop(args, train_steps[0]["minibatch_step"]["ms_per_step"],
                                     iterations=max(20, 6 * args.extra_steps)) if widen else None)
    if rk.rank == 0 and rk.world == 1 and not args.no_live_traffic and not args.no_kernel_profile:
        apply_live_traffic(out, args)
    if rk.rank == 0:
        out["workloads"] = extras
        out["train_steps"] = train_steps
        out["evaluation"] = evaluation
        out["evaluation_encodes"] = evaluation_encodes
        out["train_loop"] = train_loop
        if rk.world == 1 and args.workload == "fb237_block":
            out["cpu_baseline_reference_code"] = reference_code_baseline()
    if rk.world > 1 and rk.rank == 0 and os.path.exists(rk.rdv_path):
        os.remove(rk.rdv_path)

    # The ONE JSON line goes out last: every rank first shuts its engine down and pushes out whatever C-level
    # libraries (RCCL's debug facility writes to stdout) left in the stdio buffer, and says so through a marker
    # file next to the rendezvous file; rank 0 prints when all ranks have done that (or after a minute).
    import ctypes
    try:
        ctypes.CDLL(None).fflush(None)
    except OSError:
        pass
    sys.stdout.flush()
    if rk.world > 1:
        marker = "%s.done" % rk.rdv_path
        if rk.rank != 0:
            open("%s.%d" % (marker, rk.rank), "w").close()
        else:
            deadline = time.time() + 60
            others = ["%s.%d" % (marker, r) for r in range(1, rk.world)]
            while time.time() < deadline and not all(os.path.exists(p) for p in others):
                time.sleep(0.02)
            for p in others:
                if os.path.exists(p):
                    os.remove(p)
    if rk.rank == 0:
        out["details"] = write_details(out)
        print(json.dumps(compact_line(out), separators=(",", ":")), flush=True)


if __name__ == "__main__":
    main()
