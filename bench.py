"""Benchmark of the hot path on synthetic data at the Elec shape.

    python bench.py --gpus N --steps K --warmup W

A *step* is one full-ranking evaluation pass over ALL users: fused score + train
mask + top-20 + the six metrics at k = 5, 10, 20 from embeddings resident in HBM
(the model forward is timed separately).  `value` = users ranked per second by
the whole job.  With N > 1 the users are cut into one block per rank, each ranked
against the whole item table on its GPU (users are independent: only the metric
sums cross ranks); total work is fixed, so scaling is "strong".  The item-sharded
form (per-GPU top-K over an item shard, exchange of the lists over RCCL, merge) is
timed beside it (`item_sharded_eval`), as are the row-sharded hop and training step.

Also reported on the same JSON line: the SpMM hop (algorithmic GB/s against the
HBM roofline, `roofline_spmm`), the fused top-K call against the MFMA roofline of
its dominant kernel (`roofline`: the fp16 candidate filter at this size; the exact
fp32 kernel beside it as `roofline_exact_kernel`), one full MMFCN
training epoch timed step by step (`epoch`), and the CPU baseline (`cpu_baseline`,
the oracle's reference-style evaluation on the host cores, rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from recommend_amd import ops, synth  # noqa: E402
from recommend_amd.dist import ShardedRankingEvaluator, UserShardedRankingEvaluator  # noqa: E402
from recommend_amd.evaluator import RankingEvaluator  # noqa: E402
from recommend_amd.graph_constructor import GraphConstructor  # noqa: E402
from recommend_amd.loss import gather_dot_bpr_loss  # noqa: E402
from recommend_amd.model import ModelFactory  # noqa: E402
from recommend_amd.optim import FusedClipAdam  # noqa: E402
from recommend_amd.sampler import NegativeSampler, TrainBatchLoader, build_user_csr  # noqa: E402
from recommend_amd.train_step import GraphedTrainStep  # noqa: E402

METRICS = ["precision", "recall", "hit_rate", "ndcg", "map", "mrr"]
K_LIST = [5, 10, 20]
HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec
# Memory-side bytes per launch come from rocprofv3 PMC passes over THIS program (`--pmc-pass`, one
# counter per run: tools/pmc_bench.sh), summarised by tools/pmc_traffic.py into the newest
# profiles/r*_pmc_traffic.json.  The summary names the workload and a digest of the kernel sources it
# was collected from; a line measured on another workload or on changed sources reports null.
MFMA_F32_PEAK_TFS = 157.3    # MI355X fp32 matrix peak
MFMA_F16_PEAK_TFS = 2500.0   # MI355X dense fp16 / bf16 matrix peak
BATCH = {"baby": 2048, "clothing": 1024, "sports": 256, "elec": 256}   # reference config.py presets
EMB_DIM = {"baby": 64, "clothing": 128, "sports": 64, "elec": 64}       # config.py:188,201,212,223
LR = {"baby": 1e-3, "clothing": 5e-4, "sports": 1e-3, "elec": 1e-3}    # config.py:64,203,214,225


def kernel_source_digest():
    """sha256 over the kernel sources, so counter summaries can be tied to the code they measured."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "recommend_amd", "csrc", "*"))):
        if f.endswith((".hip", ".h", ".cpp")):
            h.update(os.path.basename(f).encode())
            h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def pmc_traffic(group, shape, D, world):
    """(bytes per launch, file it came from) of a kernel group from the newest counter summary, or
    (None, reason) when there is none for this workload and these sources."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))
    if not files:
        return None, "no profiles/r*_pmc_traffic.json"
    name = os.path.relpath(files[-1], ROOT)
    with open(files[-1]) as f:
        doc = json.load(f)
    if doc.get("shape") != shape or doc.get("emb_dim") != D or world != 1:
        return None, f"{name} was collected on {doc.get('shape')} / emb_dim {doc.get('emb_dim')} / 1 GPU"
    if doc.get("kernel_source_digest") != kernel_source_digest():
        return None, f"{name} was collected from other kernel sources ({doc.get('kernel_source_digest')})"
    return doc["groups"][group]["traffic_bytes"], name


def log(*a):
    if int(os.environ.get("RANK", "0")) == 0:
        print(*a, file=sys.stderr, flush=True)


def make_config(U, I, D, device, dropout):
    class C:
        class data:
            num_users, num_items = U, I
        class model:
            model_name, modal_num, emb_dim, layer_num = "MMFCN", 2, D, 2
        class graph:
            weight_feature = [None]
            edge_weight_type, add_self_loops, normalize_adj = "uniform", True, True
        class system:
            pass
    C.model.dropout = dropout
    C.system.device = device
    return C


def event_ms(fn, reps):
    """Average device time of fn() over reps launches, HIP events on the current stream."""
    start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    start.record()
    for _ in range(reps):
        fn()
    stop.record()
    stop.synchronize()
    return start.elapsed_time(stop) / reps


def cpu_baseline_eval(ue, ie, train, val, chunk_users, budget_s=8.0):
    """Reference-style evaluation (oracle/eval_oracle.verify: matmul -> index_put(-1e10) ->
    topk x3 -> metrics incl. the per-row python loop) on chunks of users until the time budget
    is spent, all host cores.  Metrics are per-user means, so chunking is exact."""
    from oracle import eval_oracle
    tu, ti = train
    vu, vi = val
    U, I = ue.shape[0], ie.shape[0]
    ic = torch.from_numpy(ie)
    torch.set_num_threads(min(os.cpu_count(), 16))
    done, total = 0, 0.0
    while done < U and total < budget_s:
        lo, hi = done, min(done + chunk_users, U)
        tm = (tu >= lo) & (tu < hi)
        vm = (vu >= lo) & (vu < hi)
        target = torch.zeros((hi - lo, I))
        target[torch.from_numpy(vu[vm] - lo), torch.from_numpy(vi[vm])] = 1
        row_id, col_id = (tu[tm] - lo).tolist(), ti[tm].tolist()
        uc = torch.from_numpy(ue[lo:hi])
        t0 = time.perf_counter()
        eval_oracle.verify(uc, ic, target, row_id, col_id, K_LIST, METRICS)
        total += time.perf_counter() - t0
        done = hi
    return done / total, total, torch.get_num_threads(), done


def cpu_baseline_train(model, feats, coo, batch, U, I, reps_spmm=5):
    """The oracle's CPU training step (torch CPU ops as the reference applies them, torch.sparse.mm
    standing in for DGL): one SpMM hop timed alone, and one forward + BPR loss + backward."""
    from oracle import graph_oracle, loss_oracle, mmfcn_oracle
    src, dst, w = coo
    N = U + I
    wn, _ = graph_oracle.normalize(src, dst, w, N)
    A = mmfcn_oracle.adjacency(src, dst, wn, N)
    h = torch.randn(N, model.config.model.emb_dim)
    torch.sparse.mm(A, h)
    t0 = time.perf_counter()
    for _ in range(reps_spmm):
        torch.sparse.mm(A, h)
    ms_hop = 1e3 * (time.perf_counter() - t0) / reps_spmm
    p = {k: v.detach().cpu().clone().requires_grad_(v.dtype.is_floating_point) for k, v in model.state_dict().items()}
    padded = [torch.cat([torch.zeros(U, f.shape[1]), f]) for f in feats.values()]
    embs = model.embs.detach().cpu().clone().requires_grad_(True)
    alphas = model.alphas.detach().cpu().clone().requires_grad_(True)
    for _ in range(2):       # the first step pays allocator and thread-pool start-up: the second is reported
        t0 = time.perf_counter()
        ue, ie = mmfcn_oracle.forward(p, A, padded, ["linears.0.linear", "linears.1.linear"], embs, model.ks, alphas,
                                      model.config.model.layer_num, U)
        loss = loss_oracle.trainer_loss({"user_embeddings": ue, "item_embeddings": ie},
                                        {k: batch[k].cpu() for k in ("user_ids", "item_ids", "neg_items")})
        loss.backward()
        ms_step = 1e3 * (time.perf_counter() - t0)
    return ms_hop, ms_step


def free_port():
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def launch_ranks(n, backend):
    """`python bench.py --gpus N` without a launcher: this process starts the N ranks itself (one
    child process per GPU, the same command line, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the
    environment), hands rank 0's JSON line through and exits with the worst exit code.  It runs
    before anything here has touched a GPU (counting devices does not), so no process that holds
    the GPU ever starts another program in its place."""
    import subprocess
    have = torch.cuda.device_count()
    if backend == "nccl" and have < n:
        print(f"[bench] --gpus {n} on the nccl (RCCL) backend needs {n} visible GPUs, this machine shows {have}: "
              "RCCL does not run two ranks on one device.  Nothing was measured.  (BENCH_BACKEND=gloo rehearses "
              "the N-rank code with the ranks sharing the visible GPUs and the collectives going through the host.)",
              file=sys.stderr, flush=True)
        return 2
    env = dict(os.environ, WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()),
               BENCH_BACKEND=backend)
    procs = []
    for r in range(n):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=e,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    limit = float(os.environ.get("BENCH_JOB_LIMIT_S", "1500"))
    t0, worst, grace_end = time.time(), 0, None
    try:
        while True:
            codes = [p.poll() for p in procs]
            if all(c is not None for c in codes):
                break
            now = time.time()
            if grace_end is None and any(c not in (None, 0) for c in codes):
                # a rank has failed: the others sit in a collective that will never complete
                grace_end = now + float(os.environ.get("BENCH_FAIL_GRACE_S", "60"))
                bad = [i for i, c in enumerate(codes) if c not in (None, 0)]
                print(f"[bench] rank(s) {bad} exited with {[codes[i] for i in bad]}; ending the job", file=sys.stderr, flush=True)
            if now - t0 > limit or (grace_end is not None and now > grace_end):
                if grace_end is None:
                    print(f"[bench] the {n}-rank job did not finish within {limit:.0f} s", file=sys.stderr, flush=True)
                worst = 124
                break
            time.sleep(0.2)
    finally:
        for p in procs:           # the exact processes started above, nothing else
            if p.poll() is None:
                p.kill()
        for p in procs:
            p.wait()
    # whatever rank 0 has printed (its one line fits the pipe's buffer, so it never blocked on it)
    sys.stdout.write(procs[0].stdout.read().decode())
    sys.stdout.flush()
    for p in procs:
        if worst != 124 or p.returncode not in (None, -9):     # our own kill is not a rank's failure
            worst = max(worst, abs(p.returncode or 0))
    return worst


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--shape", default="elec", choices=list(synth.SHAPES))
    ap.add_argument("--emb-dim", type=int, default=None, help="default: the reference preset of the shape")
    ap.add_argument("--epoch-steps", type=int, default=-1,
                    help="training steps to time for the epoch figure (-1: the whole epoch, 0: skip)")
    ap.add_argument("--cpu-users", type=int, default=8192, help="users in the CPU baseline sample (0: skip)")
    ap.add_argument("--dropout", type=float, default=0.3)
    ap.add_argument("--spmm-reps", type=int, default=200, help="SpMM launches per timing")
    ap.add_argument("--scaleup-slice", type=int, default=1,
                    help="1: also time the hop on one rank's row block of the large configuration (8 GB of device memory, ~10 s)")
    ap.add_argument("--pmc-pass", action="store_true",
                    help="short run for a rocprofv3 --pmc pass: few repetitions, no exact-kernel, epoch or CPU leg; "
                         "prints how many calls of each kind were made so counters can be divided per call")
    args = ap.parse_args()
    if args.pmc_pass:
        args.steps, args.warmup, args.epoch_steps, args.cpu_users, args.spmm_reps = 2, 0, 0, 0, 5
    if args.emb_dim is None:
        args.emb_dim = EMB_DIM[args.shape]

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus, os.environ.get("BENCH_BACKEND", "nccl")))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != max(args.gpus, 1) and rank == 0:
        log(f"[bench] the launcher started {world} ranks, --gpus says {args.gpus}: measuring {world}")
    local = int(os.environ.get("LOCAL_RANK", "0"))
    local = local % max(torch.cuda.device_count(), 1)   # BENCH_BACKEND=gloo rehearsals share one GPU
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # BENCH_FORCE_DIST=1: run the N > 1 code (process group, sharded evaluators, row-sharded legs) with
    # whatever world size the launcher gave, one rank included -- a rehearsal of the collectives on one GPU
    dist_on = world > 1 or os.environ.get("BENCH_FORCE_DIST") == "1"
    if dist_on:
        backend = os.environ.get("BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            # rehearsal of the N-rank code with the ranks sharing the visible GPUs: gloo moves host memory
            dist.init_process_group(backend)
            from recommend_amd import dist as rk_dist
            from tools.host_staged_transport import HostStagedTransport
            rk_dist.set_transport(HostStagedTransport())

    U, I, E = synth.SHAPES[args.shape]
    D = args.emb_dim
    t0 = time.time()
    data = synth.make_named(args.shape)
    tu, ti, tr = data.split(0)
    vu, vi, _ = data.split(1)
    log(f"[bench] {args.shape}: {U} users x {I} items, {len(tu)} train / {len(vu)} val interactions "
        f"({time.time() - t0:.1f}s to generate)")

    # ---------------------------------------------------------------- model + graph
    g = torch.Generator(device="cpu").manual_seed(1234)
    feats = {"image": torch.randn(I, 4096, generator=g), "text": torch.nn.functional.normalize(torch.randn(I, 384, generator=g))}
    torch.manual_seed(42)
    cfg = make_config(U, I, D, str(dev), args.dropout)
    model = ModelFactory.create_Model(cfg, {}, feats).to(dev)
    t0 = time.time()
    model.build_graph((tu, ti, tr))
    model.creat_feature_weight()
    torch.cuda.synchronize()
    gc: GraphConstructor = model.graph_constructor
    N, nnz = U + I, gc.graph.num_edges()
    log(f"[bench] graph: {N} nodes, {nnz} edges, long rows fwd/bwd {gc.graph.fwd.n_long}/{gc.graph.bwd.n_long} "
        f"({time.time() - t0:.1f}s to build + normalise)")

    model.eval()
    with torch.no_grad():
        out = model({})
        torch.cuda.synchronize()
        t_fwd = event_ms(lambda: model({}), 3)
    ue, ie = out["user_embeddings"].contiguous(), out["item_embeddings"].contiguous()
    if not (bool(torch.isfinite(ue).all()) and bool(torch.isfinite(ie).all())):
        raise RuntimeError("the model forward produced non-finite embeddings: nothing measured on them would mean anything")

    # ---------------------------------------------------------------- evaluation (the timed step)
    if dist_on:
        # users are independent units: every rank ranks its block of users against the whole item table,
        # no candidate lists cross ranks (the item-sharded form is timed further down, beside it)
        ev = UserShardedRankingEvaluator((vu, vi), (tu, ti), K_LIST, METRICS, U, I, rank, world)
    else:
        ev = RankingEvaluator((vu, vi), (tu, ti), K_LIST, METRICS, n_users=U)
    for _ in range(args.warmup):
        res = ev.evaluate(ue, ie)
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = ev.evaluate(ue, ie)
    torch.cuda.synchronize()
    if dist_on:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist_on:
        from recommend_amd.dist import all_reduce_tensor
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        all_reduce_tensor(t, dist.ReduceOp.MAX)
        elapsed = float(t)
    ms_per_step = 1e3 * elapsed / args.steps
    users_per_s = U * args.steps / elapsed

    # dominant kernels of the step: the fused score + mask + top-K launch sequence.  At this size
    # it is the fp16 candidate filter + exact re-scoring (results identical to the exact fp32
    # kernel, which is timed beside it against its own, fp32-input MFMA peak).
    u_loc = U if not dist_on else (ev.u_hi - ev.u_lo)     # this rank's share of the score matrix
    topk_fn = (lambda: ev.topk(ue, ie, 20)) if not dist_on else (lambda: ev.local_topk(ue, ie, 20))
    topk_fn()
    ms_topk = event_ms(topk_fn, max(3, args.steps))
    topk_calls = args.warmup + args.steps + 1 + max(3, args.steps)
    flop = 2.0 * u_loc * I * D
    tflops = flop / (ms_topk * 1e-3) / 1e12
    filtered = u_loc * I >= (1 << 27) and D in (64, 128)   # what rk_score_topk_f32 decides
    peak = MFMA_F16_PEAK_TFS if filtered else MFMA_F32_PEAK_TFS
    traffic_topk, traffic_topk_src = pmc_traffic("score_topk_call", args.shape, D, world)
    roofline = {"kernel": ("score_topk_kernel<fp16 filter> + merge + refine + exact fallback (one rk_score_topk_f32 call)"
                           if filtered else "score_topk_kernel (+topk_merge_kernel)"),
                "bound": "mfma", "achieved": round(tflops, 3), "peak": peak, "unit": "TFLOP/s",
                "frac": round(tflops / peak, 4),
                "traffic": traffic_topk, "traffic_source": traffic_topk_src,
                "traffic_unit": "B per call (L2 memory-side requests, Infinity-Cache hits included)",
                "ms_per_launch": round(ms_topk, 4),
                "algorithmic": f"2*U*I*D = {flop:.4g} flop per call" + ("" if world == 1 else " (this rank's users)"),
                "note": ("peak is the dense fp16 MFMA rate, the arithmetic of the dominant kernel; that kernel is bound "
                         "by the vector instructions of the top-K selection, not by the matrix pipe" if filtered else "")}
    roofline_exact = None
    if not dist_on and not args.pmc_pass:
        exact_fn = lambda: ev.topk(ue, ie, 20, mode="exact")
        exact_fn()
        ms_exact = event_ms(exact_fn, max(3, args.steps))
        tf_exact = flop / (ms_exact * 1e-3) / 1e12
        roofline_exact = {"kernel": "score_topk_kernel<fp32> + topk_merge (rk_score_topk_exact_f32)", "bound": "mfma",
                          "achieved": round(tf_exact, 3), "peak": MFMA_F32_PEAK_TFS, "unit": "TFLOP/s",
                          "frac": round(tf_exact / MFMA_F32_PEAK_TFS, 4), "ms_per_launch": round(ms_exact, 4)}

    # ---------------------------------------------------------------- SpMM hop
    h = torch.randn(N, D, device=dev)
    buf = torch.empty_like(h)
    gc.graph.refresh_values("weight")
    spmm_fn = lambda: ops.spmm_raw(gc.graph.fwd, h, out=buf)
    spmm_fn()
    ms_spmm = event_ms(spmm_fn, args.spmm_reps)
    traffic_spmm, traffic_spmm_src = pmc_traffic("spmm_hop", args.shape, D, 1)
    spmm_bytes = nnz * 8 + (N + 1) * 4 + 2 * N * D * 4
    gbs = spmm_bytes / (ms_spmm * 1e-3) / 1e9
    roofline_spmm = {"kernel": "spmm_rows_kernel (one launch per hop; long rows folded by the last-arriving chunk)", "bound": "hbm",
                     "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(gbs / HBM_PEAK_GBS, 4),
                     "traffic": traffic_spmm, "traffic_source": traffic_spmm_src,
                     "ms_per_launch": round(ms_spmm, 5),
                     "algorithmic": f"nnz*8 + (N+1)*4 + 2*N*D*4 = {spmm_bytes} B per hop"}
    ms_spmm_t = event_ms(lambda: ops.spmm_raw(gc.graph.bwd, h, out=buf), args.spmm_reps)
    roofline_spmm["note"] = ("at this shape the hop is bound by the L1 miss queues (63 requests of 128 B in flight per CU at "
                             "~600 cycles each), not by HBM bytes: with every gather an L1 hit the same kernel takes 52 us = "
                             "0.37 of the HBM peak, which is the ceiling of 256-B fp32 row gathers on this chip; 0.60 is "
                             "not reachable at fp32")

    # the same kernel on what ONE of eight ranks holds of the large configuration (10M x 2M x 500M,
    # emb_dim 128: 1.5M rows, 512-B rows gathered from a 6.1 GB layer): checked against float64 on a
    # sample of rows; bytes = index arrays + every distinct layer row named, once + output.  The last
    # rank's block (cold item rows reading user rows, nearly all distinct) is the one HBM bounds; the
    # first rank's (user rows reading the Zipf-popular item rows) is served from the caches.
    roofline_slice = None
    if args.scaleup_slice and not dist_on and not args.pmc_pass:
        from tools.scaleup_sizing import rank_slice_hop

        def slice_line(rank, what):
            r = rank_slice_hop(dev, rank)
            if r["max_rel_err_vs_float64"] > 1e-5:
                raise RuntimeError(f"scale-up slice hop differs from float64 by {r['max_rel_err_vs_float64']:.2e}")
            gbs_s = r["compulsory_bytes"] / (r["ms"] * 1e-3) / 1e9
            log(f"[bench] scale-up slice, rank {rank}: {r['ms']:.3f} ms, {gbs_s:.0f} GB/s of compulsory bytes")
            return {"kernel": f"spmm_rows_kernel<32> on the row block of rank {rank} of 8 of the 10M x 2M x 500M configuration ({what})",
                    "bound": "hbm", "achieved": round(gbs_s, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(gbs_s / HBM_PEAK_GBS, 4), "ms_per_launch": round(r["ms"], 4),
                    "rows": r["rows"], "nnz": r["nnz"], "distinct_rows_gathered": r["distinct_rows_gathered"],
                    "algorithmic": f"nnz*8 + (rows+1)*4 + distinct_rows*D*4 + rows*D*4 = {r['compulsory_bytes']} B per hop",
                    "gather_level_GBs": round(r["gather_level_bytes"] / (r["ms"] * 1e-3) / 1e9, 1),
                    "max_rel_err_vs_float64": float(f"{r['max_rel_err_vs_float64']:.3g}")}
        roofline_slice = slice_line(7, "cold item rows gathering user rows: nearly every gathered row is a different one")
        roofline_slice["traffic"] = None
        roofline_slice["traffic_note"] = ("profiles/r03_pmc_scaleup_slices.txt: 20.3 GB per launch leave the L2 (every gather of a "
                                          "row that 3.7 edges of the block name goes to memory: 5.2 TB/s = 0.65 of the HBM peak on the "
                                          "traffic it causes, 0.22 on the bytes it cannot avoid; the graph has no locality to reuse)")
        roofline_slice["user_rows_block"] = slice_line(0, "user rows gathering the Zipf-popular item rows, half of them L2 hits")

    # ---------------------------------------------------------------- one training epoch
    epoch = trained = None
    if args.epoch_steps != 0 and not dist_on:
        p_ptr, p_items = build_user_csr(data.users, data.items, U)
        sampler = NegativeSampler(p_ptr, p_items, I, seed=42)
        dp, di = sampler.device_csr(dev)
        loader = TrainBatchLoader(tu, ti, tr, BATCH[args.shape], sampler, device=dev, resolve_on_host=False)
        opt = FusedClipAdam(model.parameters(), lr=LR[args.shape])
        model.train()
        n_steps = len(loader) if args.epoch_steps < 0 else min(args.epoch_steps, len(loader))
        it = iter(loader)
        # the step (forward, BPR loss with in-kernel negative lookup, backward, clip, Adam) is
        # captured once as a HIP graph and replayed per batch
        first = next(it)
        stepper = GraphedTrainStep(model, opt, gather_dot_bpr_loss, first,
                                   constants={"pos_ptr": dp, "pos_items": di}, max_grad_norm=1.0)
        pending = [first]

        def train_step():
            batch = pending.pop() if pending else next(it)
            return stepper(batch)
        for _ in range(3):
            train_step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for s in range(n_steps - 3):
            loss = train_step()
        last = float(loss)
        torch.cuda.synchronize()
        t_train = time.perf_counter() - t0
        ms_train = 1e3 * t_train / max(n_steps - 3, 1)
        full = len(loader)
        epoch = {"train_steps_timed": n_steps - 3, "steps_per_epoch": full, "batch_size": BATCH[args.shape],
                 "ms_per_train_step": round(ms_train, 4), "last_loss": round(last, 6),
                 "epoch_time_s": round((ms_train * full + t_fwd + ms_per_step) / 1e3, 3),
                 "composition": "steps_per_epoch * ms_per_train_step + one eval forward + one eval step",
                 "whole_epoch_timed": n_steps == full}
        log(f"[bench] epoch: {n_steps - 3} steps timed, {ms_train:.3f} ms/step")
        # the evaluation step again, on the embeddings an evaluator sees after training: the filter's
        # speed depends on how many users it certifies, which depends on the embeddings
        model.eval()
        with torch.no_grad():
            out_t = model({})
        ue_t, ie_t = out_t["user_embeddings"].contiguous(), out_t["item_embeddings"].contiguous()
        model.train()
        for _ in range(max(args.warmup, 1)):
            res_t = ev.evaluate(ue_t, ie_t)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            res_t = ev.evaluate(ue_t, ie_t)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        trained = {"value_trained": round(U * args.steps / dt, 1), "ms_per_step_trained": round(1e3 * dt / args.steps, 4),
                   "train_steps_before": n_steps, "recall@20_trained": round(float(res_t["recall@20"]), 6),
                   "open_users_trained": ops.score_topk_open_users(ue_t, ie_t, 20, *ev.mask_csr(dev)),
                   "open_users": ops.score_topk_open_users(ue, ie, 20, *ev.mask_csr(dev)),
                   "item_norm_max_over_median": round(float(ie_t.norm(dim=1).max() / ie_t.norm(dim=1).median()), 3),
                   "note": "the same step on the model's embeddings after the timed training steps; open_users = users "
                           "the candidate filter left to the exact kernel (of them up to 512 by rule: the longest training lists)"}
        log(f"[bench] trained embeddings: {trained['ms_per_step_trained']:.3f} ms/step, open users "
            f"{trained['open_users_trained']} (at initialisation {trained['open_users']})")

    # ---------------------------------------------------------------- CPU baseline
    cpu = None
    if rank == 0 and not dist_on and args.cpu_users > 0:
        v, dt, cores, n = cpu_baseline_eval(ue.cpu().numpy(), ie.cpu().numpy(), (tu, ti), (vu, vi), args.cpu_users)
        cpu = {"value": round(v, 1), "unit": "users/s", "cores": cores, "host_cpus": os.cpu_count(), "kind": "port",
               "sample": f"reference-style verify (matmul, index_put -1e10, topk x3, metrics) on the first {n} "
                         f"users x {I} items in chunks of {args.cpu_users}, {dt:.2f} s"}
        if epoch is not None:
            from oracle import graph_oracle
            coo = graph_oracle.build_coo_vectorized(tu, ti, tr, U, I, True)
            B = BATCH[args.shape]     # any batch of the right size times the same work
            cpu_batch = {"user_ids": torch.from_numpy(tu[:B].astype(np.int64)),
                         "item_ids": torch.from_numpy(ti[:B].astype(np.int64)),
                         "neg_items": torch.randint(0, I, (B, 1))}
            ms_hop, ms_step = cpu_baseline_train(model, feats, coo, cpu_batch, U, I)
            cpu["spmm_hop_ms"] = round(ms_hop, 3)
            cpu["train_step_ms"] = round(ms_step, 1)
            cpu["train_sample"] = "one torch.sparse.mm hop (mean of 5) and one forward + BPR loss + backward of the " \
                                  "oracle's MMFCN at the same shape (the second of two steps)"

    # N > 1: the same evaluation with the ITEMS cut over the ranks instead of the users (per-GPU top-K over
    # its item shard, exchange of the per-user lists, merge; for item tables too large for one GPU): timed
    # like the headline step, reported beside it.
    user_sharded = None
    sharded_hop = sharded_train = None

    def emit(error=None):
        if rank == 0:
            line = {
                "metric": "MMFCN epoch time + full-rank top-20 eval users/sec (Elec shape), SpMM HBM GB/s",
                "value": round(users_per_s, 1), "unit": "users/s", "n_gpus": world, "steps": args.steps,
                "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
                "scaling": "strong", "vs_baseline": None,
                "dtype": "f32 (results; candidates by fp16 MFMA, certified against the fp32 chain)",
                "data": "synthetic",
                "config": {"workload": f"{args.shape} shape: {U} users x {I} items, {len(tu)} train interactions, "
                                       f"emb_dim {D}, full-rank top-20 + metrics@5/10/20 per step",
                           "parallelism": "single GPU" if world == 1 else f"users in {world} blocks (one per GPU) against the whole item table, all-reduce of the metric sums"},
                "roofline": roofline, "roofline_exact_kernel": roofline_exact, "roofline_spmm": roofline_spmm,
                "roofline_spmm_scaleup_slice": roofline_slice,
                "spmm_transpose_ms": round(ms_spmm_t, 5),
                "eval_forward_ms": round(t_fwd, 3), "epoch": epoch, "trained": trained, "cpu_baseline": cpu,
                "recall@20": round(float(res["recall@20"]), 6),
                "kernel_source_digest": kernel_source_digest(),
            }
            if trained is not None:
                line["value_trained"] = trained["value_trained"]
            if error is not None:
                line["error"] = error
            if args.pmc_pass:
                line["pmc_pass_calls"] = {"score_topk_call": topk_calls}
            if dist_on:
                line["item_sharded_eval"] = user_sharded
                line["row_sharded_hop"] = sharded_hop
                line["row_sharded_train_step"] = sharded_train
            print(json.dumps(line), flush=True)

    # The legs below are reported beside the headline and run after it.  They contain collectives; should
    # one of them never return, every rank gives up after a while: rank 0 prints the line with what has
    # been measured and a top-level "error", and the process exits non-zero.
    watchdog = None
    if dist_on:
        import threading

        def give_up():
            nonlocal user_sharded, sharded_hop, sharded_train
            note = {"error": "did not finish within the leg's time limit"}
            user_sharded = user_sharded or note
            sharded_hop = sharded_hop or note
            sharded_train = sharded_train or note
            emit(error="a leg of the N > 1 run (item_sharded_eval / row_sharded_hop / row_sharded_train_step) did not "
                       "finish within its time limit; the headline value was measured before it")
            os._exit(3)
        watchdog = threading.Timer(float(os.environ.get("BENCH_LEG_LIMIT_S", "420")), give_up)
        watchdog.daemon = True
        watchdog.start()
    if dist_on:
        try:
            from recommend_amd.dist import all_reduce_tensor
            ev_u = ShardedRankingEvaluator((vu, vi), (tu, ti), K_LIST, METRICS, U, I, rank, world)
            for _ in range(max(args.warmup, 1)):
                res_u = ev_u.evaluate(ue, ie)
            dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                res_u = ev_u.evaluate(ue, ie)
            torch.cuda.synchronize()
            dist.barrier()
            t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
            all_reduce_tensor(t, dist.ReduceOp.MAX)
            user_sharded = {"users_per_s": round(U * args.steps / float(t), 1), "ms_per_step": round(1e3 * float(t) / args.steps, 4),
                            "recall@20": round(float(res_u["recall@20"]), 6),
                            "layout": f"items in {world} shards, per-GPU top-K, all-to-all of the per-user lists, merge, "
                                      "all-reduce of the 18 metric sums"}
        except Exception as e:  # reported, never fatal for the headline line
            user_sharded = {"error": repr(e)}

    # N > 1: the row-sharded forms, outside the timed region; the slowest rank counts.
    # (a) one propagation hop (all-gather of the layer over RCCL + local SpMM on the rank's rows),
    # (b) the whole MMFCN training step with rows and batch sharded (recommend_amd.dist_train).
    if dist_on:
        from recommend_amd.dist import RowShardedGraph, all_reduce_tensor
        try:
            wn = gc.graph.edata[gc.weight].detach().cpu().numpy()
            rg = RowShardedGraph(gc.graph.src, gc.graph.dst, wn, N, rank, world, device=dev)
            h_loc = torch.randn(rg.hi - rg.lo, D, device=dev)
            rg.hop(h_loc)
            dist.barrier()
            t = torch.tensor([event_ms(lambda: rg.hop(h_loc), 20)], dtype=torch.float64, device=dev)
            all_reduce_tensor(t, dist.ReduceOp.MAX)
            sharded_hop = {"ms_per_hop": round(float(t), 5), "rows_per_rank": rg.block,
                           "all_gather_bytes_total": N * D * 4,
                           "note": "all-gather of the layer + SpMM on this rank's rows; max over ranks"}
            # the same hop with the halo exchange: one all-to-all of exactly the rows each rank's edges read
            rg.enable_halo()
            rg.hop(h_loc)
            dist.barrier()
            t = torch.tensor([event_ms(lambda: rg.hop(h_loc), 20)], dtype=torch.float64, device=dev)
            all_reduce_tensor(t, dist.ReduceOp.MAX)
            halo = torch.tensor([float(rg.halo_rows()[0]) if rg.halo_rows() else 0.0], dtype=torch.float64, device=dev)
            all_reduce_tensor(halo, dist.ReduceOp.SUM)
            sharded_hop["halo"] = {"ms_per_hop": round(float(t), 5), "rows_received_total": int(halo),
                                   "bytes_total": int(halo) * D * 4,
                                   "note": "all-to-all of the rows each rank's edges read (dist.HaloPlan) + the same SpMM"}
            del rg
        except Exception as e:  # reported, never fatal for the headline line
            sharded_hop = {"error": repr(e)}
        try:
            from recommend_amd.dist_train import ShardedTESTModel, ShardedTrainStep
            torch.manual_seed(42)
            smodel = ShardedTESTModel(cfg, {}, feats, rank, world).to(dev)
            node_i = ti + U
            mine = ((tu >= smodel.lo) & (tu < smodel.hi)) | ((node_i >= smodel.lo) & (node_i < smodel.hi))
            idx = np.flatnonzero(mine)
            smodel.build_graph((tu[idx], ti[idx], tr[idx]), index=idx, n_total=len(tu))
            smodel.creat_feature_weight()
            smodel.train()
            sopt = FusedClipAdam(smodel.parameters(), lr=LR[args.shape])
            sstep = ShardedTrainStep(smodel, sopt, gather_dot_bpr_loss, max_grad_norm=1.0)
            Bg = BATCH[args.shape] * world          # the per-GPU batch is the preset's
            gen = np.random.default_rng(7)
            def global_batch():
                sel = gen.integers(0, len(tu), Bg)
                return {"user_ids": torch.from_numpy(tu[sel]).to(dev), "item_ids": torch.from_numpy(ti[sel]).to(dev),
                        "neg_items": torch.from_numpy(gen.integers(0, I, (Bg, 1))).to(dev)}
            batches = [global_batch() for _ in range(8)]
            n_timed = 10

            def time_steps(run):
                for bt in batches[:2]:
                    run(bt)
                torch.cuda.synchronize()
                dist.barrier()
                t0 = time.perf_counter()
                for i in range(n_timed):
                    last = run(batches[i % len(batches)])
                torch.cuda.synchronize()
                dist.barrier()
                t = torch.tensor([(time.perf_counter() - t0) / n_timed * 1e3], dtype=torch.float64, device=dev)
                all_reduce_tensor(t, dist.ReduceOp.MAX)
                return round(float(t), 4), round(float(last), 6)

            layout = (f"rows of the graph and of every node matrix over {world} ranks, batch split; "
                      "clip + Adam by the library's kernel pair; ")
            # First the eager form, so that the line holds a measured step even if recording the graph
            # (the first time RCCL collectives of several ranks are captured on this machine) never returns.
            ms, last = time_steps(sstep)
            sharded_train = {"ms_per_train_step": ms, "ms_per_train_step_eager": ms, "global_batch": Bg, "steps_timed": n_timed,
                             "last_loss": last, "captured_graph": False, "capture_error": None,
                             "layout": layout + "eager launches"}
            # Then the step, collectives included, as one captured HIP graph per rank (RCCL only; the gloo
            # rehearsal has no capturable collectives); BENCH_SHARDED_EAGER=1 skips it.
            if backend == "nccl" and os.environ.get("BENCH_SHARDED_EAGER") != "1":
                try:
                    gstep = sstep.graphed(batches[0], warmup_steps=3, capture=True)
                    ms, last = time_steps(gstep)
                    sharded_train.update({"ms_per_train_step": ms, "last_loss": last, "captured_graph": True,
                                          "layout": layout + "one captured HIP graph per rank, RCCL collectives inside"})
                except Exception as e:      # the recording failed (on every rank alike): the eager numbers stand
                    sharded_train["capture_error"] = repr(e)
                    torch.cuda.synchronize()
        except Exception as e:
            sharded_train = {"error": repr(e)}
    if watchdog is not None:
        watchdog.cancel()
    emit()
    if dist_on:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
