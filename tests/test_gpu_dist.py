"""The sharded paths on the HIP kernels: two ranks with a gloo rendezvous (both on cuda:0, the
collectives staged through host memory by a transport the test injects), and one rank on the RCCL
backend ("nccl"; a one-GPU box holds no second rank) so that the product's own transport -- device
tensors handed to the collectives as they are -- runs under test, captured training step included.
Forward and gradient must equal the float64 result of the unsharded recurrence; the sharded
training step and the sharded evaluators must equal their single-process counterparts."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _case(D=64):
    rng = np.random.default_rng(9)
    N, E = 3001, 40000
    src = rng.integers(0, N, E)
    dst = rng.integers(0, N, E)
    dst[:5000] = 7                       # one row long enough for the chunked-row path
    w = (rng.random(E) / 8).astype(np.float32)
    x = rng.standard_normal((N, D)).astype(np.float32)
    g = rng.standard_normal((N, D)).astype(np.float32)
    return N, src, dst, w, x, g


BACKENDS = [("gloo", 2), ("nccl", 1)]


def _init(backend, rank, world, port):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if backend == "nccl":
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda:0"))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
        # two ranks on one GPU: RCCL refuses that, gloo moves host memory
        from recommend_amd import dist as rk_dist
        from tools.host_staged_transport import HostStagedTransport
        rk_dist.set_transport(HostStagedTransport())


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _run(target, world, *args, timeout=300):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=target, args=(r, world, port, q) + args) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=timeout) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return results


def _worker(rank, world, port, q, exchange="allgather", backend="gloo", D=64):
    _init(backend, rank, world, port)
    try:
        from recommend_amd.dist import RowShardedGraph
        N, src, dst, w, x, g = _case(D)
        graph = RowShardedGraph(src, dst, w, N, device="cuda:0", exchange=exchange)
        xl = torch.from_numpy(x[graph.lo:graph.hi]).to("cuda:0").requires_grad_(True)
        out = graph.aggrate(xl, 2, 0.1)
        out.backward(torch.from_numpy(g[graph.lo:graph.hi]).to("cuda:0"))
        torch.cuda.synchronize()
        q.put((rank, graph.lo, graph.hi, out.detach().cpu().numpy(), xl.grad.cpu().numpy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("D", [64, 128])
@pytest.mark.parametrize("backend,world", BACKENDS)
@pytest.mark.parametrize("exchange", ["allgather", "halo"])
def test_row_sharded_propagation_on_the_hip_kernels(exchange, backend, world, D):
    import scipy.sparse as sp
    results = _run(_worker, world, exchange, backend, D, timeout=240)
    N, src, dst, w, x, g = _case(D)
    A = sp.csr_matrix((w.astype(np.float64), (dst, src)), shape=(N, N))
    h0 = x.astype(np.float64)
    h = h0
    for _ in range(2):
        h = 0.1 * h0 + 0.9 * (A @ h)
    gh = g.astype(np.float64)
    gx = np.zeros_like(gh)
    for _ in range(2):
        gx += 0.1 * gh
        gh = 0.9 * (A.T @ gh)
    gx += gh
    assert sorted((lo, hi) for _r, lo, hi, _o, _g in results) == ([(0, 1501), (1501, 3001)] if world == 2 else [(0, 3001)])
    for _rank, lo, hi, out, grad in results:
        assert np.abs(out - h[lo:hi]).max() <= 1e-5 * np.abs(h).max()
        assert np.abs(grad - gx[lo:hi]).max() <= 1e-5 * np.abs(gx).max()


# ---------------------------------------------------------------- sharded training step
N_USERS, N_ITEMS = 1203, 811
# (emb_dim, layer_num): the presets' form and the large configuration's (emb_dim 128, three layers)
FORMS = [(64, 2), (128, 3)]


def _cfg(form):
    class _Cfg:
        class data:
            num_users, num_items = N_USERS, N_ITEMS

        class model:
            model_name = "MMFCN"
            modal_num, emb_dim, layer_num, dropout = 2, form[0], form[1], 0.0

        class graph:
            weight_feature = [None]
            edge_weight_type, add_self_loops, normalize_adj = "uniform", True, True

        class system:
            device = "cuda:0"
    return _Cfg


def _train_case():
    rng = np.random.default_rng(21)
    U, I = N_USERS, N_ITEMS
    pairs = np.unique(np.stack([rng.integers(0, U, 9000), rng.integers(0, I, 9000)], 1), axis=0)
    pairs[:700, 1] = 3                                    # one item row long enough for the chunked-row path
    pairs = np.unique(pairs, axis=0)
    pairs = pairs[rng.permutation(len(pairs))]
    ratings = rng.integers(1, 6, len(pairs)).astype(np.float32)
    feats = {"image": torch.from_numpy(rng.standard_normal((I, 256)).astype(np.float32)),
             "text": torch.from_numpy(rng.standard_normal((I, 96)).astype(np.float32))}
    sel = rng.choice(len(pairs), 257, replace=False)      # odd batch: the ranks' shares differ
    batch = {"user_ids": torch.from_numpy(pairs[sel, 0]), "item_ids": torch.from_numpy(pairs[sel, 1]),
             "neg_items": torch.from_numpy(rng.integers(0, I, (257, 1)))}
    return pairs[:, 0], pairs[:, 1], ratings, feats, batch


def _single_process_step(form=(64, 2)):
    from recommend_amd.loss import gather_dot_bpr_loss
    from recommend_amd.model import ModelFactory
    users, items, ratings, feats, batch = _train_case()
    torch.manual_seed(4)
    model = ModelFactory.create_Model(_cfg(form), {}, feats).to("cuda:0")
    model.build_graph((users, items, ratings))
    model.creat_feature_weight()
    model.train()
    return model, {k: v.to("cuda:0") for k, v in batch.items()}, gather_dot_bpr_loss


def _sharded_model(form, exchange, full):
    from recommend_amd.dist_train import ShardedTESTModel
    users, items, ratings, feats, _ = _train_case()
    model = ShardedTESTModel(_cfg(form), {}, feats, exchange=exchange).to("cuda:0")
    model.load_replicated_state(full)
    U = N_USERS
    mine = (((users >= model.lo) & (users < model.hi)) | ((items + U >= model.lo) & (items + U < model.hi)))
    idx = np.flatnonzero(mine)
    model.build_graph((users[idx], items[idx], ratings[idx]), index=idx, n_total=len(users))
    model.creat_feature_weight()
    model.train()
    return model


def _train_worker(rank, world, port, q, exchange="allgather", backend="gloo", form=(64, 2)):
    _init(backend, rank, world, port)
    try:
        from recommend_amd.dist_train import ShardedTrainStep
        from recommend_amd.optim import FusedClipAdam
        full, batch, loss_fn = _single_process_step(form)      # only its parameters are used here
        model = _sharded_model(form, exchange, full)
        opt = FusedClipAdam(model.parameters(), lr=1e-3)         # the product's optimizer: clips inside its update
        step = ShardedTrainStep(model, opt, loss_fn, max_grad_norm=1.0)
        loss = step(batch)
        torch.cuda.synchronize()
        grads = {k: (None if p.grad is None else p.grad.detach().cpu().numpy()) for k, p in model.named_parameters()}
        params = {k: p.detach().cpu().numpy() for k, p in model.named_parameters()}
        q.put((rank, float(loss), float(step.last_grad_norm), grads, params))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("form", FORMS, ids=lambda f: f"D{f[0]}L{f[1]}")
@pytest.mark.parametrize("backend,world", BACKENDS)
@pytest.mark.parametrize("exchange", ["allgather", "halo"])
def test_sharded_training_step_on_the_hip_kernels_equals_single_process(exchange, backend, world, form):
    """The ranks (rows of the graph and of every node matrix sharded, batch split, the product's
    clip + Adam kernels) against the single-process step on the same layer kernels with torch's
    clip_grad_norm_ and Adam: loss, clipping norm, every gradient, the parameters after one step.
    Dropout is off (the ranks' masks are drawn per local row)."""
    results = _run(_train_worker, world, exchange, backend, form)
    model, batch, loss_fn = _single_process_step(form)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    loss = loss_fn(model(batch), batch)
    loss.backward()
    with_grad = [p for p in model.parameters() if p.grad is not None]
    want_grads = {k: (None if p.grad is None else p.grad.detach().cpu().numpy().copy())
                  for k, p in model.named_parameters()}
    norm = float(torch.nn.utils.clip_grad_norm_(with_grad, 1.0))
    opt.step()
    want = float(loss)
    for rank, got_loss, gnorm, grads, params in results:
        assert abs(got_loss - want) <= 1e-5 * abs(want), (rank, got_loss, want)
        assert abs(gnorm - norm) <= 1e-4 * norm
        checked = 0
        for name, p in model.named_parameters():
            ref = want_grads[name]
            if ref is None:
                assert grads[name] is None, name
                continue
            # (the fused optimizer applies the clip factor inside its update: the gradients stay unscaled)
            tol = 2e-5 * float(np.abs(ref).max()) + 1e-12
            assert float(np.abs(grads[name] - ref).max()) <= tol, (rank, name)
            new = p.detach().cpu().numpy()
            # (Adam divides by |g| + 1e-8: gradient elements of that size turn rounding into a few percent of lr)
            assert float(np.abs(params[name] - new).max()) <= 1e-5 * float(np.abs(new).max()) + 2e-5, (rank, name)
            checked += 1
        assert checked >= 13 + 7 * form[1]
    for other in results[1:]:
        for name in results[0][4]:
            assert np.array_equal(results[0][4][name], other[4][name]), name


def _captured_worker(rank, world, port, q, exchange, form):
    _init("nccl", rank, world, port)
    try:
        from recommend_amd.dist_train import ShardedTrainStep
        from recommend_amd.optim import FusedClipAdam
        full, batch, loss_fn = _single_process_step(form)
        out = {}
        for mode in ("eager", "captured"):
            model = _sharded_model(form, exchange, full)
            opt = FusedClipAdam(model.parameters(), lr=1e-3)
            step = ShardedTrainStep(model, opt, loss_fn, max_grad_norm=1.0)
            if mode == "eager":
                for _ in range(4):
                    loss = step(batch)
            else:
                gs = step.graphed(batch, warmup_steps=2)        # two eager steps, then the recording
                assert gs.graph is not None
                for _ in range(2):
                    loss = gs(batch)                            # replays
            torch.cuda.synchronize()
            out[mode] = (float(loss), {k: p.detach().cpu().numpy() for k, p in model.named_parameters()},
                         int(opt.step_count))
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("form", FORMS, ids=lambda f: f"D{f[0]}L{f[1]}")
@pytest.mark.parametrize("exchange", ["allgather", "halo"])
def test_captured_sharded_step_on_rccl_equals_eager_steps(exchange, form):
    """ShardedTrainStep.graphed on the "nccl" backend: the whole step -- layer kernels, hops, every
    collective (RCCL), flat gradient all-reduce, clip + Adam -- recorded as one HIP graph.  Two
    warm-up steps + two replays must leave exactly the parameters of four eager steps (the same
    kernels in the same order; dropout is off)."""
    (_rank, out), = _run(_captured_worker, 1, exchange, form)
    (l_e, p_e, n_e), (l_c, p_c, n_c) = out["eager"], out["captured"]
    assert n_e == n_c == 4
    assert l_e == l_c
    for name in p_e:
        assert np.array_equal(p_e[name], p_c[name]), name


# ---------------------------------------------------------------- sharded evaluation
METRICS = ["precision", "recall", "hit_rate", "ndcg", "map", "mrr"]


def _eval_case():
    rng = np.random.default_rng(5)
    U, I, D = 2500, 3001, 64
    ue = (rng.standard_normal((U, D)) * 0.1).astype(np.float32)
    ie = (rng.standard_normal((I, D)) * 0.1).astype(np.float32)
    tr = np.unique(np.stack([rng.integers(0, U, 20000), rng.integers(0, I, 20000)], 1), axis=0)
    va = np.unique(np.stack([rng.integers(0, U, 5000), rng.integers(0, I, 5000)], 1), axis=0)
    return U, I, ue, ie, tr, va


def _eval_worker(rank, world, port, q, backend="gloo"):
    _init(backend, rank, world, port)
    try:
        from recommend_amd.dist import ShardedRankingEvaluator, UserShardedRankingEvaluator
        U, I, ue, ie, tr, va = _eval_case()
        ue_d, ie_d = torch.from_numpy(ue).to("cuda:0"), torch.from_numpy(ie).to("cuda:0")
        out = {}
        for name, cls in (("items", ShardedRankingEvaluator), ("users", UserShardedRankingEvaluator)):
            ev = cls((va[:, 0], va[:, 1]), (tr[:, 0], tr[:, 1]), [5, 10, 20], METRICS, U, I)
            res = ev.evaluate(ue_d, ie_d)
            assert all(v.is_cuda for v in res.values())      # the metrics stay on the device
            out[name] = ({k: float(v) for k, v in res.items()}, ev.u_lo, ev.u_hi, ev.last_ids.cpu().numpy())
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("backend,world", BACKENDS)
def test_sharded_evaluators_on_the_hip_kernels_equal_the_unsharded_one(backend, world):
    from recommend_amd.evaluator import RankingEvaluator
    results = _run(_eval_worker, world, backend)
    U, I, ue, ie, tr, va = _eval_case()
    ev = RankingEvaluator((va[:, 0], va[:, 1]), (tr[:, 0], tr[:, 1]), [5, 10, 20], METRICS, n_users=U)
    ue_d, ie_d = torch.from_numpy(ue).to("cuda:0"), torch.from_numpy(ie).to("cuda:0")
    want = {k: float(v) for k, v in ev.evaluate(ue_d, ie_d).items()}
    _s, want_ids = ev.topk(ue_d, ie_d, 20)
    want_ids = want_ids.cpu().numpy()
    for _rank, out in results:
        for form in ("items", "users"):
            got, u_lo, u_hi, ids = out[form]
            np.testing.assert_array_equal(ids, want_ids[u_lo:u_hi])
            for k, v in want.items():
                assert abs(got[k] - v) <= 1e-6, (form, k, got[k], v)
