"""World-size-2 check of the sharded MMFCN training step on CPU (gloo): graph partition and
normalisation, per-hop gathers, the gather / reduce-scatter of the final embeddings, the batch
split, the bucketed all-reduce of the weight gradients and the clipping must reproduce the
single-process step -- loss, every gradient, the parameters after one Adam step.

Host logic and collectives are the product's; the arithmetic of the kernels is injected (scipy /
torch stand-ins for the HIP ops, test-only) and the single-process step is the oracle's."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
U, I, E, B = 37, 30, 260, 24
# (emb_dim, layer_num): the presets' form, and the large configuration's (emb_dim 128, three layers)
FORMS = [(64, 2), (128, 3)]
FEATS = {"image": 96, "text": 32}
LR = 1e-3   # Adam divides by |g| + 1e-8: gradient elements of that size turn rounding into a fraction of LR


def make_cfg(D, L):
    class Cfg:
        class data:
            num_users, num_items = U, I

        class model:
            model_name = "MMFCN"
            modal_num, emb_dim, layer_num, dropout = 2, D, L, 0.0

        class graph:
            weight_feature = [None]
            edge_weight_type, add_self_loops, normalize_adj = "uniform", True, True

        class system:
            device = "cpu"
    return Cfg


def case():
    rng = np.random.default_rng(5)
    pairs = np.unique(np.stack([rng.integers(0, U, E), rng.integers(0, I, E)], 1), axis=0)
    pairs = pairs[rng.permutation(len(pairs))]
    ratings = rng.integers(1, 6, len(pairs)).astype(np.float32)
    feats = {k: torch.from_numpy(rng.standard_normal((I, f)).astype(np.float32)) for k, f in FEATS.items()}
    sel = rng.choice(len(pairs), B, replace=False)
    batch = {"user_ids": torch.from_numpy(pairs[sel, 0]), "item_ids": torch.from_numpy(pairs[sel, 1]),
             "neg_items": torch.from_numpy(rng.integers(0, I, (B, 1)))}
    return pairs[:, 0], pairs[:, 1], ratings, feats, batch


# ---------------------------------------------------------------- injected compute (test-only)
class ScipyRowBlock:
    def __init__(self, rowptr, col, val, device, n_cols=None):
        self.rowptr, self.col, self.n_cols = rowptr, col, n_cols
        self.set_values(val)

    def set_values(self, val):
        import scipy.sparse as sp
        v = val.numpy() if isinstance(val, torch.Tensor) else np.asarray(val)
        self.m = sp.csr_matrix((v.astype(np.float64), self.col, self.rowptr),
                               shape=(self.rowptr.shape[0] - 1, self.n_cols))

    def remap_columns(self, col, n_cols):
        import scipy.sparse as sp
        self.col, self.n_cols = np.asarray(col), int(n_cols)
        self.m = sp.csr_matrix((self.m.data, self.col, self.rowptr), shape=(self.rowptr.shape[0] - 1, self.n_cols))

    def matmul(self, h_full, h0=None, alpha=0.0, beta=1.0):
        out = beta * (self.m @ h_full.detach().numpy().astype(np.float64)[: self.n_cols])
        if h0 is not None:
            out = out + alpha * h0.detach().numpy().astype(np.float64)
        return torch.from_numpy(out.astype(np.float32))


class NumpyNormalizer:
    @staticmethod
    def in_weight_sqrt(block, w_local):
        s = np.add.reduceat(np.concatenate([w_local.numpy(), [0.0]]).astype(np.float32), block.rowptr[:-1])
        s[np.diff(block.rowptr) == 0] = 0.0
        return torch.from_numpy(np.sqrt(np.maximum(s, np.float32(1e-12))).astype(np.float32))

    @staticmethod
    def sym_scale_rows(block, w_local, row_offset, sqrt_in_w):
        sq = sqrt_in_w.numpy()
        rows = np.repeat(np.arange(block.rowptr.shape[0] - 1), np.diff(block.rowptr)) + row_offset
        return torch.from_numpy((w_local.numpy() / (sq[block.col] * sq[rows])).astype(np.float32))


def patch_layers():
    """torch stand-ins for the layer kernels (dropout is off in this test)."""
    from recommend_amd import fused, ops

    def dense_fwd_raw(x1, w1, b1=None, x2=None, w2=None, b2=None, alpha=None, p=0.0, seed=0, residual=None):
        y = F.linear(x1, w1, b1)
        if x2 is not None:
            y = y + F.linear(x2, w2, b2)
        if alpha is not None:
            y = F.prelu(y, alpha)
        return y if residual is None else y + residual

    def dense_bwd_raw(x, w, b, alpha, p, seed, g, want_dx=True, want_db=True, grads=None, ws=None, defer=None):
        dz, da = g, None
        if alpha is not None:
            z = F.linear(x, w, b)
            da = (g * z * (z <= 0)).sum().reshape(1)
            dz = torch.where(z > 0, g, alpha * g)
        dw, db = dz.t() @ x, dz.sum(0)
        if grads is not None:
            dw, db = grads[0] + dw, grads[1] + db
            da = None if da is None else grads[2] + da
        return (dz @ w if want_dx else None), dw, (db if want_db or grads is not None else None), da

    def linear_act(linear, x, active=None, dropout=None, residual=None):
        return dense_fwd_raw(x, linear.weight, linear.bias, alpha=fused.prelu_slope(active), residual=residual)

    ops.dense_fwd_raw, ops.dense_bwd_raw = dense_fwd_raw, dense_bwd_raw
    ops.dense_bwd_ws = lambda n, K, D, device: None

    class NoQueue:          # the stand-in reduces at once: nothing to defer
        usable = False

        def __init__(self, *a):
            pass

        def flush(self):
            pass
    ops.DenseReduceQueue = NoQueue
    fused.linear_act = linear_act
    ops.modal_sum = lambda feats, alphas: torch.sum(
        torch.stack(feats, 0) * torch.softmax(alphas, 0).unsqueeze(-1).unsqueeze(-1), 0)
    ops.proj = lambda x, w, b, n_users: torch.cat([b.unsqueeze(0).expand(n_users, -1), F.linear(x, w, b)])


def gather_dot_loss(out, batch):
    """train/graph_trainer.py:255-270 in torch (the loss kernel's stand-in)."""
    pu = out["user_embeddings"][batch["user_ids"]]
    pi = out["item_embeddings"][batch["item_ids"]]
    ni = out["item_embeddings"][batch["neg_items"]]
    pos = (pu * pi).sum(-1)
    neg = (pu.unsqueeze(1) * ni).sum(-1)
    return (-F.logsigmoid(pos.unsqueeze(1) - neg)).mean()


def make_reference_state(Cfg):
    """Parameters of a single-process model, as a plain dict (the product's TESTModel builds its
    graph on the GPU only; here only its parameter names and shapes are needed)."""
    torch.manual_seed(11)
    from recommend_amd.dist_train import ShardedTESTModel
    users, items, ratings, feats, _ = case()
    m = ShardedTESTModel(Cfg, {}, feats, rank=0, world=1, block_factory=ScipyRowBlock, normalizer=NumpyNormalizer)
    state = {k: v.detach().clone() for k, v in m.state_dict().items()}
    return state, m.alphas.detach().clone(), m._first_emb.detach().clone()


def _worker(rank, world, port, q, exchange="allgather", form=(64, 2)):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        patch_layers()
        from recommend_amd.dist_train import ShardedTESTModel, ShardedTrainStep
        Cfg = make_cfg(*form)
        users, items, ratings, feats, batch = case()
        state, alphas, first = make_reference_state(Cfg)
        model = ShardedTESTModel(Cfg, {}, feats, block_factory=ScipyRowBlock, normalizer=NumpyNormalizer, exchange=exchange)
        model.load_state_dict(state)
        with torch.no_grad():
            model.alphas.copy_(alphas)
        model._first_emb = first[model.lo:model.hi].clone()
        # every rank hands over only the interactions that touch its rows
        node_u, node_i = users, items + U
        mine = ((node_u >= model.lo) & (node_u < model.hi)) | ((node_i >= model.lo) & (node_i < model.hi))
        idx = np.flatnonzero(mine)
        model.build_graph((users[idx], items[idx], ratings[idx]), index=idx, n_total=len(users))
        model.creat_feature_weight()
        assert model.graph_constructor.graph.exchange == exchange
        model.train()
        opt = torch.optim.Adam(model.parameters(), lr=LR)
        step = ShardedTrainStep(model, opt, gather_dot_loss, max_grad_norm=1.0)
        loss = step(batch)
        # (numpy: tensors in a queue travel as shared-memory handles that die with the worker)
        grads = {k: (None if p.grad is None else p.grad.detach().numpy().copy()) for k, p in model.named_parameters()}
        params = {k: p.detach().numpy().copy() for k, p in model.named_parameters()}
        q.put((rank, float(loss), float(step.last_grad_norm), grads, params,
               model.graph_constructor.sqrt_in_w.numpy().copy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("form", FORMS, ids=lambda f: f"D{f[0]}L{f[1]}")
@pytest.mark.parametrize("exchange", ["allgather", "halo"])
def test_sharded_training_step_world2_equals_single_process(exchange, form):
    from oracle import graph_oracle, loss_oracle, mmfcn_oracle
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, exchange, form)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0

    # the single-process step: the oracle's forward and loss, torch's clip and Adam
    users, items, ratings, feats, batch = case()
    state, alphas, first = make_reference_state(make_cfg(*form))
    N = U + I
    src, dst, w = graph_oracle.build_coo_vectorized(users, items, ratings, U, I, True)
    wn, sq = graph_oracle.normalize(src, dst, w, N)
    A = mmfcn_oracle.adjacency(src, dst, wn, N)
    p = {k: v.clone().requires_grad_(v.dtype.is_floating_point) for k, v in state.items()}
    padded = [torch.cat([torch.zeros(U, f.shape[1]), f]) for f in feats.values()]
    embs = first.unsqueeze(0).expand(2, -1, -1)
    ue, ie = mmfcn_oracle.forward(p, A, padded, ["linears.0.linear", "linears.1.linear"], embs, [5, 5], alphas, form[1], U)
    want = loss_oracle.trainer_loss({"user_embeddings": ue, "item_embeddings": ie}, batch)
    want.backward()
    trainable = [v for v in p.values() if v.requires_grad]
    with_grad = [v for v in trainable if v.grad is not None]
    norm = torch.nn.utils.clip_grad_norm_(with_grad, 1.0)
    opt = torch.optim.Adam(trainable, lr=LR)
    opt.step()

    for rank, loss, gnorm, grads, params, sq_got in results:
        assert abs(loss - float(want)) <= 1e-5 * abs(float(want)), (rank, loss, float(want))
        assert abs(gnorm - float(norm)) <= 1e-4 * float(norm)
        np.testing.assert_allclose(sq_got, sq, rtol=1e-6)
        checked = 0
        for name, ref in p.items():
            if not ref.requires_grad:
                continue
            if ref.grad is None:
                assert grads[name] is None, name
                continue
            scale = float(ref.grad.abs().max()) + 1e-30
            assert float(np.abs(grads[name] - ref.grad.numpy()).max()) <= 2e-5 * scale, (rank, name)
            assert float(np.abs(params[name] - ref.detach().numpy()).max()) <= 1e-5 * (float(ref.abs().max()) + 1e-30) + 2e-2 * LR, (rank, name)
            checked += 1
        assert checked >= 13 + 7 * form[1]
    # replicated parameters stay equal across ranks
    for name in results[0][4]:
        assert np.array_equal(results[0][4][name], results[1][4][name]), name
