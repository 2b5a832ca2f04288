"""GPU parity: the MMFCN wrapper end to end (projection -> k-hop propagation ->
combine -> fusion -> final projection) against the torch-CPU oracle, plus a short
training run through the reference's protocol (model(batch) -> loss -> backward)."""
import numpy as np
import pytest
import torch

from oracle import graph_oracle, mmfcn_oracle
from recommend_amd import synth
from recommend_amd.loss import gather_dot_bpr_loss
from recommend_amd.model import ModelFactory
from recommend_amd.sampler import NegativeSampler, TrainBatchLoader, build_user_csr

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def make_config(U, I, D, dropout):
    class C:
        class data:
            num_users, num_items = U, I
        class model:
            model_name = "MMFCN"
            modal_num, emb_dim, layer_num = 2, D, 2
        class graph:
            weight_feature = [None]
            edge_weight_type, add_self_loops, normalize_adj = "uniform", True, True
        class system:
            device = DEV
    C.model.dropout = dropout
    return C


def build_model(U, I, D, E, dropout, seed=0):
    d = synth.make_interactions(U, I, E, seed)
    rng = np.random.default_rng(seed)
    feats = {"image": torch.from_numpy(rng.standard_normal((I, 256)).astype(np.float32)),
             "text": torch.from_numpy(rng.standard_normal((I, 96)).astype(np.float32))}
    torch.manual_seed(seed)
    model = ModelFactory.create_Model(make_config(U, I, D, dropout), {}, feats).to(DEV)
    tu, ti, tr = d.split(0)
    inter = [(int(u), int(i), float(r)) for u, i, r in zip(tu, ti, tr)]
    g = model.build_graph(inter)
    model.creat_feature_weight()
    assert g.num_nodes() == U + I and g.num_edges() == 2 * len(inter) + U + I
    return model, d, feats, (tu, ti, tr)


def test_forward_vs_oracle():
    U, I, D = 300, 200, 64
    model, d, feats, (tu, ti, tr) = build_model(U, I, D, 3000, 0.0)
    with pytest.raises(ValueError, match="Graph not built"):
        ModelFactory.create_Model(make_config(U, I, D, 0.0), {}, feats)({})
    model.eval()
    out = model({})
    assert out["user_embeddings"].shape == (U, D) and out["item_embeddings"].shape == (I, D)
    src, dst, w = graph_oracle.build_coo_vectorized(tu, ti, tr, U, I, True)
    wn, _ = graph_oracle.normalize(src, dst, w, U + I)
    A = mmfcn_oracle.adjacency(src, dst, wn, U + I)
    p = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    padded = [torch.cat([torch.zeros(U, f.shape[1]), f]) for f in feats.values()]
    ue, ie = mmfcn_oracle.forward(p, A, padded, ["linears.0.linear", "linears.1.linear"],
                                  model.embs.detach().cpu(), model.ks, model.alphas.detach().cpu(), 2, U)
    for got, want in ((out["user_embeddings"], ue), (out["item_embeddings"], ie)):
        err = (got.detach().cpu() - want).abs().max() / want.abs().max()
        assert float(err) <= 1e-5, float(err)
    info = model.get_model_info()
    assert info["num_nodes"] == U + I and info["embedding_dim"] == D


def test_training_steps_reduce_loss():
    U, I, D = 400, 250, 64
    model, d, feats, (tu, ti, tr) = build_model(U, I, D, 5000, 0.1, seed=3)
    p, it = build_user_csr(d.users, d.items, U)
    sampler = NegativeSampler(p, it, I, seed=42)
    dp, di = sampler.device_csr(DEV)
    loader = TrainBatchLoader(tu, ti, tr, 256, sampler, device=DEV, resolve_on_host=False)
    opt = torch.optim.Adam(model.parameters(), lr=0.01)
    model.train()
    losses = []
    for epoch in range(3):
        for batch in loader:
            batch["pos_ptr"], batch["pos_items"] = dp, di
            opt.zero_grad()
            loss = gather_dot_bpr_loss(model(batch), batch)
            loss.backward()
            torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
            opt.step()
            losses.append(float(loss))
    assert all(np.isfinite(losses))
    n = len(losses) // 3
    assert np.mean(losses[-n:]) < np.mean(losses[:n])
    grads = [q.grad for q in model.parameters() if q.grad is not None]
    assert len(grads) > 10


def test_graphed_train_step_matches_eager_steps():
    """The captured step replays the same arithmetic as the eager step (dropout off), and with
    dropout on every replay draws a different mask."""
    from recommend_amd.train_step import GraphedTrainStep
    from recommend_amd.optim import FusedClipAdam
    from recommend_amd import ops
    U, I, D = 300, 200, 64

    def setup(dropout, fused=False):
        model, d, feats, (tu, ti, tr) = build_model(U, I, D, 4000, dropout, seed=5)
        p, it = build_user_csr(d.users, d.items, U)
        sampler = NegativeSampler(p, it, I, seed=7)
        dp, di = sampler.device_csr(DEV)
        loader = TrainBatchLoader(tu, ti, tr, 256, sampler, shuffle=False, device=DEV, resolve_on_host=False)
        batches = [b for b in loader if b["user_ids"].shape[0] == 256][:6]
        # fused: the library's own clip + Adam kernel pair; otherwise torch's optimizer (the reference's)
        opt = (FusedClipAdam(model.parameters(), lr=0.01) if fused
               else torch.optim.Adam(model.parameters(), lr=0.01, capturable=True))
        model.train()
        return model, opt, batches, {"pos_ptr": dp, "pos_items": di}

    try:
        m1, o1, batches, consts = setup(0.0)
        eager = []
        for b in [batches[0]] + batches:     # the stepper's warm-up step trains on the first batch
            b = {**b, **consts}
            o1.zero_grad(set_to_none=True)
            loss = gather_dot_bpr_loss(m1(b), b)
            loss.backward()
            torch.nn.utils.clip_grad_norm_(m1.parameters(), 1.0)
            o1.step()
            eager.append(float(loss))
        m2, o2, batches2, consts2 = setup(0.0, fused=True)    # clipping inside the optimizer's own update kernels
        st = GraphedTrainStep(m2, o2, gather_dot_bpr_loss, batches2[0], constants=consts2, warmup_steps=1)
        graphed = [float(st(b)) for b in batches2]
        np.testing.assert_allclose(graphed, eager[1:], rtol=2e-4)
        for (n1, p1), (n2, p2) in zip(m1.named_parameters(), m2.named_parameters()):
            # (Adam divides by sqrt(v): entries with near-zero gradients amplify the float-atomic
            # ordering noise of the loss backward, hence the absolute tolerance of a fraction of lr)
            torch.testing.assert_close(p2, p1, rtol=0, atol=5e-3, msg=n1)
        # dropout on: the same batch replayed twice gives different losses (fresh masks), and
        # the counter advanced once per step
        m3, o3, batches3, consts3 = setup(0.5)
        for g in o3.param_groups:
            g["lr"] = 0.0
        st3 = GraphedTrainStep(m3, o3, gather_dot_bpr_loss, batches3[0], constants=consts3, warmup_steps=1)
        c0 = int(st3.step_counter)
        a, b_ = float(st3(batches3[0])), float(st3(batches3[0]))
        assert a != b_
        assert int(st3.step_counter) == c0 + 2
    finally:
        ops.set_dropout_step_tensor(None)


def test_training_is_bit_reproducible_run_to_run():
    """No kernel of the training step adds floats in an order that depends on timing (hop partial
    rows, layer and projection reductions, loss and its backward, clip norm): two runs from the same
    seeds -- dropout on -- end with identical parameter bits."""
    import itertools
    from recommend_amd import ops
    from recommend_amd.optim import FusedClipAdam
    U, I, D = 400, 250, 64

    def run():
        ops._seed_counter = itertools.count(1)
        model, d, feats, (tu, ti, tr) = build_model(U, I, D, 6000, 0.3, seed=11)
        # one very active user and one very popular item: rows cut into chunks, gradient rows with many terms
        p, it = build_user_csr(d.users, d.items, U)
        sampler = NegativeSampler(p, it, I, seed=3)
        dp, di = sampler.device_csr(DEV)
        loader = TrainBatchLoader(tu, ti, tr, 512, sampler, shuffle=False, device=DEV, resolve_on_host=False)
        opt = FusedClipAdam(model.parameters(), lr=0.01, max_grad_norm=1.0)
        model.train()
        losses = []
        for batch in itertools.islice(loader, 5):
            batch["pos_ptr"], batch["pos_items"] = dp, di
            opt.zero_grad()
            loss = gather_dot_bpr_loss(model(batch), batch)
            loss.backward()
            opt.step()
            losses.append(float(loss))
        return losses, [q.detach().clone() for q in model.parameters()]

    l1, p1 = run()
    l2, p2 = run()
    assert l1 == l2
    for a, b in zip(p1, p2):
        assert torch.equal(a, b)
