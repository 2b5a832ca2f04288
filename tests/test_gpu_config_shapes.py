"""GPU parity at the other benchmark configurations: the Clothing shape ranked with the item table
cut into 1 / 2 / 4 / 8 shards (per-shard top-K with global ids, then the merge) against the
unsharded ranking and the oracle, and the Sports and Elec shapes through the whole MMFCN forward (4096-d
image + 384-d text features) against the torch-CPU oracle at 1e-5."""
import numpy as np
import pytest
import torch

from oracle import eval_oracle, graph_oracle, mmfcn_oracle
from recommend_amd import ops, synth
from recommend_amd.dist import item_range
from recommend_amd.evaluator import pairs_to_csr, restrict_csr
from recommend_amd.model import ModelFactory

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("D", [64, 128])        # 128 is the Clothing preset's emb_dim (config.py:201)
@pytest.mark.parametrize("mode", ["auto", "filter"])
def test_clothing_shape_item_sharded_topk(mode, D):
    d = synth.make_named("clothing")
    rng = np.random.default_rng(2)
    # half-integer grid: plenty of exact score ties, so "lower id wins" is exercised across shards
    ue = (np.round(rng.standard_normal((d.n_users, D)) * 2) / 8).astype(np.float32)
    ie = (np.round(rng.standard_normal((d.n_items, D)) * 2) / 8).astype(np.float32)
    tu, ti, _ = d.split(0)
    ptr, cols = pairs_to_csr(tu, ti, d.n_users)
    ued, ied = torch.from_numpy(ue).to(DEV), torch.from_numpy(ie).to(DEV)
    for K in (5, 10, 20):
        want_s, want_i = ops.score_topk(ued, ied, K, torch.from_numpy(ptr).to(DEV), torch.from_numpy(cols).to(DEV),
                                        mode="exact")
        if K == 20:
            us = np.sort(rng.choice(d.n_users, 256, replace=False))
            sub_ptr = np.concatenate([[0], np.cumsum(ptr[us + 1] - ptr[us])]).astype(np.int32)
            sub_cols = np.concatenate([cols[ptr[u]:ptr[u + 1]] for u in us]).astype(np.int32)
            o_s, o_i = eval_oracle.score_topk(ue[us], ie, K, sub_ptr, sub_cols)
            np.testing.assert_array_equal(want_i.cpu().numpy()[us], o_i)
            np.testing.assert_array_equal(want_s.cpu().numpy()[us], o_s)
        for world in (1, 2, 4, 8):
            ps, pi = [], []
            for rank in range(world):
                lo, hi = item_range(d.n_items, rank, world)
                p2, c2 = restrict_csr(ptr, cols, lo, hi)
                s, i = ops.score_topk(ued, ied[lo:hi], K, torch.from_numpy(p2).to(DEV), torch.from_numpy(c2).to(DEV),
                                      item_id_offset=lo, mode=mode)
                ps.append(s), pi.append(i)
            ms, mi = ops.topk_merge(torch.stack(ps, 1), torch.stack(pi, 1), K)
            assert torch.equal(mi, want_i), (K, world)
            assert torch.equal(ms, want_s), (K, world)


# (shape, emb_dim of its preset: config.py:201,212,223)
@pytest.mark.parametrize("shape,D", [("sports", 64), ("clothing", 128), ("elec", 64)])
def test_model_forward_vs_oracle_at_shape(shape, D):
    d = synth.make_named(shape)
    U, I = d.n_users, d.n_items
    image, text = synth.make_features(I)[:2]
    feats = {"image": torch.from_numpy(image), "text": torch.from_numpy(text)}

    class C:
        class data:
            num_users, num_items = U, I
        class model:
            model_name = "MMFCN"
            modal_num, emb_dim, layer_num, dropout = 2, D, 2, 0.0
        class graph:
            weight_feature = [None]
            edge_weight_type, add_self_loops, normalize_adj = "uniform", True, True
        class system:
            device = DEV

    torch.manual_seed(5)
    model = ModelFactory.create_Model(C, {}, feats).to(DEV)
    tu, ti, tr = d.split(0)
    model.build_graph((tu, ti, tr))
    model.creat_feature_weight()
    model.eval()
    with torch.no_grad():
        out = model({})
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


def test_elec_training_step_gradients_vs_oracle():
    """One training step at the Elec shape, dropout off: loss and parameter gradients of the HIP
    path (projection, 10 hops, row-wise layers, fused sum, BPR) against the oracle's torch-CPU
    step on the same parameters and batch.  Both sides accumulate in fp32 over 255k rows and a
    20-layer chain: loss to 1e-5, gradients to 2e-4 of their largest entry."""
    from oracle import loss_oracle
    from recommend_amd.loss import gather_dot_bpr_loss
    d = synth.make_named("elec")
    U, I, D = d.n_users, d.n_items, 64
    image, text = synth.make_features(I)[:2]
    feats = {"image": torch.from_numpy(image), "text": torch.from_numpy(text)}

    class C:
        class data:
            num_users, num_items = U, I
        class model:
            model_name = "MMFCN"
            modal_num, emb_dim, layer_num, dropout = 2, D, 2, 0.0
        class graph:
            weight_feature = [None]
            edge_weight_type, add_self_loops, normalize_adj = "uniform", True, True
        class system:
            device = DEV

    torch.manual_seed(6)
    model = ModelFactory.create_Model(C, {}, feats).to(DEV)
    tu, ti, tr = d.split(0)
    model.build_graph((tu, ti, tr))
    model.creat_feature_weight()
    model.train()
    # the table the network reads is frozen in training runs (no optimizer holds it); its gradient
    # is asked for here so that the dX path of the embedding branch is checked as well
    model._first_emb.requires_grad_(True)
    rng = np.random.default_rng(1)
    sel = rng.choice(len(tu), 256, replace=False)
    batch = {"user_ids": torch.from_numpy(tu[sel]).to(DEV), "item_ids": torch.from_numpy(ti[sel]).to(DEV),
             "neg_items": torch.from_numpy(rng.integers(0, I, (256, 1))).to(DEV)}
    loss = gather_dot_bpr_loss(model(batch), batch)
    loss.backward()

    src, dst, w = graph_oracle.build_coo_vectorized(tu, ti, tr, U, I, True)
    wn, _ = graph_oracle.normalize(src, dst, w, U + I)
    A = mmfcn_oracle.adjacency(src, dst, wn, U + I)
    p = {k: v.detach().cpu().clone().requires_grad_(v.dtype.is_floating_point) for k, v in model.state_dict().items()}
    padded = [torch.cat([torch.zeros(U, f.shape[1]), f]) for f in feats.values()]
    embs = model.embs.detach().cpu().clone().requires_grad_(True)
    alphas = model.alphas.detach().cpu().clone().requires_grad_(True)
    ue, ie = mmfcn_oracle.forward(p, A, padded, ["linears.0.linear", "linears.1.linear"], embs, model.ks, alphas, 2, U)
    want = loss_oracle.trainer_loss({"user_embeddings": ue, "item_embeddings": ie},
                                    {k: v.cpu() for k, v in batch.items()})
    want.backward()
    assert abs(float(loss) - float(want)) <= 1e-5 * abs(float(want))
    got = dict(model.named_parameters())
    checked = 0
    for name, ref in p.items():
        if ref.grad is None or name not in got or got[name].grad is None:
            continue
        g, r = got[name].grad.detach().cpu().double(), ref.grad.double()
        scale = float(r.abs().max())
        if scale == 0.0:
            assert float(g.abs().max()) == 0.0, name
            continue
        assert float((g - r).abs().max()) <= 2e-4 * scale, (name, float((g - r).abs().max()) / scale)
        checked += 1
    assert checked >= 20
    ge = model._first_emb.grad.detach().cpu().double()
    re = embs.grad.double().sum(0)          # every modality reads the same table
    assert float((ge - re).abs().max()) <= 2e-4 * float(re.abs().max())
    ga, ra = model.alphas.grad.detach().cpu().double(), alphas.grad.double()
    assert float((ga - ra).abs().max()) <= 2e-4 * float(ra.abs().max())


def _cfg(U, I, D, layer_num):
    class C:
        class data:
            num_users, num_items = U, I
        class model:
            model_name = "MMFCN"
            modal_num, emb_dim, dropout = 2, D, 0.0
        class graph:
            weight_feature = [None]
            edge_weight_type, add_self_loops, normalize_adj = "uniform", True, True
        class system:
            device = DEV
    C.model.layer_num = layer_num
    return C


def test_scaled_down_large_configuration_d128_three_layers():
    """The large synthetic configuration (emb_dim 128, layer_num 3) at a size the oracle finishes in
    seconds: 100k users x 20k items x 3M interactions.  Whole forward (projection, 10 hops at
    D = 128, the 128-wide node-wise layers, three Combine layers per modality), loss and parameter
    gradients against the torch-CPU oracle; full-ranking top-20 of the resulting embeddings:
    candidate filter = exact kernel bit for bit on all users, = oracle on a sample."""
    from oracle import loss_oracle
    from recommend_amd.loss import gather_dot_bpr_loss
    U, I, D, L = 100_000, 20_000, 128, 3
    d = synth.make_interactions(U, I, 3_000_000, seed=77)
    rng = np.random.default_rng(8)
    feats = {"image": torch.from_numpy(rng.standard_normal((I, 512), dtype=np.float32)),
             "text": torch.from_numpy(rng.standard_normal((I, 128), dtype=np.float32))}   # 128 = emb_dim: Identity branch
    torch.manual_seed(9)
    model = ModelFactory.create_Model(_cfg(U, I, D, L), {}, feats).to(DEV)
    tu, ti, tr = d.split(0)
    model.build_graph((tu, ti, tr))
    model.creat_feature_weight()
    model.train()
    sel = rng.choice(len(tu), 1024, replace=False)
    batch = {"user_ids": torch.from_numpy(tu[sel]).to(DEV), "item_ids": torch.from_numpy(ti[sel]).to(DEV),
             "neg_items": torch.from_numpy(rng.integers(0, I, (1024, 1))).to(DEV)}
    out = model(batch)
    loss = gather_dot_bpr_loss(out, batch)
    loss.backward()

    src, dst, w = graph_oracle.build_coo_vectorized(tu, ti, tr, U, I, True)
    wn, _ = graph_oracle.normalize(src, dst, w, U + I)
    A = mmfcn_oracle.adjacency(src, dst, wn, U + I)
    p = {k: v.detach().cpu().clone().requires_grad_(v.dtype.is_floating_point) for k, v in model.state_dict().items()}
    padded = [torch.cat([torch.zeros(U, f.shape[1]), f]) for f in feats.values()]
    ue, ie = mmfcn_oracle.forward(p, A, padded, ["linears.0.linear", None], model.embs.detach().cpu(), model.ks,
                                  model.alphas.detach().cpu(), L, U)
    for got, want in ((out["user_embeddings"], ue), (out["item_embeddings"], ie)):
        err = (got.detach().cpu() - want.detach()).abs().max() / want.detach().abs().max()
        assert float(err) <= 1e-5, float(err)
    want_loss = loss_oracle.trainer_loss({"user_embeddings": ue, "item_embeddings": ie},
                                         {k: v.cpu() for k, v in batch.items()})
    want_loss.backward()
    assert abs(float(loss) - float(want_loss)) <= 1e-5 * abs(float(want_loss))
    got = dict(model.named_parameters())
    checked = 0
    for name, ref in p.items():
        if ref.grad is None or name not in got or got[name].grad is None:
            continue
        g, r = got[name].grad.detach().cpu().double(), ref.grad.double()
        scale = float(r.abs().max())
        if scale == 0.0:
            continue
        assert float((g - r).abs().max()) <= 2e-4 * scale, (name, float((g - r).abs().max()) / scale)
        checked += 1
    assert checked >= 25

    # evaluation at D = 128 on the embeddings just computed
    ued, ied = out["user_embeddings"].detach().contiguous(), out["item_embeddings"].detach().contiguous()
    ptr, cols = pairs_to_csr(tu, ti, U)
    pt, ct = torch.from_numpy(ptr).to(DEV), torch.from_numpy(cols).to(DEV)
    es, ei = ops.score_topk(ued, ied, 20, pt, ct, mode="exact")
    fs, fi = ops.score_topk(ued, ied, 20, pt, ct, mode="filter")
    assert torch.equal(ei, fi) and torch.equal(es.view(torch.int32), fs.view(torch.int32))
    us = np.sort(rng.choice(U, 128, replace=False))
    sub_ptr = np.concatenate([[0], np.cumsum(ptr[us + 1] - ptr[us])]).astype(np.int32)
    sub_cols = np.concatenate([cols[ptr[u]:ptr[u + 1]] for u in us]).astype(np.int32)
    o_s, o_i = eval_oracle.score_topk(ued.cpu().numpy()[us], ied.cpu().numpy(), 20, sub_ptr, sub_cols)
    np.testing.assert_array_equal(ei.cpu().numpy()[us], o_i)
    np.testing.assert_array_equal(es.cpu().numpy()[us], o_s)
