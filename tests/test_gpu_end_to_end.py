"""The pieces composed as the reference's trainer composes them (train/graph_trainer.py:133-165,
181-253; evalue/evaluator.py:36-63): the loader's batches -> the captured training step (forward,
BPR loss with in-kernel negative lookup, backward, clip + Adam) for 20+ steps -> Verifier.verify,
against the oracle's CPU model driven by torch.optim.Adam + clip_grad_norm_ on the batches the
REFERENCE's own loader produced (tests/golden/loader_batches.npz, num_workers 0 and 4).

Each piece is pinned on its own elsewhere; this checks that nothing is lost between them: the
batch order and negatives of the loader, the step counter of the optimizer, gradients flowing
through every layer for many steps, the model an evaluator then sees."""
import random

import numpy as np
import pytest
import torch

from oracle import eval_oracle, graph_oracle, loss_oracle, mmfcn_oracle
from recommend_amd import ops
from recommend_amd.evaluator import Verifier
from recommend_amd.loss import gather_dot_bpr_loss
from recommend_amd.model import ModelFactory
from recommend_amd.optim import FusedClipAdam
from recommend_amd.sampler import NegativeSampler, TrainBatchLoader, build_user_csr
from recommend_amd.train_step import GraphedTrainStep

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
METRICS = ["precision", "recall", "hit_rate", "ndcg", "map", "mrr"]
LR = 1e-3


# (fixture tag, batch size, loader workers, epochs): 22 and 24 steps
@pytest.mark.parametrize("tag,bs,workers,epochs", [("r1_b32_w4", 32, 4, 2), ("r1_b64", 64, 0, 4)])
def test_training_run_then_verify_vs_oracle(golden, tag, bs, workers, epochs):
    z = golden("loader_batches.npz")
    U, I = int(z["n_users"]), int(z["n_items"])
    label = z["label"]
    tr, va = label == 0, label == 1
    tu, ti, trat = z["users"][tr], z["items"][tr], z["rating"][tr]
    vu, vi = z["users"][va], z["items"][va]
    D = 64

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
        class evaluation:
            metrics = METRICS
            k_values = [5, 10, 20]

    g = torch.Generator().manual_seed(17)
    feats = {"image": torch.randn(I, 96, generator=g), "text": torch.randn(I, 40, generator=g)}
    torch.manual_seed(17)
    model = ModelFactory.create_Model(C, {}, feats).to(DEV)
    model.build_graph((tu, ti, trat))
    model.creat_feature_weight()
    model.train()

    # ---- the oracle's model: same parameters, CPU
    N = U + I
    src, dst, w = graph_oracle.build_coo_vectorized(tu, ti, trat, U, I, True)
    wn, _ = graph_oracle.normalize(src, dst, w, N)
    A = mmfcn_oracle.adjacency(src, dst, wn, N)
    p = {k: v.detach().cpu().clone().requires_grad_(v.dtype.is_floating_point) for k, v in model.state_dict().items()}
    padded = [torch.cat([torch.zeros(U, f.shape[1]), f]) for f in feats.values()]
    embs = model.embs.detach().cpu().clone()
    alphas = model.alphas.detach().cpu().clone()       # a plain tensor in the reference: no optimizer holds it
    trainable = [v for v in p.values() if v.requires_grad]
    o_opt = torch.optim.Adam(trainable, lr=LR)

    def oracle_forward():
        return mmfcn_oracle.forward(p, A, padded, ["linears.0.linear", "linears.1.linear"], embs, model.ks, alphas, 2, U)

    # ---- the product's loader, seeded as main.set_seed seeds the reference's
    pp, it = build_user_csr(z["users"], z["items"], U)          # positives over the FULL table
    random.seed(42), np.random.seed(42), torch.manual_seed(42)
    sampler = NegativeSampler(pp, it, I, seed=42, neg_ratio=1)
    dp, di = sampler.device_csr(DEV)
    loader = TrainBatchLoader(tu, ti, trat, bs, sampler, device=DEV, resolve_on_host=False, num_workers=workers)
    opt = FusedClipAdam(model.parameters(), lr=LR)

    stepper, losses, o_losses, steps = None, [], [], 0
    try:
        for e in range(epochs):
            batches = list(loader)
            if e < 2:   # the fixture holds the reference loader's first two epochs
                cat = lambda k: np.concatenate([b[k].cpu().numpy() for b in batches])
                np.testing.assert_array_equal(cat("user_ids"), z[f"{tag}/e{e}/user_ids"])
                np.testing.assert_array_equal(cat("item_ids"), z[f"{tag}/e{e}/item_ids"])
            off = 0
            for bi, batch in enumerate(batches):
                n = batch["user_ids"].shape[0]
                if stepper is None:
                    # warm-up steps are real steps: one, on this first batch; the recording that follows does not step
                    stepper = GraphedTrainStep(model, opt, gather_dot_bpr_loss, batch, constants={"pos_ptr": dp, "pos_items": di},
                                               max_grad_norm=1.0, warmup_steps=1)
                    loss = stepper.warmup_losses[0]
                else:
                    loss = stepper(batch)       # the last batch of an epoch is shorter: it runs eagerly
                # the oracle's step on the reference loader's own batch (ids AND negatives from the fixture)
                if e < 2:
                    sl = slice(off, off + n)
                    ob = {"user_ids": torch.from_numpy(z[f"{tag}/e{e}/user_ids"][sl]),
                          "item_ids": torch.from_numpy(z[f"{tag}/e{e}/item_ids"][sl]),
                          "neg_items": torch.from_numpy(z[f"{tag}/e{e}/neg_items"][sl])}
                else:           # past the fixture: the product loader's batch, negatives resolved by the sampler oracle's rule
                    ranks = batch["neg_rank"].cpu().numpy()
                    users = batch["user_ids"].cpu().numpy()
                    neg = np.empty_like(ranks, dtype=np.int64)
                    for s, (u, rk) in enumerate(zip(users, ranks)):
                        cand = np.setdiff1d(np.arange(I), it[pp[u]:pp[u + 1]])
                        neg[s] = cand[rk]
                    ob = {"user_ids": batch["user_ids"].cpu(), "item_ids": batch["item_ids"].cpu(),
                          "neg_items": torch.from_numpy(neg)}
                off += n
                o_opt.zero_grad(set_to_none=True)
                ue, ie = oracle_forward()
                o_loss = loss_oracle.trainer_loss({"user_embeddings": ue, "item_embeddings": ie}, ob)
                o_loss.backward()
                torch.nn.utils.clip_grad_norm_([v for v in trainable if v.grad is not None], 1.0)
                o_opt.step()
                if loss is not None:
                    losses.append(float(loss))
                    o_losses.append(float(o_loss))
                steps += 1
    finally:
        ops.set_dropout_step_tensor(None)
    assert steps >= 20 and int(opt.step_count) == steps
    np.testing.assert_allclose(losses, o_losses, rtol=1e-4)
    assert o_losses[-1] < o_losses[0]

    # ---- evaluation of the trained model through the reference's evaluator interface
    target = torch.zeros(U, I)
    target[torch.from_numpy(vu), torch.from_numpy(vi)] = 1
    mask = (tu.tolist(), ti.tolist())
    res = Verifier(C, None, target, mask).verify(model)
    with torch.no_grad():
        ue, ie = oracle_forward()
    want = eval_oracle.verify(ue, ie, target, mask[0], mask[1], [5, 10, 20], METRICS)
    assert set(res) == set(want)
    for k, v in want.items():
        assert abs(float(res[k]) - float(v)) <= 1e-4 * max(abs(float(v)), 1e-3), (k, float(res[k]), float(v))
    # ranked ids: equal wherever the oracle's own ranking is not a near-tie (adjacent gap > 1e-5 relative)
    out = model({})
    ptr_m, cols_m = ops_mask(tu, ti, U)
    _s, ids = ops.score_topk(out["user_embeddings"], out["item_embeddings"], 20, ptr_m, cols_m)
    S = (ue @ ie.T).numpy()
    S[mask[0], mask[1]] = -1e10
    o_s, o_i = eval_oracle.topk_by_sort(S, 21)
    gap = (o_s[:, :-1] - o_s[:, 1:]) / np.maximum(np.abs(o_s[:, :-1]), 1e-30)
    clear = np.minimum.accumulate(gap > 1e-5, axis=1)        # positions before the first near-tie of a row
    ids = ids.cpu().numpy()
    assert clear[:, 0].mean() > 0.9                           # the comparison is not vacuous
    assert np.array_equal(ids[clear], o_i[:, :20][clear])


def ops_mask(tu, ti, U):
    from recommend_amd.evaluator import pairs_to_csr
    ptr, cols = pairs_to_csr(tu, ti, U)
    return torch.from_numpy(ptr).to(DEV), torch.from_numpy(cols).to(DEV)
