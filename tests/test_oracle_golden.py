"""The oracle against the golden vectors captured from the reference's own
modules (tests/golden/make_golden.py) and against independent formulations."""
import json
import os
import random

import numpy as np
import pytest
import torch

from oracle import eval_oracle, graph_oracle, loss_oracle, sampler_oracle

METRICS = ["precision", "recall", "hit_rate", "ndcg", "map", "mrr"]
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def test_metric_oracle_matches_reference(golden):
    z = golden("metric_cases.npz")
    names = sorted({k.split("/")[0] for k in z.files})
    assert len(names) >= 6
    for name in names:
        s = torch.from_numpy(z[f"{name}/scores"])
        t = torch.from_numpy(z[f"{name}/targets"])
        for k in z[f"{name}/ks"]:
            k = int(k)
            got = eval_oracle.evaluate_all_at_k(s, t, k)
            vec = np.array([float(got[m]) for m in METRICS], np.float32)
            np.testing.assert_array_equal(vec, z[f"{name}/all@{k}"])
            # the single-metric functions of the reference agree with its fused one
            np.testing.assert_allclose(vec, z[f"{name}/single@{k}"], rtol=1e-6, atol=1e-7)


def test_metric_known_answers(golden):
    # values captured from the live reference for the 3x6 example
    z = golden("metric_cases.npz")
    np.testing.assert_allclose(z["tiny/all@3"], [.22222222, .22222222, .33333334, .23463936, .18518519, .33333334], rtol=1e-6)
    np.testing.assert_allclose(z["tiny/all@6"], [.22222222, .66666669, .66666669, .42074421, .32407409, .38888887], rtol=1e-6)
    np.testing.assert_allclose(z["tiny_masked/all@3"], [.22222222, .22222222, .33333334, .17690708, .12962964, .16666667], rtol=1e-6)


def test_metric_oracle_errors():
    with open(os.path.join(GOLDEN, "metric_errors.json")) as f:
        errs = json.load(f)
    S, T = torch.rand(3, 6), torch.zeros(3, 6)
    for label, args in {"k0": (S, T, 0), "k_gt": (S, T, 7), "shape": (S, T[:, :5], 2), "dim": (S[0], T[0], 1)}.items():
        with pytest.raises(ValueError) as e:
            eval_oracle.evaluate_all_at_k(*args)
        assert str(e.value) == errs[label]


def test_sparse_metric_oracle_matches_dense(golden):
    z = golden("metric_cases.npz")
    s = torch.from_numpy(z["rand_64x500/scores"])
    t = torch.from_numpy(z["rand_64x500/targets"])
    topk = torch.topk(s, 20, dim=1).indices.numpy()
    nz = t.nonzero().numpy()
    ptr = np.concatenate([[0], np.cumsum(np.bincount(nz[:, 0], minlength=64))])
    sp = eval_oracle.metrics_from_topk_sparse(topk, ptr, nz[:, 1], [5, 10, 20])
    for k in (5, 10, 20):
        np.testing.assert_allclose([sp[k][m] for m in METRICS], z[f"rand_64x500/all@{k}"], rtol=2e-6, atol=1e-7)


def test_verifier_oracle_matches_reference(golden):
    z = golden("verifier_cases.npz")
    for name in ("v96x300", "v200x1000"):
        ue, ie = torch.from_numpy(z[f"{name}/user_emb"]), torch.from_numpy(z[f"{name}/item_emb"])
        tr, va = z[f"{name}/train_pairs"], z[f"{name}/val_pairs"]
        target = torch.zeros(ue.shape[0], ie.shape[0])
        target[va[:, 0], va[:, 1]] = 1
        res = eval_oracle.verify(ue, ie, target, tr[:, 0].tolist(), tr[:, 1].tolist(), [5, 10, 20], METRICS)
        assert list(res.keys()) == [str(k) for k in z[f"{name}/keys"]]
        np.testing.assert_allclose([float(v) for v in res.values()], z[f"{name}/values"], rtol=1e-6, atol=1e-8)


def test_chain_topk_equals_reference_topk_without_near_ties(golden):
    """Chain-ordered scores + (score desc, id asc) give the reference's torch.topk ids
    whenever the reference's own top-(k+1) has no near-tie (recorded per fixture)."""
    z = golden("verifier_cases.npz")
    for name in ("v96x300", "v200x1000"):
        assert float(z[f"{name}/min_rel_gap"]) > 1e-6
        ue, ie = z[f"{name}/user_emb"], z[f"{name}/item_emb"]
        tr = z[f"{name}/train_pairs"]
        ptr = np.concatenate([[0], np.cumsum(np.bincount(tr[:, 0], minlength=ue.shape[0]))]).astype(np.int32)
        _s, ids = eval_oracle.score_topk(ue, ie, 20, ptr, tr[:, 1].astype(np.int32))
        np.testing.assert_array_equal(ids, z[f"{name}/topk20"])


def test_chain_scores_are_fp32_accurate():
    rng = np.random.default_rng(0)
    U = rng.standard_normal((20, 128)).astype(np.float32)
    I = rng.standard_normal((33, 128)).astype(np.float32)
    S = eval_oracle.scores_chain(U, I)
    ref = U.astype(np.float64) @ I.astype(np.float64).T
    bound = (np.abs(U).astype(np.float64) @ np.abs(I).astype(np.float64).T).max()
    assert np.abs(S - ref).max() <= 1e-6 * bound
    # and the C top-K agrees with a stable sort of the same matrix, ties included
    Uq = np.round(U * 4) / 4
    Iq = np.round(I * 4) / 4          # coarse values -> many exactly equal scores
    Sq = eval_oracle.scores_chain(Uq, Iq)
    s2, i2 = eval_oracle.score_topk(Uq, Iq, 7)
    s3, i3 = eval_oracle.topk_by_sort(Sq, 7)
    np.testing.assert_array_equal(i2, i3)
    np.testing.assert_array_equal(s2, s3)


def test_loss_oracle_matches_reference(golden):
    z = golden("loss_cases.npz")
    assert abs(float(z["ka/bpr_nomask"]) - 0.840064287185669) < 1e-7
    assert abs(float(z["ka/bpr_mask"]) - 0.7002001404762268) < 1e-7
    for name in ("b64r1", "b200r4", "b33r3d128"):
        ue = torch.from_numpy(z[f"{name}/user_emb"]).requires_grad_(True)
        ie = torch.from_numpy(z[f"{name}/item_emb"]).requires_grad_(True)
        batch = {"user_ids": torch.from_numpy(z[f"{name}/users"]), "item_ids": torch.from_numpy(z[f"{name}/pos"]),
                 "neg_items": torch.from_numpy(z[f"{name}/neg"])}
        loss = loss_oracle.trainer_loss({"user_embeddings": ue, "item_embeddings": ie}, batch)
        gu, gi = torch.autograd.grad(loss, [ue, ie])
        assert abs(float(loss.detach()) - float(z[f"{name}/loss"])) <= 1e-7
        # index_add order in torch's gather backward depends on the thread count
        np.testing.assert_allclose(gu.numpy(), z[f"{name}/grad_user"], rtol=1e-6, atol=1e-9)
        np.testing.assert_allclose(gi.numpy(), z[f"{name}/grad_item"], rtol=1e-6, atol=1e-9)
        lm = loss_oracle.bpr_loss(torch.from_numpy(z[f"{name}/pos_scores"]), torch.from_numpy(z[f"{name}/neg_scores"]),
                                  torch.from_numpy(z[f"{name}/mask"]))
        assert abs(float(lm) - float(z[f"{name}/loss_masked"])) <= 1e-7
    with pytest.raises(Exception, match="batch error"):
        loss_oracle.trainer_loss({"user_embeddings": ue, "item_embeddings": ie}, {})


def test_sampler_oracle_matches_reference_loader(golden):
    z = golden("loader_batches.npz")
    users, items = z["users"], z["items"]
    uii = {}
    for u, i in zip(users, items):
        uii.setdefault(int(u), set()).add(int(i))
    all_items = set(int(i) for i in np.unique(items))
    for tag, R in (("r1_b64", 1), ("r3_b50", 3), ("r8_b32", 8)):
        random.seed(42)
        for e in range(2):
            got = sampler_oracle.generate_negative_samples(z[f"{tag}/e{e}/user_ids"], uii, all_items, R)
            np.testing.assert_array_equal(np.asarray(got), z[f"{tag}/e{e}/neg_items"])


def test_graph_oracle_against_scipy_and_dense():
    import scipy.sparse as sp
    rng = np.random.default_rng(1)
    U, I, E, D = 30, 20, 150, 16
    inter = [(int(rng.integers(U)), int(rng.integers(I)), float(rng.integers(1, 6))) for _ in range(E)]
    src, dst, w = graph_oracle.build_coo(inter, U, I, True)
    N = U + I
    assert src.shape[0] == 2 * E + N and w[-1] == max(r for _, _, r in inter)
    # vectorised construction is the same edge list
    a = np.array(inter)
    s2, d2, w2 = graph_oracle.build_coo_vectorized(a[:, 0], a[:, 1], a[:, 2], U, I, True)
    np.testing.assert_array_equal(src, s2), np.testing.assert_array_equal(dst, d2), np.testing.assert_array_equal(w, w2)
    wn, in_w = graph_oracle.normalize(src, dst, w, N)
    A = sp.coo_matrix((w.astype(np.float64), (dst, src)), shape=(N, N)).toarray()   # multi-edges add up
    deg = np.sqrt(np.maximum(A.sum(1), 1e-12))
    An = A / deg[:, None] / deg[None, :]
    Aw = sp.coo_matrix((wn.astype(np.float64), (dst, src)), shape=(N, N)).toarray()
    np.testing.assert_allclose(Aw, An, rtol=1e-6, atol=1e-8)
    h = rng.standard_normal((N, D)).astype(np.float32)
    out = graph_oracle.spmm(src, dst, wn, h)
    np.testing.assert_allclose(out, An @ h.astype(np.float64), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(out, graph_oracle.spmm_f64(src, dst, wn, h), rtol=1e-5, atol=1e-6)
    k, alpha = 3, 0.2
    ref = h.astype(np.float64)
    for _ in range(k):
        ref = alpha * h + (1 - alpha) * (An @ ref)
    np.testing.assert_allclose(graph_oracle.aggrate(src, dst, wn, h, k, alpha), ref, rtol=1e-5, atol=1e-6)
    # CSR contract: stable by destination, ties by edge id
    ptr, idx, eid = graph_oracle.csr_from_coo(dst, src, N)
    for v in range(N):
        e = eid[ptr[v]:ptr[v + 1]]
        assert np.all(dst[e] == v) and np.all(np.diff(e) > 0)
        np.testing.assert_array_equal(idx[ptr[v]:ptr[v + 1]], src[e])


def test_config_presets_fixture():
    with open(os.path.join(GOLDEN, "config_presets.json")) as f:
        d = json.load(f)
    assert d["baby"]["data"]["batch_size"] == 2048 and d["baby"]["data"]["num_users"] == 19445
    assert d["elec"]["data"]["batch_size"] == 256 and d["baby"]["evaluation"]["k_values"] == [5, 10, 20]


def test_loss_variant_oracles_match_reference(golden):
    """mmgcn_loss and compute_info_bpr_loss restatements against values captured from evalue.loss."""
    z = golden("loss_cases.npz")
    for name in ("b64r1", "b200r4", "b33r3d128"):
        ue, ie = torch.from_numpy(z[f"{name}/user_emb"]), torch.from_numpy(z[f"{name}/item_emb"])
        users, pos, neg = (torch.from_numpy(z[f"{name}/{k}"]) for k in ("users", "pos", "neg"))
        got = float(loss_oracle.compute_info_bpr_loss(ue, ie, torch.stack([users, pos], 1), neg))
        assert abs(got - float(z[f"{name}/info_bpr_loss"])) <= 1e-6 * abs(float(z[f"{name}/info_bpr_loss"]))
        if f"{name}/mmgcn_loss" in z:
            outputs = {"user_embeddings": ue, "item_embeddings": ie, "id_embeddings": torch.cat([ue, ie]), "pres": ue}
            got = float(loss_oracle.mmgcn_loss(outputs, {"user_ids": users, "item_ids": pos, "neg_items": neg}))
            assert abs(got - float(z[f"{name}/mmgcn_loss"])) <= 1e-6 * abs(float(z[f"{name}/mmgcn_loss"]))


def test_propagation_variant_oracles_against_dense_fp64():
    """The restatements of MGDCF / II_GCN / IU_GCN agree with a dense float64 evaluation of the
    same recurrences (parity unpinned with respect to DGL, which is not installed)."""
    import numpy as np
    from oracle import graph_oracle
    rng = np.random.default_rng(0)
    U, I = 30, 20
    pairs = np.unique(np.stack([rng.integers(0, U, 150), rng.integers(0, I, 150)], 1), axis=0)
    src, dst, n = graph_oracle.mgdcf_sorted_homo_edges(pairs[:, 0], pairs[:, 1], U, I)
    assert src.shape[0] == 2 * pairs.shape[0] + n
    w = graph_oracle.deg_normalize(src, dst, n)
    A = np.zeros((n, n))
    np.add.at(A, (dst, src), w.astype(np.float64))
    assert np.allclose(A, A.T)                      # symmetric normalisation of a symmetric graph
    x = rng.standard_normal((n, 8)).astype(np.float32)
    k, alpha, beta = 3, 0.1, 0.9
    h, hs = x.astype(np.float64), []
    for _ in range(k):
        h = beta * (A @ h) + alpha * x
        hs.append(h)
    gam = [graph_oracle.mgdcf_gamma(alpha, beta, i + 1) for i in range(k)]
    assert abs(gam[0] - (beta + alpha)) < 1e-12
    np.testing.assert_allclose(graph_oracle.mgdcf_forward(src, dst, w, x, k, alpha, beta), hs[-1] / gam[-1], rtol=1e-5, atol=1e-6)
    for got, want, g_ in zip(graph_oracle.mgdcf_forward(src, dst, w, x, k, alpha, beta, True), hs, gam):
        np.testing.assert_allclose(got, want / g_, rtol=1e-5, atol=1e-6)
    h = x.astype(np.float64)
    for _ in range(k):
        h = (beta * (A @ h) + alpha * x) / gam[-1]
    np.testing.assert_allclose(graph_oracle.ii_gcn_forward(src, dst, w, x, k, alpha, beta), h, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(graph_oracle.iu_gcn_forward(src, dst, w, x, 2), A @ (A @ x), rtol=1e-5, atol=1e-6)
