"""GPU parity: fused score + mask + top-K, list merge and ranking metrics against
the golden vectors of the reference's Verifier and the exact oracle."""
import json
import os
import re

import numpy as np
import pytest
import torch

from oracle import eval_oracle
from recommend_amd import ops, synth
from recommend_amd.evaluator import Tester, Verifier, RankingEvaluator, pairs_to_csr

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
METRICS = ["precision", "recall", "hit_rate", "ndcg", "map", "mrr"]


class Cfg:
    class evaluation:
        metrics = METRICS
        k_values = [5, 10, 20]
    class system:
        device = DEV
    class data:
        num_users = -1


class Stub:
    def __init__(self, u, i):
        self.u, self.i, self.evaled = u, i, False

    def eval(self):
        self.evaled = True

    def __call__(self, batch):
        assert batch == {}
        return {"user_embeddings": self.u, "item_embeddings": self.i}


def dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    return t.to(DEV) if dtype is None else t.to(DEV, dtype)


@pytest.fixture(autouse=True, params=["lists", "threshold"])
def filter_form(request, monkeypatch):
    """The candidate filter has two forms (sorted lists per user in one pass; strata maxima ->
    threshold -> candidates in two) and picks one by shape; every case here runs through both
    (RK_FILTER_FORM is read by the library at every call)."""
    monkeypatch.setenv("RK_FILTER_FORM", request.param)
    return request.param


@pytest.mark.parametrize("name", ["v96x300", "v200x1000"])
def test_verifier_and_tester_vs_reference(golden, name):
    z = golden("verifier_cases.npz")
    ue, ie = dev(z[f"{name}/user_emb"]), dev(z[f"{name}/item_emb"])
    tr, va = z[f"{name}/train_pairs"], z[f"{name}/val_pairs"]
    target = torch.zeros(ue.shape[0], ie.shape[0])
    target[va[:, 0], va[:, 1]] = 1
    mask = (tr[:, 0].tolist(), tr[:, 1].tolist())
    model = Stub(ue, ie)
    res = Verifier(Cfg, None, target, mask).verify(model)
    assert model.evaled
    assert list(res.keys()) == [str(k) for k in z[f"{name}/keys"]]
    for v in res.values():
        assert isinstance(v, torch.Tensor) and v.dim() == 0
    np.testing.assert_allclose([float(v) for v in res.values()], z[f"{name}/values"], rtol=1e-5, atol=1e-7)
    res_t = Tester(Cfg, None, target, mask).test(Stub(ue, ie))
    assert {k: float(v) for k, v in res_t.items()} == {k: float(v) for k, v in res.items()}
    # ranked ids equal the reference's torch.topk ids (fixture has no near-ties) ...
    ev = RankingEvaluator(target, mask, [5, 10, 20], METRICS)
    scores, ids = ev.topk(ue, ie, 20)
    np.testing.assert_array_equal(ids.cpu().numpy(), z[f"{name}/topk20"])
    # ... and ids AND scores equal the exact oracle bit for bit
    ptr, cols = pairs_to_csr(tr[:, 0], tr[:, 1], ue.shape[0])
    o_s, o_i = eval_oracle.score_topk(z[f"{name}/user_emb"], z[f"{name}/item_emb"], 20, ptr, cols)
    np.testing.assert_array_equal(ids.cpu().numpy(), o_i)
    np.testing.assert_array_equal(scores.cpu().numpy(), o_s)
    # sparse target input gives the same result
    res_sp = RankingEvaluator((va[:, 0], va[:, 1]), mask, [5, 10, 20], METRICS, n_users=ue.shape[0]).evaluate(ue, ie)
    assert {k: float(v) for k, v in res_sp.items()} == {k: float(v) for k, v in res.items()}


@pytest.mark.parametrize("U,I,D,K", [(1, 1, 64, 1), (5, 7, 64, 7), (129, 33, 64, 20), (300, 1000, 128, 32),
                                      (257, 4099, 64, 20), (1000, 257, 64, 5), (64, 20000, 64, 10),
                                      (2500, 400, 64, 20), (1500, 3000, 128, 10)])
@pytest.mark.parametrize("mode", ["exact", "filter"])
def test_topk_exact_with_ties_and_ragged_shapes(U, I, D, K, mode):
    """Quantised embeddings give many exactly equal scores: ids must come out in
    (score desc, id asc) order, identical to the oracle, for awkward shapes."""
    rng = np.random.default_rng(U * 7 + I)
    ue = (np.round(rng.standard_normal((U, D)) * 2) / 2).astype(np.float32)
    ie = (np.round(rng.standard_normal((I, D)) * 2) / 2).astype(np.float32)
    n_mask = min(U * I // 3, U * 40)
    rows, cols = rng.integers(0, U, n_mask), rng.integers(0, I, n_mask)
    if U > 2:   # one user with everything masked, one with nothing
        rows = np.concatenate([rows[rows > 1], np.zeros(I, np.int64)])
        cols = np.concatenate([cols[:rows.shape[0] - I], np.arange(I)])
    ptr, mcols = pairs_to_csr(rows, cols, U)
    s, ids = ops.score_topk(dev(ue), dev(ie), K, dev(ptr), dev(mcols), mode=mode)
    o_s, o_i = eval_oracle.score_topk(ue, ie, K, ptr, mcols)
    np.testing.assert_array_equal(ids.cpu().numpy(), o_i)
    np.testing.assert_array_equal(s.cpu().numpy(), o_s)
    # no mask at all
    s2, ids2 = ops.score_topk(dev(ue), dev(ie), K, mode=mode)
    o_s2, o_i2 = eval_oracle.score_topk(ue, ie, K)
    np.testing.assert_array_equal(ids2.cpu().numpy(), o_i2)
    np.testing.assert_array_equal(s2.cpu().numpy(), o_s2)


@pytest.mark.parametrize("U,I,D,K,scale", [(3000, 9001, 64, 20, 0.1), (700, 5000, 128, 10, 1.0),
                                            (520, 2000, 64, 28, 30.0), (300, 2000, 64, 32, 0.1),
                                            (1300, 700, 128, 5, 0.1)])
def test_filtered_ranking_equals_exact_kernel(U, I, D, K, scale):
    """rk_score_topk_f32 (bf16 candidate filter, exact re-scoring, exact kernel for whoever is
    left) returns the bits of rk_score_topk_exact_f32; K > 28 and widths other than 64/128 take
    the exact kernel directly."""
    g = torch.Generator().manual_seed(U + I)
    ue = (torch.randn(U, D, generator=g) * scale).to(DEV)
    ie = (torch.randn(I, D, generator=g) * scale).to(DEV)
    ie[::7] *= 4.0          # a wide spread of item norms loosens the filter's error bound
    rng = np.random.default_rng(I)
    ptr, mcols = pairs_to_csr(rng.integers(0, U, 20 * U), rng.integers(0, I, 20 * U), U)
    for off in (0, 12345):
        fs, fi = ops.score_topk(ue, ie, K, dev(ptr), dev(mcols), item_id_offset=off, mode="filter")
        es, ei = ops.score_topk(ue, ie, K, dev(ptr), dev(mcols), item_id_offset=off, mode="exact")
        assert torch.equal(fi, ei)
        assert torch.equal(fs.view(torch.int32), es.view(torch.int32))
    o_s, o_i = eval_oracle.score_topk(ue[:64].cpu().numpy(), ie.cpu().numpy(), K, ptr[:65], mcols[:ptr[64]])
    np.testing.assert_array_equal(fi[:64].cpu().numpy() - 12345, o_i)
    np.testing.assert_array_equal(fs[:64].cpu().numpy(), o_s)


@pytest.mark.parametrize("U,I,D,K", [(20000, 7000, 64, 20), (6000, 9000, 128, 10), (3000, 40000, 64, 20)])
def test_filter_certifies_ordinary_inputs(U, I, D, K):
    """Parity cannot see a weak filter: whatever it fails to certify the exact kernel ranks, and the
    bits come out right, only slowly.  On ordinary inputs (Gaussian embeddings, short training
    lists) nearly every user must be settled by the filter: the library reports how many were not."""
    from ctypes import byref, c_int32, c_size_t
    from recommend_amd import _ffi
    g = torch.Generator().manual_seed(U + I)
    ue = (torch.randn(U, D, generator=g) * 0.1).to(DEV)
    ie = (torch.randn(I, D, generator=g) * 0.1).to(DEV)
    rng = np.random.default_rng(I)
    ptr, mcols = pairs_to_csr(rng.integers(0, U, 10 * U), rng.integers(0, I, 10 * U), U)
    mp, mc = dev(ptr), dev(mcols)
    lib = _ffi.load()
    ws_bytes = lib.rk_score_topk_ws_bytes(U, I, D, K)
    ws = torch.empty(ws_bytes // 4 + 4, dtype=torch.float32, device=DEV)
    out_s = torch.empty((U, K), dtype=torch.float32, device=DEV)
    out_i = torch.empty((U, K), dtype=torch.int32, device=DEV)
    _ffi.call("rk_score_topk_filtered_f32", ops.ptr(ue), c_int32(U), ops.ptr(ie), c_int32(I), c_int32(D), ops.ptr(mp),
              ops.ptr(mc), c_int32(K), c_int32(0), ops.ptr(out_s), ops.ptr(out_i), ops.ptr(ws), c_size_t(ws.numel() * 4),
              ops.stream_ptr(ue.device))
    n_open = c_int32(-1)
    _ffi.call("rk_score_topk_open_users", c_int32(U), c_int32(I), c_int32(D), c_int32(K), ops.ptr(ws),
              c_size_t(ws.numel() * 4), ops.stream_ptr(ue.device), byref(n_open))
    # (up to 512 users with the longest training lists always go to the exact kernel)
    assert 0 <= n_open.value <= 512 + U // 100, n_open.value
    es, ei = ops.score_topk(ue, ie, K, mp, mc, mode="exact")
    assert torch.equal(out_i, ei) and torch.equal(out_s.view(torch.int32), es.view(torch.int32))


@pytest.mark.parametrize("U,I,D,K", [(700, 3000, 256, 20), (130, 517, 192, 32), (65, 129, 320, 5)])
def test_other_embedding_widths_take_the_chunked_kernel(U, I, D, K):
    """Widths other than 64 / 128 (the reference's default emb_dim is 256) run the kernel that walks
    the width in chunks: the same score chain, mask, tie order and id offset, bit for bit the oracle's."""
    rng = np.random.default_rng(D + U)
    ue = (np.round(rng.standard_normal((U, D)) * 2) / 2).astype(np.float32)     # many exact ties
    ie = (np.round(rng.standard_normal((I, D)) * 2) / 2).astype(np.float32)
    ptr, mcols = pairs_to_csr(rng.integers(0, U, 30 * U), rng.integers(0, I, 30 * U), U)
    for mode in ("auto", "exact", "filter"):
        s, ids = ops.score_topk(dev(ue), dev(ie), K, dev(ptr), dev(mcols), item_id_offset=1000, mode=mode)
        o_s, o_i = eval_oracle.score_topk(ue, ie, K, ptr, mcols, id_offset=1000)
        np.testing.assert_array_equal(ids.cpu().numpy(), o_i)
        np.testing.assert_array_equal(s.cpu().numpy(), o_s)
    g = torch.Generator().manual_seed(D)
    uf, itf = torch.randn(U, D, generator=g), torch.randn(I, D, generator=g)
    s, ids = ops.score_topk(uf.to(DEV), itf.to(DEV), K)
    o_s, o_i = eval_oracle.score_topk(uf.numpy(), itf.numpy(), K)
    np.testing.assert_array_equal(ids.cpu().numpy(), o_i)
    np.testing.assert_array_equal(s.cpu().numpy(), o_s)


@pytest.mark.parametrize("dense_users", [40, 3000])
def test_users_with_long_training_lists_skip_the_filter(dense_users):
    """Users with very long training lists are not masked by the filter pass and are ranked by the
    exact kernel's compacted launch; how long is "very long" adapts so that at most 512 users
    qualify.  A few heavy users (all of them skip) and a data set where most users are heavy (the
    threshold rises, the rest are masked in the filter as usual): the exact kernel's bits either way."""
    U, I, D, K = 3200, 8192, 64, 20
    g = torch.Generator().manual_seed(dense_users)
    ue, ie = (torch.randn(U, D, generator=g) * 0.1).to(DEV), (torch.randn(I, D, generator=g) * 0.1).to(DEV)
    rng = np.random.default_rng(dense_users)
    rows, cols = [rng.integers(0, U, 10 * U)], [rng.integers(0, I, 10 * U)]
    for u in range(dense_users):                       # 100 .. 4000 masked items each
        n = int(rng.integers(100, 4000))
        rows.append(np.full(n, u * (U // dense_users)))
        cols.append(rng.choice(I, n, replace=False))
    ptr, mcols = pairs_to_csr(np.concatenate(rows), np.concatenate(cols), U)
    fs, fi = ops.score_topk(ue, ie, K, dev(ptr), dev(mcols), mode="filter")
    es, ei = ops.score_topk(ue, ie, K, dev(ptr), dev(mcols), mode="exact")
    assert torch.equal(fi, ei)
    assert torch.equal(fs.view(torch.int32), es.view(torch.int32))
    o_s, o_i = eval_oracle.score_topk(ue[:40].cpu().numpy(), ie.cpu().numpy(), K, ptr[:41], mcols[:ptr[40]])
    np.testing.assert_array_equal(fi[:40].cpu().numpy(), o_i)
    np.testing.assert_array_equal(fs[:40].cpu().numpy(), o_s)


@pytest.mark.parametrize("U", [300, 1500])
def test_filter_falls_back_when_nothing_can_be_certified(U):
    """All-zero users tie every item (nobody certified: few users go through the compacted exact
    launch, more than its 1024 slots through the flagged full-grid one); non-finite embeddings
    make the error bound useless.  Either way the exact kernel's answer comes out."""
    I, D, K = 900, 64, 20
    g = torch.Generator().manual_seed(U)
    ie = torch.randn(I, D, generator=g).to(DEV)
    ue = torch.zeros(U, D, device=DEV)
    ue[::3] = torch.randn(U, D, generator=g).to(DEV)[::3]
    rng = np.random.default_rng(U)
    ptr, mcols = pairs_to_csr(rng.integers(0, U, 5 * U), rng.integers(0, I, 5 * U), U)
    fs, fi = ops.score_topk(ue, ie, K, dev(ptr), dev(mcols), mode="filter")
    es, ei = ops.score_topk(ue, ie, K, dev(ptr), dev(mcols), mode="exact")
    assert torch.equal(fi, ei) and torch.equal(fs, es)
    o_s, o_i = eval_oracle.score_topk(ue.cpu().numpy(), ie.cpu().numpy(), K, ptr, mcols)
    np.testing.assert_array_equal(fi.cpu().numpy(), o_i)
    ie2 = ie.clone()
    ie2[5, 3] = float("inf")
    ie2[17, 0] = float("nan")
    ie2[40] = 3e38
    fs, fi = ops.score_topk(ue, ie2, K, dev(ptr), dev(mcols), mode="filter")
    es, ei = ops.score_topk(ue, ie2, K, dev(ptr), dev(mcols), mode="exact")
    assert torch.equal(fi, ei)
    assert torch.equal(fs.view(torch.int32), es.view(torch.int32))


@pytest.mark.parametrize("seed", range(24))
def test_filtered_equals_exact_on_random_configurations(seed):
    """Random shapes, K, embedding magnitudes (down to fp16-subnormal products and up to values far
    beyond the fp16 range: the per-row / per-table power-of-two scaling), mask densities, duplicate
    rows (exact ties) and sparse rows: the filtered path returns the exact kernel's bits."""
    rng = np.random.default_rng(1000 + seed)
    U = int(rng.integers(1, 700))
    I = int(rng.integers(1, 3000))
    D = int(rng.choice([64, 128]))
    K = int(rng.integers(1, min(I, 28) + 1))
    su, si = 10.0 ** rng.uniform(-18, 12), 10.0 ** rng.uniform(-18, 12)
    ue = (rng.standard_normal((U, D)) * su).astype(np.float32)
    ie = (rng.standard_normal((I, D)) * si).astype(np.float32)
    ue *= (10.0 ** rng.uniform(-3, 3, size=(U, 1))).astype(np.float32)       # users of very different norms
    if I > 4:
        ie[rng.integers(0, I, I // 4)] = ie[rng.integers(0, I, I // 4)]       # duplicate items: exact ties
        ie[rng.integers(0, I, max(1, I // 10))] *= np.float32(rng.choice([0.0, 1e-4, 50.0]))
    ie[:, rng.integers(0, D, D // 3)] = 0.0
    n_mask = int(rng.integers(0, 30 * U + 1))
    ptr, mcols = pairs_to_csr(rng.integers(0, U, n_mask), rng.integers(0, I, n_mask), U)
    mp, mc = (dev(ptr), dev(mcols)) if n_mask else (None, None)
    fs, fi = ops.score_topk(dev(ue), dev(ie), K, mp, mc, mode="filter")
    es, ei = ops.score_topk(dev(ue), dev(ie), K, mp, mc, mode="exact")
    assert torch.equal(fi, ei), (U, I, D, K, su, si)
    assert torch.equal(fs.view(torch.int32), es.view(torch.int32))


def test_filter_with_subnormal_and_huge_rows():
    """User rows whose largest entry is an fp32 subnormal (the scale 2^-e is not representable) or
    close to the fp32 maximum, next to ordinary ones."""
    g = torch.Generator().manual_seed(3)
    U, I, D, K = 200, 1500, 64, 10
    ue = torch.randn(U, D, generator=g)
    ue[0::4] *= 1e-41
    ue[1::4] *= 1e-37
    ue[2::4] *= 3e37
    ie = torch.randn(I, D, generator=g) * 0.5
    ie[::9] *= 1e-30
    rng = np.random.default_rng(3)
    ptr, mcols = pairs_to_csr(rng.integers(0, U, 2000), rng.integers(0, I, 2000), U)
    fs, fi = ops.score_topk(ue.to(DEV), ie.to(DEV), K, dev(ptr), dev(mcols), mode="filter")
    es, ei = ops.score_topk(ue.to(DEV), ie.to(DEV), K, dev(ptr), dev(mcols), mode="exact")
    assert torch.equal(fi, ei)
    assert torch.equal(fs.view(torch.int32), es.view(torch.int32))
    assert int((ei >= 0).sum()) == U * K


def test_filter_when_real_scores_fall_below_the_mask_score():
    """Scores around -6e11: the masked items (-1e10) are the best ones of the exact ranking, while
    the filter's own order puts them last.  Nobody may be certified on the filter's word."""
    g = torch.Generator().manual_seed(5)
    U, I, D, K = 150, 800, 64, 10
    ue = torch.randn(U, D, generator=g).abs() * 1e5
    ie = -torch.randn(I, D, generator=g).abs() * 1e5
    rng = np.random.default_rng(5)
    ptr, mcols = pairs_to_csr(rng.integers(0, U, 1500), rng.integers(0, I, 1500), U)
    fs, fi = ops.score_topk(ue.to(DEV), ie.to(DEV), K, dev(ptr), dev(mcols), mode="filter")
    es, ei = ops.score_topk(ue.to(DEV), ie.to(DEV), K, dev(ptr), dev(mcols), mode="exact")
    assert float(es.max()) == -1e10 and float(es.min()) < -1e10       # masked items lead
    assert torch.equal(fi, ei)
    assert torch.equal(fs.view(torch.int32), es.view(torch.int32))


def test_filter_with_fewer_items_than_candidates_and_nan_scores():
    """Every item is a candidate (I < K + 8) and some exact scores are NaN or infinite."""
    g = torch.Generator().manual_seed(8)
    U, I, D, K = 130, 20, 64, 12
    ue = torch.randn(U, D, generator=g)
    ie = torch.randn(I, D, generator=g)
    ie[3, 5] = float("nan")
    ie[7, 0] = float("inf")
    ue[11, 2] = float("-inf")
    for mode_items in (ie, torch.nan_to_num(ie, nan=0.0, posinf=1.0)):
        fs, fi = ops.score_topk(ue.to(DEV), mode_items.to(DEV), K, mode="filter")
        es, ei = ops.score_topk(ue.to(DEV), mode_items.to(DEV), K, mode="exact")
        assert torch.equal(fi, ei)
        assert torch.equal(fs.view(torch.int32), es.view(torch.int32))


def test_item_offset_and_merge_equal_unsharded():
    """Two item shards ranked separately and merged give the unsharded ranking."""
    rng = np.random.default_rng(4)
    U, I, D, K = 500, 3001, 64, 20
    ue = (np.round(rng.standard_normal((U, D)) * 4) / 4).astype(np.float32)
    ie = (np.round(rng.standard_normal((I, D)) * 4) / 4).astype(np.float32)
    rows, cols = rng.integers(0, U, 5000), rng.integers(0, I, 5000)
    ptr, mcols = pairs_to_csr(rows, cols, U)
    full_s, full_i = ops.score_topk(dev(ue), dev(ie), K, dev(ptr), dev(mcols))
    from recommend_amd.evaluator import restrict_csr
    parts_s, parts_i = [], []
    for lo, hi in ((0, 1400), (1400, I)):
        p2, c2 = restrict_csr(ptr, mcols, lo, hi)
        s, i = ops.score_topk(dev(ue), dev(ie[lo:hi]), K, dev(p2), dev(c2), item_id_offset=lo)
        parts_s.append(s), parts_i.append(i)
    ms, mi = ops.topk_merge(torch.stack(parts_s, 1), torch.stack(parts_i, 1), K)
    assert torch.equal(mi, full_i) and torch.equal(ms, full_s)
    # padding entries (id < 0) never win
    pad_s = torch.full((U, 1, K), float("inf"), device=DEV)
    pad_i = torch.full((U, 1, K), -1, dtype=torch.int32, device=DEV)
    ms2, mi2 = ops.topk_merge(torch.cat([torch.stack(parts_s, 1), pad_s], 1),
                              torch.cat([torch.stack(parts_i, 1), pad_i], 1), K)
    assert torch.equal(mi2, full_i)


def test_rank_metrics_vs_reference_cases(golden):
    z = golden("metric_cases.npz")
    for name in sorted({k.split("/")[0] for k in z.files}):
        t = z[f"{name}/targets"]
        ks = [int(k) for k in z[f"{name}/ks"]]     # many_pos: k = 40 = every item
        kmax = max(ks)
        ids = z[f"{name}/topk@{kmax}"].astype(np.int32)
        nz = np.argwhere(t != 0)
        ptr, cols = pairs_to_csr(nz[:, 0], nz[:, 1], t.shape[0])
        out = ops.rank_metrics(dev(ids), dev(ptr), dev(cols), ks).cpu().numpy()
        for j, k in enumerate(ks):
            np.testing.assert_allclose(out[j], z[f"{name}/all@{k}"], rtol=1e-5, atol=1e-7)


def test_validation_errors_match_reference(golden):
    ue, ie = torch.randn(6, 64, device=DEV), torch.randn(9, 64, device=DEV)
    target = torch.zeros(6, 9)

    class C(Cfg):
        class evaluation:
            metrics = METRICS
            k_values = [5, 10]
    with pytest.raises(ValueError, match=r"k must be in the range \[1, num_items\]"):
        Verifier(C, None, target, ([], [])).verify(Stub(ue, ie))
    with pytest.raises(ValueError, match="same shape"):
        Verifier(Cfg, None, torch.zeros(6, 8), ([], [])).verify(Stub(ue, ie))


def test_embedding_width_not_multiple_of_64():
    rng = np.random.default_rng(8)
    U, I, D = 200, 500, 48
    ue, ie = rng.standard_normal((U, D)).astype(np.float32), rng.standard_normal((I, D)).astype(np.float32)
    ev = RankingEvaluator((np.array([0]), np.array([1])), ([0], [2]), [5], METRICS, n_users=U)
    _s, ids = ev.topk(dev(ue), dev(ie), 5)
    pad = lambda x: np.pad(x, ((0, 0), (0, 16)))
    ptr, cols = pairs_to_csr([0], [2], U)
    _os, o_i = eval_oracle.score_topk(pad(ue), pad(ie), 5, ptr, cols)
    np.testing.assert_array_equal(ids.cpu().numpy(), o_i)


def test_baby_shape_end_to_end_vs_oracle():
    """Full Baby shape (19,445 x 7,050): ids exact vs the oracle on a user sample,
    metrics within 1e-5 of the oracle's on all users, idempotent across runs."""
    d = synth.make_named("baby")
    rng = np.random.default_rng(0)
    D = 64
    ue = (rng.standard_normal((d.n_users, D)) * 0.1).astype(np.float32)
    ie = (rng.standard_normal((d.n_items, D)) * 0.1).astype(np.float32)
    tu, ti, _ = d.split(0)
    vu, vi, _ = d.split(1)
    ev = RankingEvaluator((vu, vi), (tu, ti), [5, 10, 20], METRICS, n_users=d.n_users)
    res = ev.evaluate(dev(ue), dev(ie))
    s, ids = ev.topk(dev(ue), dev(ie), 20)
    s2, ids2 = ev.topk(dev(ue), dev(ie), 20)
    assert torch.equal(ids, ids2) and torch.equal(s, s2)
    ids = ids.cpu().numpy()
    sample = np.sort(rng.choice(d.n_users, 1500, replace=False))
    for a in range(0, len(sample), 500):     # oracle in chunks of users
        us = sample[a:a + 500]
        ptr, cols = pairs_to_csr(tu, ti, d.n_users)
        sub_ptr = np.concatenate([[0], np.cumsum(ptr[us + 1] - ptr[us])]).astype(np.int32)
        sub_cols = np.concatenate([cols[ptr[u]:ptr[u + 1]] for u in us]).astype(np.int32)
        _os, o_i = eval_oracle.score_topk(ue[us], ie, 20, sub_ptr, sub_cols)
        np.testing.assert_array_equal(ids[us], o_i)
    # masked items never appear; ids are unique per user
    ptr, cols = pairs_to_csr(tu, ti, d.n_users)
    width = d.n_items
    train_keys = np.repeat(np.arange(d.n_users), np.diff(ptr)) * width + cols
    got_keys = (np.arange(d.n_users)[:, None] * width + ids).ravel()
    assert not np.isin(got_keys, train_keys).any()
    assert all(len(set(r)) == 20 for r in ids[:2000])
    tp, tc = pairs_to_csr(vu, vi, d.n_users)
    want = eval_oracle.metrics_from_topk_sparse(ids, tp, tc, [5, 10, 20])
    for k in (5, 10, 20):
        for m in METRICS:
            assert abs(float(res[f"{m}@{k}"]) - want[k][m]) <= 1e-5 * max(want[k][m], 1e-3), (m, k)


def test_dense_metric_functions_vs_reference_cases(golden):
    """evaluate_all_at_k and the single-metric functions on precomputed dense scores
    (recommend_amd.metric) against the values captured from evalue/metric.py."""
    from recommend_amd import metric
    z = golden("metric_cases.npz")
    err = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "metric_errors.json")))
    for name in sorted({k.split("/")[0] for k in z.files}):
        s, t = dev(z[f"{name}/scores"]), dev(z[f"{name}/targets"])
        for k in [int(k) for k in z[f"{name}/ks"]]:
            got = metric.evaluate_all_at_k(s, t, k)
            assert list(got.keys()) == ["precision", "recall", "hit_rate", "ndcg", "map", "mrr"]
            assert all(v.dim() == 0 for v in got.values())
            np.testing.assert_allclose([float(v) for v in got.values()], z[f"{name}/all@{k}"], rtol=1e-5, atol=1e-7)
            if name == "rand_64x500":   # continuous scores: no ties, the reference's ids are the only order
                assert np.array_equal(ops.rows_topk(s, k)[1].cpu().numpy(), z[f"{name}/topk@{k}"])
        k = int(z[f"{name}/ks"][0])
        assert abs(float(metric.ndcg_at_k(s, t, k)) - float(z[f"{name}/all@{k}"][3])) <= 1e-5
    s, t = torch.rand(4, 9, device=DEV), torch.zeros(4, 9, device=DEV)
    with pytest.raises(ValueError, match=re.escape(err["k0"])):
        metric.evaluate_all_at_k(s, t, 0)
    with pytest.raises(ValueError, match=re.escape(err["k_gt"])):
        metric.evaluate_all_at_k(s, t, 10)
    with pytest.raises(ValueError, match=re.escape(err["shape"])):
        metric.evaluate_all_at_k(s, t[:, :8], 3)
    with pytest.raises(ValueError, match=re.escape(err["dim"])):
        metric.evaluate_all_at_k(s[0], t[0], 3)


def test_rows_topk_ties_and_oracle():
    """Dense top-K: ids exact against a stable sort by (score desc, id asc), with heavy ties."""
    rng = np.random.default_rng(11)
    for (B, I, K) in ((7, 40, 5), (130, 1000, 20), (33, 77, 32), (5, 3, 3)):
        s = rng.integers(0, 6, size=(B, I)).astype(np.float32)      # many equal scores
        vals, ids = ops.rows_topk(dev(s), K)
        order = np.lexsort((np.broadcast_to(np.arange(I), (B, I)), -s), axis=1)[:, :K]
        assert np.array_equal(ids.cpu().numpy(), order.astype(np.int32))
        assert np.array_equal(vals.cpu().numpy(), np.take_along_axis(s, order, 1))


def test_sparse_pairs_from_the_data_path_equal_the_dense_target():
    """The (row, col) pairs of recommend_amd.data.mask_index give the same metrics as the dense
    [U, I] target matrix and python-list mask of main.mask_index built from the same splits."""
    from recommend_amd import data as rdata
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "data_path.npz"))
    U, I = int(z["rated/num_users"]), int(z["rated/num_items"])
    splits = [rdata.InteractionSplit(z[f"rated/{n}/users"], z[f"rated/{n}/items"], z[f"rated/{n}/ratings"], n)
              for n in ("train", "val", "test")]
    target_pairs, mask = rdata.mask_index(splits[2], splits[:2])
    dense = torch.zeros(U, I)
    dense[torch.from_numpy(target_pairs[0]), torch.from_numpy(target_pairs[1])] = 1
    g = torch.Generator().manual_seed(3)
    ue = torch.randn(U, 64, generator=g).to(DEV)
    ie = torch.randn(I, 64, generator=g).to(DEV)
    ks, names = [5, 10, 20], ["precision", "recall", "hit_rate", "ndcg", "map", "mrr"]
    a = RankingEvaluator(target_pairs, mask, ks, names, n_users=U).evaluate(ue, ie)
    b = RankingEvaluator(dense, (mask[0].tolist(), mask[1].tolist()), ks, names).evaluate(ue, ie)
    assert a.keys() == b.keys()
    for key in a:
        assert torch.equal(a[key], b[key]), key
    # and against the CPU oracle on the same embeddings
    ptr, cols = pairs_to_csr(mask[0], mask[1], U)
    _s, ids = eval_oracle.score_topk(ue.cpu().numpy(), ie.cpu().numpy(), 20, ptr, cols)
    tp, tc = pairs_to_csr(target_pairs[0], target_pairs[1], U)
    want = eval_oracle.metrics_from_topk_sparse(ids, tp, tc, ks)
    for k in ks:
        assert abs(float(a[f"recall@{k}"]) - want[k]["recall"]) <= 1e-6
        assert abs(float(a[f"ndcg@{k}"]) - want[k]["ndcg"]) <= 1e-6


@pytest.mark.parametrize("U,I,D,K", [(300, 1000, 64, 50), (129, 130, 128, 128), (70, 33, 64, 33), (1000, 5000, 64, 100),
                                     (130, 400, 64, 129), (90, 300, 128, 300),
                                      (65, 400, 64, 64)])
def test_rankings_longer_than_one_list_vs_oracle(U, I, D, K):
    """K > 32: pages of 32 places, each bounded above by the last entry of the page before.
    Quantised embeddings put long runs of equal scores across the page boundaries; one user has
    every item masked (its ranking is the mask score, by id), some have most.  Ids and scores equal
    the oracle's, up to K = the number of items (evalue/metric.py:9-10: any k in [1, num_items])."""
    rng = np.random.default_rng(U + K)
    ue = (np.round(rng.standard_normal((U, D)) * 2) / 2).astype(np.float32)
    ie = (np.round(rng.standard_normal((I, D)) * 2) / 2).astype(np.float32)
    ie[I // 2:I // 2 + 40] = ie[3]                       # 41 identical items: a tie longer than a page
    rows, cols = rng.integers(2, U, U * 30), rng.integers(0, I, U * 30)
    rows = np.concatenate([rows, np.zeros(I, np.int64), np.ones(I - 5, np.int64)])
    cols = np.concatenate([cols, np.arange(I), np.arange(I - 5)])
    ptr, mcols = pairs_to_csr(rows, cols, U)
    for mode in ("auto", "exact"):
        s, ids = ops.score_topk(dev(ue), dev(ie), K, dev(ptr), dev(mcols), mode=mode)
        o_s, o_i = eval_oracle.score_topk(ue, ie, K, ptr, mcols)
        np.testing.assert_array_equal(ids.cpu().numpy(), o_i)
        np.testing.assert_array_equal(s.cpu().numpy(), o_s)
    s2, ids2 = ops.score_topk(dev(ue), dev(ie), K, item_id_offset=1000)
    o_s2, o_i2 = eval_oracle.score_topk(ue, ie, K)
    np.testing.assert_array_equal(ids2.cpu().numpy(), o_i2 + 1000)
    np.testing.assert_array_equal(s2.cpu().numpy(), o_s2)
    with pytest.raises(Exception, match=r"range \[1, num_items\]"):
        ops.score_topk(dev(ue), dev(ie), I + 1)


def test_long_rankings_of_dense_scores_and_their_metrics_vs_oracle():
    """rows_topk and the metrics at k up to 128: ids against a stable sort with heavy ties, the six
    metrics against the oracle's per-user formulas (evalue/metric.py:106-164) on the same ranking."""
    rng = np.random.default_rng(12)
    for (B, I, K) in ((9, 200, 33), (130, 1000, 100), (40, 128, 128), (300, 700, 64), (70, 500, 129), (33, 260, 260)):
        s = rng.integers(0, 9, size=(B, I)).astype(np.float32)
        vals, ids = ops.rows_topk(dev(s), K)
        order = np.lexsort((np.broadcast_to(np.arange(I), (B, I)), -s), axis=1)[:, :K]
        assert np.array_equal(ids.cpu().numpy(), order.astype(np.int32))
        assert np.array_equal(vals.cpu().numpy(), np.take_along_axis(s, order, 1))
        # targets: a few per user, one user with hundreds (the bisection path), one with none
        tr = np.concatenate([np.repeat(np.arange(2, B), 3), np.zeros(I // 2, np.int64)])
        tc = np.concatenate([rng.integers(0, I, 3 * (B - 2)), rng.choice(I, I // 2, replace=False)])
        tp, tcols = pairs_to_csr(tr, tc, B)
        ks = sorted({5, 20, 33, K})
        got = ops.rank_metrics(ids, dev(tp), dev(tcols), ks).cpu().numpy()
        if K > 128:
            # rankings beyond 128 places are walked from memory: for the k that both kernels serve, the same bits
            short = [k for k in ks if k <= 128]
            again = ops.rank_metrics(ids, dev(tp), dev(tcols), short).cpu().numpy()
            assert np.array_equal(again, got[:len(short)])
        want = eval_oracle.metrics_from_topk_sparse(order.astype(np.int32), tp, tcols, ks)
        for j, k in enumerate(ks):
            for m, name in enumerate(METRICS):
                assert abs(got[j][m] - want[k][name]) <= 1e-5 * max(want[k][name], 1e-3), (B, I, K, k, name)


@pytest.mark.parametrize("KL", [50, 200])
def test_evaluator_with_long_rankings_vs_oracle(KL):
    """k beyond one list (50) and beyond the ranked ids a metrics lane holds in registers (200)."""
    rng = np.random.default_rng(4)
    U, I, D = 900, 2500, 64
    ue = (rng.standard_normal((U, D)) * 0.1).astype(np.float32)
    ie = (rng.standard_normal((I, D)) * 0.1).astype(np.float32)
    tu, ti = rng.integers(0, U, 20000), rng.integers(0, I, 20000)
    vu, vi = rng.integers(0, U, 4000), rng.integers(0, I, 4000)
    ev = RankingEvaluator((vu, vi), (tu, ti), [10, KL], METRICS, n_users=U)
    res = ev.evaluate(dev(ue), dev(ie))
    ptr, cols = pairs_to_csr(tu, ti, U)
    _os, o_i = eval_oracle.score_topk(ue, ie, KL, ptr, cols)
    np.testing.assert_array_equal(ev.topk(dev(ue), dev(ie), KL)[1].cpu().numpy(), o_i)
    tp, tc = pairs_to_csr(vu, vi, U)
    want = eval_oracle.metrics_from_topk_sparse(o_i, tp, tc, [10, KL])
    for k in (10, KL):
        for m in METRICS:
            assert abs(float(res[f"{m}@{k}"]) - want[k][m]) <= 1e-5 * max(want[k][m], 1e-3), (m, k)


def _open_users(ue, ie, K, ptr, cols):
    return ops.score_topk_open_users(ue, ie, K, ptr, cols)


def test_filter_certifies_popularity_skewed_embeddings():
    """What an evaluator sees after training is not Gaussian: item norms follow popularity (here
    spread 10x along a Zipf curve), user vectors lean towards the popular direction, and the best
    scores of a user crowd together.  The certificate's error bound grows with the largest item
    norm, so this is the input on which a weak filter would show: the filtered ranking must still
    be the exact kernel's bit for bit AND leave few users to the exact pass."""
    U, I, D, K = 20000, 7000, 64, 20
    g = torch.Generator().manual_seed(77)
    pop = 1.0 / torch.arange(1, I + 1, dtype=torch.float32) ** 0.5          # Zipf-like popularity
    norms = 0.1 * (1.0 + 9.0 * (pop - pop.min()) / (pop.max() - pop.min()))  # 0.1 .. 1.0: a 10x spread
    direction = torch.nn.functional.normalize(torch.randn(D, generator=g), dim=0)
    ie = torch.nn.functional.normalize(torch.randn(I, D, generator=g) + 1.5 * direction, dim=1) * norms[:, None]
    ie = ie[torch.randperm(I, generator=g)]
    ue = torch.randn(U, D, generator=g) * 0.05 + direction * (0.3 * torch.rand(U, 1, generator=g))
    ue, ie = ue.to(DEV), ie.to(DEV)
    rng = np.random.default_rng(7)
    ptr, mcols = pairs_to_csr(rng.integers(0, U, 8 * U), rng.integers(0, I, 8 * U), U)
    mp, mc = dev(ptr), dev(mcols)
    n_open = _open_users(ue, ie, K, mp, mc)
    assert 0 <= n_open <= 1024, n_open
    fs, fi = ops.score_topk(ue, ie, K, mp, mc, mode="filter")
    es, ei = ops.score_topk(ue, ie, K, mp, mc, mode="exact")
    assert torch.equal(fi, ei) and torch.equal(fs.view(torch.int32), es.view(torch.int32))


def test_filter_certifies_the_embeddings_of_a_trained_model():
    """The MMFCN after 200 training steps at the Baby shape (graphed step, dropout on, the
    preset's batch size): evaluation of ITS embeddings through the filter equals the exact
    kernel's bits and the filter settles all but a few users."""
    from recommend_amd.loss import gather_dot_bpr_loss
    from recommend_amd.model import ModelFactory
    from recommend_amd.optim import FusedClipAdam
    from recommend_amd.sampler import NegativeSampler, TrainBatchLoader, build_user_csr
    from recommend_amd.train_step import GraphedTrainStep
    d = synth.make_named("baby")
    U, I = d.n_users, d.n_items

    class C:
        class data:
            num_users, num_items = U, I
        class model:
            model_name = "MMFCN"
            modal_num, emb_dim, layer_num, dropout = 2, 64, 2, 0.3
        class graph:
            weight_feature = [None]
            edge_weight_type, add_self_loops, normalize_adj = "uniform", True, True
        class system:
            device = DEV

    g = torch.Generator().manual_seed(3)
    feats = {"image": torch.randn(I, 512, generator=g), "text": torch.nn.functional.normalize(torch.randn(I, 96, generator=g))}
    torch.manual_seed(3)
    model = ModelFactory.create_Model(C, {}, feats).to(DEV)
    tu, ti, tr = d.split(0)
    model.build_graph((tu, ti, tr))
    model.creat_feature_weight()
    p, it = build_user_csr(d.users, d.items, U)
    sampler = NegativeSampler(p, it, I, seed=42)
    dp, di = sampler.device_csr(DEV)
    loader = TrainBatchLoader(tu, ti, tr, 2048, sampler, device=DEV, resolve_on_host=False)
    opt = FusedClipAdam(model.parameters(), lr=1e-3)
    model.train()
    try:
        step, done, first_loss = None, 0, None
        while done < 200:
            for batch in loader:
                if step is None:
                    step = GraphedTrainStep(model, opt, gather_dot_bpr_loss, batch, constants={"pos_ptr": dp, "pos_items": di})
                    done += 3
                loss = step(batch)
                first_loss = float(loss) if first_loss is None else first_loss
                done += 1
                if done >= 200:
                    break
        assert float(loss) < first_loss                       # it did train
    finally:
        ops.set_dropout_step_tensor(None)
    model.eval()
    with torch.no_grad():
        out = model({})
    ue, ie = out["user_embeddings"].contiguous(), out["item_embeddings"].contiguous()
    ptr, mcols = pairs_to_csr(tu, ti, U)
    mp, mc = dev(ptr), dev(mcols)
    n_open = _open_users(ue, ie, 20, mp, mc)
    assert 0 <= n_open <= 1024, n_open
    fs, fi = ops.score_topk(ue, ie, 20, mp, mc, mode="filter")
    es, ei = ops.score_topk(ue, ie, 20, mp, mc, mode="exact")
    assert torch.equal(fi, ei) and torch.equal(fs.view(torch.int32), es.view(torch.int32))
