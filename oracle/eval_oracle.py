"""Oracle: full-ranking evaluation (torch CPU + C), restating
evalue/evaluator.py:36-63 and evalue/metric.py:106-164 of the reference.
Pinned by tests/golden/metric_*.npz / verifier_*.npz captured from the
reference's own evalue modules."""
import ctypes
import os

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def _lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            raise RuntimeError("oracle/liboracle.so missing: run `make -C oracle`")
        _LIB = ctypes.CDLL(path)
        _LIB.oracle_score_one.restype = ctypes.c_float
    return _LIB


def _p(a):
    return ctypes.c_void_p(a.ctypes.data) if a is not None else None


# ---------------------------------------------------------------- metric.py
def evaluate_all_at_k(scores, targets, k):
    """evalue/metric.py:106-164, statement for statement."""
    if scores.dim() != 2 or targets.dim() != 2:
        raise ValueError("scores and targets must be 2D tensors: (batch_size, num_items)")
    if scores.shape != targets.shape:
        raise ValueError("scores and targets must have the same shape")
    if k <= 0 or k > scores.shape[1]:
        raise ValueError("k must be in the range [1, num_items]")
    topk = torch.topk(scores, k, dim=1, largest=True, sorted=True).indices
    return metrics_from_topk(topk, targets, k)


def metrics_from_topk(topk, targets, k):
    """metric.py:115-164 given the ranked ids (dense 0/1 targets)."""
    n = topk.size(0)
    batch_idx = torch.arange(n).unsqueeze(1).expand_as(topk)
    rel = targets[batch_idx, topk].float()
    hits = rel.sum(dim=1)
    precision = (hits / float(k)).mean()
    positives = targets.sum(dim=1).clamp_min(1).float()
    recall = (hits / positives).mean()
    hit_rate = (hits > 0).float().mean()
    positions = torch.arange(1, k + 1).float()
    discounts = 1.0 / torch.log2(positions + 1.0)
    dcg = (rel * discounts).sum(dim=1)
    ideal_k = torch.minimum(targets.sum(dim=1), torch.tensor(k)).long()
    ideal_rel = torch.zeros_like(rel)
    for i in range(n):                                     # :135-137
        if ideal_k[i] > 0:
            ideal_rel[i, : ideal_k[i]] = 1.0
    idcg = (ideal_rel * discounts).sum(dim=1)
    ndcg = torch.where(idcg > 0, dcg / idcg, torch.zeros_like(dcg)).mean()
    cum_rels = rel.cumsum(dim=1)
    prec_at_pos = cum_rels / positions
    ap = (prec_at_pos * rel).sum(dim=1) / positives
    mapk = ap.mean()
    rel_bool = rel.bool()
    no_hit = (~rel_bool).all(dim=1)
    first_pos = torch.full((n,), k + 1, dtype=torch.float32)
    has_hit_idx = (~no_hit).nonzero(as_tuple=False).flatten()
    if has_hit_idx.numel() > 0:
        first_hit_pos = rel_bool[has_hit_idx].float().argmax(dim=1) + 1
        first_pos[has_hit_idx] = first_hit_pos.float()
    mrr = torch.where(first_pos <= k, 1.0 / first_pos, torch.zeros_like(first_pos)).mean()
    return {"precision": precision, "recall": recall, "hit_rate": hit_rate, "ndcg": ndcg,
            "map": mapk, "mrr": mrr}


def metrics_from_topk_sparse(topk_ids, tgt_ptr, tgt_items, k_list, n_users_total=None):
    """The same six metrics from ranked ids and a CSR target, vectorised in numpy
    (float64 accumulation of per-user fp32 values); for sizes where a dense target
    does not fit.  Returns {k: {metric: float}}."""
    ids = np.asarray(topk_ids)
    U = ids.shape[0]
    n_pos = np.diff(tgt_ptr).astype(np.int64)
    width = int(max(int(ids.max()), int(tgt_items.max()) if tgt_items.size else 0)) + 2
    rows = np.repeat(np.arange(U, dtype=np.int64), n_pos)
    tkeys = rows * width + tgt_items.astype(np.int64)
    kmax = ids.shape[1]
    ckeys = (np.arange(U, dtype=np.int64)[:, None] * width + np.where(ids >= 0, ids, width - 1)).ravel()
    rel = np.isin(ckeys, tkeys).reshape(U, kmax).astype(np.float32)
    denom = float(n_users_total or U)
    out = {}
    for k in k_list:
        r = rel[:, :k]
        disc = (1.0 / torch.log2(torch.arange(1, k + 1).float() + 1.0)).numpy()
        hits = r.sum(1, dtype=np.float32)
        positives = np.maximum(n_pos, 1).astype(np.float32)
        dcg = (r * disc).sum(1, dtype=np.float32)
        cdisc = np.concatenate([[0.0], np.cumsum(disc, dtype=np.float32)]).astype(np.float32)
        idcg = cdisc[np.minimum(n_pos, k)]
        ndcg = np.where(idcg > 0, dcg / np.where(idcg > 0, idcg, 1), 0).astype(np.float32)
        cum = np.cumsum(r, axis=1, dtype=np.float32)
        ap = ((cum / np.arange(1, k + 1, dtype=np.float32)) * r).sum(1, dtype=np.float32) / positives
        first = np.where(r.any(1), r.argmax(1) + 1, 0)
        mrr = np.where(first > 0, 1.0 / np.maximum(first, 1), 0).astype(np.float32)
        out[k] = {
            "precision": float((hits / np.float32(k)).sum(dtype=np.float64) / denom),
            "recall": float((hits / positives).sum(dtype=np.float64) / denom),
            "hit_rate": float((hits > 0).sum(dtype=np.float64) / denom),
            "ndcg": float(ndcg.sum(dtype=np.float64) / denom),
            "map": float(ap.sum(dtype=np.float64) / denom),
            "mrr": float(mrr.sum(dtype=np.float64) / denom),
        }
    return out


# ------------------------------------------------------------- evaluator.py
def verify(user_embeddings, item_embeddings, target, row_id, col_id, k_list, metrics):
    """Verifier.verify / Tester.test body (evaluator.py:51-63) on CPU tensors."""
    pre_score = torch.matmul(user_embeddings, item_embeddings.T).to("cpu")
    pre_score[row_id, col_id] = -1e10
    results = {}
    for k in k_list:
        all_metric = evaluate_all_at_k(pre_score, target, k)
        for name in metrics:
            results[f"{name}@{k}"] = all_metric[name]
    return results


# ------------------------------------------------- chain-ordered exact top-K
def scores_chain(U, I):
    """Dense fp32 scores accumulated in the order the evaluation kernel documents."""
    U = np.ascontiguousarray(U, np.float32)
    I = np.ascontiguousarray(I, np.float32)
    assert U.shape[1] % 64 == 0 and U.shape[1] == I.shape[1]
    S = np.empty((U.shape[0], I.shape[0]), np.float32)
    _lib().oracle_scores(_p(U), U.shape[0], _p(I), I.shape[0], U.shape[1], _p(S))
    return S


def score_topk(U, I, K, mask_ptr=None, mask_items=None, id_offset=0, u_lo=0, u_hi=None):
    """Masked top-K per user by (score desc, id asc) on chain-ordered scores."""
    U = np.ascontiguousarray(U, np.float32)
    I = np.ascontiguousarray(I, np.float32)
    assert U.shape[1] % 64 == 0 and U.shape[1] == I.shape[1]
    u_hi = U.shape[0] if u_hi is None else u_hi
    out_s = np.empty((u_hi - u_lo, K), np.float32)
    out_i = np.empty((u_hi - u_lo, K), np.int32)
    mp = None if mask_ptr is None else np.ascontiguousarray(mask_ptr, np.int32)
    mi = None if mask_items is None else np.ascontiguousarray(mask_items, np.int32)
    _lib().oracle_score_topk(_p(U), u_lo, u_hi, _p(I), I.shape[0], U.shape[1], _p(mp), _p(mi),
                             K, id_offset, _p(out_s), _p(out_i))
    return out_s, out_i


def topk_by_sort(S, K):
    """(score desc, id asc) top-K of a dense matrix via a stable sort (independent check)."""
    order = np.argsort(-S, axis=1, kind="stable")[:, :K]
    return np.take_along_axis(S, order, 1), order.astype(np.int32)
