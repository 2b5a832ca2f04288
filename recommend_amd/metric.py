"""Ranking metrics on precomputed scores with the reference's function names and
signatures (evalue/metric.py): scores and targets are dense [batch, num_items] tensors on
the GPU; the top-K and the reductions run in librecommend_amd (rk_rows_topk_f32,
rk_rank_metrics_f32).  Ties in the scores rank the lower item id first (the reference's
torch.topk leaves that order to the implementation).  Targets are read as relevant / not
relevant (non-zero / zero), which is what the reference's callers pass.
"""
import torch

from . import ops

_ORDER = ("precision", "recall", "hit_rate", "ndcg", "map", "mrr")


def _validate_inputs(scores: torch.Tensor, targets: torch.Tensor, k: int) -> None:
    # same checks and messages as evalue/metric.py:4-10
    if scores.dim() != 2 or targets.dim() != 2:
        raise ValueError("scores and targets must be 2D tensors: (batch_size, num_items)")
    if scores.shape != targets.shape:
        raise ValueError("scores and targets must have the same shape")
    if k <= 0 or k > scores.shape[1]:
        raise ValueError("k must be in the range [1, num_items]")


def _target_csr(targets: torch.Tensor):
    nz = targets != 0
    counts = nz.sum(dim=1)
    ptr = torch.zeros(targets.shape[0] + 1, dtype=torch.int32, device=targets.device)
    ptr[1:] = torch.cumsum(counts, 0).to(torch.int32)
    items = nz.nonzero()[:, 1].to(torch.int32)      # row-major: ascending item ids per row
    return ptr, items


def evaluate_all_at_k(scores: torch.Tensor, targets: torch.Tensor, k: int) -> dict:
    """evalue/metric.py:106-164: the six metrics at one k as 0-dim tensors."""
    _validate_inputs(scores, targets, k)
    _vals, ids = ops.rows_topk(scores, k)
    ptr, items = _target_csr(targets.to(scores.device))
    table = ops.rank_metrics(ids, ptr, items, [k])
    return {name: table[0, j] for j, name in enumerate(_ORDER)}


def _single(name):
    def fn(scores: torch.Tensor, targets: torch.Tensor, k: int) -> torch.Tensor:
        return evaluate_all_at_k(scores, targets, k)[name]
    fn.__name__ = f"{name}_at_k"
    fn.__doc__ = f"{name}@k as a 0-dim tensor (evalue/metric.py, function {name}_at_k)."
    return fn


precision_at_k = _single("precision")
recall_at_k = _single("recall")
hit_rate_at_k = _single("hit_rate")
ndcg_at_k = _single("ndcg")
map_at_k = _single("map")
mrr_at_k = _single("mrr")

__all__ = ["evaluate_all_at_k", "precision_at_k", "recall_at_k", "hit_rate_at_k", "ndcg_at_k", "map_at_k",
           "mrr_at_k"]
