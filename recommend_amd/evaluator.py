"""Full-ranking evaluation with the reference's Verifier / Tester interface
(evalue/evaluator.py:20-108): same constructor, same `verify(model)` /
`test(model)` calls, same result dict `{f"{metric}@{k}": 0-dim tensor}`.

The reference materialises the user x item score matrix on the CPU, overwrites
the training pairs with -1e10, runs torch.topk once per k and computes the
metrics from a dense target matrix.  Here one fused HIP kernel ranks the top
max(k) items per user straight from the embeddings and a second one reduces all
metrics for every k; the dense `target` and the python-list `mask` of the
reference are converted to CSR once at construction.
"""
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _ffi, ops

METRIC_ORDER = ("precision", "recall", "hit_rate", "ndcg", "map", "mrr")


def pairs_to_csr(rows, cols, n_rows):
    """Sorted, de-duplicated (row, col) pairs -> (ptr int32 [n_rows+1], cols int32)."""
    rows = np.asarray(rows, dtype=np.int64).reshape(-1)
    cols = np.asarray(cols, dtype=np.int64).reshape(-1)
    if rows.shape[0] == 0:
        return np.zeros(n_rows + 1, np.int32), np.zeros(0, np.int32)
    width = int(cols.max()) + 1
    key = np.unique(rows * width + cols)
    r, c = key // width, key % width
    ptr = np.zeros(n_rows + 1, np.int64)
    ptr[1:] = np.bincount(r, minlength=n_rows)
    return np.cumsum(ptr).astype(np.int32), c.astype(np.int32)


def target_to_csr(target, n_users):
    """Accepts the reference's dense [U, I] 0/1 tensor (main.py:124-130), a sparse
    COO tensor, or a (row_ids, col_ids) pair; returns CSR numpy arrays."""
    if isinstance(target, torch.Tensor):
        if target.is_sparse:
            idx = target.coalesce().indices().cpu().numpy()
            return pairs_to_csr(idx[0], idx[1], n_users)
        nz = torch.nonzero(target, as_tuple=False).cpu().numpy()
        return pairs_to_csr(nz[:, 0], nz[:, 1], n_users)
    rows, cols = target
    return pairs_to_csr(rows, cols, n_users)


def restrict_csr(ptr, cols, lo, hi):
    """Entries with lo <= col < hi, re-based to lo (per-row order kept)."""
    n = ptr.shape[0] - 1
    rows = np.repeat(np.arange(n), np.diff(ptr))
    keep = (cols >= lo) & (cols < hi)
    counts = np.bincount(rows[keep], minlength=n)
    new_ptr = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
    return new_ptr, (cols[keep] - lo).astype(np.int32)


class RankingEvaluator:
    """Device-side state and the evaluation step shared by Verifier and Tester."""

    def __init__(self, target, mask, k_list: Sequence[int], metrics: Sequence[str], n_users=None):
        self.k_list = [int(k) for k in k_list]
        self.metrics = list(metrics)
        for m in self.metrics:
            if m not in METRIC_ORDER:
                raise KeyError(m)
        row_id, col_id = mask
        if n_users is None:
            if isinstance(target, torch.Tensor):
                n_users = target.shape[0]
            else:
                raise ValueError("n_users is required with a sparse target")
        self.n_users = int(n_users)
        self.n_items = int(target.shape[1]) if isinstance(target, torch.Tensor) else None
        self.tgt_ptr, self.tgt_items = target_to_csr(target, self.n_users)
        self.mask_ptr, self.mask_items = pairs_to_csr(row_id, col_id, self.n_users)
        self._dev = {}

    def _device_state(self, device):
        device = torch.device(device)
        st = self._dev.get(device)
        if st is None:
            to = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
            st = {"tgt_ptr": to(self.tgt_ptr), "tgt_items": to(self.tgt_items),
                  "mask_ptr": to(self.mask_ptr), "mask_items": to(self.mask_items)}
            self._dev[device] = st
        return st

    def mask_csr(self, device):
        """(ptr, items) of the training-pair mask on `device`, or (None, None) without a mask."""
        st = self._device_state(device)
        return (st["mask_ptr"], st["mask_items"]) if st["mask_items"].numel() > 0 else (None, None)

    def topk(self, user_embeddings, item_embeddings, k, mode="auto"):
        """Ranked top-k item ids per user with the training pairs masked (mode: ops.score_topk)."""
        st = self._device_state(user_embeddings.device)
        U, I = _pad64(user_embeddings), _pad64(item_embeddings)
        has_mask = st["mask_items"].numel() > 0
        return ops.score_topk(U, I, k, st["mask_ptr"] if has_mask else None,
                              st["mask_items"] if has_mask else None, mode=mode)

    def evaluate(self, user_embeddings, item_embeddings) -> Dict[str, torch.Tensor]:
        n_items = item_embeddings.shape[0]
        if user_embeddings.shape[0] != self.n_users or (self.n_items not in (None, n_items)):
            # evalue/metric.py:7-8
            raise ValueError("scores and targets must have the same shape")
        for k in self.k_list:
            if k <= 0 or k > n_items:
                raise ValueError("k must be in the range [1, num_items]")
        st = self._device_state(user_embeddings.device)
        _scores, ids = self.topk(user_embeddings, item_embeddings, max(self.k_list))
        # The metrics stay on the embeddings' device, as evaluate_all_at_k's do (evalue/metric.py:
        # 157-164: means of device tensors): nothing here waits for the GPU, so a caller that
        # evaluates again right away has its launches queued behind this call's kernels; reading a
        # value (float(), .item(), .cpu()) is what synchronises.
        table = ops.rank_metrics(ids, st["tgt_ptr"], st["tgt_items"], self.k_list)
        out = {}
        for ki, k in enumerate(self.k_list):
            row = table[ki]
            for name in self.metrics:
                out[f"{name}@{k}"] = row[METRIC_ORDER.index(name)]
        return out


def _pad64(x):
    """Zero-pad the embedding width to a multiple of 64 (adds exact zeros to every score)."""
    x = x.detach()
    d = x.shape[1]
    if d % 64 == 0:
        return x
    return torch.nn.functional.pad(x, (0, 64 - d % 64))


class Verifier:
    """verifier for graph-based recommendation models."""

    def __init__(self, config, loader, target, mask):
        self.config = config
        self.loader = loader
        self.target = target
        self.row_id, self.col_id = mask

        self.metrics = config.evaluation.metrics
        self.device = config.system.device
        self.k_list = config.evaluation.k_values
        n_users = getattr(config.data, "num_users", -1)
        self._eval = RankingEvaluator(target, mask, self.k_list, self.metrics,
                                      n_users if n_users and n_users > 0 else None)

    def _run(self, model):
        model.eval()
        outputs = model({})
        return self._eval.evaluate(outputs["user_embeddings"], outputs["item_embeddings"])

    def verify(self, model):
        return self._run(model)


class Tester(Verifier):
    """tester for graph-based recommendation models."""

    def test(self, model):
        return self._run(model)


__all__ = ["Verifier", "Tester", "RankingEvaluator", "pairs_to_csr", "target_to_csr",
           "restrict_csr", "METRIC_ORDER"]
