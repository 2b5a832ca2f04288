"""Loss functions with the reference's names and signatures
(evalue/loss.py and GraphTrainer._loss_func, train/graph_trainer.py:255-270),
computed by the BPR kernels of librecommend_amd.
"""
from typing import Optional

import torch

from . import _ffi, ops
from ._ffi import c_int32, c_int64, c_size_t, ptr, stream_ptr


class _BprScores(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pos, neg, mask):
        ops._need_cuda(pos, neg, mask)
        ctx.in_shapes = (pos.shape, neg.shape)
        pos = ops._f32c(pos).reshape(-1)
        neg = ops._f32c(neg)
        B = pos.shape[0]
        neg2 = neg.reshape(B, -1)
        R = neg2.shape[1]
        m = ops._f32c(mask).reshape(B, R) if mask is not None else None
        ws = torch.empty(_ffi.load().rk_bpr_ws_bytes(B, R) // 4 + 4 + 2 * ((B + 255) // 256),
                         dtype=torch.float32, device=pos.device)
        x = torch.empty((B, R), dtype=torch.float32, device=pos.device)
        out = torch.empty(2, dtype=torch.float32, device=pos.device)
        _ffi.call("rk_bpr_scores_fwd_f32", ptr(pos), ptr(neg2), c_int64(B), c_int32(R), ptr(m),
                  ptr(x), ptr(out), ptr(ws), c_size_t(ws.numel() * 4), stream_ptr(pos.device))
        ctx.save_for_backward(x, m, out)
        ctx.shapes = (B, R)
        return out[0]

    @staticmethod
    def backward(ctx, g):
        x, m, out = ctx.saved_tensors
        B, R = ctx.shapes
        g = ops._f32c(g).reshape(1)
        dpos = torch.empty(B, dtype=torch.float32, device=x.device)
        dneg = torch.empty((B, R), dtype=torch.float32, device=x.device)
        _ffi.call("rk_bpr_scores_bwd_f32", ptr(x), ptr(m), ptr(out), ptr(g), c_int64(B),
                  c_int32(R), ptr(dpos), ptr(dneg), stream_ptr(x.device))
        return dpos.reshape(ctx.in_shapes[0]), dneg.reshape(ctx.in_shapes[1]), None


def bpr_loss(pos_scores: torch.Tensor, neg_scores: torch.Tensor,
             mask: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Bayesian Personalized Ranking loss on precomputed scores (evalue/loss.py:7-28).

    pos_scores: (B,) or (B, 1); neg_scores: (B, N); mask: optional, same shape as
    neg_scores.  Returns the scalar mean over batch and negatives."""
    return _BprScores.apply(pos_scores, neg_scores, mask)


def gather_dot_bpr_loss(outputs, batch):
    """GraphTrainer._loss_func (train/graph_trainer.py:255-270) as one fused kernel.

    outputs: dict with `user_embeddings` [U, D] and `item_embeddings` [I, D];
    batch: dict with `user_ids`, `item_ids` and either `neg_items` [B, R] or, for
    in-kernel negative lookup, `neg_rank` [B, R] plus `pos_ptr` / `pos_items`."""
    user_embeddings = outputs.get("user_embeddings")
    item_embeddings = outputs.get("item_embeddings")
    dev = user_embeddings.device
    empty = torch.tensor([], dtype=torch.long)
    user_ids = batch.get("user_ids", empty)
    item_ids = batch.get("item_ids", empty)
    neg_items_ids = batch.get("neg_items", None)
    neg_rank = batch.get("neg_rank", None) if "pos_ptr" in batch else None
    n_neg = len(neg_items_ids) if neg_items_ids is not None else (len(neg_rank) if neg_rank is not None else 0)
    if not (len(user_ids) > 0 and len(item_ids) > 0 and n_neg > 0):
        raise Exception("batch error, recive null user or item or neg_items")
    if neg_rank is not None and neg_items_ids is None:
        loss, neg = ops.bpr_gather_dot_sampled(
            user_embeddings, item_embeddings, user_ids.to(dev), item_ids.to(dev),
            neg_rank.to(dev), batch["pos_ptr"], batch["pos_items"])
        batch["neg_items"] = neg
        return loss
    return ops.bpr_gather_dot(user_embeddings, item_embeddings, user_ids.to(dev),
                              item_ids.to(dev), neg_items_ids.to(dev))


def mmgcn_loss(outputs, batch):
    """mmgcn_loss (evalue/loss.py:153-173, = model/Main_Model.py:672-692): one negative per
    sample, -mean(log(sigmoid(pos - neg))) in the reference's two-step form, plus the
    regulariser that the reference multiplies by zero (kept: it propagates NaN/inf like there)."""
    empty = torch.tensor([], dtype=torch.long)
    batch_users = batch.get("user_ids", empty)
    pos_items = batch.get("item_ids", empty)
    neg_items = batch.get("neg_items", empty).reshape(-1)
    user_h = outputs["user_embeddings"]
    item_h = outputs["item_embeddings"]
    id_embedding = outputs["id_embeddings"]
    dev = user_h.device
    batch_users, pos_items, neg_items = batch_users.to(dev), pos_items.to(dev), neg_items.to(dev)
    loss = ops.bpr_gather_dot(user_h, item_h, batch_users, pos_items, neg_items.reshape(-1, 1),
                              mode=ops.LOSS_LOG_SIGMOID)
    # The reference adds 0 * (mean of squared id-embedding rows of the batch + mean(pres^2)): a term whose
    # only effect is to turn the loss into NaN when one of those values is not finite (or its square
    # overflows).  Its value and gradient are zero otherwise, so no gradient graph is built for it,
    # and the squares of the batch's rows are reduced without materialising the interleaved index
    # tensors.
    with torch.no_grad():
        idd = id_embedding.detach()
        reg = (idd[batch_users] ** 2).mean() + (idd[pos_items] ** 2).mean() + (idd[neg_items] ** 2).mean()
        reg = reg + (outputs["pres"].detach() ** 2).mean()
    return loss + 0 * reg


def compute_info_bpr_loss(a_embeddings, b_embeddings, pos_edges, neg_items, reduction="mean",
                          hard_negs=None):
    """compute_info_bpr_loss (evalue/loss.py:95-117): softmax cross-entropy over
    [<a, b_pos>, <a, b_neg_1>, ...] with the positive as the target class."""
    if reduction != "mean":
        raise ValueError("only reduction='mean' is built")
    if isinstance(pos_edges, list):
        pos_edges = torch.as_tensor(pos_edges)
    dev = a_embeddings.device
    pos_edges = torch.as_tensor(pos_edges).to(dev)
    neg_items = torch.as_tensor(neg_items).to(dev)
    return ops.bpr_gather_dot(a_embeddings, b_embeddings, pos_edges[:, 0], pos_edges[:, 1],
                              neg_items.reshape(pos_edges.shape[0], -1), mode=ops.LOSS_INFO_BPR)


def sgrec_loss(result, batch, reg_weight=0.0):
    """The SGrec model loss (model/Main_Model.py:902-928):
    -mean(log(sigmoid(pos - neg))) + reg_weight * (mean(ori_u^2) + mean(ori_i^2)), one negative
    per sample.  (The reference also builds a 20-random-item cross-entropy term that it never
    adds to the result; that term only consumes random numbers and is not reproduced.)"""
    dev = result["user_embeddings"].device
    empty = torch.tensor([], dtype=torch.long)
    users = batch.get("user_ids", empty).to(dev)
    pos = batch.get("item_ids", empty).to(dev)
    neg = batch.get("neg_items", empty).to(dev).reshape(-1, 1)
    loss = ops.bpr_gather_dot(result["user_embeddings"], result["item_embeddings"], users, pos, neg,
                              mode=ops.LOSS_LOG_SIGMOID)
    reg_loss = torch.mean(result["ori_u_emb"] ** 2) + torch.mean(result["ori_i_emb"] ** 2)
    return loss + reg_weight * reg_loss


__all__ = ["bpr_loss", "gather_dot_bpr_loss", "mmgcn_loss", "compute_info_bpr_loss", "sgrec_loss"]
