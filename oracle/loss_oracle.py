"""Oracle: BPR gather-dot loss (torch CPU), restating evalue/loss.py:7-28 and
GraphTrainer._loss_func, train/graph_trainer.py:255-270.  Pinned by
tests/golden/loss_*.npz captured from the reference's evalue.loss."""
import torch
import torch.nn.functional as F


def bpr_loss(pos_scores, neg_scores, mask=None):
    pos = pos_scores.unsqueeze(-1) if pos_scores.dim() == 1 else pos_scores   # loss.py:21
    x = pos - neg_scores
    loss = -F.logsigmoid(x)
    if mask is not None:
        loss = loss * mask.to(loss.dtype)
        denom = mask.to(loss.dtype).sum().clamp_min(1.0)
        return loss.sum() / denom
    return loss.mean()


def trainer_loss(outputs, batch):
    """graph_trainer.py:255-270."""
    user_embeddings = outputs.get("user_embeddings")
    item_embeddings = outputs.get("item_embeddings")
    user_ids = batch.get("user_ids", torch.tensor([], dtype=torch.long))
    item_ids = batch.get("item_ids", torch.tensor([], dtype=torch.long))
    neg_items_ids = batch.get("neg_items", torch.tensor([], dtype=torch.long))
    if not (len(user_ids) > 0 and len(item_ids) > 0 and len(neg_items_ids) > 0):
        raise Exception("batch error, recive null user or item or neg_items")
    pos_user_emb = user_embeddings[user_ids]
    pos_item_emb = item_embeddings[item_ids]
    neg_item_emb = item_embeddings[neg_items_ids]
    pos_scores = torch.sum(pos_user_emb * pos_item_emb, dim=-1)
    neg_scores = torch.sum(pos_user_emb.unsqueeze(1) * neg_item_emb, dim=-1)
    return bpr_loss(pos_scores, neg_scores)
