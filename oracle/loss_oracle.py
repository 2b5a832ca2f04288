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


def mmgcn_loss(outputs, batch):
    """evalue/loss.py:153-173 (the zero-weighted regulariser included)."""
    batch_users = batch["user_ids"]
    pos_items = batch["item_ids"]
    neg_items = batch["neg_items"].reshape(-1)
    user_tensor = batch_users.repeat_interleave(2)
    item_tensor = torch.stack((pos_items, neg_items)).t().contiguous().view(-1)
    user_emb = outputs["user_embeddings"][user_tensor]
    item_emb = outputs["item_embeddings"][item_tensor]
    score = torch.sum(user_emb * item_emb, dim=1).view((-1, 2))
    loss = -torch.mean(torch.log(torch.sigmoid(torch.matmul(score, torch.tensor([[1.0], [-1.0]])))))
    idemb = outputs["id_embeddings"]
    reg = (idemb[user_tensor] ** 2 + idemb[item_tensor] ** 2).mean() + (outputs["pres"] ** 2).mean()
    return loss + 0 * reg


def compute_info_bpr_loss(a_embeddings, b_embeddings, pos_edges, neg_items):
    """evalue/loss.py:95-117, reduction='mean'."""
    a = a_embeddings[pos_edges[:, 0]]
    b = b_embeddings[pos_edges[:, 1]]
    nb = b_embeddings[neg_items]
    comb = torch.cat([b.unsqueeze(1), nb], 1)
    logits = (comb @ a.unsqueeze(-1)).squeeze(-1)
    return F.cross_entropy(logits, torch.zeros(pos_edges.size(0), dtype=torch.int64))


def sgrec_loss(result, batch, reg_weight):
    """model/Main_Model.py:902-928 without the unused random-item term."""
    users = result["user_embeddings"][batch["user_ids"]]
    items = result["item_embeddings"][batch["item_ids"]]
    negs = result["item_embeddings"][batch["neg_items"].reshape(-1)]
    pos_score = torch.sum(users * items, dim=1)
    neg_score = torch.sum(users * negs, dim=1)
    loss = -torch.mean(torch.log(torch.sigmoid(pos_score - neg_score)))
    reg = torch.mean(result["ori_u_emb"] ** 2) + torch.mean(result["ori_i_emb"] ** 2)
    return loss + reg_weight * reg
