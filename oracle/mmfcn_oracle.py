"""Oracle: MMFCN forward (torch CPU), restating model/Main_Model.py:145-170,
model/test/out_Layer.py:20-54, Agg_Layer.py:16-26, Combin_Layer.py:16-30 and
graph_constructor.py:199-208 with torch.sparse.mm standing in for DGL's
update_all(u_mul_e, sum).  Dropout is the identity (eval mode / p = 0).
Parity unpinned (the reference's MMFCN wrapper cannot be constructed at HEAD and
no run of it is logged); this is the text of those files applied to tensors."""
import torch
import torch.nn.functional as F


def adjacency(src, dst, w, n_nodes):
    idx = torch.stack([torch.as_tensor(dst, dtype=torch.long), torch.as_tensor(src, dtype=torch.long)])
    return torch.sparse_coo_tensor(idx, torch.as_tensor(w, dtype=torch.float32),
                                   (n_nodes, n_nodes)).coalesce().to_sparse_csr()


def linear(x, p, prefix):
    return F.linear(x, p[prefix + ".weight"], p.get(prefix + ".bias"))


def agg_func(A, x, k, p, prefix):
    h = x
    for _ in range(k):                                      # graph_constructor.py:203-205
        h = F.prelu(linear(h, p, prefix + ".W"), p[prefix + ".active.alpha"])
        h = torch.sparse.mm(A, h)
    return h


def combine(x, agg_x, p, prefix):                           # Combin_Layer.py:26-30
    y1 = linear(x, p, prefix + ".W_ori_f")
    y2 = linear(agg_x, p, prefix + ".W_f")
    return F.prelu(linear(y1 + y2, p, prefix + ".W_o"), p[prefix + ".active.alpha"])


def mgcn(A, X, emb, k, p, prefix, layer_num):               # out_Layer.py:32-36
    agg_X = agg_func(A, X, k, p, prefix + ".agg")
    for l in range(layer_num):
        agg_X = combine(X, agg_X, p, f"{prefix}.combine.{l}")
    return agg_X + F.prelu(linear(emb, p, prefix + ".W_o"), p[prefix + ".active.alpha"])


def forward(p, A, padded_feats, proj_prefixes, embs, ks, alphas, layer_num, num_users):
    """p: parameter dict (names of recommend_amd.model.TESTModel.state_dict());
    padded_feats: list of [N, F] feature matrices zero-padded as Main_Model.py:65-68."""
    Xs = []
    for x, pre in zip(padded_feats, proj_prefixes):         # Main_Model.py:154
        Xs.append(x if pre is None else linear(x, p, pre))
    modal = [mgcn(A, Xs[i], embs[i], ks[i], p, f"test.MGCNs.{i}", layer_num) for i in range(len(Xs))]
    modal = torch.stack(modal, dim=0)
    a = torch.softmax(alphas, dim=0).unsqueeze(-1).unsqueeze(-1)   # out_Layer.py:53
    node = torch.sum(a * modal, dim=0)
    node = linear(node, p, "final_projection")              # Main_Model.py:162
    return node[:num_users], node[num_users:]
