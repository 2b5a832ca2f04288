"""The propagation layers of the reference's live model families on the HIP SpMM.

Mirrors, with the same class names, constructor arguments and forward signatures:
  MGDCF   model/mig/mgdcf.py:14-175   h <- beta * A h + alpha * h0, k times, / gamma
  II_GCN  model/sgrec/gcn.py:7-50     the same with / gamma and z-dropout after every hop
  IU_GCN  model/sgrec/gcn.py:55-87    k plain hops with z-dropout after each
The graph argument `g` is a DeviceGraph (graph_constructor.py) instead of a DGLGraph: the
COO edge list in the reference's order plus CSR/CSC operands in HBM; `g.edata[key]` holds
edge values in COO order as it does in DGL.  Every hop -- forward and backward -- is one
rk_spmm_csr_f32 launch with the alpha/beta epilogue fused; the three nn.Dropout modules of
the reference are kept as they are (edge dropout acts on the COO value vector, from which
both operands of that forward pass take their values).
"""
import numpy as np
import torch
import torch.nn as nn

from . import ops
from .graph_constructor import DeviceGraph


def _operands(g: DeviceGraph, key, edge_dropout: nn.Dropout):
    """CSR/CSC operands carrying edata[key], after edge dropout when it is active."""
    if g.fwd is None:
        g.to("cuda")
    w = g.edata[key]
    if edge_dropout.training and edge_dropout.p > 0:
        w = edge_dropout(w)
        return g.fwd.with_values(w), g.bwd.with_values(w)
    g.refresh_values(key)
    return g.fwd, g.bwd


class MGDCF(nn.Module):

    CACHE_KEY = "mgdcf_weight"

    def __init__(self, k, alpha, beta, x_drop_rate, edge_drop_rate, z_drop_rate, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.k = k
        self.alpha = alpha
        self.beta = beta
        self.gamma = torch.tensor(MGDCF.compute_gamma(alpha, beta, k)).float()
        self.gammas = [torch.tensor(MGDCF.compute_gamma(alpha, beta, i)).float() for i in range(1, k + 1)]
        self.x_drop_rate = x_drop_rate
        self.edge_drop_rate = edge_drop_rate
        self.z_drop_rate = z_drop_rate
        self.x_dropout = nn.Dropout(x_drop_rate)
        self.edge_dropout = nn.Dropout(edge_drop_rate)
        self.z_dropout = nn.Dropout(z_drop_rate)

    @classmethod
    def compute_gamma(cls, alpha, beta, k):
        return np.power(beta, k) + alpha * np.sum([np.power(beta, i) for i in range(k)])

    @classmethod
    def build_sorted_homo_graph(cls, user_item_edges, num_users=None, num_items=None):
        """mgdcf.py:78-104: edges u->i, then i->u, then one self loop per node, unit weights."""
        user_index, item_index = np.asarray(user_item_edges).T
        if num_users is None:
            num_users = int(np.max(user_index)) + 1
        if num_items is None:
            num_items = int(np.max(item_index)) + 1
        n = num_users + num_items
        homo_item = item_index.astype(np.int64) + num_users
        loops = np.arange(n, dtype=np.int64)
        src = np.concatenate([user_index.astype(np.int64), homo_item, loops]).astype(np.int32)
        dst = np.concatenate([homo_item, user_index.astype(np.int64), loops]).astype(np.int32)
        return DeviceGraph(src, dst, np.ones(src.shape[0], np.float32), n)

    @classmethod
    @torch.no_grad()
    def norm_adj(cls, g):
        """mgdcf.py:107-128: w_e = in_deg[src]^-0.5 * in_deg[dst]^-0.5."""
        if g.fwd is None:
            g.to("cuda")
        g.edata[MGDCF.CACHE_KEY] = ops.deg_normalize(g.fwd)

    def forward(self, g, x, return_all=False):
        fwd, bwd = _operands(g, MGDCF.CACHE_KEY, self.edge_dropout)
        h0 = self.x_dropout(x)
        h = h0
        h_list = []
        for _ in range(self.k):
            h = ops.spmm(fwd, bwd, h, h0, self.alpha, self.beta)   # h * beta + h0 * alpha, fused
            if return_all:
                h_list.append(h)
        if not return_all:
            return self.z_dropout(h / self.gamma.to(h.device))
        return [self.z_dropout(hh / gamma.to(hh.device)) for hh, gamma in zip(h_list, self.gammas)]


class II_GCN(nn.Module):
    def __init__(self, k, alpha, beta, edge_drop_rate, x_drop_rate, z_drop_rate, cache_key):
        super().__init__()
        self.k = k
        self.beta = beta
        self.alpha = alpha
        self.gamma = pow(self.beta, k) + self.alpha * sum(pow(self.beta, i) for i in range(k))
        self.x_dropout = nn.Dropout(x_drop_rate)
        self.edge_dropout = nn.Dropout(edge_drop_rate)
        self.z_dropout = nn.Dropout(z_drop_rate)
        self.cache_key = cache_key

    def forward(self, x, g):
        fwd, bwd = _operands(g, self.cache_key, self.edge_dropout)
        h0 = self.x_dropout(x)
        h = h0
        for _ in range(self.k):
            # (h * beta + h0 * alpha) / gamma with the division folded into the coefficients
            h = ops.spmm(fwd, bwd, h, h0, self.alpha / self.gamma, self.beta / self.gamma)
            h = self.z_dropout(h)
        return h


class IU_GCN(nn.Module):
    def __init__(self, edge_drop_rate, x_drop_rate, z_drop_rate, cache_key):
        super().__init__()
        self.x_dropout = nn.Dropout(x_drop_rate)
        self.edge_dropout = nn.Dropout(edge_drop_rate)
        self.z_dropout = nn.Dropout(z_drop_rate)
        self.cache_key = cache_key

    def forward(self, x, g, k):
        self.k = k
        fwd, bwd = _operands(g, self.cache_key, self.edge_dropout)
        h = self.x_dropout(x)
        for _ in range(k):
            h = self.z_dropout(ops.spmm(fwd, bwd, h))
        return h


def build_item_graph(feat: torch.Tensor, k: int, weight_key: str) -> DeviceGraph:
    """Item-item kNN graph of model/sgrec/graph.py:100-118 (build_item_g) without the dense
    I x I matrices: cosine similarities and the k largest of each row come out of one fused
    kernel (rk_score_topk_wide_f32: the feature width is a loop inside it; ties: lower item id),
    weights s / (sum of the row's k similarities + 1e-8).  Edges run from the neighbour (src) to
    the row item (dst), ordered by (dst, src) as score.nonzero() lists them; exact-zero
    similarities are dropped as nonzero() drops them.  feat: [I, F] on the GPU.  Returns the
    graph with edata[weight_key] set."""
    if not feat.is_cuda:
        from . import _ffi
        raise _ffi.RkError(_ffi.RK_ERR_NODEV, "build_item_graph: features must be on the GPU")
    if not 1 <= k <= min(32, feat.shape[0]):
        raise ValueError("k must be in the range [1, min(32, num_items)]")
    feat = feat.float()
    nf = feat / torch.sqrt(torch.sum(feat ** 2, dim=1, keepdim=True))
    n = nf.shape[0]
    vals, ids = ops.score_topk_wide(nf, nf, k)
    inv = 1.0 / (vals.sum(dim=1, keepdim=True) + 1e-8)
    w = inv * vals
    order = torch.argsort(ids, dim=1)                  # nonzero() order: ascending column
    ids, w = torch.gather(ids, 1, order), torch.gather(w, 1, order)
    rows = torch.arange(n, device=feat.device, dtype=torch.int32).unsqueeze(1).expand_as(ids)
    keep = w != 0
    src_parts, dst_parts, w_parts = [ids[keep]], [rows[keep]], [w[keep]]
    src = torch.cat(src_parts).cpu().numpy().astype(np.int32)
    dst = torch.cat(dst_parts).cpu().numpy().astype(np.int32)
    w = torch.cat(w_parts).float()
    g = DeviceGraph(src, dst, w.cpu().numpy(), n)
    g.to(feat.device)
    g.edata[weight_key] = w
    return g


class SGrecGraphOps:
    """The graph half of the reference's SGrec `Graph` module (model/sgrec/graph.py:88-135) on the
    device graph: build_graph, build_item_g and norm_adj / creat_graph_weight with the reference's
    names, argument meaning and attribute names (num_users, num_items, num_nodes, device,
    weight_cache, g).  The reference's module keeps its torch layers (feature MLPs, transformer)
    and takes these three methods from here, with II_GCN / IU_GCN above as its propagation layers;
    `dgl` is then no longer needed by that model."""

    def build_graph(self, interactions) -> DeviceGraph:
        """graph.py:88-98: edges u -> i+U, then i+U -> u, then one self loop per node, in that order."""
        self.num_inter = len(interactions)
        inter = np.asarray(interactions)
        users = inter[:, 0].astype(np.int64)
        items = inter[:, 1].astype(np.int64) + self.num_users
        loops = np.arange(self.num_nodes, dtype=np.int64)
        src = np.concatenate([users, items, loops]).astype(np.int32)
        dst = np.concatenate([items, users, loops]).astype(np.int32)
        self.g = DeviceGraph(src, dst, np.ones(src.shape[0], np.float32), self.num_nodes).to(self.device)
        return self.g

    def build_item_g(self, feat, k) -> DeviceGraph:
        """graph.py:100-118."""
        return build_item_graph(feat.to(self.device), k, self.weight_cache)

    def creat_graph_weight(self):
        self.norm_adj()

    @torch.no_grad()
    def norm_adj(self):
        """graph.py:123-135: w_e = in_deg[src]^-0.5 * in_deg[dst]^-0.5."""
        self.g.edata[self.weight_cache] = ops.deg_normalize(self.g.fwd)
