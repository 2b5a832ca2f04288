"""Oracle: graph construction, normalisation and propagation (numpy, CPU).

Follows model/test/graph_constructor.py of the reference line by line; DGL's
message passing is restated from its documented semantics:
update_all(u_mul_e(h, w, m), sum(m, h))  =>  out[dst_e] += w_e * h[src_e]
update_all(copy_e(w, m), sum(m, x))      =>  x[dst_e] += w_e
apply_edges(f)                            =>  per-edge function of src/dst data
Parity unpinned (no DGL here, no reference fixtures): cross-checked in the tests
against scipy.sparse and dense matmul.
"""
import numpy as np


def build_coo(interactions, num_users, num_items, add_self_loops=True):
    """graph_constructor.py:73-105 -> (src int64, dst int64, w float32)."""
    num_nodes = num_users + num_items
    user_indices, item_indices, edge_weights = [], [], []
    max_rating = 0
    for user_id, item_id, rating in interactions:          # :80-91
        user_indices.append(user_id)
        item_indices.append(item_id + num_users)
        edge_weights.append(rating)
        user_indices.append(item_id + num_users)
        item_indices.append(user_id)
        edge_weights.append(rating)
        max_rating = max(rating, max_rating)
    if add_self_loops:                                     # :94-98
        for i in range(num_nodes):
            user_indices.append(i)
            item_indices.append(i)
            edge_weights.append(max_rating)
    return (np.asarray(user_indices, dtype=np.int64), np.asarray(item_indices, dtype=np.int64),
            np.asarray(edge_weights, dtype=np.float32))


def build_coo_vectorized(users, items, ratings, num_users, num_items, add_self_loops=True):
    """Same edge list as build_coo for array inputs (used at sizes where the python loop is slow)."""
    users = np.asarray(users, np.int64)
    items = np.asarray(items, np.int64)
    ratings = np.asarray(ratings, np.float32)
    n, N = users.shape[0], num_users + num_items
    src = np.empty(2 * n, np.int64)
    dst = np.empty(2 * n, np.int64)
    w = np.empty(2 * n, np.float32)
    src[0::2], dst[0::2], w[0::2] = users, items + num_users, ratings
    src[1::2], dst[1::2], w[1::2] = items + num_users, users, ratings
    if add_self_loops:
        loop = np.arange(N, dtype=np.int64)
        mx = np.float32(max(0.0, float(ratings.max()))) if n else np.float32(0)
        src, dst = np.concatenate([src, loop]), np.concatenate([dst, loop])
        w = np.concatenate([w, np.full(N, mx, np.float32)])
    return src, dst, w


def csr_from_coo(key, other, n_nodes):
    """Stable sort by key, ties by edge id -> (ptr, idx, eid)."""
    order = np.argsort(key, kind="stable")
    counts = np.bincount(key, minlength=n_nodes)
    ptr = np.concatenate([[0], np.cumsum(counts)])
    return ptr.astype(np.int64), np.asarray(other)[order], order


def normalize(src, dst, w, n_nodes):
    """graph_constructor.py:177-186 in fp32: in_w = sum of incoming weights (edge order),
    clamp(min=1e-12).sqrt(), w / (in_w[src] * in_w[dst])."""
    in_w = np.zeros(n_nodes, np.float32)
    np.add.at(in_w, dst, w.astype(np.float32))             # sequential fp32 adds in edge order
    in_w = np.sqrt(np.maximum(in_w, np.float32(1e-12))).astype(np.float32)
    return (w.astype(np.float32) / (in_w[src] * in_w[dst])).astype(np.float32), in_w


def deg_normalize(src, dst, n_nodes):
    """model/sgrec/graph.py:123-135: in_degrees().pow(-0.5) on both ends."""
    deg = np.bincount(dst, minlength=n_nodes).astype(np.float32)
    with np.errstate(divide="ignore"):
        d = np.power(deg, np.float32(-0.5)).astype(np.float32)
    return (d[src] * d[dst]).astype(np.float32)


def spmm(src, dst, w, h):
    """out[dst_e] += w_e * h[src_e], fp32, edge order."""
    out = np.zeros_like(h, dtype=np.float32)
    np.add.at(out, dst, (w[:, None].astype(np.float32) * h[src].astype(np.float32)))
    return out


def spmm_f64(src, dst, w, h):
    """Same in float64 (reference value for tolerance checks at large sizes)."""
    import scipy.sparse as sp
    n = h.shape[0]
    A = sp.coo_matrix((w.astype(np.float64), (dst, src)), shape=(n, n)).tocsr()
    return A @ h.astype(np.float64)


def aggrate(src, dst, w, node_x, k, alpha):
    """graph_constructor.py:189-197."""
    h0 = node_x.astype(np.float32)
    h = h0
    for _ in range(k):
        h = spmm(src, dst, w, h)
        h = (np.float32(alpha) * h0 + np.float32(1 - alpha) * h).astype(np.float32)
    return h


def agg_func(src, dst, w, node_x, k, f):
    """graph_constructor.py:199-208 with f = the composed dropout(active(f(.))) callable."""
    h = node_x.astype(np.float32)
    for _ in range(k):
        h = f(h)
        h = spmm(src, dst, w, h)
    return h


# ---- live-model variants of the propagation primitive (dropout off: eval mode) ----
def mgdcf_gamma(alpha, beta, k):
    """model/mig/mgdcf.py:46-48."""
    return np.power(beta, k) + alpha * np.sum([np.power(beta, i) for i in range(k)])


def mgdcf_sorted_homo_edges(user_index, item_index, num_users, num_items):
    """model/mig/mgdcf.py:78-96: u->i edges, i->u edges, then one self loop per node."""
    n = num_users + num_items
    hi = np.asarray(item_index) + num_users
    ui = np.asarray(user_index)
    loops = np.arange(n)
    return np.concatenate([ui, hi, loops]), np.concatenate([hi, ui, loops]), n


def mgdcf_forward(src, dst, w, x, k, alpha, beta, return_all=False):
    """model/mig/mgdcf.py:131-166 with the three dropouts inactive."""
    h0 = x.astype(np.float32)
    h = h0
    hs = []
    for _ in range(k):
        h = spmm(src, dst, w, h)
        h = (h * np.float32(beta) + h0 * np.float32(alpha)).astype(np.float32)
        hs.append(h)
    if not return_all:
        return (h / np.float32(mgdcf_gamma(alpha, beta, k))).astype(np.float32)
    return [(hh / np.float32(mgdcf_gamma(alpha, beta, i + 1))).astype(np.float32) for i, hh in enumerate(hs)]


def ii_gcn_forward(src, dst, w, x, k, alpha, beta):
    """model/sgrec/gcn.py:32-50 (II_GCN), dropouts inactive: every hop divides by gamma."""
    gamma = np.float32(pow(beta, k) + alpha * sum(pow(beta, i) for i in range(k)))
    h0 = x.astype(np.float32)
    h = h0
    for _ in range(k):
        h = spmm(src, dst, w, h)
        h = (h * np.float32(beta) + h0 * np.float32(alpha)).astype(np.float32)
        h = (h / gamma).astype(np.float32)
    return h


def iu_gcn_forward(src, dst, w, x, k):
    """model/sgrec/gcn.py:74-87 (IU_GCN), dropouts inactive: k plain hops."""
    h = x.astype(np.float32)
    for _ in range(k):
        h = spmm(src, dst, w, h)
    return h


def build_item_graph(feat, k):
    """model/sgrec/graph.py:100-118 densely in float64 (argsort instead of torch.topk: ties go to
    the lower id).  Returns (src, dst, w) in score.nonzero() order."""
    f = feat.astype(np.float64)
    nf = f / np.sqrt((f ** 2).sum(1, keepdims=True))
    score = nf @ nf.T
    n = score.shape[0]
    idx = np.argsort(-score, axis=1, kind="stable")[:, :k]
    mask = np.zeros_like(score, dtype=bool)
    mask[np.arange(n)[:, None], idx] = True
    score = score * mask
    score = (1.0 / (score.sum(1) + 1e-8))[:, None] * score
    dst, src = np.nonzero(score)
    return src, dst, score[dst, src]


def edge_feature_weights(src, dst, x_padded, edge_weight_type):
    """graph_constructor.py:137-156 in float64: x_padded is the feature matrix zero-padded to all
    nodes (:130-135).  "cosine": clamp(<x_s,x_d> / ((|x_s|+1e-12)(|x_d|+1e-12)), 0, 1);
    "dot": sigmoid(<x_s,x_d>)."""
    x = np.asarray(x_padded, np.float64)
    sim = np.einsum("ij,ij->i", x[src], x[dst])
    if edge_weight_type == "cosine":
        norm = np.linalg.norm(x, axis=1) + 1e-12
        return np.clip(sim / (norm[src] * norm[dst]), 0.0, 1.0)
    if edge_weight_type == "dot":
        return 1.0 / (1.0 + np.exp(-sim))
    raise ValueError(f"Unknown edge weight type: {edge_weight_type}")
