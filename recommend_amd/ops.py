"""Tensor-level wrappers of the C ABI: autograd Functions for the propagation,
projection and BPR kernels, plain functions for evaluation.

PyTorch is used for device memory, streams and autograd bookkeeping only; all
arithmetic of the hot path runs in librecommend_amd.so.  Every function here
requires CUDA(HIP) tensors and raises otherwise.
"""
import ctypes
import itertools

import numpy as np
import torch

from . import _ffi
from ._ffi import c_float, c_int32, c_int64, c_size_t, ptr, stream_ptr


def _need_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise _ffi.RkError(_ffi.RK_ERR_NODEV,
                               "recommend_amd device ops need tensors on a HIP device; "
                               "there is no CPU fallback")


def _f32c(t):
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


# --------------------------------------------------------------------------
# host helpers (graph construction, sampler RNG): callable without a GPU
# --------------------------------------------------------------------------
def build_coo_host(users, items, ratings, n_users, n_items, add_self_loops):
    """rk_build_coo_host on numpy arrays -> (src, dst, w) numpy int32/int32/float32."""
    users = np.ascontiguousarray(users, dtype=np.int64)
    items = np.ascontiguousarray(items, dtype=np.int64)
    if ratings is not None:
        ratings = np.ascontiguousarray(ratings, dtype=np.float32)
    n = users.shape[0]
    nnz = 2 * n + ((n_users + n_items) if add_self_loops else 0)
    src = np.empty(nnz, np.int32)
    dst = np.empty(nnz, np.int32)
    w = np.empty(nnz, np.float32)
    _ffi.call("rk_build_coo_host", ptr(users), ptr(items), ptr(ratings), c_int64(n),
              c_int32(n_users), c_int32(n_items), c_int32(1 if add_self_loops else 0),
              ptr(src), ptr(dst), ptr(w))
    return src, dst, w


def csr_from_coo_host(key, other, n_nodes):
    """Stable sort of the COO by key -> (ptr, idx, eid) numpy int32."""
    key = np.ascontiguousarray(key, dtype=np.int32)
    other = np.ascontiguousarray(other, dtype=np.int32)
    nnz = key.shape[0]
    p = np.empty(n_nodes + 1, np.int32)
    idx = np.empty(nnz, np.int32)
    eid = np.empty(nnz, np.int32)
    _ffi.call("rk_csr_from_coo_host", ptr(key), ptr(other), c_int64(nnz), c_int32(n_nodes),
              ptr(p), ptr(idx), ptr(eid))
    return p, idx, eid


def spmm_plan_host(rowptr, long_thresh):
    """Row binning of a CSR -> dict of numpy int32 arrays (see rk_spmm_plan)."""
    rowptr = np.ascontiguousarray(rowptr, dtype=np.int32)
    n_rows = rowptr.shape[0] - 1
    nc, nm, ns = c_int32(0), c_int32(0), c_int32(0)
    _ffi.call("rk_spmm_plan_host", ptr(rowptr), c_int32(n_rows), c_int32(long_thresh),
              ctypes.byref(nc), ctypes.byref(nm), ctypes.byref(ns), None, None, None, None, None, None)
    arr = {k: np.zeros(max(nc.value, 1), np.int32) for k in ("chunk_row", "chunk_begin", "chunk_end", "chunk_slot")}
    arr["multi_rows"] = np.zeros(max(nm.value, 1), np.int32)
    arr["multi_slot_ptr"] = np.zeros(nm.value + 1, np.int32)
    _ffi.call("rk_spmm_plan_host", ptr(rowptr), c_int32(n_rows), c_int32(long_thresh),
              ctypes.byref(nc), ctypes.byref(nm), ctypes.byref(ns), ptr(arr["chunk_row"]),
              ptr(arr["chunk_begin"]), ptr(arr["chunk_end"]), ptr(arr["chunk_slot"]),
              ptr(arr["multi_rows"]), ptr(arr["multi_slot_ptr"]))
    arr.update(n_chunks=nc.value, n_multi=nm.value, n_slots=ns.value)
    return arr


# --------------------------------------------------------------------------
# G2 / G3: normalisation and propagation
# --------------------------------------------------------------------------
class CsrMatrix:
    """One sparse operand on one device: CSR arrays, edge values and the row
    binning the SpMM kernels use.  `eid` maps CSR positions to COO edge ids."""

    # Rows with more edges than this go to whole workgroups.  A 16-lane group walks its row four edges
    # at a time, one dependent gather round after the other: at 64 the longest group-tier rows were what
    # a hop waited for (hop at the Elec shape 81.5 us; 32: 78.3, 24: 77.9, 16: 78.1, 12: 83.2, 8: 105.9).
    LONG_THRESH = 32

    def __init__(self, rowptr, idx, eid, device, n_cols=None):
        self.n_rows = int(rowptr.shape[0] - 1)
        self.nnz = int(idx.shape[0])
        # rows of h the kernel may read: callers hand over h with at least this many rows
        self.n_cols = int(n_cols) if n_cols is not None else (int(np.max(idx)) + 1 if self.nnz else 0)
        self.device = torch.device(device)
        pl = spmm_plan_host(rowptr, self.LONG_THRESH)
        # empty arrays are padded to one element so that no NULL pointer reaches the library
        to = lambda a: torch.from_numpy(np.ascontiguousarray(a if len(a) else np.zeros(1, a.dtype))).to(self.device)
        self.rowptr = to(rowptr)
        self.idx = to(idx)
        self.eid = to(eid)
        self.n_long = self.n_chunks = pl["n_chunks"]
        self.n_multi, self.n_slots = pl["n_multi"], pl["n_slots"]
        self._plan_dev = {k: to(pl[k]) for k in ("chunk_row", "chunk_begin", "chunk_end", "chunk_slot",
                                                 "multi_rows", "multi_slot_ptr")} if self.n_chunks else None
        self.val = None          # fp32 [nnz] in CSR order
        self._partial = {}       # D -> scratch for the chunks of multi-chunk rows

    def set_values_from_coo(self, w_coo):
        """val[p] = w_coo[eid[p]]"""
        _need_cuda(w_coo)
        self.val = torch.zeros(max(self.nnz, 1), dtype=torch.float32, device=self.device)
        _ffi.call("rk_gather_f32", ptr(w_coo), ptr(self.eid), c_int64(self.nnz), ptr(self.val),
                  stream_ptr(self.device))

    def with_values(self, w_coo):
        """The same sparsity pattern with its own edge values (e.g. one draw of edge dropout):
        shares the index arrays and the row binning, owns `val`."""
        import copy
        m = copy.copy(self)
        m.set_values_from_coo(w_coo)
        return m

    def plan(self, D):
        if not self.n_chunks:
            return None, None
        part = self._partial.get(D)
        if part is None:
            # partial rows of the chunks of multi-chunk rows + their arrival counters (zero between launches)
            part = (torch.empty(max(self.n_slots, 1) * D, dtype=torch.float32, device=self.device),
                    torch.zeros(max(self.n_multi, 1), dtype=torch.int32, device=self.device))
            self._partial[D] = part
        d = self._plan_dev
        p = _ffi.SpmmPlan(self.LONG_THRESH, self.n_chunks, d["chunk_row"].data_ptr(),
                          d["chunk_begin"].data_ptr(), d["chunk_end"].data_ptr(),
                          d["chunk_slot"].data_ptr(), self.n_multi, d["multi_rows"].data_ptr(),
                          d["multi_slot_ptr"].data_ptr(), self.n_slots, part[0].data_ptr(), part[1].data_ptr())
        return p, part


def spmm_raw(mat, h, h0=None, alpha=0.0, beta=1.0, out=None):
    """out = beta*(A h)  (h0 is None)  or  alpha*h0 + beta*(A h); no autograd."""
    _need_cuda(h, h0)
    if mat.val is None:
        raise ValueError("sparse matrix has no edge values yet")
    h = _f32c(h)
    if h.dim() != 2 or h.shape[0] < mat.n_cols:
        raise ValueError(f"h must be [n_nodes >= {mat.n_cols}, D]: the operand gathers rows up to {mat.n_cols - 1}, "
                         f"h has shape {tuple(h.shape)}")
    if h0 is not None and (h0.dim() != 2 or h0.shape[0] < mat.n_rows or h0.shape[1] != h.shape[1]):
        raise ValueError(f"h0 must be [n_rows >= {mat.n_rows}, {h.shape[1]}], got {tuple(h0.shape)}")
    if out is not None and (not out.is_cuda or out.dtype != torch.float32 or not out.is_contiguous()
                            or out.shape[0] < mat.n_rows or out.shape[1] != h.shape[1]):
        raise ValueError("out must be a contiguous fp32 CUDA tensor [n_rows, D]")
    D = h.shape[1]
    if out is None:
        out = torch.empty((mat.n_rows, D), dtype=torch.float32, device=h.device)
    if h0 is not None:
        h0 = _f32c(h0)
    plan, _keep = mat.plan(D)
    _ffi.call("rk_spmm_csr_f32", ptr(mat.rowptr), ptr(mat.idx), ptr(mat.val), c_int32(mat.n_rows),
              ptr(h), c_int32(D), ptr(out), ptr(h0), c_float(alpha), c_float(beta),
              ctypes.byref(plan) if plan is not None else None, stream_ptr(h.device))
    return out


class _SpMM(torch.autograd.Function):
    """out = alpha*h0 + beta*(A h); backward uses the transposed (CSC) operand."""

    @staticmethod
    def forward(ctx, h, h0, fwd_mat, bwd_mat, alpha, beta):
        ctx.bwd_mat = bwd_mat
        ctx.alpha, ctx.beta = alpha, beta
        ctx.has_h0 = h0 is not None
        return spmm_raw(fwd_mat, h, h0, alpha, beta)

    @staticmethod
    def backward(ctx, g):
        g = _f32c(g)
        gh = gh0 = None
        if ctx.needs_input_grad[0]:
            gh = spmm_raw(ctx.bwd_mat, g, None, 0.0, ctx.beta)
        if ctx.has_h0 and ctx.needs_input_grad[1]:
            gh0 = g * ctx.alpha
        return gh, gh0, None, None, None, None


def spmm(fwd_mat, bwd_mat, h, h0=None, alpha=0.0, beta=1.0):
    return _SpMM.apply(h, h0, fwd_mat, bwd_mat, float(alpha), float(beta))


def sym_normalize(mat, w_coo):
    """graph_constructor.py:158-187 on the device; returns (sqrt_in_w, w_norm_coo)."""
    _need_cuda(w_coo)
    w_coo = _f32c(w_coo)
    sq = torch.empty(mat.n_rows, dtype=torch.float32, device=w_coo.device)
    out = torch.empty_like(w_coo)
    _ffi.call("rk_sym_normalize_f32", ptr(mat.rowptr), ptr(mat.idx), ptr(mat.eid), ptr(w_coo),
              c_int32(mat.n_rows), ptr(sq), ptr(out), stream_ptr(w_coo.device))
    return sq, out


def in_weight_sqrt(mat, w_local):
    """sqrt(max(sum of each row's incoming weights, 1e-12)) for the rows of `mat`
    (graph_constructor.py:177-178); w_local is indexed by mat.eid."""
    _need_cuda(w_local)
    w_local = _f32c(w_local)
    sq = torch.empty(max(mat.n_rows, 1), dtype=torch.float32, device=w_local.device)
    _ffi.call("rk_in_weight_sqrt_f32", ptr(mat.rowptr), ptr(mat.eid), ptr(w_local), c_int32(mat.n_rows), ptr(sq),
              stream_ptr(w_local.device))
    return sq[: mat.n_rows]


def sym_scale_rows(mat, w_local, row_offset, sqrt_in_w):
    """w / (sqrt_in_w[col] * sqrt_in_w[row_offset + row]) for the edges of `mat`
    (graph_constructor.py:186); sqrt_in_w covers all nodes."""
    _need_cuda(w_local, sqrt_in_w)
    w_local, sqrt_in_w = _f32c(w_local), _f32c(sqrt_in_w)
    if sqrt_in_w.numel() < max(mat.n_cols, row_offset + mat.n_rows):
        raise ValueError("sqrt_in_w is shorter than the node range the block refers to")
    out = torch.empty_like(w_local)
    _ffi.call("rk_sym_scale_rows_f32", ptr(mat.rowptr), ptr(mat.idx), ptr(mat.eid), ptr(w_local), c_int32(mat.n_rows),
              c_int32(int(row_offset)), ptr(sqrt_in_w), ptr(out), stream_ptr(w_local.device))
    return out


def edge_feature_sim(src, dst, x_rows, row_lo, mode):
    """Per-edge feature similarity (graph_constructor.py:137-156): x_rows are the features of nodes
    [row_lo, row_lo + len(x_rows)), all other nodes have zero features.  mode: "cosine" | "dot"."""
    _need_cuda(src, dst, x_rows)
    x_rows = _f32c(x_rows)
    src, dst = src.to(torch.int32).contiguous(), dst.to(torch.int32).contiguous()
    if x_rows.shape[1] % 4:
        x_rows = torch.nn.functional.pad(x_rows, (0, 4 - x_rows.shape[1] % 4))   # zero columns change nothing
    out = torch.empty(src.shape[0], dtype=torch.float32, device=src.device)
    _ffi.call("rk_edge_feature_sim_f32", ptr(src), ptr(dst), c_int64(src.shape[0]), ptr(x_rows), c_int32(int(row_lo)),
              c_int32(x_rows.shape[0]), c_int32(x_rows.shape[1]), c_int32({"cosine": 0, "dot": 1}[mode]), ptr(out),
              stream_ptr(src.device))
    return out


def deg_normalize(mat):
    """w_e = deg_in[src]^-0.5 * deg_in[dst]^-0.5 in COO order."""
    out = torch.empty(mat.nnz, dtype=torch.float32, device=mat.device)
    _ffi.require_device()
    _ffi.call("rk_deg_normalize_f32", ptr(mat.rowptr), ptr(mat.idx), ptr(mat.eid),
              c_int32(mat.n_rows), ptr(out), stream_ptr(mat.device))
    return out


# --------------------------------------------------------------------------
# W1: modality projection
# --------------------------------------------------------------------------
class _Proj(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x_items, weight, bias, n_users):
        _need_cuda(x_items, weight, bias)
        x_items, weight = _f32c(x_items), _f32c(weight)
        bias_c = _f32c(bias) if bias is not None else None
        n_items, F = x_items.shape
        D = weight.shape[0]
        if weight.shape[1] != F:
            raise ValueError(f"weight is {tuple(weight.shape)}, features are {F} wide")
        y = torch.empty((n_users + n_items, D), dtype=torch.float32, device=x_items.device)
        ws_bytes = _ffi.load().rk_proj_fwd_ws_bytes(n_items, F, D)     # the weights' split parts (0: shapes the plain kernel takes)
        ws = torch.empty(max(ws_bytes, 16) // 4 + 4, dtype=torch.float32, device=y.device)
        _ffi.call("rk_proj_fwd_ws_f32", ptr(x_items), c_int32(n_items), c_int32(F), ptr(weight),
                  ptr(bias_c), c_int32(D), c_int32(n_users), ptr(y), ptr(ws), c_size_t(ws.numel() * 4),
                  stream_ptr(y.device))
        ctx.save_for_backward(x_items)
        ctx.n_users, ctx.D, ctx.has_bias = n_users, D, bias is not None
        return y

    @staticmethod
    def backward(ctx, g):
        (x_items,) = ctx.saved_tensors
        g = _f32c(g)
        n_items, F = x_items.shape
        D, n_users = ctx.D, ctx.n_users
        lib = _ffi.load()
        ws_bytes = lib.rk_proj_bwd_ws_bytes(n_items, n_users, F, D)
        ws = torch.empty(max(ws_bytes, 16) // 4 + 4, dtype=torch.float32, device=g.device)
        dW = torch.empty((D, F), dtype=torch.float32, device=g.device)
        db = torch.empty(D, dtype=torch.float32, device=g.device) if ctx.has_bias else None
        _ffi.call("rk_proj_bwd_f32", ptr(x_items), c_int32(n_items), c_int32(F), ptr(g),
                  c_int32(D), c_int32(n_users), ptr(dW), ptr(db), ptr(ws),
                  c_size_t(ws.numel() * 4), stream_ptr(g.device))
        return None, dW, db, None


def proj(x_items, weight, bias, n_users):
    """[bias ; x_items @ weight.T + bias] as one (n_users + n_items) x D matrix."""
    return _Proj.apply(x_items, weight, bias, int(n_users))


# --------------------------------------------------------------------------
# L1 / N: BPR gather-dot loss with optional in-kernel negative lookup
# --------------------------------------------------------------------------
class _Bpr(torch.autograd.Function):
    @staticmethod
    def forward(ctx, U, I, users, pos, neg, mask, neg_rank, pos_ptr, pos_items, n_items, mode=0):
        _need_cuda(U, I, users, pos, neg, mask, neg_rank, pos_ptr, pos_items)
        U, I = _f32c(U), _f32c(I)
        users = users.to(torch.int64).contiguous()
        pos = pos.to(torch.int64).contiguous()
        B = users.shape[0]
        if neg_rank is not None:
            neg_rank = neg_rank.to(torch.int32).contiguous()
            R = neg_rank.shape[1] if neg_rank.dim() == 2 else 1
            neg = torch.empty((B, R), dtype=torch.int64, device=U.device)
        else:
            neg = neg.to(torch.int64).contiguous()
            R = neg.shape[1] if neg.dim() == 2 else 1
        if mask is not None:
            mask = _f32c(mask)
        D = U.shape[1]
        lib = _ffi.load()
        ws_bytes = lib.rk_bpr_ws_bytes(B, R)
        ws = torch.empty(ws_bytes // 4 + 4, dtype=torch.float32, device=U.device)
        x = torch.empty((max(B, 1), max(R, 1)), dtype=torch.float32, device=U.device)
        out = torch.empty(2, dtype=torch.float32, device=U.device)
        _ffi.call("rk_rank_loss_fwd_f32", ptr(U), ptr(I), c_int32(D), ptr(users), ptr(pos), ptr(neg),
                  c_int64(B), c_int32(R), ptr(mask), ptr(neg_rank), ptr(pos_ptr), ptr(pos_items),
                  c_int32(int(n_items)), c_int32(int(mode)), ptr(x), ptr(out), ptr(ws),
                  c_size_t(ws.numel() * 4), stream_ptr(U.device))
        ctx.save_for_backward(U, I, users, pos, neg, mask, x, out)
        ctx.dims = (B, R, D)
        ctx.mode = int(mode)
        ctx.mark_non_differentiable(neg)
        return out[0], neg

    @staticmethod
    def backward(ctx, g, _gneg):
        U, I, users, pos, neg, mask, x, out = ctx.saved_tensors
        B, R, D = ctx.dims
        g = _f32c(g).reshape(1)
        if _stacked_rows(U, I):
            # the two tables are the row blocks of one node matrix (split_rows): their gradients
            # are written into one matrix, which split_rows' backward hands on as it is
            both = torch.zeros((U.shape[0] + I.shape[0], D), dtype=torch.float32, device=U.device)
            dU, dI = both[: U.shape[0]], both[U.shape[0]:]
        else:
            dU = torch.zeros_like(U)
            dI = torch.zeros_like(I)
        _ffi.call("rk_rank_loss_bwd_f32", ptr(U), ptr(I), c_int32(D), ptr(users), ptr(pos), ptr(neg),
                  c_int64(B), c_int32(R), ptr(mask), ptr(x), ptr(out), ptr(g), c_int32(ctx.mode),
                  ptr(dU), ptr(dI), stream_ptr(U.device))
        return dU, dI, None, None, None, None, None, None, None, None, None


def _stacked_rows(top, bottom):
    """True when `top` and `bottom` are the upper and lower row blocks of one contiguous matrix."""
    return (top.dim() == 2 and bottom.dim() == 2 and top.shape[1] == bottom.shape[1]
            and top.dtype == bottom.dtype and top.device == bottom.device
            and top.is_contiguous() and bottom.is_contiguous()
            and top.untyped_storage().data_ptr() == bottom.untyped_storage().data_ptr()
            and bottom.storage_offset() == top.storage_offset() + top.numel())


class _SplitRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, n_top):
        ctx.n_top, ctx.shape = n_top, tuple(x.shape)
        return x[:n_top], x[n_top:]

    @staticmethod
    def backward(ctx, g_top, g_bottom):
        n_top, (n, d) = ctx.n_top, ctx.shape
        if g_top is not None and g_bottom is not None and _stacked_rows(g_top, g_bottom):
            return g_top.as_strided((n, d), (d, 1), g_top.storage_offset()), None
        ref = g_top if g_top is not None else g_bottom
        if g_top is None:
            g_top = ref.new_zeros((n_top, d))
        if g_bottom is None:
            g_bottom = ref.new_zeros((n - n_top, d))
        return torch.cat([g_top, g_bottom]), None


def split_rows(x, n_top):
    """(x[:n_top], x[n_top:]) of a node matrix (user rows, item rows: model/Main_Model.py:164-165).  Sliced
    by autograd each half's gradient becomes a zero-filled copy of the whole matrix and the two are
    added; here a consumer that wrote both gradients into one matrix (the BPR loss does) passes that
    matrix on untouched."""
    return _SplitRows.apply(x, int(n_top))


LOSS_BPR, LOSS_LOG_SIGMOID, LOSS_INFO_BPR = 0, 1, 2   # RK_LOSS_* of the header


def bpr_gather_dot(U, I, users, pos, neg, mask=None, mode=LOSS_BPR):
    """mean(-logsigmoid(<U[u],I[i]> - <U[u],I[n]>)) over B x R (train/graph_trainer.py:255-270);
    `mode` selects the other per-pair terms of evalue/loss.py (see the header)."""
    loss, _ = _Bpr.apply(U, I, users, pos, neg, mask, None, None, None, I.shape[0], mode)
    return loss


def bpr_gather_dot_sampled(U, I, users, pos, neg_rank, pos_ptr, pos_items, mask=None):
    """Same loss with negatives resolved in the kernel from per-sample ranks;
    returns (loss, neg_items)."""
    return _Bpr.apply(U, I, users, pos, None, mask, neg_rank, pos_ptr, pos_items, I.shape[0])


# --------------------------------------------------------------------------
# W2: fused node-wise layers  Y = dropout(prelu(X1 W1^T [+ X2 W2^T] + b)) [+ R]
# --------------------------------------------------------------------------
_seed_counter = itertools.count(1)


def next_dropout_seed():
    """64-bit seed for one dropout call: torch's seed mixed with a call counter, so a run
    re-seeded with torch.manual_seed reproduces its masks."""
    # (63 bits: autograd tooling that records call arguments expects them to fit an int64)
    return (torch.initial_seed() * 0x9E3779B97F4A7C15 + next(_seed_counter) * 0xD1B54A32D192ED03) & (2 ** 63 - 1)


_dropout_step = {}


def set_dropout_step_tensor(t):
    """Register (or clear, with None) a one-element int64 CUDA tensor that the layer kernels
    mix into their dropout seeds at run time.  A training step captured in a HIP graph keeps
    the seeds it was captured with; bumping this counter inside the graph gives every replay
    its own masks."""
    if t is None:
        _dropout_step.clear()
        return
    if not (t.is_cuda and t.dtype == torch.int64 and t.numel() == 1):
        raise ValueError("the dropout step counter is a one-element int64 CUDA tensor")
    _dropout_step[t.device] = t


DENSE_WIDTHS = (64, 128)   # K = D of the built rk_dense_* kernels


def dense_supported(x, weight):
    return (x.is_cuda and x.dim() == 2 and weight.shape[0] == weight.shape[1] and weight.shape[0] in DENSE_WIDTHS
            and x.shape[1] == weight.shape[1])


def dense_fwd_raw(x1, w1, b1=None, x2=None, w2=None, b2=None, alpha=None, p=0.0, seed=0, residual=None):
    """Y = dropout(prelu(x1 w1^T [+ x2 w2^T] + b1 [+ b2], alpha), p) [+ residual]; no autograd.
    Tensors are contiguous fp32 on one HIP device."""
    _need_cuda(x1, w1, b1, x2, w2, b2, alpha, residual)
    if x2 is not None and (alpha is not None or p > 0.0):
        raise ValueError("the two-input form is linear only (no activation / dropout)")
    if alpha is not None and alpha.numel() != 1:
        raise ValueError("only a single PReLU slope is supported")
    n, K = x1.shape
    D = w1.shape[0]
    y = torch.empty((n, D), dtype=torch.float32, device=x1.device)
    step = _dropout_step.get(x1.device) if p > 0.0 else None
    _ffi.call("rk_dense_fwd_f32", ptr(x1), ptr(w1), ptr(x2), ptr(w2), ptr(b1), ptr(b2), ptr(alpha),
              c_float(p), _ffi.c_uint64(seed), ptr(step), ptr(residual), c_int64(n), c_int32(K),
              c_int32(D), ptr(y), stream_ptr(y.device))
    return y


def dense_bwd_ws(n, K, D, device):
    return torch.empty(_ffi.load().rk_dense_bwd_ws_bytes(n, K, D) // 4 + 4, dtype=torch.float32, device=device)


class DenseReduceQueue:
    """The closing reductions of several dense_bwd_raw calls, run as one launch by flush().

    A call with `defer=queue` leaves its per-workgroup partial sums in a slab of its own and
    returns dW / db / dalpha tensors that hold their values only after flush(); calls that add to
    the same tensors (grads=...) become one job whose slabs are summed in call order -- the bits
    are those of the immediate calls.  Built for the 64- and 128-wide layers."""

    def __init__(self, n_rows, width, device):
        self.n_rows, self.width, self.device = int(n_rows), int(width), device
        self.slab_floats = _ffi.load().rk_dense_bwd_slab_bytes(self.n_rows, self.width, self.width) // 4
        self.jobs = []       # [dw, db, da, n_part, [slabs]]

    @property
    def usable(self):
        return self.slab_floats > 0

    def new_slab(self):
        return torch.empty(self.slab_floats, dtype=torch.float32, device=self.device)

    def add(self, slab, n_part, dw, db, da, accumulate):
        if accumulate:
            for job in self.jobs:
                if job[0] is dw:
                    if job[3] != n_part or len(job[4]) >= _ffi.DENSE_REDUCE_MAX_SLABS:
                        break
                    job[4].append(slab)
                    return
            raise ValueError("deferred accumulation needs the call that created these gradient tensors in the "
                             f"same queue, and at most {_ffi.DENSE_REDUCE_MAX_SLABS} calls per tensor")
        self.jobs.append([dw, db, da, n_part, [slab]])

    def flush(self):
        if not self.jobs:
            return
        arr = (_ffi.DenseReduceJob * len(self.jobs))()
        for j, (dw, db, da, n_part, slabs) in enumerate(self.jobs):
            for i, sl in enumerate(slabs):
                arr[j].slabs[i] = sl.data_ptr()
            arr[j].n_slabs, arr[j].n_part = len(slabs), n_part
            arr[j].dW = dw.data_ptr()
            arr[j].db = db.data_ptr() if db is not None else None
            arr[j].dalpha = da.data_ptr() if da is not None else None
        _ffi.call("rk_dense_reduce_jobs_d_f32", arr, c_int32(len(self.jobs)), c_int32(self.width), stream_ptr(self.device))
        self.jobs = []


def dense_bwd_raw(x, w, b, alpha, p, seed, g, want_dx=True, want_db=True, grads=None, ws=None, defer=None):
    """Backward of the single-input layer for upstream gradient g -> (dX or None, dW, db or None,
    dalpha or None).  grads = (dW, db, dalpha) buffers to ADD the weight gradients to (a layer
    applied several times); otherwise fresh tensors are written.  The dropout step counter
    registered for the device is mixed into the seed exactly as the forward call did.
    defer: a DenseReduceQueue that runs this call's closing reduction later (see there)."""
    _need_cuda(x, w, b, alpha, g)
    n, K = x.shape
    D = w.shape[0]
    dev = g.device
    dx = torch.empty_like(x) if want_dx else None
    acc = grads is not None
    if acc:
        dw, db, da = grads
    else:
        dw = torch.empty_like(w)
        db = torch.empty(D, dtype=torch.float32, device=dev) if want_db else None
        da = torch.empty(1, dtype=torch.float32, device=dev) if alpha is not None else None
    step = _dropout_step.get(dev) if p > 0.0 else None
    if defer is not None and defer.usable and K == defer.width and D == defer.width and n == defer.n_rows:
        slab = defer.new_slab()
        n_part = c_int32(0)
        _ffi.call("rk_dense_bwd_partials_f32", ptr(x), ptr(w), ptr(b), ptr(alpha), c_float(p), _ffi.c_uint64(seed),
                  ptr(step), ptr(g), c_int64(n), c_int32(K), c_int32(D), ptr(dx), ptr(slab),
                  c_size_t(slab.numel() * 4), ctypes.byref(n_part), stream_ptr(dev))
        defer.add(slab, n_part.value, dw, db, da, acc)
        return dx, dw, db, da
    if ws is None:
        ws = dense_bwd_ws(n, K, D, dev)
    _ffi.call("rk_dense_bwd_accum_f32", ptr(x), ptr(w), ptr(b), ptr(alpha), c_float(p), _ffi.c_uint64(seed),
              ptr(step), ptr(g), c_int64(n), c_int32(K), c_int32(D), ptr(dx), ptr(dw), ptr(db), ptr(da),
              c_int32(1 if acc else 0), ptr(ws), c_size_t(ws.numel() * 4), stream_ptr(dev))
    return dx, dw, db, da


class _Dense(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x1, w1, b1, x2, w2, b2, alpha, residual, p, seed):
        x1, w1 = _f32c(x1), _f32c(w1)
        b1 = _f32c(b1) if b1 is not None else None
        if x2 is not None:
            x2, w2 = _f32c(x2), _f32c(w2)
            b2 = _f32c(b2) if b2 is not None else None
        if residual is not None:
            residual = _f32c(residual)
        alpha_c = _f32c(alpha) if alpha is not None else None
        y = dense_fwd_raw(x1, w1, b1, x2, w2, b2, alpha_c, p, seed, residual)
        ctx.save_for_backward(x1, w1, b1, x2, w2, alpha_c)
        ctx.p, ctx.seed = p, seed
        ctx.has = (b1 is not None, b2 is not None, residual is not None)
        return y

    @staticmethod
    def backward(ctx, g):
        x1, w1, b1, x2, w2, alpha = ctx.saved_tensors
        has_b1, has_b2, has_res = ctx.has
        g = _f32c(g)
        n, K = x1.shape
        ws = dense_bwd_ws(n, K, w1.shape[0], g.device)
        need = ctx.needs_input_grad
        dx1, dw1, db, da = dense_bwd_raw(x1, w1, b1, alpha, ctx.p, ctx.seed, g, need[0], True, None, ws)
        dx2 = dw2 = None
        if x2 is not None:
            dx2, dw2, _db2, _ = dense_bwd_raw(x2, w2, None, None, ctx.p, ctx.seed, g, need[3], False, None, ws)
        return (dx1, dw1, db if has_b1 else None, dx2, dw2, db if has_b2 else None,
                da, g if has_res else None, None, None)


def dense(x, weight, bias=None, alpha=None, p=0.0, residual=None, x2=None, weight2=None, bias2=None,
          seed=None):
    """Y = dropout(prelu(x W^T [+ x2 W2^T] + b [+ b2], alpha), p) [+ residual] in one kernel
    (64 -> 64 and 128 -> 128 layers).  alpha None: no activation; p == 0: no dropout."""
    p = float(p)
    if seed is None:
        seed = next_dropout_seed() if p > 0.0 else 0
    y = _Dense.apply(x, weight, bias, x2, weight2, bias2, alpha, residual, p, int(seed))
    return y


class _ModalSum(torch.autograd.Function):
    @staticmethod
    def forward(ctx, alphas, *feats):
        _need_cuda(alphas, *feats)
        feats = [_f32c(f) for f in feats]
        alphas_c = _f32c(alphas)
        M, n = len(feats), feats[0].numel()
        out = torch.empty_like(feats[0])
        arr = (ctypes.c_void_p * M)(*[f.data_ptr() for f in feats])
        _ffi.call("rk_modal_sum_fwd_f32", arr, c_int32(M), ptr(alphas_c), c_int64(n), ptr(out),
                  stream_ptr(out.device))
        ctx.save_for_backward(alphas_c, *feats)
        return out

    @staticmethod
    def backward(ctx, g):
        alphas, *feats = ctx.saved_tensors
        g = _f32c(g)
        M, n = len(feats), feats[0].numel()
        need = ctx.needs_input_grad
        dfs = [torch.empty_like(f) if need[1 + m] else None for m, f in enumerate(feats)]
        dalpha = torch.empty(M, dtype=torch.float32, device=g.device)
        lib = _ffi.load()
        ws = torch.empty(lib.rk_modal_sum_bwd_ws_bytes() // 4, dtype=torch.float32, device=g.device)
        farr = (ctypes.c_void_p * M)(*[f.data_ptr() for f in feats])
        darr = (ctypes.c_void_p * M)(*[(d.data_ptr() if d is not None else None) for d in dfs])
        _ffi.call("rk_modal_sum_bwd_f32", farr, darr, c_int32(M), ptr(alphas), ptr(g), c_int64(n), ptr(dalpha),
                  ptr(ws), c_size_t(ws.numel() * 4), stream_ptr(g.device))
        return (dalpha if need[0] else None, *dfs)


def modal_sum_supported(feats, alphas):
    return (1 <= len(feats) <= 4 and alphas.dim() == 1 and alphas.numel() == len(feats) and alphas.is_cuda
            and all(f.is_cuda and f.dtype == torch.float32 and f.shape == feats[0].shape for f in feats)
            and feats[0].numel() % 4 == 0)


def modal_sum(feats, alphas):
    """sum_m softmax(alphas)_m * feats[m] (model/test/out_Layer.py:52-54) in one pass."""
    if not modal_sum_supported(feats, alphas):
        raise _ffi.RkError(_ffi.RK_ERR_UNSUPPORTED,
                           f"modality-weighted sum of {len(feats)} matrices: built for 1..4 equal-shaped fp32 CUDA "
                           "matrices (element count a multiple of 4) and one weight per matrix")
    return _ModalSum.apply(alphas, *feats)


# --------------------------------------------------------------------------
# E1 / M: evaluation
# --------------------------------------------------------------------------
_TOPK_ENTRY = {"auto": "rk_score_topk_f32", "exact": "rk_score_topk_exact_f32",
               "filter": "rk_score_topk_filtered_f32"}


def score_topk(U, I, K, mask_ptr=None, mask_items=None, item_id_offset=0, mode="auto"):
    """Top-K items per user of U @ I.T with masked pairs scored -1e10, without
    materialising the score matrix.  Returns (scores [U,K] fp32, ids [U,K] int32).
    mode: "exact" ranks every score on the exact kernel; "filter" ranks fp16 scores first and
    re-scores the candidates exactly (same result, faster on large problems); "auto" picks by
    size."""
    _need_cuda(U, I, mask_ptr, mask_items)
    U, I = _f32c(U.detach()), _f32c(I.detach())
    if mask_items is None or mask_items.numel() == 0:
        mask_ptr = mask_items = None
    n_users, D = U.shape
    n_items = I.shape[0]
    lib = _ffi.load()
    ws_bytes = lib.rk_score_topk_ws_bytes(n_users, n_items, D, K)
    ws = torch.empty(ws_bytes // 4 + 4, dtype=torch.float32, device=U.device)
    out_s = torch.empty((n_users, K), dtype=torch.float32, device=U.device)
    out_i = torch.empty((n_users, K), dtype=torch.int32, device=U.device)
    _ffi.call(_TOPK_ENTRY[mode],
              ptr(U), c_int32(n_users), ptr(I), c_int32(n_items), c_int32(D),
              ptr(mask_ptr), ptr(mask_items), c_int32(K), c_int32(item_id_offset), ptr(out_s),
              ptr(out_i), ptr(ws), c_size_t(ws.numel() * 4), stream_ptr(U.device))
    return out_s, out_i


def score_topk_open_users(U, I, K, mask_ptr=None, mask_items=None):
    """How many users one filtered ranking of these inputs leaves to the exact kernel (those the
    candidate filter cannot certify, and the users with the longest training lists, which skip
    it by rule).  The ranking's results do not depend on it -- its speed does.  Runs one
    rk_score_topk_filtered_f32 call and synchronises."""
    _need_cuda(U, I, mask_ptr, mask_items)
    U, I = _f32c(U.detach()), _f32c(I.detach())
    if mask_items is None or mask_items.numel() == 0:
        mask_ptr = mask_items = None
    n_users, D = U.shape
    n_items = I.shape[0]
    lib = _ffi.load()
    ws = torch.empty(lib.rk_score_topk_ws_bytes(n_users, n_items, D, K) // 4 + 4, dtype=torch.float32, device=U.device)
    out_s = torch.empty((n_users, K), dtype=torch.float32, device=U.device)
    out_i = torch.empty((n_users, K), dtype=torch.int32, device=U.device)
    _ffi.call("rk_score_topk_filtered_f32", ptr(U), c_int32(n_users), ptr(I), c_int32(n_items), c_int32(D),
              ptr(mask_ptr), ptr(mask_items), c_int32(K), c_int32(0), ptr(out_s), ptr(out_i), ptr(ws),
              c_size_t(ws.numel() * 4), stream_ptr(U.device))
    n_open = c_int32(-1)
    _ffi.call("rk_score_topk_open_users", c_int32(n_users), c_int32(n_items), c_int32(D), c_int32(K), ptr(ws),
              c_size_t(ws.numel() * 4), stream_ptr(U.device), ctypes.byref(n_open))
    return int(n_open.value)


def rows_topk(scores, K):
    """torch.topk(scores, K, dim=1, largest=True, sorted=True) with ties broken by the lower
    column id.  Returns (values [B,K] fp32, ids [B,K] int32)."""
    _need_cuda(scores)
    scores = _f32c(scores.detach())
    n_rows, n_items = scores.shape
    lib = _ffi.load()
    ws_bytes = lib.rk_rows_topk_ws_bytes(n_rows, K)
    ws = torch.empty(ws_bytes // 4 + 4, dtype=torch.float32, device=scores.device)
    out_s = torch.empty((n_rows, K), dtype=torch.float32, device=scores.device)
    out_i = torch.empty((n_rows, K), dtype=torch.int32, device=scores.device)
    _ffi.call("rk_rows_topk_f32", ptr(scores), c_int32(n_rows), c_int32(n_items), c_int64(scores.stride(0)),
              c_int32(K), ptr(out_s), ptr(out_i), ptr(ws), c_size_t(ws.numel() * 4),
              stream_ptr(scores.device))
    return out_s, out_i


def score_topk_wide(A, B, K):
    """The K largest <A[a], B[b]> over b for every row a of A (score descending, ties by lower
    b), for feature rows of any width: the width is zero-padded to a multiple of 64 and walked in
    chunks inside the kernel, no block of A @ B.T exists.  model/sgrec/graph.py:102-105.
    Returns (values [n_a, K] fp32, ids [n_a, K] int32)."""
    _need_cuda(A, B)
    A, B = _f32c(A.detach()), _f32c(B.detach())
    if A.dim() != 2 or B.dim() != 2 or A.shape[1] != B.shape[1]:
        raise ValueError("score_topk_wide: A [n_a, F] and B [n_b, F] must share the width")
    F = A.shape[1]
    if F % 64:
        same = A.data_ptr() == B.data_ptr()
        A = torch.nn.functional.pad(A, (0, 64 - F % 64))
        B = A if same else torch.nn.functional.pad(B, (0, 64 - F % 64))
    n_a, n_b = A.shape[0], B.shape[0]
    lib = _ffi.load()
    ws_bytes = lib.rk_score_topk_wide_ws_bytes(n_a, n_b, K)
    ws = torch.empty(ws_bytes // 4 + 4, dtype=torch.float32, device=A.device)
    out_s = torch.empty((n_a, K), dtype=torch.float32, device=A.device)
    out_i = torch.empty((n_a, K), dtype=torch.int32, device=A.device)
    _ffi.call("rk_score_topk_wide_f32", ptr(A), c_int32(n_a), ptr(B), c_int32(n_b), c_int32(A.shape[1]), c_int32(K),
              ptr(out_s), ptr(out_i), ptr(ws), c_size_t(ws.numel() * 4), stream_ptr(A.device))
    return out_s, out_i


def topk_merge(scores, ids, K):
    """Merge candidate lists [U, P, Kin] into the best K per user by (score desc, id asc)."""
    _need_cuda(scores, ids)
    scores = _f32c(scores)
    ids = ids.to(torch.int32).contiguous()
    n_users, P, Kin = scores.shape
    out_s = torch.empty((n_users, K), dtype=torch.float32, device=scores.device)
    out_i = torch.empty((n_users, K), dtype=torch.int32, device=scores.device)
    _ffi.call("rk_topk_merge_f32", ptr(scores), ptr(ids), c_int32(n_users), c_int32(P),
              c_int32(Kin), c_int32(K), ptr(out_s), ptr(out_i), stream_ptr(scores.device))
    return out_s, out_i


_disc_cache = {}


def _discounts(kmax, dev):
    """1 / log2(rank + 1) for ranks 1..kmax exactly as evalue/metric.py:130-131 builds them (fp32
    torch ops on the host), kept on the device: a host-to-device copy per evaluation would make the
    host wait for everything queued before it."""
    key = (int(kmax), str(dev))
    d = _disc_cache.get(key)
    if d is None:
        d = (1.0 / torch.log2(torch.arange(1, kmax + 1).float() + 1.0)).to(dev)
        _disc_cache[key] = d
    return d


def rank_metrics(topk_ids, tgt_ptr, tgt_items, k_list, mean_denominator=None, return_sums=False):
    """evaluate_all_at_k for all k at once -> fp32 tensor [len(k_list), 6]
    (precision, recall, hit_rate, ndcg, map, mrr), means over the rows."""
    _need_cuda(topk_ids, tgt_ptr, tgt_items)
    topk_ids = topk_ids.to(torch.int32).contiguous()
    if tgt_items.numel() == 0:
        tgt_items = torch.zeros(1, dtype=torch.int32, device=topk_ids.device)
    n_users, stride = topk_ids.shape
    nk = len(k_list)
    dev = topk_ids.device
    kmax = max(k_list)
    disc = _discounts(kmax, dev)
    lib = _ffi.load()
    ws_bytes = lib.rk_rank_metrics_ws_bytes(n_users, nk)
    ws = torch.empty(ws_bytes // 8 + 2, dtype=torch.float64, device=dev)
    out = torch.empty((nk, 6), dtype=torch.float32, device=dev)
    sums = torch.empty((nk, 6), dtype=torch.float64, device=dev) if return_sums else None
    kl = (c_int32 * nk)(*[int(k) for k in k_list])
    denom = int(mean_denominator) if mean_denominator is not None else n_users
    _ffi.call("rk_rank_metrics_f32", ptr(topk_ids), c_int32(n_users), c_int32(stride),
              ptr(tgt_ptr), ptr(tgt_items), kl, c_int32(nk), ptr(disc), ptr(out), ptr(sums),
              c_int64(denom), ptr(ws), c_size_t(ws.numel() * 8), stream_ptr(dev))
    return (out, sums) if return_sums else out
