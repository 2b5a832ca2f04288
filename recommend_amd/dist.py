"""Multi-GPU partitioning of the hot path: one process per GPU, torch.distributed
(backend "nccl" = RCCL over xGMI on the GPU box).

Full-ranking evaluation shards the ITEM table: rank r scores every user against
items [lo_r, hi_r) only and keeps its local top-K (ids are global).  Every rank
then merges the lists of its own slice of users, so the exchange is one
all-to-all: rank r sends rank q the rows of q's user slice (scores and ids packed
in one buffer), U*K*8*(W-1)/W bytes out and in per rank; an all-gather would move W
times as much for lists the receiver never reads.  Each rank reduces the metric
sums of its slice and one all-reduce of 6*len(k_list) doubles finishes the means.
Contiguous item ranges keep "lower id wins" valid across shards.

Propagation over graphs too large for one GPU shards the ROWS (RowShardedGraph at the
end of this file): one exchange of layer rows per hop, forward and backward -- an all-gather
of the whole layer, or (exchange="halo") one all-to-all that brings every rank exactly the
rows its own edges reference.

The reference is single-process (no collective anywhere in its tree); this file
has no counterpart there.  `compute` is the module that provides score_topk /
topk_merge / rank_metrics: recommend_amd.ops (HIP) in production.
"""
from typing import Dict, Optional, Sequence

import numpy as np
import torch
import torch.distributed as dist

from .evaluator import METRIC_ORDER, pairs_to_csr, restrict_csr, target_to_csr, _pad64


class DirectTransport:
    """How tensors reach the collectives: as they are.  On the GPU box the process group is "nccl"
    (= RCCL) and device tensors go over xGMI; CPU tensors with a gloo group work the same way."""

    @staticmethod
    def collective_device(device):
        """Where index tensors exchanged while a plan is built should live."""
        return torch.device(device)

    @staticmethod
    def all_gather(out_list, t, group):
        dist.all_gather(out_list, t, group=group)

    @staticmethod
    def all_gather_into(out, t, group):
        dist.all_gather_into_tensor(out, t, group=group)

    @staticmethod
    def reduce_scatter(out, full, group):
        dist.reduce_scatter_tensor(out, full, group=group)

    @staticmethod
    def all_reduce(t, op, group):
        dist.all_reduce(t, op=op, group=group)

    @staticmethod
    def all_to_all(out, src, out_splits, in_splits, group):
        dist.all_to_all_single(out, src, out_splits, in_splits, group=group)


_transport = DirectTransport()


def set_transport(transport):
    """Replace the object the collectives below go through (same methods as DirectTransport).
    Rehearsals that run several ranks on ONE GPU cannot use RCCL and hand in a transport that
    stages device tensors through host memory for a gloo group; production never calls this."""
    global _transport
    old, _transport = _transport, (transport or DirectTransport())
    return old


def all_gather_tensor(t, world, group=None):
    out = [torch.empty_like(t) for _ in range(world)]
    _transport.all_gather(out, t.contiguous(), group)
    return out


def all_gather_rows_into(buf, x_local, group=None):
    """buf[r * n : (r + 1) * n] = rank r's x_local ([n, D], the same n on every rank); buf is the
    caller's preallocated [world * n, D] tensor (no list of pieces, no concatenation)."""
    _transport.all_gather_into(buf, x_local.detach().contiguous(), group)
    return buf


def reduce_scatter_rows(out_local, full, group=None):
    """out_local ([n, D]) = sum over ranks of their full[rank * n : (rank + 1) * n]."""
    _transport.reduce_scatter(out_local, full.detach().contiguous(), group)
    return out_local


def all_reduce_tensor(t, op=dist.ReduceOp.SUM, group=None):
    _transport.all_reduce(t, op, group)
    return t


def exchange_user_slices(packed, n_users, rank, world, group=None):
    """packed: [n_users, C] rows of every user on this rank.  Returns [world, rows_of_my_slice, C]:
    the rows of this rank's user slice as computed by every rank (source-major)."""
    bounds = [user_range(n_users, r, world) for r in range(world)]
    in_split = [hi - lo for lo, hi in bounds]
    mine = in_split[rank]
    src = packed.contiguous()
    out = torch.empty((world * mine, packed.shape[1]), dtype=src.dtype, device=src.device)
    _transport.all_to_all(out, src, [mine] * world, in_split, group)
    return out.view(world, mine, packed.shape[1])


def item_range(n_items, rank, world):
    return (n_items * rank) // world, (n_items * (rank + 1)) // world


def user_range(n_users, rank, world):
    return (n_users * rank) // world, (n_users * (rank + 1)) // world


class ShardedRankingEvaluator:
    def __init__(self, target, mask, k_list: Sequence[int], metrics: Sequence[str], n_users: int,
                 n_items: int, rank: Optional[int] = None, world: Optional[int] = None,
                 group=None, compute=None):
        if compute is None:
            from . import ops as compute  # the HIP path; raises without a GPU
        self.compute = compute
        self.group = group
        self.rank = dist.get_rank(group) if rank is None else rank
        self.world = dist.get_world_size(group) if world is None else world
        self.k_list = [int(k) for k in k_list]
        self.metrics = list(metrics)
        self.n_users, self.n_items = int(n_users), int(n_items)
        self.lo, self.hi = item_range(self.n_items, self.rank, self.world)
        self.u_lo, self.u_hi = user_range(self.n_users, self.rank, self.world)
        row_id, col_id = mask
        mp, mc = pairs_to_csr(row_id, col_id, self.n_users)
        self.mask_ptr, self.mask_items = restrict_csr(mp, mc, self.lo, self.hi)
        tp, tc = target_to_csr(target, self.n_users)
        # targets of this rank's user slice only
        self.tgt_ptr = (tp[self.u_lo:self.u_hi + 1] - tp[self.u_lo]).astype(np.int32)
        self.tgt_items = tc[tp[self.u_lo]:tp[self.u_hi]].astype(np.int32)
        self._dev = {}

    def _state(self, device):
        device = torch.device(device)
        st = self._dev.get(device)
        if st is None:
            to = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
            st = {k: to(getattr(self, k)) for k in ("mask_ptr", "mask_items", "tgt_ptr", "tgt_items")}
            self._dev[device] = st
        return st

    def local_topk(self, user_embeddings, item_embeddings_full, k):
        st = self._state(user_embeddings.device)
        shard = item_embeddings_full[self.lo:self.hi]
        return self.compute.score_topk(_pad64(user_embeddings), _pad64(shard), k, st["mask_ptr"],
                                       st["mask_items"], item_id_offset=self.lo)

    def evaluate(self, user_embeddings, item_embeddings_full) -> Dict[str, torch.Tensor]:
        kmax = max(self.k_list)
        if kmax > self.hi - self.lo:
            raise ValueError("k must be in the range [1, num_items of the smallest shard]")
        st = self._state(user_embeddings.device)
        s, ids = self.local_topk(user_embeddings, item_embeddings_full, kmax)
        # one exchange step: every rank receives the lists of its own user slice from every rank
        packed = torch.cat([s.contiguous(), ids.contiguous().view(torch.float32)], dim=1)     # [U, 2K]
        got = exchange_user_slices(packed, self.n_users, self.rank, self.world, self.group)  # [W, u, 2K]
        cand_s = got[:, :, :kmax].permute(1, 0, 2).contiguous()                              # [u, W, K]
        cand_i = got[:, :, kmax:].contiguous().view(torch.int32).permute(1, 0, 2).contiguous()
        sums = torch.zeros((len(self.k_list), 6), dtype=torch.float64, device=s.device)
        self.last_ids = None
        if self.u_hi > self.u_lo:
            _ms, mi = self.compute.topk_merge(cand_s, cand_i, kmax)
            self.last_ids = mi
            _out, sums = self.compute.rank_metrics(mi, st["tgt_ptr"], st["tgt_items"], self.k_list,
                                                   mean_denominator=self.n_users, return_sums=True)
        all_reduce_tensor(sums, dist.ReduceOp.SUM, self.group)
        # (on the sums' device, like RankingEvaluator.evaluate: reading a value is what synchronises)
        table = (sums / float(self.n_users)).to(torch.float32)
        out = {}
        for ki, k in enumerate(self.k_list):
            row = table[ki]
            for name in self.metrics:
                out[f"{name}@{k}"] = row[METRIC_ORDER.index(name)]
        return out


class UserShardedRankingEvaluator:
    """Full-ranking evaluation with the USERS cut over the ranks: every rank ranks its own block of
    users against the whole item table (which every rank holds) and only the 18 metric sums cross
    ranks.  No candidate lists are exchanged or merged and every user is re-scored on one rank
    only; what it needs instead is the whole item table on every GPU (16 MB at Elec), so it is the
    form to use when the item table fits one GPU, and the item-sharded evaluator above when it
    does not.  Same constructor, `evaluate` signature and result as ShardedRankingEvaluator."""

    def __init__(self, target, mask, k_list: Sequence[int], metrics: Sequence[str], n_users: int,
                 n_items: int, rank: Optional[int] = None, world: Optional[int] = None,
                 group=None, compute=None):
        if compute is None:
            from . import ops as compute  # the HIP path; raises without a GPU
        self.compute = compute
        self.group = group
        self.rank = dist.get_rank(group) if rank is None else rank
        self.world = dist.get_world_size(group) if world is None else world
        self.k_list = [int(k) for k in k_list]
        self.metrics = list(metrics)
        self.n_users, self.n_items = int(n_users), int(n_items)
        self.u_lo, self.u_hi = user_range(self.n_users, self.rank, self.world)
        row_id, col_id = mask
        mp, mc = pairs_to_csr(row_id, col_id, self.n_users)
        self.mask_ptr = (mp[self.u_lo:self.u_hi + 1] - mp[self.u_lo]).astype(np.int32)
        self.mask_items = mc[mp[self.u_lo]:mp[self.u_hi]].astype(np.int32)
        tp, tc = target_to_csr(target, self.n_users)
        self.tgt_ptr = (tp[self.u_lo:self.u_hi + 1] - tp[self.u_lo]).astype(np.int32)
        self.tgt_items = tc[tp[self.u_lo]:tp[self.u_hi]].astype(np.int32)
        self._dev = {}

    _state = ShardedRankingEvaluator._state

    def local_topk(self, user_embeddings, item_embeddings_full, k):
        """Top-k of this rank's users over all items (global user rows in, local rows out)."""
        st = self._state(user_embeddings.device)
        return self.compute.score_topk(_pad64(user_embeddings[self.u_lo:self.u_hi]), _pad64(item_embeddings_full), k,
                                       st["mask_ptr"], st["mask_items"])

    def evaluate(self, user_embeddings, item_embeddings_full) -> Dict[str, torch.Tensor]:
        kmax = max(self.k_list)
        if kmax > self.n_items:
            raise ValueError("k must be in the range [1, num_items]")
        st = self._state(user_embeddings.device)
        sums = torch.zeros((len(self.k_list), 6), dtype=torch.float64, device=user_embeddings.device)
        self.last_ids = None
        if self.u_hi > self.u_lo:
            _s, ids = self.local_topk(user_embeddings, item_embeddings_full, kmax)
            self.last_ids = ids
            _out, sums = self.compute.rank_metrics(ids, st["tgt_ptr"], st["tgt_items"], self.k_list,
                                                   mean_denominator=self.n_users, return_sums=True)
        all_reduce_tensor(sums, dist.ReduceOp.SUM, self.group)
        # (on the sums' device, like RankingEvaluator.evaluate: reading a value is what synchronises)
        table = (sums / float(self.n_users)).to(torch.float32)
        out = {}
        for ki, k in enumerate(self.k_list):
            row = table[ki]
            for name in self.metrics:
                out[f"{name}@{k}"] = row[METRIC_ORDER.index(name)]
        return out


# --------------------------------------------------------------------------
# Row-sharded propagation: rank r owns the destination rows [lo_r, hi_r) of the adjacency and the
# matching rows of every layer's embeddings.  One hop needs every source row, so it is one
# all-gather of the layer (N*D*4 bytes in total) into a buffer that is allocated once, followed by
# the local SpMM with its fused alpha/beta epilogue; the backward pass is the same step with the
# transposed operand (rank r owns rows [lo_r, hi_r) of A^T), i.e. one all-gather of the gradient.
# Blocks are equal (ceil(N / world), the last one short) so that the gather needs no size exchange.
# --------------------------------------------------------------------------
def node_range(n_nodes, rank, world):
    block = -(-n_nodes // world)
    lo = min(rank * block, n_nodes)
    return lo, min(lo + block, n_nodes)


def _rows_csr(rows, cols, lo, hi, order_key=None):
    """CSR of the edges whose row falls in [lo, hi), rows re-based to lo, stable in edge order
    (order_key: the edges' positions in the full list when `rows` is a subset of it).
    Returns (rowptr int32 [hi-lo+1], col int32, positions into the given arrays)."""
    rows = np.asarray(rows, np.int64)
    sel = np.nonzero((rows >= lo) & (rows < hi))[0]
    if order_key is not None:
        sel = sel[np.argsort(np.asarray(order_key)[sel], kind="stable")]
    order = sel[np.argsort(rows[sel], kind="stable")]
    ptr = np.zeros(hi - lo + 1, np.int64)
    np.cumsum(np.bincount(rows[order] - lo, minlength=hi - lo), out=ptr[1:])
    return ptr.astype(np.int32), np.asarray(cols, np.int32)[order], order


class HipRowBlock:
    """The production compute of a row block: recommend_amd.ops SpMM kernels."""

    def __init__(self, rowptr, col, val, device, n_cols=None):
        from . import ops
        self._ops = ops
        eid = np.arange(col.shape[0], dtype=np.int32)
        self.mat = ops.CsrMatrix(rowptr, col, eid, device, n_cols=n_cols)
        self.set_values(val)

    def set_values(self, val):
        """Edge values in the block's CSR order (numpy or a tensor on the block's device)."""
        if not isinstance(val, torch.Tensor):
            val = torch.from_numpy(np.ascontiguousarray(val, np.float32))
        self.mat.set_values_from_coo(val.to(self.mat.device, torch.float32))

    def matmul(self, h_full, h0=None, alpha=0.0, beta=1.0):
        """beta * (A_rows h_full) [+ alpha * h0], the epilogue fused into the SpMM kernel."""
        return self._ops.spmm_raw(self.mat, h_full, h0, alpha, beta)

    def remap_columns(self, col, n_cols):
        """The same rows and values with other column ids (same order): the operand then gathers
        from a matrix of n_cols rows (HaloPlan)."""
        self.mat.idx = torch.from_numpy(np.ascontiguousarray(col if len(col) else np.zeros(1, np.int32), np.int32)).to(self.mat.device)
        self.mat.n_cols = int(n_cols)


class HaloPlan:
    """Which rows of the other ranks' blocks the columns of one row block reference, and the
    exchange that fetches exactly those: rank r sends rank q the rows of its block that q's edges
    read, in one all_to_all_single per hop (sum over ranks of their halo rows x D x 4 bytes
    instead of the N x D x 4 of an all-gather).  The block's column ids are renumbered to the
    compact matrix [own rows | rows from rank 0 | rows from rank 1 | ...] the exchange fills, so no
    rank holds a buffer of N rows.

    What it buys depends on the graph: a rank's halo is the set of distinct columns outside its own
    block, at most min(N - n_local, edges of the block).  A graph whose row blocks reference most
    of the other nodes (the synthetic scale-up shape: 61 M edges per block into 12 M nodes) moves
    nearly as much as the all-gather; graphs with locality, or many ranks with few edges each, move
    a fraction."""

    def __init__(self, cols, lo, hi, block, rank, world, group=None, device="cpu"):
        cols = np.asarray(cols, np.int64)
        self.lo, self.hi, self.block, self.rank, self.world, self.group = lo, hi, block, rank, world, group
        self.n_local = hi - lo
        outside = cols[(cols < lo) | (cols >= hi)]
        self.need = np.unique(outside)                      # ascending global ids = grouped by owner rank
        owner = self.need // block
        recv_counts = np.bincount(owner, minlength=world).astype(np.int64)
        cdev = _transport.collective_device(device)
        # every owner learns how many, then which, of its rows this rank reads
        rc = torch.from_numpy(recv_counts).to(cdev)
        sc = torch.empty_like(rc)
        _transport.all_to_all(sc, rc, None, None, group)
        send_counts = sc.cpu().numpy()
        want = torch.from_numpy(self.need - owner * block).to(cdev)      # row numbers inside the owner's block
        give = torch.empty(int(send_counts.sum()), dtype=torch.int64, device=cdev)
        _transport.all_to_all(give, want, [int(c) for c in send_counts], [int(c) for c in recv_counts], group)
        self.send_idx = give.to(device)                     # rows of x_local to pack, by destination rank
        self.send_splits = [int(c) for c in send_counts]
        self.recv_splits = [int(c) for c in recv_counts]
        self.n_halo = int(self.need.shape[0])
        self.n_send = int(send_counts.sum())
        self.device = torch.device(device)
        self._bufs = {}

    def remap(self, cols):
        """Global column ids -> rows of the compact matrix."""
        cols = np.asarray(cols, np.int64)
        own = (cols >= self.lo) & (cols < self.hi)
        out = np.empty(cols.shape[0], np.int64)
        out[own] = cols[own] - self.lo
        out[~own] = self.n_local + np.searchsorted(self.need, cols[~own])
        return out.astype(np.int32)

    def exchange(self, x_local):
        """[n_local, D] on every rank -> [n_local + n_halo, D]: the rank's own rows followed by the
        rows it reads from the others.  The result lives in a buffer that the next exchange of the
        same width overwrites."""
        x = x_local.detach()
        D = x.shape[1]
        key = (D, x.dtype)
        bufs = self._bufs.get(key)
        if bufs is None:
            bufs = (torch.empty((self.n_local + self.n_halo, D), dtype=x.dtype, device=x.device),
                    torch.empty((self.n_send, D), dtype=x.dtype, device=x.device))
            self._bufs[key] = bufs
        buf, send = bufs
        buf[: self.n_local].copy_(x)
        if self.n_send:
            torch.index_select(x, 0, self.send_idx, out=send)
        _transport.all_to_all(buf[self.n_local:], send, self.recv_splits, self.send_splits, self.group)
        return buf


class _ShardedHop(torch.autograd.Function):
    """out_local = alpha * h0_local + beta * (A_rows gather(h_local)).  Backward:
    d h_local = beta * ((A^T)_rows gather(g_local)), d h0_local = alpha * g_local."""

    @staticmethod
    def forward(ctx, h_local, h0_local, graph, alpha, beta):
        ctx.graph, ctx.alpha, ctx.beta = graph, alpha, beta
        ctx.has_h0 = h0_local is not None
        return graph.fwd.matmul(graph.inputs(h_local, "fwd"), h0_local, alpha, beta)

    @staticmethod
    def backward(ctx, g_local):
        graph = ctx.graph
        g_local = g_local.contiguous()
        gh = gh0 = None
        if ctx.needs_input_grad[0]:
            gh = graph.bwd.matmul(graph.inputs(g_local, "bwd"), None, 0.0, ctx.beta)
        if ctx.has_h0 and ctx.needs_input_grad[1]:
            gh0 = g_local * ctx.alpha
        return gh, gh0, None, None, None


class RowShardedGraph:
    """Rank r's rows of a graph and of its transpose.

    RowShardedGraph(src, dst, w, n_nodes, ...): every rank reads the same full COO edge list (with
    its normalised values) and keeps its rows only.  RowShardedGraph.from_local_edges(...): the rank
    hands over only the edges it owns (no rank holds the whole list).
    `block_factory(rowptr, col, val, device, n_cols=...)` returns an object with
    `matmul(h_full, h0, alpha, beta)` and `remap_columns(col, n_cols)`; HipRowBlock in production.
    exchange: "allgather" (every hop gathers the whole layer into an N-row buffer) or "halo" (every
    hop brings exactly the rows this rank's edges read: HaloPlan; `enable_halo()` switches a graph
    over after anything that needs global column ids -- the normalisation -- has run)."""

    def __init__(self, src, dst, w, n_nodes, rank=None, world=None, group=None, device="cuda",
                 block_factory=None, exchange="allgather"):
        self._init_partition(n_nodes, rank, world, group, device)
        factory = block_factory or HipRowBlock
        w = np.asarray(w, np.float32)
        p, c, e = _rows_csr(dst, src, self.lo, self.hi)       # out[v] = sum over edges u -> v
        self.fwd = factory(p, c, w[e], device, n_cols=self.n_nodes)
        self._cols["fwd"] = c
        p, c, e = _rows_csr(src, dst, self.lo, self.hi)       # rows of the transposed operand
        self.bwd = factory(p, c, w[e], device, n_cols=self.n_nodes)
        self._cols["bwd"] = c
        if exchange == "halo":
            self.enable_halo()
        elif exchange != "allgather":
            raise ValueError(f"unknown exchange {exchange!r} (allgather, halo)")

    def _init_partition(self, n_nodes, rank, world, group, device):
        self.group = group
        self.rank = dist.get_rank(group) if rank is None else rank
        self.world = dist.get_world_size(group) if world is None else world
        self.n_nodes = int(n_nodes)
        self.block = -(-self.n_nodes // self.world)
        self.lo, self.hi = node_range(self.n_nodes, self.rank, self.world)
        self.device = torch.device(device)
        self._gather_buf = {}
        self._cols = {}          # global column ids of the two blocks (host), until a halo plan renumbers them
        self._plans = {}
        self.exchange = "allgather"

    @classmethod
    def from_local_edges(cls, fwd_rows, bwd_rows, n_nodes, rank=None, world=None, group=None, device="cuda",
                         block_factory=None):
        """fwd_rows = (rowptr, col, val) of this rank's destination rows (columns = global source
        ids), bwd_rows = the same for its rows of the transpose (rows = sources it owns, columns =
        global destination ids)."""
        g = cls.__new__(cls)
        g._init_partition(n_nodes, rank, world, group, device)
        factory = block_factory or HipRowBlock
        g.fwd = factory(*fwd_rows, device, n_cols=g.n_nodes)
        g.bwd = factory(*bwd_rows, device, n_cols=g.n_nodes)
        g._cols = {"fwd": np.asarray(fwd_rows[1]), "bwd": np.asarray(bwd_rows[1])}
        return g

    def enable_halo(self):
        """Switch the hops to the halo exchange: the blocks' columns are renumbered to the compact
        [own rows | halo rows] matrices the exchanges fill (a collective: every rank calls it)."""
        if self.world == 1 or self.exchange == "halo":
            return self
        for name, block in (("fwd", self.fwd), ("bwd", self.bwd)):
            plan = HaloPlan(self._cols[name], self.lo, self.hi, self.block, self.rank, self.world, self.group, self.device)
            block.remap_columns(plan.remap(self._cols[name]), plan.n_local + plan.n_halo)
            self._plans[name] = plan
        self._cols = {}
        self.exchange = "halo"
        return self

    def halo_rows(self):
        """(rows received per forward hop, rows received per backward hop) of this rank; an
        all-gather receives n_nodes - n_local."""
        return tuple(self._plans[n].n_halo for n in ("fwd", "bwd")) if self.exchange == "halo" else None

    def inputs(self, x_local, which):
        """The matrix the block `which` ("fwd" / "bwd") multiplies: this rank's rows of the layer
        completed by the rows of the other ranks it reads."""
        if self.exchange == "halo":
            return self._plans[which].exchange(x_local)
        return self.gather_rows(x_local)

    def local_rows(self, x_full):
        return x_full[self.lo:self.hi]

    def gather_rows(self, x_local):
        """[hi-lo, D] on every rank -> [world * block, D] on every rank: rows [0, N) are the full
        matrix, the rows behind them padding of the last rank's short block.  The result lives in
        a buffer that the next gather of the same width overwrites."""
        if self.world == 1:
            return x_local
        n_loc, D = x_local.shape
        key = (D, x_local.dtype, x_local.device)
        bufs = self._gather_buf.get(key)
        if bufs is None:
            bufs = (torch.zeros((self.world * self.block, D), dtype=x_local.dtype, device=x_local.device),
                    torch.zeros((self.block, D), dtype=x_local.dtype, device=x_local.device))
            self._gather_buf[key] = bufs
        full, send = bufs
        if n_loc < self.block:        # the last rank's block is short: send it padded (the pad stays zero)
            send[:n_loc].copy_(x_local.detach())
            x_local = send
        return all_gather_rows_into(full, x_local, self.group)

    def hops(self):
        """The propagation on raw tensors for callers that run their own backward pass
        (recommend_amd.mgcn_op): forward(z) = A_rows gather(z), backward(g) = (A^T)_rows gather(g)."""
        graph = self

        class _Hops:
            @staticmethod
            def forward(z):
                return graph.fwd.matmul(graph.inputs(z, "fwd"))

            @staticmethod
            def backward(g):
                return graph.bwd.matmul(graph.inputs(g, "bwd"))
        return _Hops

    def hop(self, h_local, h0_local=None, alpha=0.0, beta=1.0):
        """One hop on this rank's rows, h <- alpha*h0 + beta*(A h) (differentiable)."""
        return _ShardedHop.apply(h_local, h0_local, self, float(alpha), float(beta))

    def aggrate(self, x_local, k, alpha):
        """k hops of h <- alpha*h0 + (1-alpha)*A h (graph_constructor.py:189-197), rows sharded."""
        h = x_local
        for _ in range(k):
            h = self.hop(h, x_local, alpha, 1.0 - alpha)
        return h

    def agg_func(self, x_local, k, f, dropout, active, layer=None):
        """k hops of h <- A dropout(active(f(h))) (graph_constructor.py:199-208) on this rank's rows:
        the node-wise layer is local, the hop gathers.  `layer(f, h, active, dropout)`: the fused
        layer (recommend_amd.fused.linear_act unless given)."""
        if layer is None:
            from . import fused
            layer = fused.linear_act
        h = x_local
        for _ in range(k):
            h = self.hop(layer(f, h, active, dropout))
        return h
