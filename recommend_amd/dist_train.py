"""The MMFCN training step over several GPUs: rows of the graph and of every node matrix sharded,
the batch split, gradients exchanged with RCCL collectives (torch.distributed, backend "nccl").

The reference trains in one process (train/graph_trainer.py:133-165; no collective anywhere in
its tree); this file has no counterpart there.  What it keeps is the step's arithmetic: with the
same parameters and batch, the ranks together compute the loss, gradients and updated parameters
of the single-process step.

Layout.  Rank r owns the node rows [lo_r, hi_r) (equal blocks, users first as in the reference's
node numbering): its rows of the adjacency and of the transpose (ShardedGraphConstructor), of the
projected features, of the id-embedding table and of every layer's activations.  The node-wise
layers and the modality projection are row-local; their weights are replicated.  One hop is one
all-gather of the layer plus the local SpMM (dist.RowShardedGraph).  The final node embeddings are
gathered once more (the loss reads arbitrary rows); the batch is split, every rank scatter-adds
its samples' gradients into a full-height buffer and one reduce-scatter hands every rank the
gradient rows it owns.  Weight gradients are partial sums over a rank's rows: one all-reduce of
all of them in one flat bucket.  The clipping norm is computed from the reduced gradients, so
every rank scales alike and the replicated parameters stay equal without a broadcast.
"""
from typing import Callable, Dict, List, Optional

import numpy as np
import torch
import torch.distributed as dist
import torch.nn as nn

from . import fused
from .dist import (HipRowBlock, RowShardedGraph, _rows_csr, all_gather_rows_into, all_reduce_tensor,
                   node_range, reduce_scatter_rows)
from .model import TEST, TESTModel, _ItemProjection


class _HipNormalizer:
    """Production compute of the sharded normalisation: the two halves of rk_sym_normalize_f32."""

    @staticmethod
    def in_weight_sqrt(block, w_local):
        from . import ops
        return ops.in_weight_sqrt(block.mat, w_local)

    @staticmethod
    def sym_scale_rows(block, w_local, row_offset, sqrt_in_w):
        from . import ops
        return ops.sym_scale_rows(block.mat, w_local, row_offset, sqrt_in_w)


class ShardedGraphConstructor:
    """GraphConstructor (model/test/graph_constructor.py:18-208) for one rank's rows.

    build_graph takes (users, items, ratings) arrays -- the whole interaction list or any part of
    it that contains every interaction touching this rank's nodes, with `index` giving the rows'
    positions in the whole list (edge order inside a row follows the reference's edge ids).  No
    rank needs the whole edge list.  creat_feature_weight normalises as graph_constructor.py:
    158-187 does: incoming-weight sums of the local rows, one all-gather of the N square roots,
    then w / (sqrt_in[src] * sqrt_in[dst]) on the local edges of both operands."""

    def __init__(self, rank=None, world=None, group=None, device="cuda", add_self_loops=True,
                 normalize_adj=True, block_factory=None, normalizer=None, exchange="allgather"):
        if exchange not in ("allgather", "halo"):
            raise ValueError(f"unknown exchange {exchange!r} (allgather, halo)")
        self.exchange = exchange        # how a hop gets the rows of other ranks: dist.RowShardedGraph
        self.group = group
        self.rank = dist.get_rank(group) if rank is None else rank
        self.world = dist.get_world_size(group) if world is None else world
        self.device = torch.device(device)
        self.add_self_loops = add_self_loops
        self.normalize_adj = normalize_adj
        self.block_factory = block_factory or HipRowBlock
        self.normalizer = normalizer or _HipNormalizer
        self.graph: Optional[RowShardedGraph] = None
        self.weight = "weight"

    def build_graph(self, interactions, num_users: int, num_items: int, index=None, n_total=None):
        users = np.asarray(interactions[0], np.int64)
        items = np.asarray(interactions[1], np.int64)
        ratings = (np.ones(users.shape[0], np.float32) if interactions[2] is None
                   else np.asarray(interactions[2], np.float32))
        self.num_users, self.num_items = int(num_users), int(num_items)
        self.num_nodes = N = self.num_users + self.num_items
        self.lo, self.hi = node_range(N, self.rank, self.world)
        index = np.arange(users.shape[0], dtype=np.int64) if index is None else np.asarray(index, np.int64)
        n_total = int(n_total) if n_total is not None else int(index.max()) + 1 if index.size else 0
        # the self-loop weight is the largest rating of the WHOLE list (graph_constructor.py:91-98)
        mx = torch.tensor([float(ratings.max()) if ratings.size else 0.0], dtype=torch.float64)
        if self.world > 1:
            dist.all_reduce(mx, op=dist.ReduceOp.MAX, group=self.group)
        max_rating = np.float32(max(0.0, float(mx)))
        inode = items + self.num_users
        # edge 2j: u -> i+U, edge 2j+1: i+U -> u, edge 2E+n: n -> n (the reference's edge ids)
        src = np.concatenate([users, inode])
        dst = np.concatenate([inode, users])
        w = np.concatenate([ratings, ratings])
        gid = np.concatenate([2 * index, 2 * index + 1])
        if self.add_self_loops:
            own = np.arange(self.lo, self.hi, dtype=np.int64)
            src, dst = np.concatenate([src, own]), np.concatenate([dst, own])
            w = np.concatenate([w, np.full(own.shape[0], max_rating, np.float32)])
            gid = np.concatenate([gid, 2 * n_total + own])
        self._w_raw = {}
        blocks = {}
        for name, rows, cols in (("fwd", dst, src), ("bwd", src, dst)):
            p, c, e = _rows_csr(rows, cols, self.lo, self.hi, order_key=gid)
            self._w_raw[name] = torch.from_numpy(np.ascontiguousarray(w[e], np.float32))
            blocks[name] = (p, c, w[e])
        self.graph = RowShardedGraph.from_local_edges(blocks["fwd"], blocks["bwd"], N, self.rank, self.world,
                                                      self.group, self.device, self.block_factory)
        self.num_local_edges = int(blocks["fwd"][1].shape[0])
        return self.graph

    def move_to_device(self, device=None):
        return None      # the blocks are built on their device

    def creat_feature_weight(self, feature=None):
        if feature is not None:
            raise NotImplementedError("feature-similarity edge weights are not built for the sharded graph")
        if not self.normalize_adj:
            if self.exchange == "halo":
                self.graph.enable_halo()
            return
        g, nz = self.graph, self.normalizer
        w_f = self._w_raw["fwd"].to(self.device)
        w_b = self._w_raw["bwd"].to(self.device)
        sq_local = nz.in_weight_sqrt(g.fwd, w_f)                       # [hi - lo]
        send = torch.ones(g.block, dtype=torch.float32, device=sq_local.device)
        send[: sq_local.shape[0]] = sq_local
        sq_full = torch.empty(g.world * g.block, dtype=torch.float32, device=sq_local.device)
        if g.world > 1:
            all_gather_rows_into(sq_full.view(-1, 1), send.view(-1, 1), self.group)
        else:
            sq_full.copy_(send)
        self.sqrt_in_w = sq_full[: self.num_nodes]
        g.fwd.set_values(nz.sym_scale_rows(g.fwd, w_f, self.lo, sq_full))
        g.bwd.set_values(nz.sym_scale_rows(g.bwd, w_b, self.lo, sq_full))
        # (the normalisation reads the blocks' GLOBAL column ids; the halo plan renumbers them)
        if self.exchange == "halo":
            g.enable_halo()

    # the calls the network blocks make on the graph object
    def agg_func(self, node_x, k, f, dropout, active):
        return self.graph.agg_func(node_x, k, f, dropout, active)

    def aggrate(self, node_x, k, alpha):
        return self.graph.aggrate(node_x, k, alpha)

    def propagate(self, node_x):
        return self.graph.hop(node_x)

    def hops(self):
        return self.graph.hops()


class _GatherRows(torch.autograd.Function):
    """[hi-lo, D] on every rank -> the full [N, D] matrix on every rank.  Backward: every rank holds
    a gradient of the full matrix (its own samples' contributions); the owner of a row needs their
    sum, which is one reduce-scatter."""

    @staticmethod
    def forward(ctx, x_local, graph):
        ctx.graph, ctx.n_loc = graph, x_local.shape[0]
        if graph.world == 1:
            return x_local.clone()
        D = x_local.shape[1]
        send = x_local
        if x_local.shape[0] < graph.block:
            send = torch.zeros((graph.block, D), dtype=x_local.dtype, device=x_local.device)
            send[: x_local.shape[0]] = x_local
        full = torch.empty((graph.world * graph.block, D), dtype=x_local.dtype, device=x_local.device)
        all_gather_rows_into(full, send, graph.group)
        return full[: graph.n_nodes]

    @staticmethod
    def backward(ctx, g_full):
        graph = ctx.graph
        if graph.world == 1:
            return g_full, None
        D = g_full.shape[1]
        padded = g_full
        if g_full.shape[0] < graph.world * graph.block:
            padded = torch.zeros((graph.world * graph.block, D), dtype=g_full.dtype, device=g_full.device)
            padded[: g_full.shape[0]] = g_full
        out = torch.empty((graph.block, D), dtype=g_full.dtype, device=g_full.device)
        reduce_scatter_rows(out, padded, graph.group)
        return out[: ctx.n_loc], None


def gather_rows(x_local, graph):
    return _GatherRows.apply(x_local, graph)


class ShardedTESTModel(nn.Module):
    """TESTModel (model/Main_Model.py:32-170) on one rank's node rows.  Same parameters (names and
    shapes: a TESTModel state dict loads), same forward contract: `forward(batch)` returns the full
    `user_embeddings [U, D]` / `item_embeddings [I, D]` on every rank, autograd-connected to the
    rank's rows through the gather."""

    def __init__(self, config, user_features=None, item_features=None, rank=None, world=None, group=None,
                 block_factory=None, normalizer=None, exchange="allgather"):
        super().__init__()
        if user_features:
            raise NotImplementedError("user-side features are not sharded yet")
        self.config = config
        self.group = group
        self.rank = dist.get_rank(group) if rank is None else rank
        self.world = dist.get_world_size(group) if world is None else world
        self.num_users, self.num_items = config.data.num_users, config.data.num_items
        self.num_nodes = N = self.num_users + self.num_items
        self.modal_num = config.model.modal_num
        device = config.system.device
        emb_dim = config.model.emb_dim
        self.lo, self.hi = node_range(N, self.rank, self.world)
        U = self.num_users
        self.users_local = max(0, min(self.hi, U) - min(self.lo, U))          # user rows of the block
        i_lo, i_hi = max(self.lo, U) - U, max(self.hi, U) - U                  # item ids of the block
        self.item_features = item_features or {}
        self.feats = [f[i_lo:i_hi].to(device).float().contiguous() for f in self.item_features.values()]
        self.linears = nn.ModuleList([
            _ItemProjection(x.shape[1], emb_dim, self.users_local) if x.shape[1] != emb_dim else nn.Identity()
            for x in self.feats])
        self.ks = [5] * self.modal_num
        self.alphas = torch.randn(self.modal_num, requires_grad=True, device=device)
        # the same draws as the single-process model makes, of which this rank keeps its rows
        self._first_emb = nn.Embedding(N, emb_dim).weight.detach()[self.lo:self.hi].to(device).contiguous()
        self.test = TEST(modal_num=self.modal_num, emb_num=emb_dim, layer_num=getattr(config.model, "layer_num", 2),
                         dropout=getattr(config.model, "dropout", 0.3))
        self.graph_constructor = ShardedGraphConstructor(
            self.rank, self.world, group, device, add_self_loops=config.graph.add_self_loops,
            normalize_adj=config.graph.normalize_adj, block_factory=block_factory, normalizer=normalizer,
            exchange=exchange)
        self.emb = nn.Embedding(N, embedding_dim=emb_dim)     # registered, unused: as in the reference
        self.final_projection = nn.Linear(emb_dim, emb_dim)
        self.dropout = nn.Dropout(getattr(config.model, "dropout", 0.3))
        TESTModel._initialize_parameters(self)
        self._graph_cache = None

    def load_replicated_state(self, full_model: TESTModel):
        """Parameters, modality weights and this rank's rows of the id table from a single-process
        model (tests, checkpoints written by one)."""
        self.load_state_dict(full_model.state_dict())
        with torch.no_grad():
            self.alphas.copy_(full_model.alphas)
            self._first_emb = full_model._first_emb.detach()[self.lo:self.hi].to(self._first_emb.device).contiguous()

    def build_graph(self, interactions, index=None, n_total=None):
        self._graph_cache = self.graph_constructor.build_graph(interactions, self.num_users, self.num_items,
                                                               index=index, n_total=n_total)
        return self._graph_cache

    def creat_feature_weight(self):
        for feature in getattr(self.config.graph, "weight_feature", [None]):
            self.graph_constructor.creat_feature_weight(feature=feature)

    def forward_local(self):
        """This rank's rows of the final node embeddings."""
        if self._graph_cache is None:
            raise ValueError("Graph not built. Call build_graph() first.")
        g = self.graph_constructor
        Xs = []
        for linear, x in zip(self.linears, self.feats):
            if isinstance(linear, nn.Identity):
                pad = torch.zeros((self.users_local, x.shape[1]), dtype=x.dtype, device=x.device)
                Xs.append(torch.cat([pad, x]))
            else:
                Xs.append(linear(x))
        node = self.test(g, Xs, [self._first_emb] * self.modal_num, self.ks, self.alphas)
        return fused.linear_act(self.final_projection, node, None, self.dropout)

    def forward(self, batch: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        full = gather_rows(self.forward_local(), self._graph_cache)
        return {"user_embeddings": full[: self.num_users], "item_embeddings": full[self.num_users:]}

    loss_func = None


class ShardedTrainStep:
    """train/graph_trainer.py:140-157 (zero_grad, forward, loss, backward, clip_grad_norm_, step)
    for a ShardedTESTModel: every rank passes the SAME global batch; rank r trains on samples
    r, r + W, r + 2W, ..., its loss weighted by its share, so the ranks' losses add up to the
    global mean.

    That weighting reproduces the single-process loss only for losses that are a plain mean over
    the samples (the trainer's BPR loss, `gather_dot_bpr_loss`, and the log-sigmoid variants).  A
    masked mean (`bpr_loss(..., mask)` divides by the number of unmasked pairs) or a softmax over
    the batch is not linear in the split: batches carrying a `mask` are refused.

    With an optimizer that clips inside its update (recommend_amd.optim.FusedClipAdam) the step has
    no per-parameter launches after the backward pass: one flat all-reduce whose slices BECOME the
    gradients, then the optimizer's two launches.  `graphed()` records the step, collectives
    included, as one HIP graph."""

    PER_SAMPLE_KEYS = ("user_ids", "item_ids", "neg_items", "neg_rank", "ratings")

    def __init__(self, model: ShardedTESTModel, optimizer: torch.optim.Optimizer,
                 loss_fn: Callable[[Dict, Dict], torch.Tensor], max_grad_norm: Optional[float] = 1.0,
                 sharded_params: Optional[List[torch.Tensor]] = None, per_sample_keys=None):
        self.model, self.opt, self.loss_fn = model, optimizer, loss_fn
        self.max_grad_norm = max_grad_norm
        self.group, self.world, self.rank = model.group, model.world, model.rank
        self.per_sample_keys = tuple(per_sample_keys or self.PER_SAMPLE_KEYS)
        # as clip_grad_norm_(model.parameters()) and the optimizer see them (train/graph_trainer.py:151-156)
        self.params = [p for p in model.parameters() if p.requires_grad]
        # parameters whose rows are sharded (none in the MMFCN as the reference defines it: its
        # trainable tensors are the replicated layer weights); their gradients are already whole
        self.sharded = list(sharded_params or [])
        sh = {id(p) for p in self.sharded}
        self.replicated = [p for p in self.params if id(p) not in sh]
        self._synced = None      # the replicated parameters some rank has a gradient for
        self.last_grad_norm = None
        # an optimizer that clips inside its own update and sums the shards' norms over the ranks itself
        self._opt_clips = hasattr(optimizer, "max_grad_norm")
        if self._opt_clips:
            optimizer.max_grad_norm = max_grad_norm
            if self.sharded and not getattr(optimizer, "_sharded", None):
                raise ValueError("build the optimizer with sharded_params=... and group=... for row-sharded parameters")

    def split(self, batch):
        """This rank's samples of the global batch: the per-sample entries (`per_sample_keys`) are
        strided, everything else (CSR constants) passes."""
        if "mask" in batch:
            raise ValueError("a masked loss is not a plain mean over the samples: it cannot be split over ranks by weight")
        B = batch["user_ids"].shape[0]
        out = {}
        for k, v in batch.items():
            out[k] = v[self.rank::self.world] if k in self.per_sample_keys else v
        return out, B

    def sync_grads(self):
        """Sum of the replicated parameters' gradients over the ranks: one flat bucket, one all-reduce;
        the bucket's slices are the parameters' gradients from then on (no copy back)."""
        if self.world == 1:
            return
        if self._synced is None:
            # which parameters take part is fixed by the network, not by the batch: decided once, by
            # all ranks together (a registered but unused table has no gradient anywhere and stays out)
            has = torch.tensor([0.0 if p.grad is None else 1.0 for p in self.replicated],
                               device=self.replicated[0].device)
            all_reduce_tensor(has, dist.ReduceOp.MAX, self.group)
            self._synced = [p for p, h in zip(self.replicated, has.tolist()) if h > 0]
        grads = []
        for p in self._synced:
            if p.grad is None:           # no local row touched it in this step
                p.grad = torch.zeros_like(p)
            grads.append(p.grad)
        flat = torch.cat([g.reshape(-1) for g in grads])
        all_reduce_tensor(flat, dist.ReduceOp.SUM, self.group)
        off = 0
        for p in self._synced:
            n = p.numel()
            p.grad = flat[off:off + n].view_as(p)
            off += n

    def clip(self):
        """clip_grad_norm_(parameters, max_norm) for optimizers that do not clip themselves: the norm
        over replicated gradients is the same on every rank after sync_grads; squared norms of
        row-sharded gradients are summed over ranks."""
        sq = torch.zeros((), dtype=torch.float32, device=self.replicated[0].device)
        for p in self.replicated:
            if p.grad is not None:
                sq = sq + p.grad.float().pow(2).sum()
        if self.sharded:
            sq_sh = torch.zeros_like(sq)
            for p in self.sharded:
                if p.grad is not None:
                    sq_sh = sq_sh + p.grad.float().pow(2).sum()
            if self.world > 1:
                all_reduce_tensor(sq_sh, dist.ReduceOp.SUM, self.group)
            sq = sq + sq_sh
        total = sq.sqrt()
        coef = torch.clamp(self.max_grad_norm / (total + 1e-6), max=1.0)
        for p in self.params:
            if p.grad is not None:
                p.grad.mul_(coef)
        self.last_grad_norm = total
        return total

    def _step(self, batch):
        """Everything of the step behind zero_grad; returns the global loss (on every rank)."""
        local, B = self.split(batch)
        b_loc = local["user_ids"].shape[0]
        out = self.model(local)
        if b_loc > 0:
            loss = self.loss_fn(out, local) * (float(b_loc) / float(B))
        else:   # a rank without samples still takes part in every collective of the backward pass
            loss = (out["user_embeddings"].sum() + out["item_embeddings"].sum()) * 0.0
        loss.backward()
        self.sync_grads()
        if self._opt_clips:
            self.opt.step()
            self.last_grad_norm = self.opt.grad_norm
        else:
            if self.max_grad_norm is not None:
                self.clip()
            self.opt.step()
        total = loss.detach().clone()
        if self.world > 1:
            all_reduce_tensor(total, dist.ReduceOp.SUM, self.group)
        return total

    def __call__(self, batch):
        self.opt.zero_grad(set_to_none=True)
        return self._step(batch)

    def graphed(self, example_batch, constants=None, warmup_steps=3, capture=None):
        """The step as one captured HIP graph (train_step.GraphedStep): every rank records its own
        graph of the same sequence -- layer kernels, hops, the collectives between them, the flat
        gradient all-reduce, the optimizer's two launches -- and replays it per global batch.
        Collectives are capturable on the "nccl" (RCCL) backend only: with any other backend, or
        capture=False, the returned object runs the step eagerly behind the same interface.
        The optimizer must be capturable (FusedClipAdam)."""
        from .train_step import GraphedStep
        if capture is None:
            capture = (self.world == 1 and not dist.is_initialized()) or dist.get_backend(self.group) == "nccl"
        return GraphedStep(self._step, lambda: self.opt.zero_grad(set_to_none=True), example_batch, constants,
                           warmup_steps, capture=capture, capture_error_mode="thread_local")


__all__ = ["ShardedGraphConstructor", "ShardedTESTModel", "ShardedTrainStep", "gather_rows"]
