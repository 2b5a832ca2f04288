"""MMFCN network blocks and model wrapper with the reference's interface.

Mirrors model/test/out_Layer.py (TEST :38-54, MGCN :20-36), Agg_Layer.py (AGG
:16-26), Combin_Layer.py (Combine :16-30), model/common/my_mlp.py (MyLinear
:6-17, MyPReLU :31-49) and the wrapper TESTModel, model/Main_Model.py:32-267:
`build_graph(interactions)`, `creat_feature_weight()`, `forward(batch) ->
{'user_embeddings','item_embeddings'}`, `get_model_info()`.

What changes is where the arithmetic runs: the graph is a recommend_amd
GraphConstructor (CSR/CSC SpMM kernels) and the modality projection of the
zero-padded feature matrix is the item-rows-only MFMA GEMM of ops.proj.
"""
from typing import Any, Dict, List, Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import fused, mgcn_op, ops
from .graph_constructor import GraphConstructor


class MyLinear(nn.Linear):
    def reset_parameters(self) -> None:
        nn.init.xavier_uniform_(self.weight)
        if self.bias is not None:
            nn.init.zeros_(self.bias)


class MyPReLU(nn.Module):
    def __init__(self, num_parameters: int = 1, init: float = 0.25, device=None, dtype=None):
        super().__init__()
        self.num_parameters = num_parameters
        self.alpha = nn.Parameter(torch.empty(num_parameters, device=device, dtype=dtype).fill_(init))

    def forward(self, input: torch.Tensor) -> torch.Tensor:
        return F.prelu(input, self.alpha)


class AGG(nn.Module):
    def __init__(self, emb_num, dropout, agg_type):
        super().__init__()
        self.W = MyLinear(emb_num, emb_num)
        self.dropout = nn.Dropout(dropout)
        self.active = MyPReLU()
        self.agg_type = agg_type

    def forward(self, x, g, k):
        if self.agg_type == "mean":
            return g.agg_func(x, k, self.W, self.dropout, self.active)


class Combine(nn.Module):
    def __init__(self, emb_num, dropout):
        super().__init__()
        self.W_ori_f = MyLinear(emb_num, emb_num)
        self.W_f = MyLinear(emb_num, emb_num)
        self.W_o = MyLinear(emb_num, emb_num)
        self.dropout = nn.Dropout(dropout)
        self.active = MyPReLU()

    def forward(self, ori_feature_x, feature_x):
        # dropout(active(W_o(W_ori_f(x) + W_f(agg_x)))), Combin_Layer.py:26-30
        y = fused.linear_sum(self.W_ori_f, ori_feature_x, self.W_f, feature_x)
        return fused.linear_act(self.W_o, y, self.active, self.dropout)


class MGCN(nn.Module):
    def __init__(self, emb_num, layer_num, dropout):
        super().__init__()
        self.agg = AGG(emb_num, dropout, "mean")
        self.W_o = MyLinear(emb_num, emb_num)
        self.combine = nn.ModuleList([Combine(emb_num, dropout) for _ in range(layer_num)])
        self.dropout = nn.Dropout(dropout)
        self.active = MyPReLU()

    def forward(self, g, X, emb, k):
        # agg -> Combine layers -> agg_X + dropout(active(W_o(emb))) (out_Layer.py:32-36) as one
        # autograd node over the layer and hop kernels
        return mgcn_op.mgcn_forward(self, g.hops(), X, emb, k)


class TEST(nn.Module):
    def __init__(self, modal_num, emb_num, layer_num, dropout):
        super().__init__()
        self.modal_num = modal_num
        self.MGCNs = nn.ModuleList([MGCN(emb_num, layer_num, dropout) for _ in range(modal_num)])

    def forward(self, g, Xs, embs, ks, alphas):
        modal_features = [self.MGCNs[i](g, Xs[i], embs[i], ks[i]) for i in range(self.modal_num)]
        # sum_m softmax(alphas)_m * F_m (out_Layer.py:52-54) without stacking the modalities
        return ops.modal_sum(modal_features, alphas)


class _ItemProjection(nn.Module):
    """nn.Linear(feat_dim, emb_dim) over a feature matrix whose user rows are zero."""

    def __init__(self, feat_dim, emb_dim, n_users):
        super().__init__()
        self.linear = nn.Linear(feat_dim, emb_dim)
        self.n_users = n_users

    def forward(self, x_items):
        return ops.proj(x_items, self.linear.weight, self.linear.bias, self.n_users)


class TESTModel(nn.Module):
    """Enhanced MMFCN model wrapper for graph-based recommendation."""

    def __init__(self, config, user_features: Optional[Dict[str, torch.Tensor]] = None,
                 item_features: Optional[Dict[str, torch.Tensor]] = None):
        super().__init__()
        self.config = config
        self.num_users = config.data.num_users
        self.num_items = config.data.num_items
        self.num_nodes = self.num_users + self.num_items
        self.modal_num = config.model.modal_num
        self.weight_feature = getattr(config.graph, "weight_feature", [None])
        device = config.system.device
        emb_dim = config.model.emb_dim
        layer_num = getattr(config.model, "layer_num", 2)
        dropout = getattr(config.model, "dropout", 0.3)

        self.user_features = user_features or {}
        self.item_features = item_features or {}
        # Main_Model.py:65-68 zero-pads every feature to all nodes.  Item features keep only
        # their own rows here (the padded user rows project to the bias); user features are
        # padded as in the reference and go through the same kernel with the roles swapped.
        self.feats: List[torch.Tensor] = []
        self._feat_is_item: List[bool] = []
        for feat in self.user_features.values():
            self.feats.append(feat.to(device).float().contiguous())
            self._feat_is_item.append(False)
        for feat in self.item_features.values():
            self.feats.append(feat.to(device).float().contiguous())
            self._feat_is_item.append(True)

        self.linears = nn.ModuleList([
            (_ItemProjection(x.shape[1], emb_dim, self.num_users) if is_item
             else nn.Linear(x.shape[1], emb_dim)) if x.shape[1] != emb_dim else nn.Identity()
            for x, is_item in zip(self.feats, self._feat_is_item)
        ])

        self.ks = [5] * self.modal_num
        self.alphas = torch.randn(self.modal_num, requires_grad=True, device=device)
        # Main_Model.py:78-79,98: the id embedding the network reads is the first table's
        # weight; the table registered afterwards is a separate parameter
        # (kept as a bare leaf tensor and expanded per forward: a view built once here would keep
        # its autograd nodes, and the stream they were created on, alive across steps)
        # a plain tensor, not a Parameter: assigning a Parameter would register it, and the reference's
        # first table is replaced as a submodule (Main_Model.py:98) and so is never handed to the optimizer
        # (and frozen: no optimizer ever sees it, so a gradient would only be computed, accumulated and never read)
        self._first_emb = nn.Embedding(self.num_nodes, emb_dim, device=device).weight.detach()

        self.test = TEST(modal_num=self.modal_num, emb_num=emb_dim, layer_num=layer_num,
                         dropout=dropout)
        self.graph_constructor = GraphConstructor(
            user_features=self.user_features,
            item_features=self.item_features,
            edge_weight_type=config.graph.edge_weight_type,
            add_self_loops=config.graph.add_self_loops,
            normalize_adj=config.graph.normalize_adj,
        )
        self.emb = nn.Embedding(self.num_nodes, embedding_dim=emb_dim)
        self.final_projection = nn.Linear(emb_dim, emb_dim)
        self.dropout = nn.Dropout(dropout)
        self._initialize_parameters()
        self._graph_cache = None
        self._graph_features_cache = None

    @property
    def embs(self):
        """Main_Model.py:79: the first embedding table's weight, once per modality."""
        return self._first_emb.unsqueeze(0).expand(self.modal_num, -1, -1)

    def _initialize_parameters(self):
        for module in self.modules():
            if isinstance(module, nn.Linear):
                nn.init.xavier_uniform_(module.weight)
                if module.bias is not None:
                    nn.init.zeros_(module.bias)
            elif isinstance(module, nn.Embedding):
                nn.init.normal_(module.weight, std=0.1)

    def build_graph(self, interactions: List[Tuple[int, int, float]]):
        self._graph_cache = self.graph_constructor.build_graph(
            interactions, self.num_users, self.num_items)
        return self._graph_cache

    def creat_feature_weight(self):
        for feature in self.weight_feature:
            self.graph_constructor.creat_feature_weight(feature=feature)

    def get_graph(self):
        if self._graph_cache is None:
            raise ValueError("Graph not built. Call build_graph() first.")
        return self._graph_cache

    def _project(self, linear, x, is_item):
        if isinstance(linear, nn.Identity):
            pad = torch.zeros((self.num_users if is_item else self.num_items, x.shape[1]),
                              dtype=x.dtype, device=x.device)
            return torch.cat([pad, x] if is_item else [x, pad])
        if is_item:
            return linear(x)
        # user-side features: rows [0, U) carry data, item rows are zero -> bias
        y_users = linear(x)
        return torch.cat([y_users, linear.bias.unsqueeze(0).expand(self.num_items, -1)])

    def forward(self, batch: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        if self._graph_cache is None:
            raise ValueError("Graph not built. Call build_graph() first.")
        g = self.graph_constructor
        g.move_to_device(self.config.system.device)

        self.Xs = [self._project(linear, x, is_item)
                   for linear, x, is_item in zip(self.linears, self.feats, self._feat_is_item)]
        # every modality reads the same table: handing over the tensor itself (not slices of the
        # expanded view `embs`) makes autograd add the two gradients instead of materialising [M, N, D]
        node_embeddings = self.test(g, self.Xs, [self._first_emb] * self.modal_num, self.ks, self.alphas)
        # dropout(final_projection(.)), Main_Model.py:162-163
        node_embeddings = fused.linear_act(self.final_projection, node_embeddings, None, self.dropout)
        user_embeddings, item_embeddings = ops.split_rows(node_embeddings, self.num_users)
        return {"user_embeddings": user_embeddings, "item_embeddings": item_embeddings}

    def get_model_info(self) -> Dict[str, Any]:
        total_params = sum(p.numel() for p in self.parameters())
        trainable_params = sum(p.numel() for p in self.parameters() if p.requires_grad)
        return {
            "model_name": "TEST",
            "total_parameters": total_params,
            "trainable_parameters": trainable_params,
            "num_users": self.num_users,
            "num_items": self.num_items,
            "num_nodes": self.num_nodes,
            "embedding_dim": self.config.model.emb_dim,
            "num_modalities": self.config.model.modal_num,
            "user_features": list(self.user_features.keys()),
            "item_features": list(self.item_features.keys()),
        }

    loss_func = None  # the trainer's built-in BPR loss applies (train/graph_trainer.py:53-54)


class ModelFactory:
    """Factory with the reference's entry point (model/Main_Model.py:954-1011)."""

    @staticmethod
    def create_Model(config, user_features=None, item_features=None) -> nn.Module:
        name = getattr(config.model, "model_name", "MMFCN")
        if name in ("MMFCN", "TEST", "mmfcn", "test"):
            return TESTModel(config, user_features, item_features)
        raise ValueError(f"Unknown model: {name} (this package builds the MMFCN path)")


__all__ = ["TESTModel", "ModelFactory", "TEST", "MGCN", "AGG", "Combine", "MyLinear", "MyPReLU"]
