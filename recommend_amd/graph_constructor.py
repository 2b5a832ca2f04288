"""GraphConstructor with the reference's call protocol, holding the bipartite
graph as CSR/CSC arrays in HBM instead of a DGL graph.

Mirrors model/test/graph_constructor.py of the reference (class
GraphConstructor: build_graph :51-117, move_to_device :119, creat_feature_weight
:122-128, _normalize_graph :158-187, aggrate :189-197, agg_func :199-208,
get_graph_statistics :222-240): same constructor arguments, method names,
argument meaning and errors.  Edge construction and the CSR/CSC sort run in the
host part of librecommend_amd; normalisation and every propagation hop run in
its HIP kernels.
"""
from typing import Any, Dict, List, Optional, Tuple

import numpy as np
import torch

from . import _ffi, fused, ops


class DeviceGraph:
    """What build_graph returns: the edge list in the reference's order plus
    the device-resident sparse operands.  Offers the few DGLGraph calls the
    reference's callers make on the returned object (main.py:110)."""

    def __init__(self, src, dst, w, n_nodes):
        self._n_nodes = int(n_nodes)
        self.src = src          # numpy int32 [nnz], COO order of the reference
        self.dst = dst
        self.w_raw = w          # numpy fp32 [nnz]
        self.device = torch.device("cpu")
        # stable sorts by destination (forward operand) and by source (transpose)
        self.csr_host = ops.csr_from_coo_host(dst, src, n_nodes)
        self.csc_host = ops.csr_from_coo_host(src, dst, n_nodes)
        self.fwd: Optional[ops.CsrMatrix] = None
        self.bwd: Optional[ops.CsrMatrix] = None
        self.edata: Dict[str, torch.Tensor] = {"weight": torch.from_numpy(w.copy())}
        self._val_key_version = None

    def num_nodes(self):
        return self._n_nodes

    def num_edges(self):
        return int(self.src.shape[0])

    def to(self, device):
        device = torch.device(device)
        if device.type == "cuda" and device.index is None:
            device = torch.device("cuda", torch.cuda.current_device())
        if device.type != "cuda":
            if self.fwd is not None and device.type == "cpu":
                return self  # tensors stay where the kernels can reach them
            raise _ffi.RkError(_ffi.RK_ERR_NODEV,
                               f"the graph lives in HBM; cannot move it to '{device}' "
                               "(no CPU propagation path exists)")
        if self.fwd is not None and self.device == device:
            return self
        _ffi.require_device()
        self.device = device
        self.fwd = ops.CsrMatrix(*self.csr_host, device)
        self.bwd = ops.CsrMatrix(*self.csc_host, device)
        self.edata = {k: v.to(device) for k, v in self.edata.items()}
        self._val_key_version = None
        return self

    def refresh_values(self, key="weight"):
        """(Re)build the CSR/CSC value arrays from the COO-ordered edge data."""
        w = self.edata[key]
        version = (key, w.data_ptr(), w._version)
        if self._val_key_version != version:
            self.fwd.set_values_from_coo(w)
            self.bwd.set_values_from_coo(w)
            self._val_key_version = version


class GraphConstructor:
    """Constructs and manages recommendation graphs with multi-modal features."""

    def __init__(
        self,
        user_features: Optional[Dict[str, torch.Tensor]] = None,
        item_features: Optional[Dict[str, torch.Tensor]] = None,
        edge_weight_type: str = "uniform",
        feature: str = "text",
        add_self_loops: bool = True,
        normalize_adj: bool = True,
        max_neighbors: int = 50,
    ):
        self.user_features = user_features or {}
        self.item_features = item_features or {}
        self.edge_weight_type = edge_weight_type
        self.add_self_loops = add_self_loops
        self.normalize_adj = normalize_adj
        self.max_neighbors = max_neighbors

        self.graph: Optional[DeviceGraph] = None
        self.num_users = 0
        self.num_items = 0
        self.num_nodes = 0
        self.weight = "weight"

    # ------------------------------------------------------------------ G1
    def build_graph(self, interactions, num_users: int, num_items: int) -> DeviceGraph:
        """interactions: list of (user_id, item_id, rating) tuples in dataset order
        (or a tuple of three arrays).  Edge order follows graph_constructor.py:80-98."""
        self.num_users = num_users
        self.num_items = num_items
        self.num_nodes = num_users + num_items
        self.num_interactions = len(interactions[0]) if _is_array_triplet(interactions) else len(interactions)

        if _is_array_triplet(interactions):
            users, items = np.asarray(interactions[0]), np.asarray(interactions[1])
            ratings = None if interactions[2] is None else np.asarray(interactions[2])
        elif len(interactions) == 0:
            users = np.zeros(0, np.int64)
            items = np.zeros(0, np.int64)
            ratings = np.zeros(0, np.float32)
        else:
            arr = np.asarray(interactions, dtype=np.float64)
            users = arr[:, 0].astype(np.int64)
            items = arr[:, 1].astype(np.int64)
            ratings = arr[:, 2].astype(np.float32)
        src, dst, w = ops.build_coo_host(users, items, ratings, num_users, num_items,
                                         self.add_self_loops)
        self.graph = DeviceGraph(src, dst, w, self.num_nodes)
        return self.graph

    def move_to_device(self, device="cpu"):
        self.graph = self.graph.to(device)

    def _ensure_on_device(self):
        if self.graph is None:
            raise ValueError("No graph built")
        if self.graph.fwd is None:
            # the reference normalises on the CPU graph before the first forward; here the
            # arithmetic exists only as HIP kernels, so the graph goes to the GPU now
            self.graph.to("cuda")

    # ------------------------------------------------------------------ G2
    def creat_feature_weight(self, feature=None):
        self._compute_edge_weights(feature=feature)
        if self.normalize_adj:
            self._normalize_graph(feature=feature)

    def _node_feature(self, feature):
        # graph_constructor.py:130-135: features zero-padded to all nodes
        if feature in self.user_features:
            f = self.user_features[feature]
            return torch.cat([f, torch.zeros(self.num_items, f.shape[1], dtype=f.dtype)], 0)
        if feature in self.item_features:
            f = self.item_features[feature]
            return torch.cat([torch.zeros(self.num_users, f.shape[1], dtype=f.dtype), f], 0)
        raise KeyError(feature)

    def _compute_edge_weights(self, feature=None):
        """Feature-similarity edge weights, graph_constructor.py:137-156."""
        if self.edge_weight_type == "uniform" or feature is None:
            return
        if self.edge_weight_type not in ("cosine", "dot"):
            raise ValueError(f"Unknown edge weight type: {self.edge_weight_type}")
        self._ensure_on_device()
        g = self.graph
        # the zero-padded rows of _node_feature are never materialised: the kernel is told which
        # node range carries features
        if feature in self.user_features:
            rows, row_lo = self.user_features[feature], 0
        elif feature in self.item_features:
            rows, row_lo = self.item_features[feature], self.num_users
        else:
            raise KeyError(feature)
        src = torch.from_numpy(g.src).to(g.device)
        dst = torch.from_numpy(g.dst).to(g.device)
        out = ops.edge_feature_sim(src, dst, rows.to(g.device, torch.float32), row_lo, self.edge_weight_type)
        g.edata[f"{feature}_weight"] = out

    def _normalize_graph(self, feature=None):
        """w_ij / (sqrt(sum_in_i) * sqrt(sum_in_j)), graph_constructor.py:158-187."""
        if self.edge_weight_type == "uniform" or feature is None:
            self.weight = "weight"
        else:
            self.weight = f"{feature}_weight"
        self._ensure_on_device()
        g = self.graph
        _sq, w_norm = ops.sym_normalize(g.fwd, g.edata[self.weight])
        g.edata[self.weight] = w_norm

    # ------------------------------------------------------------------ G3
    def _operands(self):
        self._ensure_on_device()
        g = self.graph
        g.refresh_values("weight")  # hops always read edata['weight'] (:194,:205)
        return g.fwd, g.bwd

    def aggrate(self, node_x, k, alpha):
        """k hops of h <- alpha*h0 + (1-alpha)*A h, graph_constructor.py:189-197."""
        fwd, bwd = self._operands()
        h0 = node_x
        h = node_x
        for _ in range(k):
            h = ops.spmm(fwd, bwd, h, h0, alpha, 1.0 - alpha)
        return h

    def agg_func(self, node_x, k, f, dropout, active):
        """k hops of h <- A dropout(active(f(h))), graph_constructor.py:199-208."""
        fwd, bwd = self._operands()
        h = node_x
        for _ in range(k):
            h = fused.linear_act(f, h, active, dropout)   # dropout(active(f(h)))
            h = ops.spmm(fwd, bwd, h)
        return h

    def hops(self):
        """The propagation on raw tensors, forward (A z) and backward (A^T g), for callers that run
        their own backward pass (recommend_amd.mgcn_op)."""
        from .mgcn_op import SpmmHops
        return SpmmHops(*self._operands())

    def propagate(self, node_x):
        """One hop h <- A h (the primitive every model variant shares)."""
        fwd, bwd = self._operands()
        return ops.spmm(fwd, bwd, node_x)

    # --------------------------------------------------------------- misc
    def get_node_features(self, modal_name: Optional[str] = None):
        if modal_name:
            x = self._node_feature(modal_name)
            if modal_name.startswith("user_"):
                return x[: self.num_users]
            if modal_name.startswith("item_"):
                return x[self.num_users:]
            return x
        return list(self.user_features.keys()) + list(self.item_features.keys())

    def get_graph_statistics(self) -> Dict[str, Any]:
        if self.graph is None:
            return {}
        w = self.graph.edata[self.weight].float()
        return {
            "num_nodes": self.graph.num_nodes(),
            "num_edges": self.graph.num_edges(),
            "num_users": self.num_users,
            "num_items": self.num_items,
            "edge_weight_stats": {
                "mean": w.mean().item(),
                "std": w.std().item(),
                "min": w.min().item(),
                "max": w.max().item(),
            },
        }

    def save_graph(self, filepath: str):
        if self.graph is None:
            raise ValueError("No graph to save")
        g = self.graph
        np.savez(filepath, src=g.src, dst=g.dst, n_nodes=g.num_nodes(),
                 num_users=self.num_users, num_items=self.num_items,
                 **{f"edata_{k}": v.detach().cpu().numpy() for k, v in g.edata.items()})

    def load_graph(self, filepath: str):
        z = np.load(filepath if filepath.endswith(".npz") else filepath + ".npz")
        n = int(z["n_nodes"])
        edata = {k[6:]: z[k] for k in z.files if k.startswith("edata_")}
        self.graph = DeviceGraph(z["src"], z["dst"], edata["weight"].astype(np.float32), n)
        self.graph.edata = {k: torch.from_numpy(v) for k, v in edata.items()}
        self.num_nodes = n
        self.num_users = int(z["num_users"])
        self.num_items = int(z["num_items"])


def _is_array_triplet(x):
    return (isinstance(x, tuple) and len(x) == 3 and hasattr(x[0], "shape")
            and hasattr(x[1], "shape") and (x[2] is None or hasattr(x[2], "shape")))
