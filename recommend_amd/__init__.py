"""recommend_amd: MI355X-native hot path of the multi-modal graph recommender.

Graph propagation (CSR/CSC SpMM), BPR gather-dot loss with negative sampling and
full-ranking evaluation, as hand-written gfx950 kernels behind a C ABI
(include/recommend_amd.h), under host classes that keep the reference's call
protocols (GraphConstructor, TESTModel, Verifier/Tester, loss functions).
"""
from . import _ffi
from ._ffi import RkError
from .graph_constructor import GraphConstructor, DeviceGraph
from .evaluator import Verifier, Tester, RankingEvaluator
from .loss import bpr_loss, gather_dot_bpr_loss
from .sampler import NegativeSampler, TrainBatchLoader, build_user_csr
from .model import TESTModel, ModelFactory
from .data import InteractionTable, InteractionSplit, mask_index

__all__ = ["GraphConstructor", "DeviceGraph", "Verifier", "Tester", "RankingEvaluator",
           "bpr_loss", "gather_dot_bpr_loss", "NegativeSampler", "TrainBatchLoader",
           "build_user_csr", "TESTModel", "ModelFactory", "RkError", "InteractionTable",
           "InteractionSplit", "mask_index"]
