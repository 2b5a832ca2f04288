"""World-size-2 check of the item-sharded evaluation path on CPU (gloo).

The partitioning, the all-gather exchange, the per-slice merge and the metric
all-reduce are host logic; the three compute calls are injected.  Here they are
backed by the oracle (test-only stand-in for the HIP ops), so the test pins the
distributed plumbing: the 2-rank result must equal the unsharded oracle."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
METRICS = ["precision", "recall", "hit_rate", "ndcg", "map", "mrr"]


class OracleCompute:
    """score_topk / topk_merge / rank_metrics with the signatures of recommend_amd.ops."""

    @staticmethod
    def score_topk(U, I, K, mask_ptr=None, mask_items=None, item_id_offset=0):
        from oracle import eval_oracle
        mp_ = None if mask_ptr is None else mask_ptr.numpy()
        mi_ = None if mask_items is None else mask_items.numpy()
        if mi_ is not None and mi_.size == 0:
            mp_ = mi_ = None
        s, i = eval_oracle.score_topk(U.numpy(), I.numpy(), K, mp_, mi_, id_offset=item_id_offset)
        return torch.from_numpy(s), torch.from_numpy(i)

    @staticmethod
    def topk_merge(scores, ids, K):
        U, P, Kin = scores.shape
        s = scores.reshape(U, P * Kin).numpy()
        i = ids.reshape(U, P * Kin).numpy().astype(np.int64)
        key_i = np.where(i < 0, np.iinfo(np.int64).max, i)
        order = np.lexsort((key_i, -s.astype(np.float64), i < 0), axis=1)[:, :K]
        return (torch.from_numpy(np.take_along_axis(s, order, 1)),
                torch.from_numpy(np.take_along_axis(i, order, 1).astype(np.int32)))

    @staticmethod
    def rank_metrics(topk_ids, tgt_ptr, tgt_items, k_list, mean_denominator=None, return_sums=False):
        from oracle import eval_oracle
        n = topk_ids.shape[0]
        d = eval_oracle.metrics_from_topk_sparse(topk_ids.numpy(), tgt_ptr.numpy(), tgt_items.numpy(),
                                                 k_list, n_users_total=1)   # sums
        sums = torch.tensor([[d[k][m] for m in METRICS] for k in k_list], dtype=torch.float64)
        out = (sums / float(mean_denominator or n)).float()
        return (out, sums) if return_sums else out


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from recommend_amd.dist import ShardedRankingEvaluator, item_range
        rng = np.random.default_rng(0)
        U, I, D = 150, 333, 64
        ue = (np.round(rng.standard_normal((U, D)) * 2) / 2).astype(np.float32)   # ties on purpose
        ie = (np.round(rng.standard_normal((I, D)) * 2) / 2).astype(np.float32)
        tr = np.unique(np.stack([rng.integers(0, U, 900), rng.integers(0, I, 900)], 1), axis=0)
        va = np.unique(np.stack([rng.integers(0, U, 300), rng.integers(0, I, 300)], 1), axis=0)
        ev = ShardedRankingEvaluator((va[:, 0], va[:, 1]), (tr[:, 0], tr[:, 1]), [5, 10, 20], METRICS,
                                     U, I, compute=OracleCompute)
        assert (ev.lo, ev.hi) == item_range(I, rank, world)
        res = ev.evaluate(torch.from_numpy(ue), torch.from_numpy(ie))
        ids = ev.last_ids.numpy()
        q.put((rank, {k: float(v) for k, v in res.items()}, ev.u_lo, ev.u_hi, ids))
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_item_sharded_eval_world2_equals_unsharded():
    from oracle import eval_oracle
    from recommend_amd.evaluator import pairs_to_csr
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # unsharded oracle on the same inputs
    rng = np.random.default_rng(0)
    U, I, D = 150, 333, 64
    ue = (np.round(rng.standard_normal((U, D)) * 2) / 2).astype(np.float32)
    ie = (np.round(rng.standard_normal((I, D)) * 2) / 2).astype(np.float32)
    tr = np.unique(np.stack([rng.integers(0, U, 900), rng.integers(0, I, 900)], 1), axis=0)
    va = np.unique(np.stack([rng.integers(0, U, 300), rng.integers(0, I, 300)], 1), axis=0)
    mp_, mi_ = pairs_to_csr(tr[:, 0], tr[:, 1], U)
    _s, want_ids = eval_oracle.score_topk(ue, ie, 20, mp_, mi_)
    tp, tc = pairs_to_csr(va[:, 0], va[:, 1], U)
    want = eval_oracle.metrics_from_topk_sparse(want_ids, tp, tc, [5, 10, 20])
    for rank, res, u_lo, u_hi, ids in results:
        np.testing.assert_array_equal(ids, want_ids[u_lo:u_hi])          # merged ids exact, ties included
        for k in (5, 10, 20):
            for m in METRICS:
                assert abs(res[f"{m}@{k}"] - want[k][m]) <= 1e-6, (rank, m, k)
    assert results[0][1] == results[1][1]                                  # every rank holds the same metrics
