"""CPU-side checks of the product: the C-ABI library loads and exports every
symbol the header declares, and its host entry points (edge list, CSR sort, row
binning, MT19937 negative sampler, batch loader) are bit-exact against the
oracle and against the batches captured from the reference's own loader.
No device entry point is called here."""
import ctypes
import os
import random
import re

import numpy as np
import pytest
import torch

from oracle import graph_oracle, sampler_oracle
from recommend_amd import _ffi, ops
from recommend_amd.sampler import NegativeSampler, TrainBatchLoader, build_user_csr

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, "include", "recommend_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rk_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = _ffi.load()
    names = header_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in the header but not exported"
    # and every symbol the Python side binds is declared in the header
    assert sorted(_ffi.SIGNATURES) == names
    assert lib.rk_abi_version() == 1


def test_device_entry_points_fail_loudly_without_gpu():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    assert _ffi.device_count() == 0
    with pytest.raises(_ffi.RkError):
        _ffi.require_device()
    with pytest.raises(_ffi.RkError):
        ops.spmm_raw(None, torch.zeros(4, 64))
    with pytest.raises(_ffi.RkError):
        ops.score_topk(torch.zeros(4, 64), torch.zeros(8, 64), 2)


def test_argument_errors_are_reported():
    lib = _ffi.load()
    rc = lib.rk_spmm_csr_f32(None, None, None, 5, None, 64, None, None, 0.0, 1.0, None, None)
    assert rc == _ffi.RK_ERR_ARG and "null" in _ffi.last_error()
    st = np.zeros(625, np.uint32)
    with pytest.raises(_ffi.RkError, match="Negative ratio must be greater than zero"):
        _ffi.call("rk_neg_sample_host", _ffi.ptr(st), _ffi.ptr(np.zeros(1, np.int64)), 1, 0,
                  _ffi.ptr(np.zeros(2, np.int32)), _ffi.ptr(np.zeros(1, np.int32)), 5, None, None)
    with pytest.raises(_ffi.RkError, match="out of range"):
        ops.build_coo_host([0, 7], [0, 0], [1.0, 1.0], 3, 2, True)


@pytest.mark.parametrize("self_loops", [True, False])
def test_coo_and_csr_bit_exact_vs_oracle(self_loops):
    rng = np.random.default_rng(3)
    U, I, E = 50, 37, 400
    users, items = rng.integers(0, U, E), rng.integers(0, I, E)      # duplicates kept (multi-edges)
    ratings = rng.integers(1, 6, E).astype(np.float32)
    inter = [(int(u), int(i), float(r)) for u, i, r in zip(users, items, ratings)]
    o_src, o_dst, o_w = graph_oracle.build_coo(inter, U, I, self_loops)
    src, dst, w = ops.build_coo_host(users, items, ratings, U, I, self_loops)
    np.testing.assert_array_equal(src, o_src), np.testing.assert_array_equal(dst, o_dst)
    np.testing.assert_array_equal(w, o_w)
    for key, other in ((dst, src), (src, dst)):
        p, idx, eid = ops.csr_from_coo_host(key, other, U + I)
        op, oidx, oeid = graph_oracle.csr_from_coo(key, other, U + I)
        np.testing.assert_array_equal(p, op), np.testing.assert_array_equal(idx, oidx)
        np.testing.assert_array_equal(eid, oeid)


def test_empty_graph_and_no_ratings():
    src, dst, w = ops.build_coo_host([], [], None, 3, 2, True)
    np.testing.assert_array_equal(src, np.arange(5)), np.testing.assert_array_equal(w, np.zeros(5, np.float32))
    p, idx, eid = ops.csr_from_coo_host(dst, src, 5)
    np.testing.assert_array_equal(p, np.arange(6))
    src, dst, w = ops.build_coo_host([1], [0], None, 2, 1, False)
    np.testing.assert_array_equal(w, [1.0, 1.0])


def test_spmm_plan_covers_long_rows():
    """Rows above the threshold are cut into chunks of 512 edges; a row of one chunk needs no partial
    slot, a row of several gets one slot per chunk."""
    ptr = np.array([0, 3, 3, 403, 408, 1608], np.int32)       # rows of 3, 0, 400, 5, 1200 edges
    pl = ops.spmm_plan_host(ptr, 64)
    assert (pl["n_chunks"], pl["n_multi"], pl["n_slots"]) == (4, 1, 3)
    np.testing.assert_array_equal(pl["chunk_row"], [2, 4, 4, 4])
    np.testing.assert_array_equal(pl["chunk_begin"], [3, 408, 920, 1432])
    np.testing.assert_array_equal(pl["chunk_end"], [403, 920, 1432, 1608])
    np.testing.assert_array_equal(pl["chunk_slot"], [-1, 0, 1, 2])
    np.testing.assert_array_equal(pl["multi_rows"], [4])
    np.testing.assert_array_equal(pl["multi_slot_ptr"], [0, 3])
    assert ops.spmm_plan_host(np.array([0, 2, 5], np.int32), 64)["n_chunks"] == 0


def test_mt19937_stream_is_cpythons():
    st = np.zeros(625, np.uint32)
    for seed in (0, 1, 42, 2 ** 32 + 5, 2 ** 63 + 11):
        _ffi.call("rk_mt_seed_host", _ffi.ptr(st), ctypes.c_uint64(seed))
        random.seed(seed)
        version, internal, _ = random.getstate()
        np.testing.assert_array_equal(st, np.array(internal, dtype=np.uint32))


@pytest.mark.parametrize("R", [1, 3, 6, 20])
def test_negative_sampler_bit_exact_vs_python_random(R):
    rng = np.random.default_rng(R)
    U, I = 40, 57
    users = rng.integers(0, U, 300)
    items = rng.integers(0, I, 300)
    # user 0 has touched everything but 2 items, user 1 everything (no candidates)
    users = np.concatenate([users, np.zeros(I - 2, np.int64), np.ones(I, np.int64)])
    items = np.concatenate([items, np.arange(I - 2), np.arange(I)])
    p, it = build_user_csr(users, items, U)
    uii = {}
    for u, i in zip(users, items):
        uii.setdefault(int(u), set()).add(int(i))
    batch = np.concatenate([rng.integers(0, U, 200), [0, 1, 0, 1]])
    s = NegativeSampler(p, it, I, seed=123, neg_ratio=R)
    random.seed(123)
    for _ in range(3):
        ranks, neg = s.sample(batch)
        want = np.asarray(sampler_oracle.generate_negative_samples(batch, uii, set(range(I)), R))
        np.testing.assert_array_equal(neg, want)
    # the generator state stays in lock-step with CPython's
    np.testing.assert_array_equal(s.state, np.array(random.getstate()[1], dtype=np.uint32))


def test_batches_bit_exact_vs_reference_loader(golden):
    """Shuffle order, batch contents and negative items equal what the reference's
    GraphDataLoader produced for two epochs: three settings with num_workers=0, two with the
    reference's preset of four workers (batch b collated in worker b mod 4, each worker's `random`
    seeded base_seed + worker id)."""
    z = golden("loader_batches.npz")
    U, I = int(z["n_users"]), int(z["n_items"])
    p, it = build_user_csr(z["users"], z["items"], U)          # positives over the FULL table
    label = z["label"]
    tr = label == 0
    for tag, R, bs, workers in (("r1_b64", 1, 64, 0), ("r3_b50", 3, 50, 0), ("r8_b32", 8, 32, 0),
                                ("r1_b32_w4", 1, 32, 4), ("r2_b48_w4", 2, 48, 4)):
        np.testing.assert_array_equal(z["users"][tr], z[f"{tag}/train_users"])
        random.seed(42), np.random.seed(42), torch.manual_seed(42)
        sampler = NegativeSampler(p, it, I, seed=42, neg_ratio=R)
        loader = TrainBatchLoader(z["users"][tr], z["items"][tr], z["rating"][tr], bs, sampler,
                                  num_workers=workers)
        for e in range(2):
            batches = list(loader)
            assert len(batches) == int(z[f"{tag}/e{e}/n_batches"]) == len(loader)
            cat = lambda k: np.concatenate([b[k].numpy() for b in batches])
            np.testing.assert_array_equal(cat("user_ids"), z[f"{tag}/e{e}/user_ids"])
            np.testing.assert_array_equal(cat("item_ids"), z[f"{tag}/e{e}/item_ids"])
            np.testing.assert_array_equal(cat("ratings"), z[f"{tag}/e{e}/ratings"])
            np.testing.assert_array_equal(cat("neg_items"), z[f"{tag}/e{e}/neg_items"])


def test_split_rows_passes_a_joint_gradient_matrix_through():
    """ops.split_rows: a consumer that returns the two gradients as the row blocks of one matrix
    has that matrix handed on as it is; anything else is concatenated (missing halves are zero)."""
    x = torch.randn(10, 4, requires_grad=True)

    class Joint(torch.autograd.Function):
        seen = None

        @staticmethod
        def forward(ctx, a, b):
            ctx.n = a.shape[0]
            return a.sum() + 2 * b.sum()

        @staticmethod
        def backward(ctx, g):
            both = torch.cat([torch.full((ctx.n, 4), 1.0), torch.full((10 - ctx.n, 4), 2.0)])
            Joint.seen = both
            return both[: ctx.n], both[ctx.n:]

    top, bottom = ops.split_rows(x, 3)
    assert ops._stacked_rows(top, bottom) and not ops._stacked_rows(bottom, top)
    Joint.apply(top, bottom).backward()
    assert x.grad.data_ptr() == Joint.seen.data_ptr()      # no copy, no add
    assert torch.equal(x.grad, Joint.seen)
    x.grad = None
    top, bottom = ops.split_rows(x, 3)
    (top.sum() * 3).backward()
    assert torch.equal(x.grad[:3], torch.full((3, 4), 3.0)) and not x.grad[3:].any()
    x.grad = None
    top, bottom = ops.split_rows(x, 3)
    ((top * top).sum() + (bottom * bottom * bottom).sum()).backward()
    assert torch.allclose(x.grad[:3], 2 * x.detach()[:3]) and torch.allclose(x.grad[3:], 3 * x.detach()[3:] ** 2)


@pytest.mark.parametrize("R", [1, 3, 8])
def test_cached_candidate_orders_leave_the_sampler_stream_unchanged(R):
    """Users holding a quarter of the items or more get their candidate list from CPython's set
    order (emulated); NegativeSampler builds those lists once and hands them to the sampling call.
    Same ranks, ids and generator state as the call that rebuilds them per sample, and as the
    oracle's random.sample over list(all_items - pos)."""
    rng = np.random.default_rng(3)
    n_items, n_users = 300, 40
    lens = rng.integers(1, 60, n_users)
    lens[[3, 11, 29]] = [75, 200, 299]          # 75 = n_items >> 2: the first length that is emulated
    lens[7] = 74                                # the last one that is not
    pairs_u = np.concatenate([np.full(l, u) for u, l in enumerate(lens)])
    pairs_i = np.concatenate([rng.choice(n_items, l, replace=False) for l in lens])
    # every item id has to occur
    pairs_u = np.concatenate([pairs_u, np.zeros(n_items, np.int64)])
    pairs_i = np.concatenate([pairs_i, np.arange(n_items)])
    ptr_, items_ = build_user_csr(pairs_u, pairs_i, n_users)
    users = rng.integers(1, n_users, 500).astype(np.int64)   # user 0 holds every item: a separate path
    users[:6] = [3, 11, 29, 7, 29, 3]
    s = NegativeSampler(ptr_, items_, n_items, seed=9, neg_ratio=R)
    assert s.cached_users.tolist() == [3, 11, 29]
    ranks_c, neg_c = s.sample(users)
    st = np.zeros(625, np.uint32)
    _ffi.call("rk_mt_seed_host", _ffi.ptr(st), _ffi.c_uint64(9))
    ranks_u = np.empty((500, R), np.int32)
    neg_u = np.empty((500, R), np.int64)
    _ffi.call("rk_neg_sample_host", _ffi.ptr(st), _ffi.ptr(users), _ffi.c_int64(500), _ffi.c_int32(R), _ffi.ptr(ptr_),
              _ffi.ptr(items_), _ffi.c_int32(n_items), _ffi.ptr(ranks_u), _ffi.ptr(neg_u))
    assert np.array_equal(ranks_c, ranks_u) and np.array_equal(neg_c, neg_u) and np.array_equal(s.state, st)
    random.seed(9)
    all_items = set(range(n_items))
    for b, u in enumerate(users[:60]):
        pos = set(items_[ptr_[u]:ptr_[u + 1]].tolist())
        cands = list(all_items - pos)
        want = random.sample(cands, R) if len(cands) >= R else random.choices(cands, k=R)
        assert neg_c[b].tolist() == want


def test_sampler_over_a_sparse_id_space_and_its_cache_budget():
    """Item ids with holes (or positives from a part of the table): without `all_items` the sampler
    refuses -- the reference's candidate list is then its set's slot order, not range(n_items) --
    and with it the candidates are exactly all_items minus the user's positives, drawn by the same
    random.sample arithmetic over their ascending list.  A zero cache budget changes nothing but
    where the heavy users' candidate orders are rebuilt."""
    import random
    rng = np.random.default_rng(4)
    ids = np.sort(rng.choice(500, 120, replace=False))           # 120 of 500 ids occur
    U = 40
    users = rng.integers(0, U, 900)
    items = ids[rng.integers(0, len(ids), 900)]
    items[:110] = ids[:110]
    users[:110] = 3                                              # a heavy user: candidate order emulated
    p, it = build_user_csr(users, items, U)
    with pytest.raises(ValueError, match="never occur"):
        NegativeSampler(p, it, 500, seed=1)
    batch = np.concatenate([rng.integers(0, U, 60), [3, 3, 3]])
    got = {}
    for budget in (1 << 24, 0):
        s = NegativeSampler(p, it, 500, seed=1, neg_ratio=2, all_items=ids, cache_entries=budget)
        assert (s.cached_users.size > 0) == (budget > 0)
        ranks, neg = s.sample(batch)
        got[budget] = neg
        for u, row in zip(batch, neg):
            assert set(row.tolist()) <= set(ids.tolist()) - set(it[p[u]:p[u + 1]].tolist())
        with pytest.raises(ValueError, match="resolve the negatives on the host"):
            s.device_csr("cpu")
    np.testing.assert_array_equal(got[1 << 24], got[0])
    # light users: random.sample over the ascending candidate list, the same stream
    random.seed(1)
    for u, row in zip(batch, got[0]):
        pos = set(it[p[u]:p[u + 1]].tolist())
        cands = [int(i) for i in ids if int(i) not in pos]
        if len(pos) < (len(ids) >> 2):
            assert random.sample(cands, 2) == row.tolist()
        else:       # (heavy user: CPython's order differs from ascending; consume the same draws)
            random.sample(cands, 2)
