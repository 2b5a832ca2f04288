"""Negative sampling and training-batch production.

Restates the behaviour of GraphDataLoader (data/graph_data_loader.py:321-447 of
the reference) without pandas/DataLoader workers: the shuffle order comes from
the same torch RNG calls the DataLoader makes, the negative draws from the same
MT19937 stream CPython's `random` module uses (host part of librecommend_amd),
and the negative item ids are either resolved on the host (reference batch
layout, `neg_items`) or left as ranks that the BPR kernel resolves (`neg_rank`).
"""
import numpy as np
import torch

from . import _ffi
from ._ffi import c_int32, c_int64, ptr


def build_user_csr(users, items, n_users):
    """Per-user sorted unique item lists as CSR (ptr int32 [U+1], items int32)."""
    users = np.asarray(users, dtype=np.int64)
    items = np.asarray(items, dtype=np.int64)
    if users.shape[0] == 0:
        return np.zeros(n_users + 1, np.int32), np.zeros(0, np.int32)
    key = np.unique(users * (int(items.max()) + 1) + items)
    m = int(items.max()) + 1
    u = key // m
    it = (key % m).astype(np.int32)
    ptr_ = np.zeros(n_users + 1, np.int64)
    ptr_[1:] = np.bincount(u, minlength=n_users)
    return np.cumsum(ptr_).astype(np.int32), it


class NegativeSampler:
    """random.sample over the ascending list of items the user never touched
    (graph_data_loader.py:373-397), with CPython's generator state kept here."""

    def __init__(self, pos_ptr, pos_items, n_items, seed=42, neg_ratio=1, all_items=None,
                 cache_entries=1 << 24):
        """all_items: the item ids that count as candidates when not every id below n_items occurs
        in the interaction table (see below); cache_entries: how many candidate ids may be kept
        for the users whose candidate order is emulated (4 bytes each; users beyond the budget
        are emulated per sample)."""
        if neg_ratio <= 0:
            raise ValueError("Negative ratio must be greater than zero")
        self.pos_ptr = np.ascontiguousarray(pos_ptr, dtype=np.int32)
        self.pos_items = np.ascontiguousarray(pos_items, dtype=np.int32)
        self.n_items = int(n_items)
        self.cache_entries = int(cache_entries)
        self.item_ids = None      # rank -> item id when the candidates are a subset of range(n_items)
        # The reference draws from all_items = the ids that OCCUR in the interaction table
        # (data/graph_data_loader.py:173); this sampler ranks candidates over range(n_items).  The
        # two agree (and the random stream stays the reference's) only if every id occurs.
        if all_items is None:
            present = np.zeros(self.n_items, bool)
            present[self.pos_items[(self.pos_items >= 0) & (self.pos_items < self.n_items)]] = True
            if not present.all():
                missing = np.flatnonzero(~present)
                raise ValueError(
                    f"{missing.size} item ids below n_items={self.n_items} never occur in the interaction table "
                    f"(first: {missing[:5].tolist()}); the reference samples negatives from the ids that occur, "
                    "so its candidate lists and random stream differ from range(n_items): remap item ids densely, "
                    "or pass all_items=<the ids that occur> to sample from exactly those")
        else:
            # Sparse id space, or positives from a part of the table: candidates are `all_items`
            # minus the user's positives, in ASCENDING id order.  The reference lists them in the
            # slot order of its Python set, which for sparse ids is not ascending and is not
            # reproduced here: the same candidate sets and the same random stream, other picks.
            ids = np.unique(np.asarray(all_items, dtype=np.int64))
            if ids.size == 0 or ids[0] < 0:
                raise ValueError("all_items must hold non-negative item ids")
            self.item_ids = ids
            lens = np.diff(self.pos_ptr.astype(np.int64))
            rows = np.repeat(np.arange(lens.shape[0]), lens)
            pos = self.pos_items.astype(np.int64)
            where = np.searchsorted(ids, pos)
            where[where >= ids.size] = ids.size - 1
            known = ids[where] == pos                  # positives outside all_items are no candidates anyway
            new_ptr = np.zeros(lens.shape[0] + 1, np.int64)
            np.cumsum(np.bincount(rows[known], minlength=lens.shape[0]), out=new_ptr[1:])
            self.pos_ptr = new_ptr.astype(np.int32)
            self.pos_items = np.ascontiguousarray(where[known], dtype=np.int32)   # ranks: ascending per user
            self.n_items = int(ids.size)
        self.neg_ratio = int(neg_ratio)
        self.state = np.zeros(625, np.uint32)
        self.seed(seed)
        self._dev = {}
        self._cache_orders()

    def _cache_orders(self):
        """Users holding a quarter of all items or more have their candidates listed in CPython's
        hash-table order, which costs a pass over every item id to rebuild; there are few of them
        and they recur in batch after batch, so their orders are built once here (up to
        `cache_entries` ids in all)."""
        lens = np.diff(self.pos_ptr.astype(np.int64))
        heavy = np.flatnonzero((lens >= (self.n_items >> 2)) & (lens < self.n_items))
        sizes = self.n_items - lens[heavy]
        keep = np.cumsum(sizes) <= self.cache_entries
        heavy, sizes = heavy[keep], sizes[keep]
        self.cached_users = np.ascontiguousarray(heavy, dtype=np.int64)
        self.cached_ptr = np.zeros(heavy.size + 1, np.int64)
        np.cumsum(sizes, out=self.cached_ptr[1:])
        self.cached_order = np.empty(max(int(self.cached_ptr[-1]), 1), np.int32)
        for i, u in enumerate(heavy):
            lo, hi = int(self.pos_ptr[u]), int(self.pos_ptr[u + 1])
            _ffi.call("rk_neg_order_host", ptr(self.pos_items[lo:hi]), c_int32(hi - lo), c_int32(self.n_items),
                      ptr(self.cached_order[int(self.cached_ptr[i]):int(self.cached_ptr[i + 1])]))

    def seed(self, seed):
        """random.seed(seed) for a non-negative int below 2**64."""
        _ffi.call("rk_mt_seed_host", ptr(self.state), _ffi.c_uint64(int(seed)))

    def sample(self, user_ids, want_ids=True):
        """-> (ranks int32 [B,R], neg_items int64 [B,R] or None) as numpy arrays;
        consumes the MT19937 stream exactly like one _generate_negative_samples call."""
        users = np.ascontiguousarray(
            user_ids.cpu().numpy() if isinstance(user_ids, torch.Tensor) else user_ids,
            dtype=np.int64)
        B = users.shape[0]
        ranks = np.empty((B, self.neg_ratio), np.int32)
        neg = np.empty((B, self.neg_ratio), np.int64) if want_ids else None
        _ffi.call("rk_neg_sample_cached_host", ptr(self.state), ptr(users), c_int64(B),
                  c_int32(self.neg_ratio), ptr(self.pos_ptr), ptr(self.pos_items),
                  c_int32(self.n_items), ptr(self.cached_users), ptr(self.cached_ptr), ptr(self.cached_order),
                  c_int32(self.cached_users.shape[0]), ptr(ranks), ptr(neg))
        if neg is not None and self.item_ids is not None:
            neg = self.item_ids[neg]
        return ranks, neg

    def device_csr(self, device):
        """User->positives CSR in HBM for the in-kernel negative lookup."""
        if self.item_ids is not None:
            raise ValueError("the in-kernel negative lookup ranks candidates over range(n_items); with all_items=... "
                             "resolve the negatives on the host (TrainBatchLoader(resolve_on_host=True))")
        device = torch.device(device)
        if device not in self._dev:
            self._dev[device] = (torch.from_numpy(self.pos_ptr).to(device),
                                 torch.from_numpy(self.pos_items).to(device))
        return self._dev[device]


class TrainBatchLoader:
    """Iterates shuffled training batches as dicts with the reference's keys
    (`user_ids`, `item_ids`, `ratings`, `neg_items`) plus `neg_rank`.

    Shuffle order: the reference wraps a torch DataLoader(shuffle=True); creating
    its iterator draws one int64 (the loader's base seed) and the RandomSampler
    draws a second one to seed the generator behind torch.randperm.  The same two
    draws are made here so a run seeded like the reference sees the same batches.

    num_workers = 0: negatives come from the process-wide `random` stream, batch after batch
    (the sampler's own state, seeded like main.set_seed seeds `random`).
    num_workers = W > 0 (the reference's preset is 4, config.py:30): torch's DataLoader hands
    batch b to worker b mod W, and every worker process starts with random.seed(base_seed + w)
    (torch/utils/data/_utils/worker.py), base_seed being the first of the two draws above; new
    workers, hence new seeds, every epoch.  The W streams are reproduced here in one process."""

    def __init__(self, users, items, ratings, batch_size, sampler=None, shuffle=True,
                 device=None, resolve_on_host=True, num_workers=0):
        self.users = torch.as_tensor(np.asarray(users), dtype=torch.int64)
        self.items = torch.as_tensor(np.asarray(items), dtype=torch.int64)
        self.ratings = None if ratings is None else torch.as_tensor(np.asarray(ratings), dtype=torch.float32)
        self.batch_size = int(batch_size)
        self.sampler = sampler
        self.shuffle = shuffle
        self.device = device
        self.resolve_on_host = resolve_on_host
        self.num_workers = int(num_workers)

    def __len__(self):
        return (self.users.shape[0] + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        n = self.users.shape[0]
        # the iterator's base seed is drawn whether or not the loader shuffles
        base_seed = int(torch.empty((), dtype=torch.int64).random_().item())
        if self.shuffle:
            seed = int(torch.empty((), dtype=torch.int64).random_().item())
            g = torch.Generator()
            g.manual_seed(seed)
            order = torch.randperm(n, generator=g)
        else:
            order = torch.arange(n)
        worker_states = None
        if self.num_workers > 0 and self.sampler is not None:
            worker_states = []
            for w in range(self.num_workers):
                st = np.zeros(625, np.uint32)
                _ffi.call("rk_mt_seed_host", ptr(st), _ffi.c_uint64(base_seed + w))
                worker_states.append(st)
            own_state = self.sampler.state
        for bi, b in enumerate(range(0, n, self.batch_size)):
            idx = order[b:b + self.batch_size]
            if worker_states is not None:
                self.sampler.state = worker_states[bi % self.num_workers]
            batch = {"user_ids": self.users[idx], "item_ids": self.items[idx]}
            if self.ratings is not None:
                batch["ratings"] = self.ratings[idx]
            if self.sampler is not None:
                ranks, neg = self.sampler.sample(batch["user_ids"], want_ids=self.resolve_on_host)
                batch["neg_rank"] = torch.from_numpy(ranks)
                if neg is not None:
                    batch["neg_items"] = torch.from_numpy(neg)
            if worker_states is not None:
                self.sampler.state = own_state
            if self.device is not None:
                batch = {k: v.to(self.device, non_blocking=True) for k, v in batch.items()}
            yield batch
