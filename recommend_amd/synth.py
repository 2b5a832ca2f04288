"""Synthetic interaction tables at the published dataset shapes.

There is no dataset in the reference tree, so benchmarks and large parity tests
run on generated data: unique (user, item) pairs, power-law user activity and
item popularity (Zipf exponent 1), at least 5 interactions per user and 1 per
item (item ids dense, which the bit-exact negative sampler needs), ratings in
1..5 with p = (.05, .05, .10, .20, .60), split label 0/1/2 (train/val/test) with
p = (.8, .1, .1) and at least one training row per user.
"""
from dataclasses import dataclass

import numpy as np

SHAPES = {
    # name: (users, items, interactions)
    "baby": (19445, 7050, 160792),
    "sports": (35598, 18357, 296337),
    "clothing": (39387, 23033, 278677),
    "elec": (192403, 63001, 1689188),
}
CONFIG_INDEX = {"baby": 0, "sports": 1, "clothing": 2, "elec": 3, "scaleup": 4}


@dataclass
class Interactions:
    n_users: int
    n_items: int
    users: np.ndarray    # int64 [E]
    items: np.ndarray    # int64 [E]
    ratings: np.ndarray  # float32 [E]
    label: np.ndarray    # int8 [E]  0 train / 1 val / 2 test

    def split(self, which):
        m = self.label == which
        return self.users[m], self.items[m], self.ratings[m]


def _zipf_probs(n, rng):
    p = 1.0 / np.arange(1, n + 1, dtype=np.float64)
    p /= p.sum()
    return p[rng.permutation(n)]


def make_interactions(n_users, n_items, n_inter, seed=42, min_per_user=5):
    rng = np.random.default_rng(seed)
    if n_inter < n_users * min_per_user:
        min_per_user = max(1, n_inter // n_users)
    min_per_user = min(min_per_user, n_items)
    deg = np.full(n_users, min_per_user, np.int64)
    extra = n_inter - int(deg.sum())
    if extra > 0:
        deg += rng.multinomial(extra, _zipf_probs(n_users, rng))
    deg = np.minimum(deg, n_items)
    users = np.repeat(np.arange(n_users, dtype=np.int64), deg)
    E = users.shape[0]
    cdf = np.cumsum(_zipf_probs(n_items, rng))
    cdf[-1] = 1.0
    items = np.searchsorted(cdf, rng.random(E), side="right").astype(np.int64)
    # re-draw duplicate (user, item) pairs until all pairs are unique
    for attempt in range(200):
        key = users * n_items + items
        order = np.argsort(key, kind="stable")
        dup = np.zeros(E, bool)
        dup[order[1:]] = key[order[1:]] == key[order[:-1]]
        n_dup = int(dup.sum())
        if n_dup == 0:
            break
        if attempt >= 20:
            # near-saturated users: give each duplicate an item the user does not have yet
            for u in np.unique(users[dup]):
                mine = np.flatnonzero(users == u)
                d_u = mine[dup[mine]]
                free = np.setdiff1d(np.arange(n_items), items[mine[~dup[mine]]])
                items[d_u] = rng.choice(free, d_u.shape[0], replace=False)
            continue
        # heavy users exhaust the popular items: fall back to uniform draws for them
        items[dup] = rng.integers(0, n_items, n_dup)
    else:
        raise RuntimeError("could not make interactions unique")
    # every item id must occur at least once
    counts = np.bincount(items, minlength=n_items)
    missing = np.flatnonzero(counts == 0)
    if missing.size:
        cand = rng.permutation(E)
        taken = 0
        for pos in cand:
            if taken == missing.size:
                break
            if counts[items[pos]] > 1:
                counts[items[pos]] -= 1
                items[pos] = missing[taken]
                counts[missing[taken]] += 1
                taken += 1
        if taken != missing.size:
            raise RuntimeError("could not cover every item")
    perm = rng.permutation(E)  # dataset order is not sorted by user
    users, items = users[perm], items[perm]
    ratings = rng.choice(np.arange(1, 6), size=E, p=[.05, .05, .10, .20, .60]).astype(np.float32)
    label = rng.choice(np.arange(3), size=E, p=[.8, .1, .1]).astype(np.int8)
    # at least one training row per user
    has_train = np.zeros(n_users, bool)
    has_train[users[label == 0]] = True
    first = np.full(n_users, -1, np.int64)
    idx = np.arange(E)[::-1]
    first[users[idx]] = idx
    fix = np.flatnonzero(~has_train & (first >= 0))
    label[first[fix]] = 0
    return Interactions(n_users, n_items, users, items, ratings, label)


def make_named(name, seed=42):
    n_users, n_items, n_inter = SHAPES[name]
    return make_interactions(n_users, n_items, n_inter, seed + CONFIG_INDEX[name])


def make_features(n_items, seed=42, image_dim=4096, text_dim=384):
    """item_image_feat ~ N(0,1); item_text_feat ~ N(0,1) with L2-normalised rows (fp32)."""
    rng = np.random.default_rng(seed + 1000)
    image = rng.standard_normal((n_items, image_dim), dtype=np.float32)
    text = rng.standard_normal((n_items, text_dim), dtype=np.float32)
    text /= np.linalg.norm(text, axis=1, keepdims=True)
    return image, text
