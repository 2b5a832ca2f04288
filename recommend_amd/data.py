"""Interaction tables in the reference's on-disk format, read into flat arrays.

Same files and meaning as data/graph_data_loader.py of the reference (class
GraphRecDataset: _load_interactions :86-112, _calculate_num :114-128,
_load_features :130-151, _build_user_item_interaction :163-173, split :209-247)
and the two extraction loops of main.py (build_graph_and_model :87-100,
mask_index :117-139): a `*.inter` file with a header row (tab separated by
default) and `*item*.npy` / `*user*.npy` feature matrices next to it.

What changes is the form of the result.  The reference walks the DataFrame one
row at a time (`df.iloc[idx]`, `iterrows`) and builds Python lists, sets and a
dense [U, I] target matrix; here every step is a column operation and what comes
out are the arrays the kernels take: (users, items, ratings) in file order for
the graph, CSR positives for the negative sampler, and (row, col) id pairs for
the evaluator's target and mask.
"""
import os
import warnings
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np


@dataclass
class InteractionSplit:
    """One of train / val / test: the rows of the table with that label, in file order."""
    users: np.ndarray                 # int64 [n]
    items: np.ndarray                 # int64 [n]
    ratings: Optional[np.ndarray]     # float32 [n], or None when the table has no rating column
    mode: str = "train"

    def __len__(self):
        return int(self.users.shape[0])

    def triplets(self):
        """What main.py:87-100 collects for build_graph, as three arrays: a missing
        rating is 1 (main.py:98-99)."""
        r = self.ratings if self.ratings is not None else np.ones(len(self), np.float32)
        return self.users, self.items, r


class InteractionTable:
    """The full `.inter` table plus the features found beside it."""

    def __init__(self, users, items, labels, ratings=None, num_users=-1, num_items=-1,
                 filter_out_new_users=True):
        self.users = np.ascontiguousarray(users, dtype=np.int64)
        self.items = np.ascontiguousarray(items, dtype=np.int64)
        self.labels = np.ascontiguousarray(labels, dtype=np.int64)
        self.ratings = None if ratings is None else np.ascontiguousarray(ratings, dtype=np.float32)
        if not (self.users.shape == self.items.shape == self.labels.shape) or self.users.ndim != 1:
            raise ValueError("users, items and labels must be 1-D arrays of one length")
        if self.ratings is not None and self.ratings.shape != self.users.shape:
            raise ValueError("ratings must have one value per interaction")
        # graph_data_loader.py:121,128: without a mapping file the counts are the numbers of
        # distinct ids (NOT max id + 1)
        self.num_users = int(num_users) if num_users != -1 else int(np.unique(self.users).shape[0])
        self.num_items = int(num_items) if num_items != -1 else int(np.unique(self.items).shape[0])
        self.filter_out_new_users = filter_out_new_users
        self.user_features: Dict[str, np.ndarray] = {}
        self.item_features: Dict[str, np.ndarray] = {}

    # ------------------------------------------------------------------ reading
    @classmethod
    def from_dir(cls, data_path, user_col="userID", item_col="itemID", rating_col=None,
                 splitting_label="x_label", sep="\t", num_users=-1, num_items=-1,
                 filter_out_new_users=True, load_features=True):
        """Reads the first `*.inter` file of `data_path` (directory order, as :92-95)."""
        import pandas as pd

        inter = next((f for f in os.listdir(data_path) if f.endswith(".inter")), None)
        if inter is None:
            raise FileNotFoundError(f"No interaction file found in {data_path}")
        df = pd.read_csv(os.path.join(data_path, inter), sep=sep)
        required = [user_col, item_col, splitting_label] + ([rating_col] if rating_col else [])
        missing = [c for c in required if c not in df.columns]
        if missing:
            raise ValueError(f"Missing required columns: {missing}")
        if num_users == -1:
            p = os.path.join(data_path, "u_id_mapping.csv")
            if os.path.exists(p):
                num_users = len(pd.read_csv(p))
        if num_items == -1:
            p = os.path.join(data_path, "i_id_mapping.csv")
            if os.path.exists(p):
                num_items = len(pd.read_csv(p))
        t = cls(df[user_col].to_numpy(), df[item_col].to_numpy(), df[splitting_label].to_numpy(),
                df[rating_col].to_numpy() if rating_col else None, num_users, num_items,
                filter_out_new_users)
        if load_features:
            t.load_features(data_path)
        return t

    @classmethod
    def from_config(cls, config, load_features=True):
        d = config.data
        return cls.from_dir(d.data_path, d.user_col, d.item_col, d.rating_col, d.splitting_label,
                            d.sep, d.num_users, d.num_items, d.filter_out_new_users, load_features)

    def load_features(self, data_path):
        """`.npy` files with 'user' / 'item' in the name (:134-148); the feature name is the file
        name without those words.  Arrays only: a pickled object file is refused, not executed."""
        for file in os.listdir(data_path):
            if not file.endswith(".npy"):
                continue
            low = file.lower()
            if "user" not in low and "item" not in low:
                continue
            try:
                arr = np.load(os.path.join(data_path, file), allow_pickle=False)
            except Exception as e:  # same policy as :147-148
                warnings.warn(f"Failed to load feature file {file}: {e}")
                continue
            arr = np.ascontiguousarray(arr, dtype=np.float32)
            stem = file.replace(".npy", "")
            if "user" in low:
                name = stem.replace("user_", "").replace("_user", "").replace("user", "")
                self.user_features[name] = arr
                if arr.shape[0] != self.num_users:
                    warnings.warn(f"User feature {name} has {arr.shape[0]} rows, expected {self.num_users}")
            else:
                name = stem.replace("item_", "").replace("_item", "").replace("item", "")
                self.item_features[name] = arr
                if arr.shape[0] != self.num_items:
                    warnings.warn(f"Item feature {name} has {arr.shape[0]} rows, expected {self.num_items}")

    def get_features(self):
        import torch
        return ({k: torch.from_numpy(v) for k, v in self.user_features.items()},
                {k: torch.from_numpy(v) for k, v in self.item_features.items()})

    # ------------------------------------------------------------------ splitting
    def split(self) -> Tuple[InteractionSplit, InteractionSplit, InteractionSplit]:
        """Rows with label 0 / 1 / 2 in file order; val and test lose the rows of users that
        never occur in train when filter_out_new_users is set (:223-237)."""
        out = []
        train_users = None
        for label, mode in enumerate(("train", "val", "test")):
            keep = self.labels == label
            if label == 0:
                train_users = np.unique(self.users[keep])
            elif self.filter_out_new_users:
                known = np.isin(self.users, train_users)
                removed = int(np.count_nonzero(keep & ~known))
                if removed > 0:
                    warnings.warn(f"Filtered out {removed} interactions with new users in {mode} set")
                keep &= known
            out.append(InteractionSplit(self.users[keep], self.items[keep],
                                        None if self.ratings is None else self.ratings[keep], mode))
        return tuple(out)

    def __len__(self):
        return int(self.users.shape[0])

    # ------------------------------------------------------------------ derived indices
    def user_positive_csr(self):
        """All items a user has in the FULL table (the sets of :166-170), as CSR with sorted,
        distinct columns: what recommend_amd.sampler.NegativeSampler takes."""
        return positives_csr(self.users, self.items, self.num_users)

    def get_statistics(self):
        n = len(self)
        stats = {
            "num_users": self.num_users,
            "num_items": self.num_items,
            "num_interactions": n,
            "sparsity": 1 - (n / (self.num_users * self.num_items)),
            "user_features": list(self.user_features.keys()),
            "item_features_dimensions": {k: v.shape[1] for k, v in self.item_features.items()},
            "user_feature_dimensions": {k: v.shape[1] for k, v in self.user_features.items()},
        }
        for label, name in enumerate(("train_ratio", "val_ratio", "test_ratio")):
            stats[name] = float(np.count_nonzero(self.labels == label)) / max(n, 1)
        return stats


def positives_csr(users, items, num_users):
    """CSR of distinct (user, item) pairs, columns ascending within a row."""
    users = np.asarray(users, np.int64)
    items = np.asarray(items, np.int64)
    if users.size and (users.min() < 0 or users.max() >= num_users):
        raise ValueError("user id outside [0, num_users)")
    order = np.lexsort((items, users))
    u, i = users[order], items[order]
    first = np.ones(u.shape[0], bool)
    first[1:] = (u[1:] != u[:-1]) | (i[1:] != i[:-1])
    u, i = u[first], i[first]
    ptr = np.zeros(num_users + 1, np.int64)
    np.cumsum(np.bincount(u, minlength=num_users), out=ptr[1:])
    return ptr, i


def grouped_pairs(users, items):
    """(row_id, col_id) in the order mask_index produces them (main.py:127-137): users
    ascending (pandas groupby sorts its keys), a user's items in file order."""
    users = np.asarray(users, np.int64)
    order = np.argsort(users, kind="stable")
    return users[order], np.asarray(items, np.int64)[order]


def mask_index(target: InteractionSplit, need_mask: Sequence[InteractionSplit]):
    """main.py:117-139 without the dense matrix: returns (target, mask) where target is the
    (row_ids, col_ids) of the ones of the reference's [U, I] matrix and mask is its
    (row_id, col_id), both as int64 arrays in the reference's order.  Both forms are accepted
    by recommend_amd.evaluator.Verifier / Tester."""
    t_rows, t_cols = grouped_pairs(target.users, target.items)
    rows: List[np.ndarray] = []
    cols: List[np.ndarray] = []
    for s in need_mask:
        r, c = grouped_pairs(s.users, s.items)
        rows.append(r)
        cols.append(c)
    if rows:
        mask = (np.concatenate(rows), np.concatenate(cols))
    else:
        mask = (np.zeros(0, np.int64), np.zeros(0, np.int64))
    return (t_rows, t_cols), mask


__all__ = ["InteractionTable", "InteractionSplit", "positives_csr", "grouped_pairs", "mask_index"]
