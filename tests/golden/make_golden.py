"""Generates tests/golden/* by running the reference's own modules on seeded inputs.

Run in the build container only (needs /root/reference; the GPU box never runs
this).  Imports evalue.metric, evalue.loss, evalue.evaluator, config and
data.graph_data_loader from the reference tree, feeds them synthetic inputs and
stores inputs + outputs as small .npz / .json fixtures.  Nothing of the
reference's source is copied; the fixtures are data.

    python tests/golden/make_golden.py
"""
import json
import os
import random
import shutil
import sys
import tempfile

import numpy as np
import torch

REF = os.environ.get("REFERENCE_ROOT", "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REF)

from evalue.metric import (evaluate_all_at_k, precision_at_k, recall_at_k, hit_rate_at_k,  # noqa: E402
                           ndcg_at_k, map_at_k, mrr_at_k)
from evalue.loss import bpr_loss, compute_info_bpr_loss, mmgcn_loss  # noqa: E402
from evalue.evaluator import Verifier, Tester  # noqa: E402
import config as ref_config  # noqa: E402
from data.graph_data_loader import create_data_loaders  # noqa: E402

METRICS = ["precision", "recall", "hit_rate", "ndcg", "map", "mrr"]


def metric_vec(d):
    return np.array([float(d[m]) for m in METRICS], np.float32)


def min_rel_gap(scores, k):
    """Smallest relative gap between neighbours in the sorted top-(k+1) of every row."""
    k1 = min(k + 1, scores.shape[1])
    top = torch.topk(scores, k1, dim=1).values.double()
    gap = (top[:, :-1] - top[:, 1:]) / top[:, :-1].abs().clamp_min(1e-30)
    return float(gap.min())


def golden_metrics():
    cases = {}
    # hand-built edge cases
    S = torch.tensor([[0.9, 0.1, 0.8, 0.3, 0.7, 0.2], [0.1, 0.2, 0.3, 0.4, 0.5, 0.6],
                      [0.5, 0.4, 0.3, 0.2, 0.1, 0.0]])
    T = torch.tensor([[1, 0, 0, 1, 1, 0], [0, 0, 0, 0, 0, 0], [0, 0, 0, 0, 0, 1.]])
    cases["tiny"] = (S, T, [1, 3, 6])
    S2 = S.clone()
    S2[0, 0] = -1e10
    cases["tiny_masked"] = (S2, T, [3])
    g = torch.Generator().manual_seed(7)
    Sa = torch.rand(8, 40, generator=g)
    all_hit = torch.zeros(8, 40)
    all_hit.scatter_(1, torch.topk(Sa, 5, dim=1).indices, 1.0)
    cases["all_hit"] = (Sa, all_hit, [1, 5])
    cases["no_hit"] = (Sa, torch.zeros(8, 40), [5, 10])
    many = (torch.rand(8, 40, generator=g) < 0.6).float()   # positives > k
    cases["many_pos"] = (Sa, many, [5, 10, 20, 40])
    Sr = torch.randn(64, 500, generator=g)
    Tr = (torch.rand(64, 500, generator=g) < 0.01).float()
    Tr[5] = 0
    cases["rand_64x500"] = (Sr, Tr, [5, 10, 20])
    out = {}
    for name, (s, t, ks) in cases.items():
        out[f"{name}/scores"] = s.numpy()
        out[f"{name}/targets"] = t.numpy()
        out[f"{name}/ks"] = np.array(ks)
        for k in ks:
            d = evaluate_all_at_k(s, t, k)
            out[f"{name}/all@{k}"] = metric_vec(d)
            single = {"precision": precision_at_k(s, t, k), "recall": recall_at_k(s, t, k),
                      "hit_rate": hit_rate_at_k(s, t, k), "ndcg": ndcg_at_k(s, t, k),
                      "map": map_at_k(s, t, k), "mrr": mrr_at_k(s, t, k)}
            out[f"{name}/single@{k}"] = metric_vec(single)
            out[f"{name}/topk@{k}"] = torch.topk(s, k, dim=1, largest=True, sorted=True).indices.numpy()
    # error behaviour of _validate_inputs
    errs = {}
    for label, (s, t, k) in {"k0": (S, T, 0), "k_gt": (S, T, 7), "shape": (S, T[:, :5], 2),
                             "dim": (S[0], T[0], 1)}.items():
        try:
            evaluate_all_at_k(s, t, k)
            errs[label] = ""
        except ValueError as e:
            errs[label] = str(e)
    np.savez_compressed(os.path.join(OUT, "metric_cases.npz"), **out)
    with open(os.path.join(OUT, "metric_errors.json"), "w") as f:
        json.dump(errs, f, indent=1)


class StubModel:
    def __init__(self, u, i):
        self.u, self.i = u, i

    def eval(self):
        return self

    def __call__(self, batch):
        return {"user_embeddings": self.u, "item_embeddings": self.i}


def golden_verifier():
    """Verifier.verify / Tester.test end to end with fixed embeddings."""
    out = {}
    for name, (U, I, D, seed) in {"v96x300": (96, 300, 64, 11), "v200x1000": (200, 1000, 128, 12)}.items():
        g = torch.Generator().manual_seed(seed)
        ue = torch.randn(U, D, generator=g) * 0.3
        ie = torch.randn(I, D, generator=g) * 0.3
        rng = np.random.default_rng(seed)
        # interactions: train pairs (mask) and validation pairs (target), main.py:117-139
        n_tr, n_va = U * 6, U * 2
        tr = np.unique(np.stack([rng.integers(0, U, n_tr), rng.integers(0, I, n_tr)], 1), axis=0)
        va = np.unique(np.stack([rng.integers(0, U, n_va), rng.integers(0, I, n_va)], 1), axis=0)
        va = va[va[:, 0] % 7 != 3]                      # some users without validation items
        target = torch.zeros((U, I))
        target[va[:, 0], va[:, 1]] = 1
        row_id, col_id = tr[:, 0].tolist(), tr[:, 1].tolist()
        cfg = ref_config.get_config("baby")
        cfg.system.device = "cpu"
        res_v = Verifier(cfg, None, target, (row_id, col_id)).verify(StubModel(ue, ie))
        res_t = Tester(cfg, None, target, (row_id, col_id)).test(StubModel(ue, ie))
        assert all(float(res_v[k]) == float(res_t[k]) for k in res_v)
        S = torch.matmul(ue, ie.T)
        S[row_id, col_id] = -1e10
        gap = min_rel_gap(S, 20)
        assert gap > 1e-6, (name, gap)
        out[f"{name}/user_emb"] = ue.numpy()
        out[f"{name}/item_emb"] = ie.numpy()
        out[f"{name}/train_pairs"] = tr
        out[f"{name}/val_pairs"] = va
        out[f"{name}/keys"] = np.array(list(res_v.keys()))
        out[f"{name}/values"] = np.array([float(v) for v in res_v.values()], np.float32)
        out[f"{name}/topk20"] = torch.topk(S, 20, dim=1, largest=True, sorted=True).indices.numpy()
        out[f"{name}/min_rel_gap"] = np.array(gap)
    np.savez_compressed(os.path.join(OUT, "verifier_cases.npz"), **out)


def golden_losses():
    out = {}
    g = torch.Generator().manual_seed(3)
    # known answers on hand-written scores
    out["ka/bpr_nomask"] = np.array(float(bpr_loss(torch.tensor([1., 2., -1.]),
                                                   torch.tensor([[.5, 0.], [3., 1.], [0., 0.]]))))
    out["ka/bpr_mask"] = np.array(float(bpr_loss(torch.tensor([1., 2.]), torch.tensor([[.5, 0.], [3., 1.]]),
                                                 torch.tensor([[1., 0.], [1., 1.]]))))
    for name, (U, I, D, B, R) in {"b64r1": (50, 80, 64, 64, 1), "b200r4": (120, 300, 64, 200, 4),
                                  "b33r3d128": (40, 70, 128, 33, 3)}.items():
        ue = (torch.randn(U, D, generator=g) * 0.4).requires_grad_(True)
        ie = (torch.randn(I, D, generator=g) * 0.4).requires_grad_(True)
        users = torch.randint(0, U, (B,), generator=g)
        pos = torch.randint(0, I, (B,), generator=g)
        neg = torch.randint(0, I, (B, R), generator=g)
        # GraphTrainer._loss_func body (train/graph_trainer.py:264-269) around the reference's bpr_loss
        pu, pi, ni = ue[users], ie[pos], ie[neg]
        pos_scores = torch.sum(pu * pi, dim=-1)
        neg_scores = torch.sum(pu.unsqueeze(1) * ni, dim=-1)
        loss = bpr_loss(pos_scores, neg_scores)
        gu, gi = torch.autograd.grad(loss, [ue, ie], retain_graph=True)
        mask = (torch.rand(B, R, generator=g) < 0.7).float()
        loss_m = bpr_loss(pos_scores, neg_scores, mask)
        gum, gim = torch.autograd.grad(loss_m, [ue, ie])
        out.update({f"{name}/user_emb": ue.detach().numpy(), f"{name}/item_emb": ie.detach().numpy(),
                    f"{name}/users": users.numpy(), f"{name}/pos": pos.numpy(), f"{name}/neg": neg.numpy(),
                    f"{name}/pos_scores": pos_scores.detach().numpy(),
                    f"{name}/neg_scores": neg_scores.detach().numpy(),
                    f"{name}/loss": np.array(float(loss)), f"{name}/grad_user": gu.numpy(),
                    f"{name}/grad_item": gi.numpy(), f"{name}/mask": mask.numpy(),
                    f"{name}/loss_masked": np.array(float(loss_m)),
                    f"{name}/grad_user_masked": gum.numpy(), f"{name}/grad_item_masked": gim.numpy()})
        if R == 1:
            outputs = {"user_embeddings": ue.detach(), "item_embeddings": ie.detach(),
                       "id_embeddings": torch.cat([ue, ie]).detach(), "pres": ue.detach()}
            batch = {"user_ids": users, "item_ids": pos, "neg_items": neg}
            out[f"{name}/mmgcn_loss"] = np.array(float(mmgcn_loss(outputs, batch)))
        edges = torch.stack([users, pos], 1)
        out[f"{name}/info_bpr_loss"] = np.array(float(compute_info_bpr_loss(ue.detach(), ie.detach(), edges, neg)))
    np.savez_compressed(os.path.join(OUT, "loss_cases.npz"), **out)


def golden_loader():
    """Batches of the reference GraphDataLoader (num_workers=0) on a dense-id synthetic table."""
    U, I, E = 60, 45, 420
    rng = np.random.default_rng(5)
    pairs = np.unique(np.stack([rng.integers(0, U, E * 2), rng.integers(0, I, E * 2)], 1), axis=0)
    pairs = pairs[rng.permutation(len(pairs))][:E]
    # make every user and item id occur
    pairs[:U, 0] = np.arange(U)
    pairs[U:U + I, 1] = np.arange(I)
    pairs = np.unique(pairs, axis=0)
    pairs = pairs[rng.permutation(len(pairs))]
    rating = rng.integers(1, 6, len(pairs)).astype(np.float64)
    label = rng.choice(3, len(pairs), p=[.8, .1, .1])
    tmp = tempfile.mkdtemp()
    try:
        with open(os.path.join(tmp, "tiny.inter"), "w") as f:
            f.write("userID\titemID\trating\tx_label\n")
            for (u, i), r, l in zip(pairs, rating, label):
                f.write(f"{u}\t{i}\t{r}\t{l}\n")
        np.save(os.path.join(tmp, "item_image_feat.npy"), rng.standard_normal((I, 8)).astype(np.float32))
        out = {"users": pairs[:, 0], "items": pairs[:, 1], "rating": rating.astype(np.float32),
               "label": label, "n_users": np.array(U), "n_items": np.array(I)}
        # (neg_ratio, batch size, DataLoader workers): 0 workers and the reference's preset of 4
        # (config.py:30), where batch b is collated in worker b mod 4
        for neg_ratio, bs, workers in [(1, 64, 0), (3, 50, 0), (8, 32, 0), (1, 32, 4), (2, 48, 4)]:
            cfg = ref_config.get_config("baby")
            cfg.data.data_path = tmp
            cfg.data.num_users, cfg.data.num_items = U, I
            cfg.data.batch_size, cfg.data.num_workers, cfg.data.neg_ratio = bs, workers, neg_ratio
            cfg.system.device = "cpu"
            # main.set_seed (main.py:31-39)
            random.seed(42)
            np.random.seed(42)
            torch.manual_seed(42)
            train_loader, val_loader, test_loader = create_data_loaders(cfg)
            tag = f"r{neg_ratio}_b{bs}" + (f"_w{workers}" if workers else "")
            ds = train_loader.dataset
            out[f"{tag}/train_users"] = ds.df[cfg.data.user_col].to_numpy()
            out[f"{tag}/train_items"] = ds.df[cfg.data.item_col].to_numpy()
            out[f"{tag}/val_len"] = np.array(len(val_loader.dataset))
            out[f"{tag}/test_len"] = np.array(len(test_loader.dataset))
            for epoch in range(2):
                batches = list(train_loader)
                out[f"{tag}/e{epoch}/user_ids"] = np.concatenate([b["user_ids"].numpy() for b in batches])
                out[f"{tag}/e{epoch}/item_ids"] = np.concatenate([b["item_ids"].numpy() for b in batches])
                out[f"{tag}/e{epoch}/ratings"] = np.concatenate([b["ratings"].numpy() for b in batches])
                out[f"{tag}/e{epoch}/neg_items"] = np.concatenate([b["neg_items"].numpy() for b in batches])
                out[f"{tag}/e{epoch}/n_batches"] = np.array(len(batches))
        np.savez_compressed(os.path.join(OUT, "loader_batches.npz"), **out)
    finally:
        shutil.rmtree(tmp)


def golden_data_path():
    """The reference's dataset class on a table with users that only occur in val / test, a
    mapping file that makes the id space larger than the ids seen, and two feature files: the
    splits (graph_data_loader.py:209-247), the per-sample extraction loop of main.py:92-100 run
    on the reference's __getitem__, the positives index (:163-173) and the groupby expressions of
    main.mask_index (main.py:127-137; main itself imports DGL and cannot be imported here, so
    those two pandas expressions are evaluated on the reference's own split DataFrames)."""
    from data.graph_data_loader import GraphRecDataset
    U, I = 50, 30
    rng = np.random.default_rng(17)
    pairs = np.unique(np.stack([rng.integers(0, U - 4, 700), rng.integers(0, I - 2, 700)], 1), axis=0)
    pairs = pairs[rng.permutation(len(pairs))][:360]
    label = rng.choice(3, len(pairs), p=[.7, .15, .15])
    # users 40..43 never in train
    label[np.isin(pairs[:, 0], [40, 41, 42, 43]) & (label == 0)] = 1
    rating = rng.integers(1, 6, len(pairs)).astype(np.float64)
    tmp = tempfile.mkdtemp()
    try:
        with open(os.path.join(tmp, "tiny.inter"), "w") as f:
            f.write("userID\titemID\trating\tx_label\n")
            for (u, i), r, l in zip(pairs, rating, label):
                f.write(f"{u}\t{i}\t{r}\t{l}\n")
        with open(os.path.join(tmp, "u_id_mapping.csv"), "w") as f:
            f.write("asin,userID\n" + "".join(f"a{k},{k}\n" for k in range(U)))
        with open(os.path.join(tmp, "i_id_mapping.csv"), "w") as f:
            f.write("asin,itemID\n" + "".join(f"b{k},{k}\n" for k in range(I)))
        np.save(os.path.join(tmp, "item_image_feat.npy"), rng.standard_normal((I, 6)).astype(np.float32))
        np.save(os.path.join(tmp, "text_feat_item.npy"), rng.standard_normal((I, 4)).astype(np.float32))
        out = {"inter_text": np.frombuffer(open(os.path.join(tmp, "tiny.inter"), "rb").read(), np.uint8)}
        for tag, rating_col, filt in [("rated", "rating", True), ("unrated", None, True), ("nofilter", "rating", False)]:
            cfg = ref_config.get_config("baby")
            cfg.data.data_path = tmp
            cfg.data.num_users, cfg.data.num_items = -1, -1
            cfg.data.rating_col = rating_col
            cfg.data.filter_out_new_users = filt
            import warnings as _w
            with _w.catch_warnings():
                _w.simplefilter("ignore")
                full = GraphRecDataset(cfg)
                splits = full.split()
            out[f"{tag}/num_users"] = np.array(full.user_num)
            out[f"{tag}/num_items"] = np.array(full.item_num)
            out[f"{tag}/item_feature_names"] = np.array(sorted(full.item_features.keys()))
            for name, ds in zip(("train", "val", "test"), splits):
                # main.py:92-100 on the reference's own __getitem__
                us, its, rs = [], [], []
                for idx in range(len(ds)):
                    sample = ds[idx]
                    us.append(sample["user_id"].item())
                    its.append(sample["item_id"].item())
                    rs.append(sample["rating"].item() if "rating" in sample else 1)
                out[f"{tag}/{name}/users"] = np.array(us, np.int64)
                out[f"{tag}/{name}/items"] = np.array(its, np.int64)
                out[f"{tag}/{name}/ratings"] = np.array(rs, np.float32)
            # positives index of the full table, as sorted CSR
            ptr = [0]
            cols = []
            for u in range(full.user_num):
                its = sorted(int(v) for v in full.user_item_interaction.get(u, ()))
                cols.extend(its)
                ptr.append(len(cols))
            out[f"{tag}/pos_ptr"] = np.array(ptr, np.int64)
            out[f"{tag}/pos_items"] = np.array(cols, np.int64)
            # mask_index(config, val, [train]) and (config, test, [train, val])
            ucol, icol = cfg.data.user_col, cfg.data.item_col
            for case, target, need in [("val", splits[1], [splits[0]]), ("test", splits[2], [splits[0], splits[1]])]:
                tr, tc = [], []
                for user_id, item_id in target.df.groupby(ucol)[icol].apply(list).to_dict().items():
                    tc.extend(item_id)
                    tr.extend([user_id] * len(item_id))
                row_id, col_id = [], []
                for ds in need:
                    for user_id, item_id in ds.df.groupby(ucol)[icol].apply(list).to_dict().items():
                        col_id.extend(item_id)
                        row_id.extend([user_id] * len(item_id))
                out[f"{tag}/mask_{case}/target_rows"] = np.array(tr, np.int64)
                out[f"{tag}/mask_{case}/target_cols"] = np.array(tc, np.int64)
                out[f"{tag}/mask_{case}/row_id"] = np.array(row_id, np.int64)
                out[f"{tag}/mask_{case}/col_id"] = np.array(col_id, np.int64)
        out["item_image_feat"] = np.load(os.path.join(tmp, "item_image_feat.npy"))
        out["text_feat_item"] = np.load(os.path.join(tmp, "text_feat_item.npy"))
        np.savez_compressed(os.path.join(OUT, "data_path.npz"), **out)
    finally:
        shutil.rmtree(tmp)


def golden_config():
    d = {name: ref_config.get_config(name).to_dict() for name in ["baby", "clothing", "sports", "elec"]}
    with open(os.path.join(OUT, "config_presets.json"), "w") as f:
        json.dump(d, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    torch.set_num_threads(4)
    golden_metrics()
    golden_verifier()
    golden_losses()
    golden_loader()
    golden_data_path()
    golden_config()
    print("golden fixtures written to", OUT)
