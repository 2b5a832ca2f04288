"""Host data path against fixtures produced by the reference's own dataset class
(tests/golden/make_golden.py::golden_data_path): the `.inter` table, its splits
with and without the new-user filter, the interaction triplets that go into
build_graph, the positives index of the negative sampler and the (row, col)
pairs of main.mask_index.  Integer work: everything must match exactly."""
import os
import warnings

import numpy as np
import pytest

from recommend_amd import data as rdata

GOLD = os.path.join(os.path.dirname(__file__), "golden", "data_path.npz")


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD)


@pytest.fixture()
def data_dir(tmp_path, gold):
    (tmp_path / "tiny.inter").write_bytes(gold["inter_text"].tobytes())
    n_u, n_i = int(gold["rated/num_users"]), int(gold["rated/num_items"])
    (tmp_path / "u_id_mapping.csv").write_text("asin,userID\n" + "".join(f"a{k},{k}\n" for k in range(n_u)))
    (tmp_path / "i_id_mapping.csv").write_text("asin,itemID\n" + "".join(f"b{k},{k}\n" for k in range(n_i)))
    np.save(tmp_path / "item_image_feat.npy", gold["item_image_feat"])
    np.save(tmp_path / "text_feat_item.npy", gold["text_feat_item"])
    return str(tmp_path)


@pytest.mark.parametrize("tag,rating_col,filt", [("rated", "rating", True), ("unrated", None, True),
                                                 ("nofilter", "rating", False)])
def test_table_splits_and_indices_match_the_reference(gold, data_dir, tag, rating_col, filt):
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        t = rdata.InteractionTable.from_dir(data_dir, rating_col=rating_col, filter_out_new_users=filt)
        splits = t.split()
    assert t.num_users == int(gold[f"{tag}/num_users"]) and t.num_items == int(gold[f"{tag}/num_items"])
    assert sorted(t.item_features) == list(gold[f"{tag}/item_feature_names"])
    assert np.array_equal(t.item_features["image_feat"], gold["item_image_feat"])
    assert np.array_equal(t.item_features["text_feat"], gold["text_feat_item"])
    for name, s in zip(("train", "val", "test"), splits):
        u, i, r = s.triplets()
        assert np.array_equal(u, gold[f"{tag}/{name}/users"]), name
        assert np.array_equal(i, gold[f"{tag}/{name}/items"]), name
        assert np.array_equal(r, gold[f"{tag}/{name}/ratings"]), name
    ptr, items = t.user_positive_csr()
    assert np.array_equal(ptr, gold[f"{tag}/pos_ptr"]) and np.array_equal(items, gold[f"{tag}/pos_items"])
    for case, target, need in [("val", splits[1], [splits[0]]), ("test", splits[2], [splits[0], splits[1]])]:
        (tr, tc), (mr, mc) = rdata.mask_index(target, need)
        assert np.array_equal(tr, gold[f"{tag}/mask_{case}/target_rows"])
        assert np.array_equal(tc, gold[f"{tag}/mask_{case}/target_cols"])
        assert np.array_equal(mr, gold[f"{tag}/mask_{case}/row_id"])
        assert np.array_equal(mc, gold[f"{tag}/mask_{case}/col_id"])


def test_new_user_filter_warns_and_counts(gold, data_dir):
    with pytest.warns(UserWarning, match="new users in val"):
        rdata.InteractionTable.from_dir(data_dir).split()


def test_counts_without_mapping_files_are_distinct_ids(tmp_path):
    (tmp_path / "x.inter").write_text("userID\titemID\tx_label\n5\t9\t0\n5\t2\t0\n7\t9\t1\n")
    t = rdata.InteractionTable.from_dir(str(tmp_path))
    assert (t.num_users, t.num_items) == (2, 2)      # graph_data_loader.py:121,128
    assert len(t.split()[1]) == 0                    # user 7 is new in val


def test_missing_file_and_columns_raise(tmp_path):
    with pytest.raises(FileNotFoundError):
        rdata.InteractionTable.from_dir(str(tmp_path))
    (tmp_path / "x.inter").write_text("userID\titemID\n1\t2\n")
    with pytest.raises(ValueError, match="Missing required columns"):
        rdata.InteractionTable.from_dir(str(tmp_path))


def test_empty_mask_list_and_empty_split():
    s = rdata.InteractionSplit(np.zeros(0, np.int64), np.zeros(0, np.int64), None)
    (tr, tc), (mr, mc) = rdata.mask_index(s, [])
    assert tr.size == tc.size == mr.size == mc.size == 0
    ptr, items = rdata.positives_csr(np.zeros(0, np.int64), np.zeros(0, np.int64), 3)
    assert ptr.tolist() == [0, 0, 0, 0] and items.size == 0
