"""Oracle: negative sampling, restating GraphDataLoader._generate_negative_samples
(data/graph_data_loader.py:355-397) with the stdlib `random` module it uses.
Pinned by tests/golden/loader_batches.npz captured from the reference loader."""
import random


def generate_negative_samples(user_ids, user_item_interaction, all_items, neg_ratio):
    if neg_ratio <= 0:
        raise ValueError("Negative ratio must be greater than zero")
    neg_items = []
    for user_id in user_ids:
        user_id = int(user_id)
        pos_items = user_item_interaction.get(user_id, set())
        neg_candidates = list(all_items - pos_items)
        if len(neg_candidates) == 0:
            neg_candidates = [0]
        if len(neg_candidates) >= neg_ratio:
            sampled_neg = random.sample(neg_candidates, k=neg_ratio)
        else:
            sampled_neg = random.choices(neg_candidates, k=neg_ratio)
        neg_items.append(sampled_neg)
    return neg_items
