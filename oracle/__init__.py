"""CPU oracle for the recommend_amd hot path -- TEST INFRASTRUCTURE ONLY.

A plain restatement (numpy / torch-CPU / C) of the reference's algorithms for
graph construction and propagation, the BPR gather-dot loss, negative sampling
and full-ranking evaluation.  Only tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py may import it; nothing under recommend_amd/ does.

Pinning: the evaluation, loss and sampler restatements are checked against
outputs of the reference's own modules (evalue/metric.py, evalue/loss.py,
evalue/evaluator.py, data/graph_data_loader.py) captured by
tests/golden/make_golden.py into tests/golden/*.npz|json.  The propagation
restatement follows model/test/graph_constructor.py as text; its arithmetic
lives in DGL, which is not part of the reference tree and not installed
(version unpinned in the reference's README.md:42), and the reference has no
tests or golden vectors for it: PARITY UNPINNED for graph propagation, checked
against independent scipy / dense-matmul formulations instead.
"""
