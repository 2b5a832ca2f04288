"""Dump the forward CSR of a synthetic shape for tools/spmm_lab (binary, little endian):
int64 N, nnz, U ; int32 rowptr[N+1] ; int32 col[nnz] ; float val[nnz]."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recommend_amd import ops, synth  # noqa: E402

shape = sys.argv[1] if len(sys.argv) > 1 else "elec"
d = synth.make_named(shape)
tu, ti, tr = d.split(0)
src, dst, w = ops.build_coo_host(tu, ti, tr, d.n_users, d.n_items, True)
N = d.n_users + d.n_items
# symmetric normalisation in numpy (a timing input: the values need not be the library's bits)
root = np.sqrt(np.maximum(np.bincount(dst, weights=w, minlength=N), 1e-12))
wn = (w / (root[src] * root[dst])).astype(np.float32)
ptr, idx, eid = ops.csr_from_coo_host(dst.astype(np.int32), src.astype(np.int32), N)
val = wn[eid].astype(np.float32)
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_variants", f"{shape}_csr.bin")
with open(out, "wb") as f:
    np.array([N, idx.shape[0], d.n_users], np.int64).tofile(f)
    ptr.astype(np.int32).tofile(f)
    idx.astype(np.int32).tofile(f)
    val.tofile(f)
print(out, N, idx.shape[0])
