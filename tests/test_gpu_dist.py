"""Two ranks (gloo rendezvous, both on cuda:0) run the row-sharded propagation on the HIP SpMM
kernels; forward and gradient must equal the float64 result of the unsharded recurrence."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _case():
    rng = np.random.default_rng(9)
    N, E, D = 3001, 40000, 64
    src = rng.integers(0, N, E)
    dst = rng.integers(0, N, E)
    dst[:5000] = 7                       # one row long enough for the chunked-row path
    w = (rng.random(E) / 8).astype(np.float32)
    x = rng.standard_normal((N, D)).astype(np.float32)
    g = rng.standard_normal((N, D)).astype(np.float32)
    return N, src, dst, w, x, g


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from recommend_amd.dist import RowShardedGraph
        N, src, dst, w, x, g = _case()
        graph = RowShardedGraph(src, dst, w, N, device="cuda:0")
        xl = torch.from_numpy(x[graph.lo:graph.hi]).to("cuda:0").requires_grad_(True)
        out = graph.aggrate(xl, 2, 0.1)
        out.backward(torch.from_numpy(g[graph.lo:graph.hi]).to("cuda:0"))
        torch.cuda.synchronize()
        q.put((rank, graph.lo, graph.hi, out.detach().cpu().numpy(), xl.grad.cpu().numpy()))
    finally:
        dist.destroy_process_group()


def test_row_sharded_propagation_on_the_hip_kernels():
    import scipy.sparse as sp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    N, src, dst, w, x, g = _case()
    A = sp.csr_matrix((w.astype(np.float64), (dst, src)), shape=(N, N))
    h0 = x.astype(np.float64)
    h = h0
    for _ in range(2):
        h = 0.1 * h0 + 0.9 * (A @ h)
    gh = g.astype(np.float64)
    gx = np.zeros_like(gh)
    for _ in range(2):
        gx += 0.1 * gh
        gh = 0.9 * (A.T @ gh)
    gx += gh
    assert sorted((lo, hi) for _r, lo, hi, _o, _g in results) == [(0, 1501), (1501, 3001)]
    for _rank, lo, hi, out, grad in results:
        assert np.abs(out - h[lo:hi]).max() <= 1e-5 * np.abs(h).max()
        assert np.abs(grad - gx[lo:hi]).max() <= 1e-5 * np.abs(gx).max()
