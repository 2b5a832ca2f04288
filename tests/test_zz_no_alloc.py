"""rk_bpr_fwd_noalloc_f32 (built-in scratch) against rk_bpr_fwd_f32 (caller workspace): same
loss bits, same saved differences, same resolved negatives -- on two calls in a row, the second
with a smaller batch, through the C ABI."""
import numpy as np
import pytest
import torch

from recommend_amd import _ffi
from recommend_amd._ffi import c_int32, c_int64, c_size_t, ptr, stream_ptr

pytestmark = pytest.mark.gpu


def _inputs(B, R, n_users, n_items, D, seed, sampled):
    g = torch.Generator().manual_seed(seed)
    dev = torch.device("cuda", 0)
    U = torch.randn(n_users, D, generator=g).to(dev)
    I = torch.randn(n_items, D, generator=g).to(dev)
    users = torch.randint(0, n_users, (B,), generator=g).to(dev)
    pos = torch.randint(0, n_items, (B,), generator=g).to(dev)
    rng = np.random.default_rng(seed)
    lens = rng.integers(1, 12, n_users)
    pp = np.zeros(n_users + 1, np.int32)
    pp[1:] = np.cumsum(lens)
    items = np.concatenate([np.sort(rng.choice(n_items, l, replace=False)) for l in lens]).astype(np.int32)
    pos_ptr, pos_items = torch.from_numpy(pp).to(dev), torch.from_numpy(items).to(dev)
    if sampled:
        n_cand = (n_items - lens)[users.cpu().numpy()]
        rank = torch.from_numpy((rng.random((B, R)) * n_cand[:, None]).astype(np.int32)).to(dev)
        neg = torch.zeros((B, R), dtype=torch.int64, device=dev)
    else:
        rank = None
        neg = torch.randint(0, n_items, (B, R), generator=g).to(dev)
    mask = (torch.rand(B, R, generator=g) > 0.2).float().to(dev)
    return U, I, users, pos, neg, rank, pos_ptr, pos_items, mask


def _run(entry, U, I, users, pos, neg, rank, pos_ptr, pos_items, mask, n_items):
    B, R = neg.shape
    D = U.shape[1]
    dev = U.device
    neg = neg.clone()
    x = torch.full((B, R), float("nan"), device=dev)
    out = torch.full((2,), float("nan"), device=dev)
    head = [ptr(U), ptr(I), c_int32(D), ptr(users), ptr(pos), ptr(neg), c_int64(B), c_int32(R), ptr(mask),
            ptr(rank), ptr(pos_ptr if rank is not None else None), ptr(pos_items if rank is not None else None),
            c_int32(n_items), ptr(x), ptr(out)]
    if entry == "rk_bpr_fwd_f32":
        nbytes = _ffi.load().rk_bpr_ws_bytes(B, R)
        ws = torch.empty(nbytes // 4 + 4, dtype=torch.float32, device=dev)
        _ffi.call(entry, *head, ptr(ws), c_size_t(ws.numel() * 4), stream_ptr(dev))
    else:
        _ffi.call(entry, *head, stream_ptr(dev))
    torch.cuda.synchronize()
    return out.cpu(), x.cpu(), neg.cpu()


def test_noalloc_bpr_forward_matches_workspace_form_twice():
    n_users, n_items, D = 3000, 900, 64
    # first a large batch with in-kernel negative lookup, then a smaller one with given negatives
    # and a mask: the second call must not see anything the first left in the built-in array
    for B, R, sampled, seed in ((20000, 2, True, 1), (777, 3, False, 2)):
        args = _inputs(B, R, n_users, n_items, D, seed, sampled)
        ref = _run("rk_bpr_fwd_f32", *args, n_items)
        got = _run("rk_bpr_fwd_noalloc_f32", *args, n_items)
        assert torch.isfinite(ref[0]).all()
        for a, b in zip(ref, got):
            assert torch.equal(a, b)


def test_noalloc_bpr_forward_recorded_in_a_hip_graph():
    """The callers this entry point is for: after one eager call on the device (which looks the
    built-in array's address up), a call recorded in a HIP graph and replayed on new inputs gives
    the workspace form's bits."""
    n_users, n_items, D, B, R = 2000, 700, 64, 1500, 2
    dev = torch.device("cuda", 0)
    U, I, users, pos, neg, rank, pos_ptr, pos_items, mask = _inputs(B, R, n_users, n_items, D, 5, False)
    _run("rk_bpr_fwd_noalloc_f32", U, I, users, pos, neg, rank, pos_ptr, pos_items, mask, n_items)   # eager warm-up
    x = torch.zeros((B, R), device=dev)
    out = torch.zeros(2, device=dev)
    s = torch.cuda.Stream(device=dev)
    s.wait_stream(torch.cuda.current_stream(dev))
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=s):
        _ffi.call("rk_bpr_fwd_noalloc_f32", ptr(U), ptr(I), c_int32(D), ptr(users), ptr(pos), ptr(neg), c_int64(B),
                  c_int32(R), ptr(mask), None, None, None, c_int32(n_items), ptr(x), ptr(out), stream_ptr(dev))
    for seed in (6, 7):       # new inputs in the recorded buffers, one replay each
        U2, I2, users2, pos2, neg2, _r, _pp, _pi, mask2 = _inputs(B, R, n_users, n_items, D, seed, False)
        for dst, src in ((U, U2), (I, I2), (users, users2), (pos, pos2), (neg, neg2), (mask, mask2)):
            dst.copy_(src)
        graph.replay()
        torch.cuda.synchronize()
        ref = _run("rk_bpr_fwd_f32", U, I, users, pos, neg, None, pos_ptr, pos_items, mask, n_items)
        assert torch.equal(out.cpu(), ref[0]) and torch.equal(x.cpu(), ref[1])


def test_noalloc_bpr_forward_rejects_oversize_batch():
    lim = int(_ffi.load().rk_bpr_fwd_noalloc_max_batch())
    dev = torch.device("cuda", 0)
    z = torch.zeros(4, device=dev)
    zi = torch.zeros(4, dtype=torch.int64, device=dev)
    with pytest.raises(_ffi.RkError) as e:
        _ffi.call("rk_bpr_fwd_noalloc_f32", ptr(z), ptr(z), c_int32(4), ptr(zi), ptr(zi), ptr(zi),
                  c_int64(lim + 1), c_int32(1), None, None, None, None, c_int32(1), ptr(z), ptr(z), stream_ptr(dev))
    assert e.value.code == _ffi.RK_ERR_UNSUPPORTED
