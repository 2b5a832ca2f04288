"""Probe: fused score + mask + top-K at a named shape with random embeddings; the filtered entry
point against the exact one (time, equality, users left to the exact pass)."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ctypes import byref, c_int32, c_size_t
import numpy as np, torch
from recommend_amd import ops, synth, _ffi
from recommend_amd.evaluator import pairs_to_csr

ap = argparse.ArgumentParser()
ap.add_argument("--shape", default="elec"); ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--k", type=int, default=20); ap.add_argument("--dim", type=int, default=64)
ap.add_argument("--scale", type=float, default=0.1); ap.add_argument("--nomask", action="store_true"); ap.add_argument("--users", type=int, default=0)
ap.add_argument("--items", type=int, default=0, help="keep the first N items only (an item shard)")
a = ap.parse_args()
d = synth.make_named(a.shape)
tu, ti, _ = d.split(0)
if a.items:
    keep = ti < a.items; tu, ti = tu[keep], ti[keep]; d.n_items = a.items
ptr, cols = pairs_to_csr(tu, ti, d.n_users)
if a.users:
    d.n_users = a.users; ptr = ptr[:a.users + 1]; cols = cols[:ptr[-1]]
dev = "cuda"
g = torch.Generator().manual_seed(0)
ue = (torch.randn(d.n_users, a.dim, generator=g) * a.scale).to(dev)
ie = (torch.randn(d.n_items, a.dim, generator=g) * a.scale).to(dev)
mp, mc = torch.from_numpy(ptr).to(dev), torch.from_numpy(cols).to(dev)
if a.nomask: mp = mc = None
lib = _ffi.load()
ws_bytes = lib.rk_score_topk_ws_bytes(d.n_users, d.n_items, a.dim, a.k)
ws = torch.zeros(ws_bytes // 4 + 4, dtype=torch.float32, device=dev)
out = {}
for name in ("rk_score_topk_exact_f32", "rk_score_topk_filtered_f32"):
    os_ = torch.empty((d.n_users, a.k), dtype=torch.float32, device=dev)
    oi = torch.empty((d.n_users, a.k), dtype=torch.int32, device=dev)
    def run():
        _ffi.call(name, ops.ptr(ue), c_int32(d.n_users), ops.ptr(ie), c_int32(d.n_items), c_int32(a.dim),
                  ops.ptr(mp), ops.ptr(mc), c_int32(a.k), c_int32(0), ops.ptr(os_), ops.ptr(oi), ops.ptr(ws),
                  c_size_t(ws.numel() * 4), ops.stream_ptr(ue.device))
    run(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(a.reps): run()
    e.record(); e.synchronize()
    ms = s.elapsed_time(e) / a.reps
    out[name] = (os_, oi)
    print(f"{name} {a.shape} K={a.k} D={a.dim}: {ms:.3f} ms  {2.0*d.n_users*d.n_items*a.dim/ms/1e9:.1f} TFLOP/s equiv")
    if name.endswith("filtered_f32"):
        n_open = c_int32(0)
        _ffi.call("rk_score_topk_open_users", c_int32(d.n_users), c_int32(d.n_items), c_int32(a.dim), c_int32(a.k), ops.ptr(ws),
                  c_size_t(ws.numel() * 4), ops.stream_ptr(ue.device), byref(n_open))
        print(f"users left to the exact pass: {n_open.value} of {d.n_users}")
(es, ei), (fs, fi) = out["rk_score_topk_exact_f32"], out["rk_score_topk_filtered_f32"]
print("ids equal:", bool(torch.equal(ei, fi)), " scores bit-equal:", bool(torch.equal(es.view(torch.int32), fs.view(torch.int32))))
if not torch.equal(ei, fi):
    bad = (ei != fi).any(1).nonzero().flatten()
    print("differing users:", bad.numel(), bad[:5].tolist())
    u = int(bad[0]); print(ei[u].tolist(), fi[u].tolist()); print(es[u].tolist(), fs[u].tolist())
