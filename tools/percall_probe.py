"""Per-call times of the top-K entry points at a named shape, eight calls each in the order given:
  python tools/percall_probe.py baby rk_score_topk_filtered_f32 rk_score_topk_exact_f32 rk_score_topk_filtered_f32
(what shows a filter whose thresholds went weak: right results, calls several times slower than the exact kernel's)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ctypes import c_int32, c_size_t
import numpy as np, torch
from recommend_amd import ops, synth, _ffi
from recommend_amd.evaluator import pairs_to_csr
shape = sys.argv[1]
d = synth.make_named(shape)
tu, ti, _ = d.split(0)
ptr, cols = pairs_to_csr(tu, ti, d.n_users)
dev = "cuda"
g = torch.Generator().manual_seed(0)
ue = (torch.randn(d.n_users, 64, generator=g) * 0.1).to(dev)
ie = (torch.randn(d.n_items, 64, generator=g) * 0.1).to(dev)
mp, mc = torch.from_numpy(ptr).to(dev), torch.from_numpy(cols).to(dev)
lib = _ffi.load()
ws_bytes = lib.rk_score_topk_ws_bytes(d.n_users, d.n_items, 64, 20)
ws = torch.zeros(ws_bytes // 4 + 4, dtype=torch.float32, device=dev)
os_ = torch.empty((d.n_users, 20), dtype=torch.float32, device=dev)
oi = torch.empty((d.n_users, 20), dtype=torch.int32, device=dev)
def run(name):
    _ffi.call(name, ops.ptr(ue), c_int32(d.n_users), ops.ptr(ie), c_int32(d.n_items), c_int32(64),
              ops.ptr(mp), ops.ptr(mc), c_int32(20), c_int32(0), ops.ptr(os_), ops.ptr(oi), ops.ptr(ws),
              c_size_t(ws.numel() * 4), ops.stream_ptr(ue.device))
for name in sys.argv[2:]:
    ts = []
    for _ in range(8):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); run(name); e.record(); e.synchronize()
        ts.append(round(s.elapsed_time(e), 3))
    print(name, ts)
