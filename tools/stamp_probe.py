"""Phase cycles of the filter kernel from the diagnostic (RK_STAMP) build of the library."""
import ctypes, os, sys
os.environ["RECOMMEND_AMD_LIB"] = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_variants", "lib_stamp.so")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from recommend_amd import ops, synth, _ffi
from recommend_amd.evaluator import pairs_to_csr
dim = int(sys.argv[1]) if len(sys.argv) > 1 else 64
d = synth.make_named("elec")
tu, ti, _ = d.split(0)
ptr, cols = pairs_to_csr(tu, ti, d.n_users)
g = torch.Generator().manual_seed(0)
ue = (torch.randn(d.n_users, dim, generator=g) * 0.1).cuda()
ie = (torch.randn(d.n_items, dim, generator=g) * 0.1).cuda()
mp, mc = torch.from_numpy(ptr).cuda(), torch.from_numpy(cols).cuda()
lib = _ffi.load()
lib.rk_debug_stamps.restype = ctypes.c_int
ops.score_topk(ue, ie, 20, mp, mc, mode="filter"); torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 10)()
lib.rk_debug_stamps(buf, 1)
ops.score_topk(ue, ie, 20, mp, mc, mode="filter"); torch.cuda.synchronize()
lib.rk_debug_stamps(buf, 0)
v = list(buf)
tiles, waves = v[8], v[9]
names = ["vmcnt(0) wait", "barrier", "DMA issue", "zero+frags+MFMA+swap", "range+mask", "selection"]
print(f"wavefronts {waves}, wave-tiles {tiles}")
for n, c in zip(names[:3], v[:3]):
    print(f"  {n:24s} {c / tiles:9.1f} cycles per wave-tile (per stage of 4: {4 * c / tiles:8.1f})")
for n, c in zip(names[3:], v[3:6]):
    print(f"  {n:24s} {c / tiles:9.1f} cycles per wave-tile")
print(f"  total                    {sum(v[:6]) / tiles:9.1f}")
print(f"  of the selection: parking candidates {v[7] / tiles:9.1f}, sorted insertion (flush) {v[6] / tiles:9.1f}, gates and the rest {(v[5] - v[6] - v[7]) / tiles:9.1f}")
