import sys; sys.path.insert(0, "/root/repo")
import numpy as np, torch
from recommend_amd import ops
from oracle import eval_oracle
rng = np.random.default_rng(1)
U, I, D, K = 70, 40, 64, 8
ue = rng.standard_normal((U, D)).astype(np.float32)
ie = rng.standard_normal((I, D)).astype(np.float32)
s, ids = ops.score_topk(torch.from_numpy(ue).cuda(), torch.from_numpy(ie).cuda(), K)
os_, oi = eval_oracle.score_topk(ue, ie, K)
s = s.cpu().numpy(); ids = ids.cpu().numpy()
for u in [0, 1, 31, 32, 33, 63, 64, 69]:
    print(u, ids[u], oi[u], np.round(s[u][:4], 3), np.round(os_[u][:4], 3))
