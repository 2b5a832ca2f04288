"""Adam with global-norm gradient clipping on the library's two-launch kernel pair
(rk_adam_clip_step_f32): the optimizer half of train/graph_trainer.py:151-156,
`clip_grad_norm_(model.parameters(), max_norm); optimizer.step()`, for fp32 parameters.

The update is torch.optim.Adam's (same constants, same order of operations); moments and the step
count live in device memory, the learning rate is a device scalar refreshed from
`param_groups[0]['lr']`, so a step captured in a HIP graph keeps working when a scheduler changes
the rate between replays.  Parameters whose `.grad` is None take no part, as in torch.

Checkpoints (train/graph_trainer.py:279,315 save and restore `optimizer.state_dict()`): the state
dict carries `step`, `exp_avg`, `exp_avg_sq` per parameter exactly as torch.optim.Adam's does, so a
checkpoint written by either optimizer loads into the other and a resumed run continues the bias
correction where it stopped.

Several GPUs: parameters listed in `sharded_params` are row shards (every rank holds other rows);
their squared gradient norms are summed over the ranks of `group` by one all-reduce of one float
between the two launches (rk_adam_sqnorm_f32 / rk_adam_update_f32).  All other parameters are
taken to be replicated with gradients already summed over the ranks."""
import torch

from . import _ffi
from ._ffi import c_float, c_int32, ptr, stream_ptr

_MAX_TABLES = 4     # descriptor tables kept (one per set of gradient buffers: eager steps, a captured graph)


class FusedClipAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, max_grad_norm=None,
                 sharded_params=None, group=None, all_reduce=None):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        if len(self.param_groups) != 1:
            raise ValueError("FusedClipAdam takes one parameter group")
        self.max_grad_norm = max_grad_norm
        self._sharded = {id(p) for p in (sharded_params or [])}
        self._group = group
        self._all_reduce = all_reduce    # all_reduce(tensor, group): sums in place (default: torch.distributed)
        self._tables = {}        # pointer set -> (descriptor array, workspace, replicated count)
        self._dev = None
        self._lr_host = None
        self._restored_step = None
        self.grad_norm = None    # device scalar: total gradient norm of the last step, before clipping

    def __setstate__(self, state):
        super().__setstate__(state)
        self._tables = {}

    def _state_for(self, p):
        st = self.state[p]
        if "exp_avg" not in st:
            st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
            st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
        return st

    def _init_device(self, dev):
        self._dev = dev
        self.step_count = torch.full((1,), int(self._restored_step or 0), dtype=torch.int64, device=dev)
        self.lr_dev = torch.zeros(1, dtype=torch.float32, device=dev)
        self.grad_norm = torch.zeros(1, dtype=torch.float32, device=dev)
        self.shard_sq = torch.zeros(1, dtype=torch.float32, device=dev)

    def _prepare(self, params):
        """Descriptor table + workspace for this set of parameter / gradient / moment buffers
        (replicated tensors first, row shards behind them)."""
        states = [self._state_for(p) for p in params]
        key = tuple((p.data_ptr(), p.grad.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), p.numel())
                    for p, st in zip(params, states))
        tab = self._tables.get(key)
        if tab is None:
            order = sorted(range(len(params)), key=lambda i: id(params[i]) in self._sharded)   # stable
            arr = (_ffi.AdamTensor * len(params))()
            for j, i in enumerate(order):
                p, st = params[i], states[i]
                arr[j] = _ffi.AdamTensor(p.data_ptr(), p.grad.data_ptr(), st["exp_avg"].data_ptr(),
                                         st["exp_avg_sq"].data_ptr(), p.numel())
            n_rep = sum(1 for p in params if id(p) not in self._sharded)
            nbytes = _ffi.load().rk_adam_ws_bytes(arr, len(params))
            tab = (arr, torch.empty(nbytes // 4 + 4, dtype=torch.float32, device=self._dev), n_rep)
            while len(self._tables) >= _MAX_TABLES:         # oldest first: dicts keep insertion order
                self._tables.pop(next(iter(self._tables)))
            self._tables[key] = tab
        return tab

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        group = self.param_groups[0]
        params = [p for p in group["params"] if p.grad is not None]
        if not params:
            return loss
        for p in params:
            if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and p.grad.is_contiguous()
                    and p.grad.dtype == torch.float32):
                raise _ffi.RkError(_ffi.RK_ERR_UNSUPPORTED, "FusedClipAdam: contiguous fp32 parameters and gradients "
                                   "on a HIP device only")
        if self._dev is None:
            self._init_device(params[0].device)
        arr, ws, n_rep = self._prepare(params)     # (moments are created outside any graph capture: first eager step)
        if self._lr_host != group["lr"]:
            if torch.cuda.is_current_stream_capturing():
                if self._lr_host is None:
                    raise RuntimeError("run one optimizer step before capturing it in a graph")
            else:
                self.lr_dev.fill_(float(group["lr"]))
                self._lr_host = group["lr"]
        b1, b2 = group["betas"]
        n = len(params)
        max_norm = c_float(self.max_grad_norm if self.max_grad_norm else 0.0)
        if n_rep == n:
            _ffi.call("rk_adam_clip_step_f32", arr, c_int32(n), ptr(self.step_count), ptr(self.lr_dev),
                      c_float(b1), c_float(b2), c_float(group["eps"]), c_float(group["weight_decay"]),
                      max_norm, ptr(self.grad_norm), ptr(ws), _ffi.c_size_t(ws.numel() * 4), stream_ptr(self._dev))
            return loss
        _ffi.call("rk_adam_sqnorm_f32", arr, c_int32(n), c_int32(n_rep), ptr(self.step_count), ptr(self.shard_sq),
                  ptr(ws), _ffi.c_size_t(ws.numel() * 4), stream_ptr(self._dev))
        if self._all_reduce is not None:
            self._all_reduce(self.shard_sq, self._group)
        else:
            import torch.distributed as dist
            if dist.is_initialized() and dist.get_world_size(self._group) > 1:
                dist.all_reduce(self.shard_sq, group=self._group)
        _ffi.call("rk_adam_update_f32", arr, c_int32(n), c_int32(n_rep), ptr(self.step_count), ptr(self.lr_dev),
                  c_float(b1), c_float(b2), c_float(group["eps"]), c_float(group["weight_decay"]), max_norm,
                  ptr(self.shard_sq), ptr(self.grad_norm), ptr(ws), _ffi.c_size_t(ws.numel() * 4),
                  stream_ptr(self._dev))
        return loss

    def state_dict(self):
        """torch.optim.Adam's layout: every parameter's state carries `step` beside its moments
        (reading the device counter synchronises)."""
        if self._dev is not None:
            t = float(self.step_count.item())
            for st in self.state.values():
                if "exp_avg" in st:
                    st["step"] = torch.tensor(t)
        return super().state_dict()

    def load_state_dict(self, state_dict):
        """Moments and step count of a checkpoint (this optimizer's or torch.optim.Adam's).  Moment
        buffers that exist already keep their memory and take the checkpoint's VALUES, so a training
        step recorded in a HIP graph before the load (its kernels hold the buffers' addresses) goes on
        updating the moments the optimizer reports; the learning rate of the checkpoint reaches the
        device scalar at once for the same reason.  Call it outside graph capture."""
        old = {p: (st["exp_avg"], st["exp_avg_sq"]) for p, st in self.state.items() if "exp_avg" in st}
        super().load_state_dict(state_dict)
        for q, (m, v) in old.items():
            st = self.state.get(q)
            if st is None or "exp_avg" not in st:
                continue
            if st["exp_avg"].shape == m.shape and st["exp_avg"].device == m.device:
                m.copy_(st["exp_avg"])
                v.copy_(st["exp_avg_sq"])
                st["exp_avg"], st["exp_avg_sq"] = m, v
        self._tables = {}        # descriptor tables are rebuilt from the state at the next step
        steps = [float(st["step"]) for st in self.state.values() if "step" in st]
        self._restored_step = int(max(steps)) if steps else 0
        if self._dev is not None:
            self.step_count.fill_(self._restored_step)
        self._lr_host = None
        self.sync_lr()

    def sync_lr(self):
        """Push a learning rate changed by a scheduler to the device scalar (call outside graph capture)."""
        lr = self.param_groups[0]["lr"]
        if self._dev is not None and self._lr_host != lr:
            self.lr_dev.fill_(float(lr))
            self._lr_host = lr
