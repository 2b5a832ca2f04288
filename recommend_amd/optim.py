"""Adam with global-norm gradient clipping on the library's two-launch kernel pair
(rk_adam_clip_step_f32): the optimizer half of train/graph_trainer.py:151-156,
`clip_grad_norm_(model.parameters(), max_norm); optimizer.step()`, for fp32 parameters on one GPU.

The update is torch.optim.Adam's (same constants, same order of operations); moments and the step
count live in device memory, the learning rate is a device scalar refreshed from
`param_groups[0]['lr']`, so a step captured in a HIP graph keeps working when a scheduler changes
the rate between replays.  Parameters whose `.grad` is None take no part, as in torch."""
import torch

from . import _ffi
from ._ffi import c_float, c_int32, ptr, stream_ptr


class FusedClipAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, max_grad_norm=None):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        if len(self.param_groups) != 1:
            raise ValueError("FusedClipAdam takes one parameter group")
        self.max_grad_norm = max_grad_norm
        self._tables = {}        # pointer set -> (descriptor array, workspace)
        self._dev = None
        self._lr_host = None
        self.grad_norm = None    # device scalar: total gradient norm of the last step, before clipping

    def _state_for(self, p):
        st = self.state[p]
        if not st:
            st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
            st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
        return st

    def _prepare(self, params):
        dev = params[0].device
        if self._dev is None:
            self._dev = dev
            self.step_count = torch.zeros(1, dtype=torch.int64, device=dev)
            self.lr_dev = torch.zeros(1, dtype=torch.float32, device=dev)
            self.grad_norm = torch.zeros(1, dtype=torch.float32, device=dev)
        key = tuple((p.data_ptr(), p.grad.data_ptr(), p.numel()) for p in params)
        tab = self._tables.get(key)
        if tab is None:
            arr = (_ffi.AdamTensor * len(params))()
            for i, p in enumerate(params):
                st = self._state_for(p)
                arr[i] = _ffi.AdamTensor(p.data_ptr(), p.grad.data_ptr(), st["exp_avg"].data_ptr(),
                                         st["exp_avg_sq"].data_ptr(), p.numel())
            nbytes = _ffi.load().rk_adam_ws_bytes(arr, len(params))
            tab = (arr, torch.empty(nbytes // 4 + 4, dtype=torch.float32, device=dev))
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
        for p in params:          # moments are created outside any graph capture (first eager step)
            self._state_for(p)
        arr, ws = self._prepare(params)
        if self._lr_host != group["lr"]:
            if torch.cuda.is_current_stream_capturing():
                if self._lr_host is None:
                    raise RuntimeError("run one optimizer step before capturing it in a graph")
            else:
                self.lr_dev.fill_(float(group["lr"]))
                self._lr_host = group["lr"]
        b1, b2 = group["betas"]
        _ffi.call("rk_adam_clip_step_f32", arr, c_int32(len(params)), ptr(self.step_count), ptr(self.lr_dev),
                  c_float(b1), c_float(b2), c_float(group["eps"]), c_float(group["weight_decay"]),
                  c_float(self.max_grad_norm if self.max_grad_norm else 0.0), ptr(self.grad_norm), ptr(ws),
                  _ffi.c_size_t(ws.numel() * 4), stream_ptr(self._dev))
        return loss

    def sync_lr(self):
        """Push a learning rate changed by a scheduler to the device scalar (call outside graph capture)."""
        lr = self.param_groups[0]["lr"]
        if self._dev is not None and self._lr_host != lr:
            self.lr_dev.fill_(float(lr))
            self._lr_host = lr
