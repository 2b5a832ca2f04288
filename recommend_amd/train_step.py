"""One MMFCN training step as a captured HIP graph.

The reference's step (train/graph_trainer.py:140-157: zero_grad, forward, loss, backward,
clip_grad_norm_, optimizer.step) launches a few hundred small kernels; issued one by one from
Python the step is bound by the host, not by the GPU.  `GraphedTrainStep` records the whole step
once with torch.cuda.graphs (hipGraph underneath) and replays it per batch: the batch is copied
into fixed device buffers, the dropout masks advance through a step counter in device memory,
and the only host work per step is one graph launch.  `GraphedStep` is the recording and replay
machinery for any step body (recommend_amd.dist_train captures the row-sharded step, collectives
included, with it).

Batches whose size differs from the captured one (the last batch of an epoch) run eagerly.
"""
from typing import Callable, Dict, Optional

import torch

from . import ops


class GraphedStep:
    """body(batch) -> loss tensor: one whole training step (zero_grad excluded) on the current
    stream.  Warm-up steps and the capture run on ONE side stream, so autograd's gradient
    accumulators see the stream they were created on."""

    def __init__(self, body: Callable[[Dict], torch.Tensor], zero_grad: Callable[[], None],
                 example_batch: Dict[str, torch.Tensor], constants: Optional[Dict[str, torch.Tensor]] = None,
                 warmup_steps: int = 3, capture: bool = True, capture_error_mode: str = "global"):
        if warmup_steps < 1:
            # optimizers create their state on the first step; created inside the capture it would
            # be re-zeroed by every replay
            raise ValueError("at least one eager warm-up step is needed before the capture")
        self._body_fn, self._zero_grad = body, zero_grad
        dev = next(iter(example_batch.values())).device
        self.constants = dict(constants or {})
        self.static = {k: v.clone() for k, v in example_batch.items() if k not in self.constants}
        self.shapes = {k: tuple(v.shape) for k, v in self.static.items()}
        self.step_counter = torch.zeros(1, dtype=torch.int64, device=dev)
        ops.set_dropout_step_tensor(self.step_counter)
        self.graph = None
        # warm-up on a side stream, as graph capture requires (allocator pools, lazy state)
        s = self._stream = torch.cuda.Stream(device=dev)
        s.wait_stream(torch.cuda.current_stream(dev))
        self.warmup_losses = []     # the warm-up steps are real training steps on the example batch
        with torch.cuda.stream(s):
            for _ in range(warmup_steps):
                self.warmup_losses.append(self._eager(self.static).clone())
        torch.cuda.current_stream(dev).wait_stream(s)
        if not capture:
            return
        self.graph = torch.cuda.CUDAGraph()
        self._zero_grad()
        # (capture_error_mode "thread_local": other threads -- the collective backend's watchdog -- may
        # query events while this thread records; the default forbids that for every thread)
        with torch.cuda.graph(self.graph, stream=s, capture_error_mode=capture_error_mode):
            self.static_loss = self._body({**self.static, **self.constants})

    def _body(self, batch):
        self.step_counter.add_(1)
        return self._body_fn(batch)

    def _eager(self, batch):
        self._zero_grad()
        return self._body({**batch, **self.constants})

    def __call__(self, batch: Dict[str, torch.Tensor]) -> torch.Tensor:
        """Runs one step on `batch`; returns the loss (a device scalar that the next call
        overwrites)."""
        if self.graph is None or any(tuple(batch[k].shape) != self.shapes[k] for k in self.shapes):
            # on the stream the recorded step's autograd nodes were created on (gradient accumulators
            # outlive an iteration and remember their stream)
            cur = torch.cuda.current_stream(self._stream.device)
            self._stream.wait_stream(cur)
            with torch.cuda.stream(self._stream):
                loss = self._eager(batch)
            cur.wait_stream(self._stream)
            return loss
        for k, dst in self.static.items():
            dst.copy_(batch[k], non_blocking=True)
        self.graph.replay()
        return self.static_loss


class GraphedTrainStep(GraphedStep):
    def __init__(self, model: torch.nn.Module, optimizer: torch.optim.Optimizer,
                 loss_fn: Callable[[Dict, Dict], torch.Tensor], example_batch: Dict[str, torch.Tensor],
                 constants: Optional[Dict[str, torch.Tensor]] = None,
                 max_grad_norm: Optional[float] = 1.0, warmup_steps: int = 3):
        """example_batch: a batch dict as the loader yields it (device tensors); its tensors fix
        the captured shapes, and the warm-up steps train on it (they are real steps).  constants: entries every batch carries unchanged (e.g. the
        user->positives CSR for in-kernel negative lookup); they are not copied per step.
        The optimizer must be capturable (recommend_amd.optim.FusedClipAdam, or e.g.
        torch.optim.Adam(capturable=True))."""
        self.model, self.opt, self.loss_fn = model, optimizer, loss_fn
        self.max_grad_norm = max_grad_norm
        # an optimizer that clips inside its own update (recommend_amd.optim.FusedClipAdam)
        self._opt_clips = hasattr(optimizer, "max_grad_norm")
        if self._opt_clips:
            optimizer.max_grad_norm = max_grad_norm
        self.params = [p for p in model.parameters() if p.requires_grad]
        super().__init__(self._train_body, lambda: self.opt.zero_grad(set_to_none=True), example_batch, constants,
                         warmup_steps)

    def _train_body(self, batch):
        loss = self.loss_fn(self.model(batch), batch)
        loss.backward()
        if self.max_grad_norm is not None and not self._opt_clips:
            self._clip()
        self.opt.step()
        return loss.detach()

    def _clip(self):
        """clip_grad_norm_(params, max_norm) (train/graph_trainer.py:154).  With a fused optimizer the
        scaling rides on the optimizer kernel's gradient-scale input (grad /= scale inside the
        update) instead of one multiply launch per parameter."""
        if not self.opt.defaults.get("fused"):
            torch.nn.utils.clip_grad_norm_(self.params, self.max_grad_norm)
            return
        grads = [p.grad for p in self.params if p.grad is not None]
        total = torch.linalg.vector_norm(torch.stack(torch._foreach_norm(grads)))
        # clip_coef = min(1, max_norm / (total + 1e-6)); the optimizer divides by its inverse
        self.opt.grad_scale = torch.clamp((total + 1e-6) / self.max_grad_norm, min=1.0).reshape(())
        if getattr(self.opt, "found_inf", None) is None:
            self.opt.found_inf = torch.zeros((), dtype=torch.float32, device=total.device)
