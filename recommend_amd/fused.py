"""Routes `dropout(active(linear(x)))` compositions of the reference's blocks to
the fused node-wise layer kernel (rk_dense_*).  The modules must be the ones the
MMFCN blocks use (square nn.Linear of a built width, one-slope PReLU, nn.Dropout)
and the tensors must live on the GPU: anything else raises, there is no second
implementation behind these calls."""
import torch.nn as nn

from . import _ffi, ops


def prelu_slope(active):
    """The slope parameter of a one-parameter PReLU module (MyPReLU.alpha / nn.PReLU.weight)."""
    if active is None:
        return None
    p = getattr(active, "alpha", None)
    if p is None and isinstance(active, nn.PReLU):
        p = active.weight
    if p is None or not hasattr(p, "numel") or p.numel() != 1:
        return None
    return p


def check_fusable(linear, x, active=None, dropout=None):
    """Raise unless the composition is one the layer kernels compute."""
    if not x.is_cuda:
        raise _ffi.RkError(_ffi.RK_ERR_NODEV, "the node-wise layers run only on a HIP device; there is no CPU fallback")
    if not isinstance(linear, nn.Linear):
        raise _ffi.RkError(_ffi.RK_ERR_UNSUPPORTED, f"node-wise layer: {type(linear).__name__} is not an nn.Linear")
    if not ops.dense_supported(x, linear.weight):
        raise _ffi.RkError(_ffi.RK_ERR_UNSUPPORTED,
                           f"node-wise layer {tuple(x.shape)} x {tuple(linear.weight.shape)}^T: built for square "
                           f"layers of width {ops.DENSE_WIDTHS}")
    if active is not None and prelu_slope(active) is None:
        raise _ffi.RkError(_ffi.RK_ERR_UNSUPPORTED, "node-wise layer: the activation must be a one-slope PReLU")
    if dropout is not None and not isinstance(dropout, nn.Dropout):
        raise _ffi.RkError(_ffi.RK_ERR_UNSUPPORTED, "node-wise layer: the dropout must be an nn.Dropout")


def drop_p(dropout):
    return float(dropout.p) if (dropout is not None and dropout.training) else 0.0


def linear_act(linear, x, active=None, dropout=None, residual=None):
    """dropout(active(linear(x))) [+ residual]"""
    check_fusable(linear, x, active, dropout)
    return ops.dense(x, linear.weight, linear.bias, alpha=prelu_slope(active), p=drop_p(dropout),
                     residual=residual)


def linear_sum(linear1, x1, linear2, x2):
    """linear1(x1) + linear2(x2)"""
    check_fusable(linear1, x1)
    check_fusable(linear2, x2)
    return ops.dense(x1, linear1.weight, linear1.bias, x2=x2, weight2=linear2.weight, bias2=linear2.bias)
