"""Routes `dropout(active(linear(x)))` compositions of the reference's blocks to
the fused node-wise layer kernel (rk_dense_*) when the modules are the ones the
MMFCN blocks use (nn.Linear 64 -> 64, one-slope PReLU, nn.Dropout) and the
tensors live on the GPU; any other combination of callables runs as given."""
import torch.nn as nn

from . import ops


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


def can_fuse(linear, x, active=None, dropout=None):
    if not isinstance(linear, nn.Linear) or not ops.dense_supported(x, linear.weight):
        return False
    if active is not None and prelu_slope(active) is None:
        return False
    if dropout is not None and not isinstance(dropout, nn.Dropout):
        return False
    return True


def drop_p(dropout):
    return float(dropout.p) if (dropout is not None and dropout.training) else 0.0


def linear_act(linear, x, active=None, dropout=None, residual=None):
    """dropout(active(linear(x))) [+ residual]"""
    if can_fuse(linear, x, active, dropout):
        return ops.dense(x, linear.weight, linear.bias, alpha=prelu_slope(active), p=drop_p(dropout),
                         residual=residual)
    y = linear(x)
    if active is not None:
        y = active(y)
    if dropout is not None:
        y = dropout(y)
    return y if residual is None else residual + y


def linear_sum(linear1, x1, linear2, x2):
    """linear1(x1) + linear2(x2)"""
    if can_fuse(linear1, x1) and can_fuse(linear2, x2):
        return ops.dense(x1, linear1.weight, linear1.bias, x2=x2, weight2=linear2.weight, bias2=linear2.bias)
    return linear1(x1) + linear2(x2)
