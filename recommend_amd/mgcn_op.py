"""One modality branch of the MMFCN network (MGCN.forward, model/test/out_Layer.py:32-36, with
AGG / agg_func, Agg_Layer.py:24-26 and graph_constructor.py:199-208, and the Combine layers,
Combin_Layer.py:26-30) as ONE autograd node over the layer and hop kernels.

Composed from per-layer autograd nodes the branch costs, per training step, one add launch for
every extra use of a tensor (the aggregation's Linear is applied k times, the projected features
X feed the aggregation and every Combine layer) -- the largest of them over N x D matrices.  Here
the backward pass is written out: the weight gradients of the k applications accumulate inside
the layer kernel's reduction (one launch for the reductions of all the branch's layer calls), and the gradient of X is one launch of the two-input layer kernel,
    dX = gY_0 W_ori,0 + gY_1 W_ori,1 + dX_aggregation
(the forward kernel with the transposed weights and the aggregation's gradient as its residual).

`hops` supplies the propagation: hops.forward(z) = A z, hops.backward(g) = A^T g on raw tensors
(single GPU: the SpMM kernels; row-sharded: all-gather + local SpMM).
"""
import torch

from . import ops


class _MGCN(torch.autograd.Function):
    @staticmethod
    def forward(ctx, X, emb, hops, k, p_agg, p_comb, p_out, L, *params):
        X, emb = ops._f32c(X), ops._f32c(emb)
        params = [None if t is None else ops._f32c(t) for t in params]
        Wa, ba, aa = params[0:3]
        comb = [params[3 + 7 * l: 3 + 7 * (l + 1)] for l in range(L)]
        Wm, bm, am = params[3 + 7 * L: 6 + 7 * L]
        seeds = [ops.next_dropout_seed() for _ in range(k + L + 1)]
        h, hs = X, []
        for t in range(k):                                   # graph_constructor.py:203-205
            hs.append(h)
            z = ops.dense_fwd_raw(h, Wa, ba, alpha=aa, p=p_agg, seed=seeds[t])
            h = hops.forward(z)
        a, ys, a_in = h, [], []
        for l in range(L):                                   # Combin_Layer.py:26-30
            W_ori, b_ori, W_f, b_f, W_o, b_o, al = comb[l]
            a_in.append(a)
            y = ops.dense_fwd_raw(X, W_ori, b_ori, x2=a, w2=W_f, b2=b_f)
            ys.append(y)
            a = ops.dense_fwd_raw(y, W_o, b_o, alpha=al, p=p_comb[l], seed=seeds[k + l])
        out = ops.dense_fwd_raw(emb, Wm, bm, alpha=am, p=p_out, seed=seeds[k + L], residual=a)   # out_Layer.py:36
        ctx.save_for_backward(X, emb, *hs[1:], *a_in, *ys, *params)
        ctx.meta = (hops, k, p_agg, tuple(p_comb), p_out, L, seeds)
        return out

    @staticmethod
    def backward(ctx, g):
        hops, k, p_agg, p_comb, p_out, L, seeds = ctx.meta
        saved = ctx.saved_tensors
        X, emb = saved[0], saved[1]
        hs = [X] + list(saved[2:2 + k - 1])
        a_in = list(saved[2 + k - 1: 2 + k - 1 + L])
        ys = list(saved[2 + k - 1 + L: 2 + k - 1 + 2 * L])
        params = list(saved[2 + k - 1 + 2 * L:])
        Wa, ba, aa = params[0:3]
        comb = [params[3 + 7 * l: 3 + 7 * (l + 1)] for l in range(L)]
        Wm, bm, am = params[3 + 7 * L: 6 + 7 * L]
        need_X, need_emb = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        g = ops._f32c(g)
        n, D = X.shape
        # the closing reductions of the 13 layer calls below run as one launch at the end
        rq = ops.DenseReduceQueue(n, D, g.device)
        ws = None if rq.usable else ops.dense_bwd_ws(n, D, D, g.device)

        # out = a_L + dropout(prelu(W_m emb + b_m)): the residual passes g through
        d_emb, dWm, dbm, dam = ops.dense_bwd_raw(emb, Wm, bm, am, p_out, seeds[k + L], g, need_emb, True, None, ws, rq)
        g_a = g
        comb_grads = [None] * L
        g_ys = [None] * L
        for l in reversed(range(L)):
            W_ori, b_ori, W_f, b_f, W_o, b_o, al = comb[l]
            g_y, dW_o, db_o, dal = ops.dense_bwd_raw(ys[l], W_o, b_o, al, p_comb[l], seeds[k + l], g_a, True, True, None, ws, rq)
            # y = X W_ori^T + a W_f^T + b_ori + b_f: both biases get the column sums of g_y
            _, dW_ori, db_lin, _ = ops.dense_bwd_raw(X, W_ori, None, None, 0.0, 0, g_y, False, True, None, ws, rq)
            g_a, dW_f, _, _ = ops.dense_bwd_raw(a_in[l], W_f, None, None, 0.0, 0, g_y, True, False, None, ws, rq)
            g_ys[l] = g_y
            comb_grads[l] = [dW_ori, db_lin if b_ori is not None else None, dW_f, db_lin if b_f is not None else None,
                             dW_o, db_o if b_o is not None else None, dal]
        # the aggregation: one gradient buffer for the k applications of the same layer
        g_h = g_a
        agg = None
        for t in reversed(range(k)):
            g_z = hops.backward(g_h)
            g_h, dWa, dba, daa = ops.dense_bwd_raw(hs[t], Wa, ba, aa, p_agg, seeds[t], g_z, t > 0 or need_X, True, agg, ws, rq)
            agg = (dWa, dba, daa)
        dX = None
        if need_X:
            dX = g_h
            l = 0
            while l < L:                  # two Combine layers per launch, the running sum as the residual
                if l + 1 < L:
                    dX = ops.dense_fwd_raw(g_ys[l], comb[l][0].t().contiguous(), None, x2=g_ys[l + 1],
                                           w2=comb[l + 1][0].t().contiguous(), residual=dX)
                    l += 2
                else:
                    dX = ops.dense_fwd_raw(g_ys[l], comb[l][0].t().contiguous(), None, residual=dX)
                    l += 1
        rq.flush()
        grads = [agg[0], agg[1] if ba is not None else None, agg[2]]
        for cg in comb_grads:
            grads += cg
        grads += [dWm, dbm if bm is not None else None, dam]
        return (dX, d_emb, None, None, None, None, None, None, *grads)


class SpmmHops:
    """hops on one GPU: the CSR operand and its transpose."""

    def __init__(self, fwd_mat, bwd_mat):
        self.fwd_mat, self.bwd_mat = fwd_mat, bwd_mat

    def forward(self, z):
        return ops.spmm_raw(self.fwd_mat, z)

    def backward(self, g):
        return ops.spmm_raw(self.bwd_mat, g)


def mgcn_forward(mgcn, hops, X, emb, k):
    """mgcn: the MGCN module (agg, combine, W_o, active, dropout); returns the branch's output."""
    from . import fused
    agg = mgcn.agg
    if agg.agg_type != "mean":
        raise ValueError(f"unknown aggregation type {agg.agg_type!r}")
    params = [agg.W.weight, agg.W.bias, fused.prelu_slope(agg.active)]
    for c in mgcn.combine:
        params += [c.W_ori_f.weight, c.W_ori_f.bias, c.W_f.weight, c.W_f.bias, c.W_o.weight, c.W_o.bias,
                   fused.prelu_slope(c.active)]
    params += [mgcn.W_o.weight, mgcn.W_o.bias, fused.prelu_slope(mgcn.active)]
    if any(t is None for t in (params[2], params[-1])) or any(fused.prelu_slope(c.active) is None for c in mgcn.combine):
        raise ValueError("the branch's activations must be one-slope PReLU modules")
    return _MGCN.apply(X, emb, hops, int(k), fused.drop_p(agg.dropout), [fused.drop_p(c.dropout) for c in mgcn.combine],
                       fused.drop_p(mgcn.dropout), len(mgcn.combine), *params)
