"""Numpy stand-in for pyfem_gpu_testflight_amd.dist.HipLocalOps -- TEST DOUBLE, lives in tests/.

Lets the partition / interface / single-reduction CG host logic of dist.py run under gloo on a
CPU-only machine.  Local matrices are scipy CSR built by the oracle; vectors are CPU torch tensors
(so torch.distributed collectives apply unchanged).  Never imported by the product package."""
import numpy as np
import torch


class NumpyLocalOps:
    def __init__(self, n):
        self.torch = torch
        self.device = torch.device("cpu")
        self.n = n

    def zeros(self, n, dtype=None):
        return torch.zeros(int(n), dtype=dtype or torch.float64)

    def from_numpy(self, a, dtype=None):
        t = torch.from_numpy(np.ascontiguousarray(a))
        return t.to(dtype) if dtype is not None else t

    def diag(self, K):
        return torch.from_numpy(K.diagonal().copy())

    def spmv_dot(self, K, x, y):
        y[:] = torch.from_numpy(K.dot(x.numpy()))
        self._delta = float(x.dot(y))

    def copy_indexed(self, n, src_idx, dst_idx, src, dst):
        v = src[src_idx.long()] if src_idx is not None else src[:n]
        if dst_idx is not None:
            dst[dst_idx.long()] = v
        else:
            dst[:n] = v

    def dot(self, a, b, w, out):
        out[0] = (a * b * w).sum() if w is not None else a.dot(b)

    def cgcg_update_xrz(self, state, p, s, dinv, w, x, r, z):
        alpha = state[0]
        x += alpha * p
        r -= alpha * s
        z[:] = dinv * r if dinv is not None else r
        ww = w if w is not None else 1.0
        self._gamma = float((ww * r * z).sum())
        self._rr = float((ww * r * r).sum())

    def cgcg_reduce(self, red, zero, nzero):
        if zero is not None:
            zero[:nzero] = 0.0
        red[0], red[1], red[2] = self._gamma, self._delta, self._rr

    def cgcg_update_ps(self, state_in, state_out, red, first, z, w, p, s):
        g, d = float(red[0]), float(red[1])
        if first:
            beta, alpha = 0.0, g / d
        else:
            beta = g / float(state_in[2])
            alpha = g / (d - beta * g / float(state_in[0]))
        state_out[0], state_out[1], state_out[2] = alpha, beta, g
        p.mul_(beta).add_(z)
        s.mul_(beta).add_(w)
