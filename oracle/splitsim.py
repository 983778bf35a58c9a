"""CPU *noise model* of split-operand matrix products.  TEST INFRASTRUCTURE ONLY.

Not a restatement of anything in the reference (which computes in fp32, tuatara.cpp:363-376, :443-446, :307).
Like bf16sim.py this patches the fp32 oracle (models.py) so that every convolution / linear / attention product is
formed the way a split-operand MFMA kernel forms it: each fp32 operand x is written as a sum of `terms` narrow
floats (x0 = rn(x), x1 = rn(x - x0), ...), and the product keeps the partial products x_i * w_j with i + j < terms
(3 of 4 for two terms, 6 of 9 for three), accumulated in fp32.  It answers, before any kernel is written, which
split reaches north_star's 1e-3 on the logits: bf16 pairs, fp16 pairs, bf16 triples.
"""
from __future__ import annotations

import contextlib

import torch
import torch.nn.functional as F

from . import models

_DT = {"bf16": torch.bfloat16, "f16": torch.float16}


def split(t: torch.Tensor, kind: str, terms: int, ftz: bool = False):
    """x -> [x0, x1, ...] with x0 + x1 + ... ~= x, every part exactly representable in `kind`."""
    dt = _DT[kind]
    parts, r = [], t
    for _ in range(terms):
        p = r.to(dt).to(torch.float32)
        if ftz and kind == "f16":                       # model an MFMA that flushes f16 subnormal inputs
            p = torch.where(p.abs() < 2.0 ** -14, torch.zeros_like(p), p)
        parts.append(p)
        r = r - p
    return parts


@contextlib.contextmanager
def split_noise(kind: str = "bf16", terms: int = 2, ftz: bool = False, attn: bool = True):
    lin, conv, sdpa = F.linear, F.conv2d, F.scaled_dot_product_attention
    attn_fwd = models._Attention.forward
    fast = torch.backends.mha.get_fastpath_enabled()

    def sp(t):
        return split(t, kind, terms, ftz)

    def combine(fn, xs, ws):
        out = None
        for i, xi in enumerate(xs):
            for j, wj in enumerate(ws):
                if i + j < terms:
                    y = fn(xi, wj)
                    out = y if out is None else out + y
        return out

    def linear(x, w, b=None):
        y = combine(lambda a, c: lin(a, c), sp(x), sp(w))
        return y if b is None else y + b

    def conv2d(x, w, b=None, *a, **k):
        y = combine(lambda p, q: conv(p, q, None, *a, **k), sp(x), sp(w))
        return y if b is None else y + b.view(1, -1, 1, 1)

    def mm(a, b):
        return combine(lambda p, q: p @ q, sp(a), sp(b))

    def sdp(q, k, v, attn_mask=None, dropout_p=0.0, is_causal=False, scale=None, **kw):
        if not attn:
            return sdpa(q, k, v, attn_mask=attn_mask, dropout_p=dropout_p, is_causal=is_causal, scale=scale, **kw)
        s = (q.shape[-1] ** -0.5) if scale is None else scale
        a = mm(q * s, k.transpose(-2, -1))
        if attn_mask is not None:
            a = a.masked_fill(~attn_mask, float("-inf")) if attn_mask.dtype == torch.bool else a + attn_mask
        return mm(a.softmax(-1), v)

    def enc_attention(self, x):
        B, N, C = x.shape
        qkv = self.qkv(x).reshape(B, N, 3, self.num_heads, C // self.num_heads).permute(2, 0, 3, 1, 4)
        q, k, v = qkv.unbind(0)
        a = mm(q * self.scale, k.transpose(-2, -1)).softmax(dim=-1)
        x = mm(a, v).transpose(1, 2).reshape(B, N, C)
        return self.proj(x)

    F.linear, F.conv2d, F.scaled_dot_product_attention = linear, conv2d, sdp
    if attn:
        models._Attention.forward = enc_attention
    torch.backends.mha.set_fastpath_enabled(False)
    try:
        yield
    finally:
        F.linear, F.conv2d, F.scaled_dot_product_attention = lin, conv, sdpa
        models._Attention.forward = attn_fwd
        torch.backends.mha.set_fastpath_enabled(fast)
