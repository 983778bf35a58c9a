"""CPU *noise model* of the engine's bf16 mode.  TEST INFRASTRUCTURE ONLY.

Not a restatement of anything in the reference: the reference computes in fp32
(tuatara.cpp:363-370, :443-446).  This context manager makes the fp32 oracle
(models.py) round the operands of every convolution / linear / attention product
to bfloat16 (fp32 accumulation), i.e. it puts a rounding error of the engine's
size at the places where the engine's bf16 mode rounds.  It is used only to
*calibrate* the synthetic weights (tuatara_amd/weights.py) on the CPU — how far
are the oracle's decisions (argmax margins, heat-map thresholds) from bf16 noise —
and by the CPU test that pins that calibration.  The GPU parity tests compare the
real engine with the unmodified fp32 oracle.
"""
from __future__ import annotations

import contextlib

import torch
import torch.nn.functional as F

from . import models


def _r(t: torch.Tensor) -> torch.Tensor:
    return t.to(torch.bfloat16).to(torch.float32) if t is not None and t.dtype == torch.float32 else t


@contextlib.contextmanager
def bf16_noise():
    lin, conv, sdpa = F.linear, F.conv2d, F.scaled_dot_product_attention
    attn_fwd = models._Attention.forward
    fast = torch.backends.mha.get_fastpath_enabled()

    def linear(x, w, b=None):
        return lin(_r(x), _r(w), b)

    def conv2d(x, w, b=None, *a, **k):
        return conv(_r(x), _r(w), b, *a, **k)

    def sdp(q, k, v, *a, **kw):
        return sdpa(_r(q), _r(k), _r(v), *a, **kw)

    def enc_attention(self, x):
        B, N, C = x.shape
        qkv = _r(self.qkv(x)).reshape(B, N, 3, self.num_heads, C // self.num_heads).permute(2, 0, 3, 1, 4)
        q, k, v = qkv.unbind(0)
        attn = ((q @ k.transpose(-2, -1)) * self.scale).softmax(dim=-1)
        x = (_r(attn) @ v).transpose(1, 2).reshape(B, N, C)
        return self.proj(x)

    F.linear, F.conv2d, F.scaled_dot_product_attention = linear, conv2d, sdp
    torch.nn.functional.linear = linear
    models._Attention.forward = enc_attention
    torch.backends.mha.set_fastpath_enabled(False)
    try:
        yield
    finally:
        F.linear, F.conv2d, F.scaled_dot_product_attention = lin, conv, sdpa
        models._Attention.forward = attn_fwd
        torch.backends.mha.set_fastpath_enabled(fast)
