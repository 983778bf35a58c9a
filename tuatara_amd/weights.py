"""Weight files for the engine: tensor specs, seeded synthetic weights, BN folding, ``.ttrw`` I/O.

The reference loads two TorchScript archives from ``weights_dir``
(tuatara.cpp:333 ``craft_traced_torchscript_model.pt``, tuatara.cpp:423
``parseq_torchscript.bin``).  Those archives are not in the reference tree and
cannot be fetched here, so the engine reads a flat file per model
(``craft.ttrw`` / ``parseq.ttrw``) that ``tools/convert_weights.py`` produces
from the archives' ``state_dict`` and that ``synth_*`` below produces from a
seed for tests and the benchmark.  State-dict key names are the upstream
(clovaai/CRAFT-pytorch, baudm/parseq) names so real checkpoints drop in.

numpy only — no torch, no oracle import.
"""
from __future__ import annotations

import math
import os
import struct
from typing import Dict, List, Tuple

import numpy as np

State = Dict[str, np.ndarray]

CRAFT_FILE = "craft.ttrw"
PARSEQ_FILE = "parseq.ttrw"

# ----------------------------------------------------------------------------- CRAFT spec
# (engine layer name, upstream conv key, upstream BN key or None, cin, cout, ksize)
_VGG = [  # slice, conv index in torchvision vgg16_bn.features, cin, cout
    ("slice1", 0, 3, 64), ("slice1", 3, 64, 64), ("slice1", 7, 64, 128), ("slice1", 10, 128, 128),
    ("slice2", 14, 128, 256), ("slice2", 17, 256, 256),
    ("slice3", 20, 256, 256), ("slice3", 24, 256, 512), ("slice3", 27, 512, 512),
    ("slice4", 30, 512, 512), ("slice4", 34, 512, 512), ("slice4", 37, 512, 512),
]


def craft_layers() -> List[Tuple[str, str, str | None, int, int, int]]:
    """Every conv of CRAFT in execution order."""
    out = []
    for sl, idx, cin, cout in _VGG:
        out.append((f"{sl}.{idx}", f"basenet.{sl}.{idx}", f"basenet.{sl}.{idx + 1}", cin, cout, 3))
    out.append(("slice5.1", "basenet.slice5.1", None, 512, 1024, 3))
    out.append(("slice5.2", "basenet.slice5.2", None, 1024, 1024, 1))
    for n, (cin, mid, cout) in enumerate([(1024, 512, 256), (512, 256, 128), (256, 128, 64), (128, 64, 32)], 1):
        out.append((f"upconv{n}.0", f"upconv{n}.conv.0", f"upconv{n}.conv.1", cin + mid, mid, 1))
        out.append((f"upconv{n}.3", f"upconv{n}.conv.3", f"upconv{n}.conv.4", mid, cout, 3))
    for idx, cin, cout, k in [(0, 32, 32, 3), (2, 32, 32, 3), (4, 32, 16, 3), (6, 16, 16, 1), (8, 16, 2, 1)]:
        out.append((f"conv_cls.{idx}", f"conv_cls.{idx}", None, cin, cout, k))
    return out


def craft_spec() -> List[Tuple[str, Tuple[int, ...]]]:
    spec = []
    for _, conv, bn, cin, cout, k in craft_layers():
        spec.append((conv + ".weight", (cout, cin, k, k)))
        spec.append((conv + ".bias", (cout,)))
        if bn:
            for s in ("weight", "bias", "running_mean", "running_var"):
                spec.append((f"{bn}.{s}", (cout,)))
    return spec


# ----------------------------------------------------------------------------- PARSeq spec
EMBED, ENC_DEPTH, ENC_HEADS, DEC_HEADS, FFN = 384, 12, 6, 12, 1536
N_PATCH, MAX_LEN, N_CLASSES, N_TOKENS = 128, 25, 95, 97


def parseq_spec() -> List[Tuple[str, Tuple[int, ...]]]:
    E = EMBED
    spec = [("encoder.pos_embed", (1, N_PATCH, E)),
            ("encoder.patch_embed.proj.weight", (E, 3, 4, 8)), ("encoder.patch_embed.proj.bias", (E,))]
    for i in range(ENC_DEPTH):
        p = f"encoder.blocks.{i}."
        spec += [(p + "norm1.weight", (E,)), (p + "norm1.bias", (E,)),
                 (p + "attn.qkv.weight", (3 * E, E)), (p + "attn.qkv.bias", (3 * E,)),
                 (p + "attn.proj.weight", (E, E)), (p + "attn.proj.bias", (E,)),
                 (p + "norm2.weight", (E,)), (p + "norm2.bias", (E,)),
                 (p + "mlp.fc1.weight", (4 * E, E)), (p + "mlp.fc1.bias", (4 * E,)),
                 (p + "mlp.fc2.weight", (E, 4 * E)), (p + "mlp.fc2.bias", (E,))]
    spec += [("encoder.norm.weight", (E,)), ("encoder.norm.bias", (E,))]
    p = "decoder.layers.0."
    for a in ("self_attn", "cross_attn"):
        spec += [(p + a + ".in_proj_weight", (3 * E, E)), (p + a + ".in_proj_bias", (3 * E,)),
                 (p + a + ".out_proj.weight", (E, E)), (p + a + ".out_proj.bias", (E,))]
    spec += [(p + "linear1.weight", (FFN, E)), (p + "linear1.bias", (FFN,)),
             (p + "linear2.weight", (E, FFN)), (p + "linear2.bias", (E,))]
    for n in ("norm1", "norm2", "norm_q", "norm_c"):
        spec += [(p + n + ".weight", (E,)), (p + n + ".bias", (E,))]
    spec += [("decoder.norm.weight", (E,)), ("decoder.norm.bias", (E,)),
             ("head.weight", (N_CLASSES, E)), ("head.bias", (N_CLASSES,)),
             ("text_embed.embedding.weight", (N_TOKENS, E)), ("pos_queries", (1, MAX_LEN + 1, E))]
    return spec


# ----------------------------------------------------------------------------- synthetic weights
def synth_craft(seed: int = 0, structured: bool = True) -> State:
    """Seeded CRAFT weights.  He-scaled random convs keep activations O(1) through
    the 27 layers.  ``structured`` additionally wires a hand-designed "ink
    density" pathway (darkness -> blur -> region / affinity channels) on top of
    the random features so a real page yields word-shaped blobs: channel 0 of
    every layer on the relu2_2 -> upconv4 -> conv_cls route carries it."""
    rng = np.random.default_rng(seed)
    st: State = {}
    for name, conv, bn, cin, cout, k in craft_layers():
        fan_in = cin * k * k
        gain = 2.0 if (bn or name.startswith("conv_cls")) and name != "conv_cls.8" else 1.0
        st[conv + ".weight"] = (rng.standard_normal((cout, cin, k, k)) * math.sqrt(gain / fan_in)).astype(np.float32)
        st[conv + ".bias"] = (rng.standard_normal(cout) * 0.05).astype(np.float32)
        if bn:
            st[bn + ".weight"] = rng.uniform(0.8, 1.2, cout).astype(np.float32)
            st[bn + ".bias"] = (rng.standard_normal(cout) * 0.1).astype(np.float32)
            st[bn + ".running_mean"] = (rng.standard_normal(cout) * 0.1).astype(np.float32)
            st[bn + ".running_var"] = rng.uniform(0.8, 1.2, cout).astype(np.float32)
    if structured:
        _wire_ink_pathway(st)
    return st


def _wire_ink_pathway(st: State, gain: float = 6.0) -> None:
    """Hand-designed "text-ness" detector on channels 0..3 of the route
    slice1 -> relu2_2 skip -> upconv4 -> conv_cls, so a real page gives word-shaped blobs.

    ink = 1 - mean(rgb) (input is /255).  conv1_2 takes two high-passes of it,
    a = relu(ink - hmean3(ink)) (vertical strokes) and b = relu(ink - vmean3(ink))
    (horizontal strokes); after the 2x2 max-pool they are blurred (3x3 box for
    the region pair, 1x3 horizontal for the affinity pair).  conv_cls.4/.6/.8
    form min(gain * min(A, B), 1) = 1 - relu(1 - gain*(A - relu(A - B))): text has
    strokes both ways, ruled lines and the black padding margin do not.  The four
    designed channels read only themselves, their BatchNorms are the identity and
    no random channel reads them; random features add a 2 % texture in conv_cls.8."""
    D = 4

    def ident_bn(bn: str) -> None:
        for c in range(D):
            st[bn + ".weight"][c] = 1.0
            st[bn + ".bias"][c] = 0.0
            st[bn + ".running_mean"][c] = 0.0
            st[bn + ".running_var"][c] = 1.0 - BN_EPS

    def isolate(conv: str, bn: str | None = None, col0: int = 0):
        w, b = st[conv + ".weight"], st[conv + ".bias"]
        w[:, col0:col0 + D] = 0.0   # nobody reads the designed channels ...
        w[0:D] = 0.0                # ... and they read nobody
        b[0:D] = 0.0
        if bn:
            ident_bn(bn)
        return w, b

    k = w = None
    ident = np.zeros((3, 3), np.float32); ident[1, 1] = 1.0
    box = np.full((3, 3), 1.0 / 9.0, np.float32)
    wide = np.zeros((3, 3), np.float32); wide[1, :] = 1.0 / 3.0
    hp_h = np.zeros((3, 3), np.float32); hp_h[1, :] = (-1.0 / 3.0, 2.0 / 3.0, -1.0 / 3.0)
    hp_v = np.ascontiguousarray(hp_h.T)

    def taps(conv: str, bn: str | None, per_channel, col0: int = 0) -> None:
        w, _ = isolate(conv, bn, col0)
        for c, t in enumerate(per_channel):
            if w.shape[-1] == 1:
                w[c, col0 + c, 0, 0] = t[1, 1]
            else:
                w[c, col0 + c] = t

    # conv1_1: channel 0 = ink; channels 1..3 dead
    w, b = st["basenet.slice1.0.weight"], st["basenet.slice1.0.bias"]
    w[0:D] = 0.0
    b[0:D] = 0.0
    w[0, :, 1, 1] = -1.0 / 3.0
    b[0] = 1.0
    ident_bn("basenet.slice1.1")
    # conv1_2: a, b, a, b from channel 0
    w, _ = isolate("basenet.slice1.3", "basenet.slice1.4")
    w[0, 0], w[1, 0], w[2, 0], w[3, 0] = hp_h, hp_v, hp_h, hp_v
    taps("basenet.slice1.7", "basenet.slice1.8", [box, box, wide, wide])
    taps("basenet.slice1.10", "basenet.slice1.11", [ident, ident, wide, wide])
    st["basenet.slice2.14.weight"][:, 0:D] = 0.0  # the trunk does not read them
    taps("upconv4.conv.0", "upconv4.conv.1", [ident] * D, col0=64)  # skip channels sit after up(y)'s 64
    taps("upconv4.conv.3", "upconv4.conv.4", [ident] * D)
    taps("conv_cls.0", None, [ident] * D)
    taps("conv_cls.2", None, [ident] * D)
    # conv_cls.4: (A_r, relu(A_r - B_r), A_l, relu(A_l - B_l))
    w, _ = isolate("conv_cls.4")
    w[0, 0, 1, 1] = 1.0
    w[1, 0, 1, 1], w[1, 1, 1, 1] = 1.0, -1.0
    w[2, 2, 1, 1] = 1.0
    w[3, 2, 1, 1], w[3, 3, 1, 1] = 1.0, -1.0
    # conv_cls.6: u = relu(1 - gain * min(A, B))
    w, b = isolate("conv_cls.6")
    w[0, 0, 0, 0], w[0, 1, 0, 0] = -gain, gain
    w[1, 2, 0, 0], w[1, 3, 0, 0] = -gain, gain
    b[0:2] = 1.0
    # conv_cls.8: y = 1 - u (+ faint random texture from the other channels)
    w = st["conv_cls.8.weight"]
    w *= 0.02
    w[:, 0:D] = 0.0
    w[0, 0, 0, 0] = -1.0
    w[1, 1, 0, 0] = -1.0
    st["conv_cls.8.bias"][:] = 1.0


def synth_parseq(seed: int = 0, eos_shift: float = 1.7, head_gain: float = 6.0, sharp: float = 3.0, structured: bool = True) -> State:
    """Seeded PARSeq weights.  1/sqrt(fan_in) linears (O(1) residual updates),
    LayerNorm near identity, q/k projections scaled by ``sharp`` so attention is
    peaked, a head gain so logits have std ~``head_gain`` and an EOS shift.

    A purely random model decides every character by the top of 95 Gaussian
    scores: the top-2 margin is below twice the bf16 error of the whole network
    at ~7 % of the positions, so greedy decoding forks on more than half of the
    crops and string equality says little.  ``structured`` (the default) therefore
    wires a confident, "trained-like" read-out on 64 reserved channels on top of
    the random model (``_wire_parseq_dfa``) and turns the random head down to a
    texture of std ~0.5."""
    rng = np.random.default_rng(seed + 1000)
    st: State = {}
    E = EMBED
    for name, shape in parseq_spec():
        if ("norm" in name) and name.endswith(".weight"):
            v = rng.uniform(0.8, 1.2, shape)
        elif name.endswith("bias"):
            v = rng.standard_normal(shape) * 0.05
        elif name == "encoder.pos_embed":
            v = rng.standard_normal(shape) * 0.5
        elif name == "pos_queries":
            v = rng.standard_normal(shape) * 0.2
        elif name == "text_embed.embedding.weight":
            v = rng.standard_normal(shape) * (1.0 / math.sqrt(E))  # x sqrt(E) at lookup -> O(1)
        elif name == "encoder.patch_embed.proj.weight":
            v = rng.standard_normal(shape) * (2.0 / math.sqrt(96))
        else:
            v = rng.standard_normal(shape) * (1.0 / math.sqrt(shape[-1]))
        v = v.astype(np.float32)
        if name.endswith("attn.qkv.weight") or name.endswith("in_proj_weight"):
            v[: 2 * E] *= np.float32(sharp)
        if name.endswith("cross_attn.out_proj.weight"):
            v *= np.float32(sharp)
        st[name] = v
    if structured:
        st["head.weight"] *= np.float32(0.5)          # texture: logit std ~0.5 under the designed read-out
        _wire_parseq_dfa(st)
    else:
        st["head.weight"] *= np.float32(head_gain)
        st["head.bias"][0] += np.float32(eos_shift * head_gain)
    return st


# ---- the designed PARSeq read-out ---------------------------------------------------------------
# Reserved model channels (the last 64 of 384).  Random weights neither read nor write them; every
# LayerNorm is the identity on them (gamma 1, beta 0), so a reserved value v leaves a LayerNorm as
# (v - mu) / sigma with the token's mu and sigma.  Every designed row reads differences against the
# ZERO channel (weights sum to zero: mu cancels) and states its thresholds as multiples of a reference
# channel that went through the same LayerNorms (sigma cancels): the logic is ratio-metric.
RES0 = 320
_ZERO, _ONE = 320, 321
# encoder stream
_RG, _C1, _D, _S, _B1, _B2, _B3 = 322, 330, 331, 339, 347, 356, 365       # RG/D/S: 8 channels, B*: 8 bits + reference
_DA, _SA = 374, 375                                                          # |contrast| per patch, and its mean over the crop
# decoder streams: context stream (token + position embeddings) ...
_POS, _BOSF, _TREF, _TOK = 322, 348, 349, 350                                # POS: 26, TOK: 7
# ... and query stream
_PREV, _PBOS, _PREF, _CONT, _CREF, _OUT = 357, 364, 365, 366, 374, 375       # PREV: 7, CONT: 8, OUT: 7
N_BITS, N_CODE = 8, 7
A_ONE, A_POS, A_TOK = 2.0, 3.0, 1.0
DFA_FIRST_BITS = 6                     # content bits 0..5 choose the first character (bits 6..7 are computed and not read)


def dfa_tables(seed: int = 0):
    """The string a crop decodes to under the designed read-out: first[v] (v = content bits 0..5 as an integer) is
    the first class id, nxt[t] the class that follows class t; class 0 is EOS.  first is Gray-like in the content bits -
    flipping one content bit flips exactly one bit of the 7-bit class code (first[v] = v) - so that near a content-bit
    tie the logits move between two neighbouring classes only, with a gentle slope.  nxt reaches ids up to 94: the
    reference tokenizer's shifted ids 69..94 and its eos_id 88 (tuatara.cpp:31-48) all occur in decoded strings."""
    rng = np.random.default_rng(seed + 4242)
    first = np.arange(64, dtype=np.int64)
    # every class has a level 1..10 = the length of the string that starts with it (config 5 draws words of 3-10 characters);
    # nxt[t] is a random class one level down, level-1 classes end the string: every chain reaches EOS within 10 steps, so a
    # batch's autoregressive loop can stop early the way upstream PARSeq's does
    level = np.zeros(95, np.int64)
    level[1:] = 1 + rng.permutation(94) % 10
    nxt = np.zeros(95, np.int64)
    for t in range(1, 95):
        if level[t] > 1:
            nxt[t] = int(rng.choice(np.nonzero(level == level[t] - 1)[0]))
    return first, nxt


def _code(c: int) -> np.ndarray:
    return np.array([1.0 if (c >> k) & 1 else -1.0 for k in range(N_CODE)], np.float32)


def _wire_parseq_dfa(st: State, seed: int = 0) -> None:
    """Hand-wired confident read-out (the PARSeq counterpart of ``_wire_ink_pathway``), every stage through the
    same kernels as the random part:

    encoder  patch-embed: C1 = left-minus-right contrast of the patch.  block 0 MLP: D_j = C1 gated by the patch's
             region j (8 regions of 2 patch columns) - GELU(z) - GELU(-z) = z exactly.  block 1 attention,
             head 5: q = k = 0, uniform pooling over the 128 tokens -> S_j = region sums, at every token.  blocks 1..3
             MLP: three soft sign stages B = sat(g S) (four GELU units per bit); every knee is a fraction of a reference
             that scales with the crop's contrast (SA = pooled |C1|, then the previous stage's reference bit), so the
             soft zone tracks the bf16 noise of the pooled sums whatever the crop.  memory carries 8 bits +-BREF.
    decoder  self-attention head 11: query i matches context slot i by a one-hot position code and copies the
             previous token's 7-bit code (PREV), a was-BOS flag and a reference.  cross-attention head 11: uniform
             pooling of the memory's bits (CONT) and reference.  FFN: exact-match detectors - 64 for the first
             character (BOS and content bits 0..5), 94 for the transitions (previous class) - write the next class's
             code to OUT; no detector firing means EOS.  head: logit_c = G <code_c, OUT> + texture.

    Near a content-bit tie (|S_j| within a few bf16 errors of 0) the bit is soft, the detectors fire partly and the
    logit margin shrinks continuously, so a decision that bf16 noise can flip shows a small fp32 margin."""
    E = EMBED
    R = slice(RES0, E)
    first, nxt = dfa_tables(seed)

    def put(w: np.ndarray, row: int, cols: Dict[int, float]) -> None:
        """w[row, c] = v for the reserved input channels c, and minus their sum on ZERO (mean-free row)."""
        tot = 0.0
        for c, v in cols.items():
            w[row, c] = v
            tot += v
        w[row, _ZERO] -= tot

    # ---------------- isolate the reserved channels from the random model
    st["encoder.patch_embed.proj.weight"][R] = 0.0
    st["encoder.patch_embed.proj.bias"][R] = 0.0
    st["encoder.pos_embed"][:, :, R] = 0.0
    for i in range(ENC_DEPTH):
        p = f"encoder.blocks.{i}."
        st[p + "attn.qkv.weight"][:, R] = 0.0
        st[p + "attn.proj.weight"][R, :] = 0.0
        st[p + "attn.proj.bias"][R] = 0.0
        st[p + "mlp.fc1.weight"][:, R] = 0.0
        st[p + "mlp.fc2.weight"][R, :] = 0.0
        st[p + "mlp.fc2.bias"][R] = 0.0
        for n in ("norm1", "norm2"):
            st[p + n + ".weight"][R] = 1.0
            st[p + n + ".bias"][R] = 0.0
    st["encoder.norm.weight"][R] = 1.0
    st["encoder.norm.bias"][R] = 0.0
    d = "decoder.layers.0."
    for a in ("self_attn", "cross_attn"):
        st[d + a + ".in_proj_weight"][:, R] = 0.0
        st[d + a + ".out_proj.weight"][R, :] = 0.0
        st[d + a + ".out_proj.bias"][R] = 0.0
    st[d + "linear1.weight"][:, R] = 0.0
    st[d + "linear2.weight"][R, :] = 0.0
    st[d + "linear2.bias"][R] = 0.0
    for n in (d + "norm1", d + "norm2", d + "norm_q", d + "norm_c", "decoder.norm"):
        st[n + ".weight"][R] = 1.0
        st[n + ".bias"][R] = 0.0
    st["head.weight"][:, R] = 0.0
    st["text_embed.embedding.weight"][:, R] = 0.0
    st["pos_queries"][:, :, R] = 0.0

    # ---------------- encoder
    w = st["encoder.patch_embed.proj.weight"]                 # [E, 3, 4, 8]
    w[_C1, :, :, 0:4] = 1.0 / 16.0
    w[_C1, :, :, 4:8] = -1.0 / 16.0
    pe = st["encoder.pos_embed"]                               # [1, 128, E], token = py * 16 + px
    pe[0, :, _ONE] = A_ONE
    for tok in range(N_PATCH):
        j = (tok & 15) >> 1
        pe[0, tok, _RG + j] = A_ONE

    def mlp_units(block: int, n: int):
        p = f"encoder.blocks.{block}.mlp."
        st[p + "fc1.weight"][0:n] = 0.0
        st[p + "fc1.bias"][0:n] = 0.0
        st[p + "fc2.weight"][:, 0:n] = 0.0
        return st[p + "fc1.weight"], st[p + "fc1.bias"], st[p + "fc2.weight"]

    # block 0 MLP: D_j = clip(g1 C1, +-kc) / sigma where the patch lies in region j (gate: M (RG_j - ONE) = 0 or -M A_ONE / sigma),
    # DA = |D_j|.  clip(z) = (h(z) - h(z-k)) - (h(-z) - h(-z-k)), |clip(z)| = the same four units with a + : exact for GELU's h up to
    # its rounding at 0.  The clip (at |C1| = 1; a random-noise crop has sigma 0.18, a black/white edge up to 3) bounds how much a
    # high-contrast crop can scale the bits - the decoder's detectors see content amplitudes within a factor ~3 - and a clipped patch
    # carries at most 3x the bf16 rounding error of an unclipped one (h(z) - h(z - k) cancels).
    g1, M, kc = 32.0, 64.0, 1.0 * 32.0 / A_ONE                # GELU arguments of several units: the relu-like regime
    w1, b1, w2 = mlp_units(0, 4 * N_BITS)
    for j in range(N_BITS):
        for s, (sz, sk, so) in enumerate(((1, 0, 1), (1, -1, -1), (-1, 0, -1), (-1, -1, 1))):
            u = 4 * j + s
            put(w1, u, {_C1: sz * g1, _RG + j: M, _ONE: -M + sk * kc})
            w2[_D + j, u] = so / 8.0
            w2[_DA, u] = so * sz / 8.0
    # block 1 attention head 5: uniform pooling over all tokens: S_j = region sums of D_j, SA = mean of DA, written to every token
    p = "encoder.blocks.1.attn."
    qkv, qb = st[p + "qkv.weight"], st[p + "qkv.bias"]
    for part in range(3):
        qkv[part * E + 320: part * E + 384] = 0.0
        qb[part * E + 320: part * E + 384] = 0.0
    pw = st[p + "proj.weight"]
    pw[:, 320:384] = 0.0
    for j in range(N_BITS):
        put(qkv, 2 * E + 320 + j, {_D + j: 1.0})
        pw[_S + j, 320 + j] = 8.0                              # 16 of the 128 tokens carry region j
    put(qkv, 2 * E + 320 + N_BITS, {_DA: 1.0})
    pw[_SA, 320 + N_BITS] = 1.0

    def sat_stage(block: int, src: int, ref: int, dst: int, gain: float, kappa: float, ref_gain: float) -> None:
        """dst_j = w [ (h(z+k) - h(z-k)) - (h(-z+k) - h(-z-k)) ], z = gain * src_j / sigma, k = kappa * ref / sigma: a soft sign of
        amplitude 2 kappa w ref / sigma whose knee sits at |src_j| = (kappa / gain) ref - a fraction of the reference, which went
        through the same LayerNorms and scales with the crop's contrast like the bits (and like their bf16 noise): the knee is
        scale-free.  Bit 8 = the next stage's reference: the same units driven by ref itself (always saturated to +)."""
        w1, b1, w2 = mlp_units(block, 4 * (N_BITS + 1))
        wout = 0.25
        for j in range(N_BITS + 1):
            cin, g = (src + j, gain) if j < N_BITS else (ref, ref_gain)
            for s, (sz, sk, so) in enumerate(((1, 1, 1), (1, -1, -1), (-1, 1, -1), (-1, -1, 1))):
                u = 4 * j + s
                cols = {cin: sz * g}
                cols[ref] = cols.get(ref, 0.0) + sk * kappa
                put(w1, u, cols)
                w2[dst + j, u] = so * wout

    sat_stage(1, _S, _SA, _B1, 20.0, 16.0, 64.0)               # knee |S_j| = 0.8 SA ~ 3 sigma(S_j): linear
    sat_stage(2, _B1, _B1 + N_BITS, _B2, 16.0, 4.0, 16.0)      # knee at 1/4 of the reference amplitude
    sat_stage(3, _B2, _B2 + N_BITS, _B3, 16.0, 4.0, 16.0)      # -> bits saturate beyond |S_j| ~ SA / 20

    # ---------------- decoder: embeddings
    pq = st["pos_queries"]                                      # [1, 26, E]
    pq[0, :, _ONE] = A_ONE
    for i in range(MAX_LEN + 1):
        pq[0, i, _POS + i] = A_POS
    te = st["text_embed.embedding.weight"]                      # [97, E], scaled by sqrt(E) at lookup
    rs = 1.0 / math.sqrt(E)
    for t in range(N_CLASSES):
        te[t, _TOK:_TOK + N_CODE] = _code(t) * A_TOK * rs
        te[t, _TREF] = A_TOK * rs
    te[95, _BOSF] = A_TOK * rs                                  # [B]
    te[95, _TREF] = A_TOK * rs

    # self-attention head 11: one-hot position match -> previous token's code
    H0 = 352
    w, b = st[d + "self_attn.in_proj_weight"], st[d + "self_attn.in_proj_bias"]
    for part in range(3):
        w[part * E + H0: part * E + H0 + 32] = 0.0
        b[part * E + H0: part * E + H0 + 32] = 0.0
    s_q, s_k = NOM["self_sq"], NOM["self_sk"]
    for m in range(MAX_LEN):                                    # query i (>= 1) lights dim i - 1; context slot j (>= 1) carries POS_{j-1}
        put(w, H0 + m, {_POS + m + 1: s_q})
        put(w, E + H0 + m, {_POS + m: s_k})
    put(w, H0 + 26, {_POS + 0: s_q})                            # query 0 <-> the BOS slot
    put(w, E + H0 + 26, {_BOSF: s_k * A_POS / A_TOK})
    for k in range(N_CODE):
        put(w, 2 * E + H0 + k, {_TOK + k: 1.0})
    put(w, 2 * E + H0 + 7, {_BOSF: 1.0})
    put(w, 2 * E + H0 + 8, {_TREF: 1.0})
    ow = st[d + "self_attn.out_proj.weight"]
    ow[:, H0:H0 + 32] = 0.0
    for k in range(N_CODE):
        ow[_PREV + k, H0 + k] = 1.0
    ow[_PBOS, H0 + 7] = 1.0
    ow[_PREF, H0 + 8] = 1.0

    # cross-attention head 11: uniform pooling of the memory's bits
    w, b = st[d + "cross_attn.in_proj_weight"], st[d + "cross_attn.in_proj_bias"]
    for part in range(3):
        w[part * E + H0: part * E + H0 + 32] = 0.0
        b[part * E + H0: part * E + H0 + 32] = 0.0
    for j in range(N_BITS + 1):
        put(w, 2 * E + H0 + j, {_B3 + j: 1.0})
    ow = st[d + "cross_attn.out_proj.weight"]
    ow[:, H0:H0 + 32] = 0.0
    for j in range(N_BITS):
        ow[_CONT + j, H0 + j] = 1.0
    ow[_CREF, H0 + N_BITS] = 1.0

    # FFN detectors, each a clipped pair h(pre) - h(pre - cap): the output is w_o cap whenever pre > cap + 3 - it does not pass on
    # the rounding noise of pre, nor the crop's contrast
    n_det = 64 + 94
    l1, lb, l2 = st[d + "linear1.weight"], st[d + "linear1.bias"], st[d + "linear2.weight"]
    l1[0:2 * n_det] = 0.0
    lb[0:2 * n_det] = 0.0
    l2[:, 0:2 * n_det] = 0.0
    a_c, a_p, Mg, cap, o = NOM["ffn_ac"], NOM["ffn_ap"], NOM["ffn_gate"], NOM["ffn_cap"], NOM["out_amp"]
    w_o = o / cap
    eos = _code(0)
    st[d + "linear2.bias"][_OUT:_OUT + N_CODE] = o * eos

    def detector(u: int, cols: Dict[int, float], nxt_class: int) -> None:
        for s in range(2):
            put(l1, 2 * u + s, cols)
            lb[2 * u + s] = -cap * s
            l2[_OUT:_OUT + N_CODE, 2 * u + s] = (1.0 - 2.0 * s) * w_o * (_code(nxt_class) - eos)

    u = 0
    for v in range(64):                                         # first character: BOS and content bits 0..5 = v
        # a_c (<u, CONT> - 4 CREF): 2 b on a match, b when one bit sits at 0, 0 one bit off, < 0 beyond
        cols = {_CREF: -4.0 * a_c, _PBOS: Mg, _PREF: -Mg}
        for j in range(DFA_FIRST_BITS):
            cols[_CONT + j] = a_c if (v >> j) & 1 else -a_c
        detector(u, cols, int(first[v]))
        u += 1
    for t in range(1, 95):                                      # transitions: previous class t
        # a_p (<code_t, PREV> - 6 PREF): a on a match, <= -a otherwise.  (Content bits are not read here: PREV's and PREF's bf16
        # roundings differ coherently over the 7 code bits, and a threshold that has to cancel them exactly passes that on.)
        cols = {_PREF: -6.0 * a_p}
        for k, ck in enumerate(_code(t)):
            cols[_PREV + k] = a_p * float(ck)
        detector(u, cols, int(nxt[t]))
        u += 1
    # head: logit_c = G <code_c, OUT> (ZERO column keeps the rows mean-free)
    hw = st["head.weight"]
    G = NOM["head_gain"]
    for c in range(N_CLASSES):
        put(hw, c, {_OUT + k: G * float(ck) for k, ck in enumerate(_code(c))})


# Nominal gains of the designed decoder rows.  They only set how sharp the detectors are (the thresholds are
# ratio-metric); measured with tools/calibrate_parseq.py on the seed-0 random model (LayerNorm sigmas of the
# query / context streams) and rounded.
NOM = {
    "self_sq": 1.5, "self_sk": 1.5,          # match score (1.5 * 3 / 0.26) (1.5 * 3 / 0.95) / sqrt(32) ~ 14.5
    "ffn_ac": 48.0, "ffn_ap": 28.0,         # content bit ~0.25 / 2.47 (random-noise crops; it scales with the crop's contrast) and token bit 1.03 / 2.47 -> detector steps b ~5 (up to ~15) and a ~11.5
    "ffn_gate": 1024.0, "ffn_cap": 5.0, "out_amp": 2.5,   # gate: -1024 x 0.42 when the previous token is not BOS (content amplitude x a_c never gets there)
    "head_gain": 2.0,                        # OUT 2.5 / 2.55 -> +-2 per code bit: top-2 margin ~3.9 (bf16 |dlogit| ~0.05), top logit ~14
}


# ----------------------------------------------------------------------------- export (engine layout)
BN_EPS = 1e-5


def fold_craft(st: State) -> State:
    """Fold eval-mode BatchNorm into the preceding conv (fp64 fold, fp32 store) and
    permute OIHW -> O,kh,kw,I (K contiguous = tap-major, channel-minor)."""
    out: State = {}
    for name, conv, bn, cin, cout, k in craft_layers():
        w = st[conv + ".weight"].astype(np.float64)
        b = st[conv + ".bias"].astype(np.float64)
        if bn:
            g = st[bn + ".weight"].astype(np.float64)
            beta = st[bn + ".bias"].astype(np.float64)
            mu = st[bn + ".running_mean"].astype(np.float64)
            var = st[bn + ".running_var"].astype(np.float64)
            s = g / np.sqrt(var + BN_EPS)
            w = w * s[:, None, None, None]
            b = (b - mu) * s + beta
        out[name + ".w"] = np.ascontiguousarray(w.transpose(0, 2, 3, 1)).astype(np.float32)
        out[name + ".b"] = b.astype(np.float32)
    return out


def pack_parseq(st: State) -> State:
    """Engine layout: linears stay [out,in]; patch-embed weight becomes
    [384, (dy,dx,c)=96] to match NHWC crops; token embedding pre-scaled by
    sqrt(E) exactly as ``TokenEmbedding.forward`` does in fp32."""
    out: State = {k: np.ascontiguousarray(v, dtype=np.float32) for k, v in st.items()}
    w = st["encoder.patch_embed.proj.weight"]  # [E,3,4,8]
    out["encoder.patch_embed.proj.weight"] = np.ascontiguousarray(w.transpose(0, 2, 3, 1).reshape(EMBED, 96))
    out["encoder.pos_embed"] = np.ascontiguousarray(st["encoder.pos_embed"].reshape(N_PATCH, EMBED))
    out["pos_queries"] = np.ascontiguousarray(st["pos_queries"].reshape(MAX_LEN + 1, EMBED))
    out["text_embed.embedding.weight"] = (np.float32(math.sqrt(EMBED)) * st["text_embed.embedding.weight"]).astype(np.float32)
    return out


MAGIC = b"TTRW0001"


def write_ttrw(path: str, tensors: State) -> None:
    names = list(tensors.keys())
    table = bytearray()
    off = 0
    blobs = []
    for n in names:
        a = np.ascontiguousarray(tensors[n], dtype=np.float32)
        nb = a.nbytes
        nm = n.encode()
        table += struct.pack("<H", len(nm)) + nm + struct.pack("<BB", 0, a.ndim)
        table += struct.pack(f"<{a.ndim}I", *a.shape) + struct.pack("<QQ", off, nb)
        blobs.append(a.tobytes())
        off += (nb + 63) // 64 * 64
    head = MAGIC + struct.pack("<I", len(names)) + bytes(table)
    pad = (-len(head) - 8) % 64
    tmp = path + ".tmp"
    with open(tmp, "wb") as f:
        f.write(head)
        f.write(struct.pack("<Q", len(head) + 8 + pad))  # absolute data start
        f.write(b"\0" * pad)
        for bl in blobs:
            f.write(bl)
            f.write(b"\0" * ((-len(bl)) % 64))
    os.replace(tmp, path)


def read_ttrw(path: str) -> State:
    with open(path, "rb") as f:
        buf = f.read()
    assert buf[:8] == MAGIC, "not a .ttrw file"
    (n,) = struct.unpack_from("<I", buf, 8)
    p = 12
    ent = []
    for _ in range(n):
        (ln,) = struct.unpack_from("<H", buf, p); p += 2
        name = buf[p:p + ln].decode(); p += ln
        dt, nd = struct.unpack_from("<BB", buf, p); p += 2
        dims = struct.unpack_from(f"<{nd}I", buf, p); p += 4 * nd
        off, nb = struct.unpack_from("<QQ", buf, p); p += 16
        ent.append((name, dims, off, nb))
    (data0,) = struct.unpack_from("<Q", buf, p)
    return {name: np.frombuffer(buf, np.float32, nb // 4, data0 + off).reshape(dims).copy() for name, dims, off, nb in ent}


def export_craft(st: State, weights_dir: str) -> str:
    os.makedirs(weights_dir, exist_ok=True)
    path = os.path.join(weights_dir, CRAFT_FILE)
    write_ttrw(path, fold_craft(st))
    return path


def export_parseq(st: State, weights_dir: str) -> str:
    os.makedirs(weights_dir, exist_ok=True)
    path = os.path.join(weights_dir, PARSEQ_FILE)
    write_ttrw(path, pack_parseq(st))
    return path


def make_synthetic_weights(weights_dir: str, seed: int = 0, structured: bool = True) -> Tuple[State, State]:
    """Write ``craft.ttrw`` + ``parseq.ttrw`` for ``seed`` (idempotent) and return the raw state dicts."""
    c, p = synth_craft(seed, structured), synth_parseq(seed, structured=structured)
    export_craft(c, weights_dir)
    export_parseq(p, weights_dir)
    return c, p
