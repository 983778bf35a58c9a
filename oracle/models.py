"""CPU oracle — model half.  TEST INFRASTRUCTURE ONLY (never imported by the product path).

PARITY UNPINNED: the reference (jackvial/tuatara) executes two *un-vendored*
TorchScript archives (``craft_traced_torchscript_model.pt`` loaded at
tuatara.cpp:333-336 and ``parseq_torchscript.bin`` loaded at tuatara.cpp:423-428,
both fetched by setup.sh:6 from huggingface.co/jackvial/tuatara-ocr-craft-and-parseq,
revision unpinned).  Neither archive nor any golden vector for them exists under
/root/reference, so this file restates the *published* upstream architectures
those archives were exported from:

  * CRAFT  — clovaai/CRAFT-pytorch (craft.py, basenet/vgg16_bn.py): VGG16-BN trunk
    cut into 5 slices, 4 U-Net ``double_conv`` up-blocks, 5-conv ``conv_cls`` head.
    Output tuple ``(y.permute(0,2,3,1), feature)``; the reference consumes
    element 0 only (tuatara.cpp:376-394).
  * PARSeq — baudm/parseq (strhub/models/parseq/{system,modules}.py) "base"
    config: ViT encoder (embed 384, depth 12, 6 heads, 4x8 patches on 32x128),
    one two-stream decoder layer (12 heads, FFN 1536), ``decode_ar=True``,
    ``refine_iters=1``, ``max_label_length=25``.  Called at tuatara.cpp:307.

Parameter names follow upstream so a real ``state_dict`` loads unchanged
(tools/convert_weights.py).  Parameter counts (20,770,466 / 23,832,671) match
the publicly quoted 20.8 M / 23.8 M — the only cross-check available.

Everything here is plain fp32 eager PyTorch on CPU.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn
import torch.nn.functional as F


# --------------------------------------------------------------------------- CRAFT
# torchvision vgg16_bn.features layout: index -> layer.  'M' = MaxPool2d(2,2).
_VGG16_CFG = [64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512, "M"]


def _vgg16_bn_features() -> nn.Sequential:
    layers: list[nn.Module] = []
    cin = 3
    for v in _VGG16_CFG:
        if v == "M":
            layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
        else:
            layers += [nn.Conv2d(cin, v, kernel_size=3, padding=1), nn.BatchNorm2d(v), nn.ReLU(inplace=False)]
            cin = v
    return nn.Sequential(*layers)


class VGG16BN(nn.Module):
    """Upstream ``basenet/vgg16_bn.py``: slices cut *between BN and ReLU*, so the
    four skip tensors are pre-ReLU BatchNorm outputs."""

    def __init__(self) -> None:
        super().__init__()
        feats = _vgg16_bn_features()
        self.slice1 = nn.Sequential()
        self.slice2 = nn.Sequential()
        self.slice3 = nn.Sequential()
        self.slice4 = nn.Sequential()
        self.slice5 = nn.Sequential()
        for x in range(12):  # conv2_2 (+BN)
            self.slice1.add_module(str(x), feats[x])
        for x in range(12, 19):  # conv3_2 (+BN); upstream names keep the global index
            self.slice2.add_module(str(x), feats[x])
        for x in range(19, 29):  # conv4_2 (+BN)
            self.slice3.add_module(str(x), feats[x])
        for x in range(29, 39):  # conv5_2 (+BN)
            self.slice4.add_module(str(x), feats[x])
        self.slice5 = nn.Sequential(
            nn.MaxPool2d(kernel_size=3, stride=1, padding=1),
            nn.Conv2d(512, 1024, kernel_size=3, padding=6, dilation=6),
            nn.Conv2d(1024, 1024, kernel_size=1),
        )

    def forward(self, x):
        h = self.slice1(x)
        relu2_2 = h
        h = self.slice2(h)
        relu3_2 = h
        h = self.slice3(h)
        relu4_3 = h
        h = self.slice4(h)
        relu5_3 = h
        h = self.slice5(h)
        fc7 = h
        return fc7, relu5_3, relu4_3, relu3_2, relu2_2


class DoubleConv(nn.Module):
    def __init__(self, in_ch: int, mid_ch: int, out_ch: int) -> None:
        super().__init__()
        self.conv = nn.Sequential(
            nn.Conv2d(in_ch + mid_ch, mid_ch, kernel_size=1),
            nn.BatchNorm2d(mid_ch),
            nn.ReLU(inplace=False),
            nn.Conv2d(mid_ch, out_ch, kernel_size=3, padding=1),
            nn.BatchNorm2d(out_ch),
            nn.ReLU(inplace=False),
        )

    def forward(self, x):
        return self.conv(x)


class CRAFT(nn.Module):
    def __init__(self) -> None:
        super().__init__()
        self.basenet = VGG16BN()
        self.upconv1 = DoubleConv(1024, 512, 256)
        self.upconv2 = DoubleConv(512, 256, 128)
        self.upconv3 = DoubleConv(256, 128, 64)
        self.upconv4 = DoubleConv(128, 64, 32)
        self.conv_cls = nn.Sequential(
            nn.Conv2d(32, 32, kernel_size=3, padding=1), nn.ReLU(inplace=False),
            nn.Conv2d(32, 32, kernel_size=3, padding=1), nn.ReLU(inplace=False),
            nn.Conv2d(32, 16, kernel_size=3, padding=1), nn.ReLU(inplace=False),
            nn.Conv2d(16, 16, kernel_size=1), nn.ReLU(inplace=False),
            nn.Conv2d(16, 2, kernel_size=1),
        )

    def forward(self, x):
        sources = self.basenet(x)
        y = torch.cat([sources[0], sources[1]], dim=1)
        y = self.upconv1(y)
        y = F.interpolate(y, size=sources[2].shape[2:], mode="bilinear", align_corners=False)
        y = torch.cat([y, sources[2]], dim=1)
        y = self.upconv2(y)
        y = F.interpolate(y, size=sources[3].shape[2:], mode="bilinear", align_corners=False)
        y = torch.cat([y, sources[3]], dim=1)
        y = self.upconv3(y)
        y = F.interpolate(y, size=sources[4].shape[2:], mode="bilinear", align_corners=False)
        y = torch.cat([y, sources[4]], dim=1)
        feature = self.upconv4(y)
        y = self.conv_cls(feature)
        return y.permute(0, 2, 3, 1), feature


# --------------------------------------------------------------------------- PARSeq
class _Attention(nn.Module):  # timm.models.vision_transformer.Attention
    def __init__(self, dim: int, num_heads: int) -> None:
        super().__init__()
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=True)
        self.proj = nn.Linear(dim, dim)

    def forward(self, x):
        B, N, C = x.shape
        qkv = self.qkv(x).reshape(B, N, 3, self.num_heads, C // self.num_heads).permute(2, 0, 3, 1, 4)
        q, k, v = qkv.unbind(0)
        attn = (q * self.scale) @ k.transpose(-2, -1)
        attn = attn.softmax(dim=-1)
        x = (attn @ v).transpose(1, 2).reshape(B, N, C)
        return self.proj(x)


class _Mlp(nn.Module):
    def __init__(self, dim: int, hidden: int) -> None:
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.act = nn.GELU()
        self.fc2 = nn.Linear(hidden, dim)

    def forward(self, x):
        return self.fc2(self.act(self.fc1(x)))


class _Block(nn.Module):
    def __init__(self, dim: int, num_heads: int, mlp_ratio: float) -> None:
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=1e-6)
        self.attn = _Attention(dim, num_heads)
        self.norm2 = nn.LayerNorm(dim, eps=1e-6)
        self.mlp = _Mlp(dim, int(dim * mlp_ratio))

    def forward(self, x):
        x = x + self.attn(self.norm1(x))
        x = x + self.mlp(self.norm2(x))
        return x


class _PatchEmbed(nn.Module):
    def __init__(self, patch=(4, 8), in_chans=3, embed_dim=384) -> None:
        super().__init__()
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch, stride=patch)

    def forward(self, x):
        return self.proj(x).flatten(2).transpose(1, 2)


class Encoder(nn.Module):
    """timm VisionTransformer(num_classes=0, global_pool='', class_token=False)."""

    def __init__(self, img_size=(32, 128), patch=(4, 8), embed_dim=384, depth=12, num_heads=6, mlp_ratio=4.0):
        super().__init__()
        self.patch_embed = _PatchEmbed(patch, 3, embed_dim)
        n = (img_size[0] // patch[0]) * (img_size[1] // patch[1])
        self.pos_embed = nn.Parameter(torch.zeros(1, n, embed_dim))
        self.blocks = nn.Sequential(*[_Block(embed_dim, num_heads, mlp_ratio) for _ in range(depth)])
        self.norm = nn.LayerNorm(embed_dim, eps=1e-6)

    def forward(self, x):
        x = self.patch_embed(x) + self.pos_embed
        x = self.blocks(x)
        return self.norm(x)


class DecoderLayer(nn.Module):
    """baudm/parseq modules.DecoderLayer (two-stream; with one layer only the
    query stream is ever updated)."""

    def __init__(self, d_model: int, nhead: int, dim_feedforward: int, layer_norm_eps: float = 1e-5):
        super().__init__()
        self.self_attn = nn.MultiheadAttention(d_model, nhead, batch_first=True)
        self.cross_attn = nn.MultiheadAttention(d_model, nhead, batch_first=True)
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.norm1 = nn.LayerNorm(d_model, eps=layer_norm_eps)
        self.norm2 = nn.LayerNorm(d_model, eps=layer_norm_eps)
        self.norm_q = nn.LayerNorm(d_model, eps=layer_norm_eps)
        self.norm_c = nn.LayerNorm(d_model, eps=layer_norm_eps)

    def forward_stream(self, tgt, tgt_norm, tgt_kv, memory, tgt_mask, tgt_key_padding_mask):
        tgt2, _ = self.self_attn(tgt_norm, tgt_kv, tgt_kv, attn_mask=tgt_mask,
                                 key_padding_mask=tgt_key_padding_mask, need_weights=False)
        tgt = tgt + tgt2
        tgt2, _ = self.cross_attn(self.norm1(tgt), memory, memory, need_weights=False)
        tgt = tgt + tgt2
        tgt2 = self.linear2(F.gelu(self.linear1(self.norm2(tgt))))
        return tgt + tgt2

    def forward(self, query, content, memory, query_mask=None, content_mask=None,
                content_key_padding_mask=None, update_content=True):
        query_norm = self.norm_q(query)
        content_norm = self.norm_c(content)
        query = self.forward_stream(query, query_norm, content_norm, memory, query_mask, content_key_padding_mask)
        if update_content:
            content = self.forward_stream(content, content_norm, content_norm, memory, content_mask,
                                          content_key_padding_mask)
        return query, content


class Decoder(nn.Module):
    def __init__(self, d_model: int, nhead: int, dim_feedforward: int, num_layers: int = 1):
        super().__init__()
        self.layers = nn.ModuleList([DecoderLayer(d_model, nhead, dim_feedforward) for _ in range(num_layers)])
        self.norm = nn.LayerNorm(d_model)

    def forward(self, query, content, memory, query_mask=None, content_mask=None, content_key_padding_mask=None):
        for i, mod in enumerate(self.layers):
            last = i == len(self.layers) - 1
            query, content = mod(query, content, memory, query_mask, content_mask, content_key_padding_mask,
                                 update_content=not last)
        return self.norm(query)


class TokenEmbedding(nn.Module):
    def __init__(self, charset_size: int, embed_dim: int):
        super().__init__()
        self.embedding = nn.Embedding(charset_size, embed_dim)
        self.embed_dim = embed_dim

    def forward(self, tokens):
        return math.sqrt(self.embed_dim) * self.embedding(tokens)


class PARSeq(nn.Module):
    """Inference-only restatement of ``PARSeq.forward`` (system.py), ``decode_ar=True``,
    ``refine_iters=1``.  ``early_exit`` reproduces the upstream data-dependent
    break; the final (refined) logits are invariant to it."""

    EOS, BOS, PAD = 0, 95, 96  # upstream Tokenizer: [E] + 94 chars + [B] + [P]
    NUM_TOKENS = 97

    def __init__(self, embed_dim=384, max_label_length=25):
        super().__init__()
        self.max_label_length = max_label_length
        self.encoder = Encoder(embed_dim=embed_dim)
        self.decoder = Decoder(embed_dim, 12, embed_dim * 4, num_layers=1)
        self.head = nn.Linear(embed_dim, self.NUM_TOKENS - 2)
        self.text_embed = TokenEmbedding(self.NUM_TOKENS, embed_dim)
        self.pos_queries = nn.Parameter(torch.zeros(1, max_label_length + 1, embed_dim))

    def encode(self, img):
        return self.encoder(img)

    def decode(self, tgt, memory, tgt_mask=None, tgt_padding_mask=None, tgt_query=None, tgt_query_mask=None):
        N, L = tgt.shape
        null_ctx = self.text_embed(tgt[:, :1])
        tgt_emb = self.pos_queries[:, : L - 1] + self.text_embed(tgt[:, 1:])
        tgt_emb = torch.cat([null_ctx, tgt_emb], dim=1)
        if tgt_query is None:
            tgt_query = self.pos_queries[:, :L].expand(N, -1, -1)
        return self.decoder(tgt_query, tgt_emb, memory, tgt_query_mask, tgt_mask, tgt_padding_mask)

    @torch.no_grad()
    def forward(self, images, early_exit: bool = False, return_ar: bool = False):
        bs = images.shape[0]
        num_steps = self.max_label_length + 1
        memory = self.encode(images)
        pos_queries = self.pos_queries[:, :num_steps].expand(bs, -1, -1)
        tgt_mask = query_mask = torch.triu(torch.full((num_steps, num_steps), float("-inf")), 1)

        tgt_in = torch.full((bs, num_steps), self.PAD, dtype=torch.long)
        tgt_in[:, 0] = self.BOS
        logits = []
        for i in range(num_steps):
            j = i + 1
            tgt_out = self.decode(tgt_in[:, :j], memory, tgt_mask[:j, :j], tgt_query=pos_queries[:, i:j],
                                  tgt_query_mask=query_mask[i:j, :j])
            p_i = self.head(tgt_out)
            logits.append(p_i)
            if j < num_steps:
                tgt_in[:, j] = p_i.squeeze(1).argmax(-1)
                if early_exit and (tgt_in == self.EOS).any(dim=-1).all():
                    break
        logits = torch.cat(logits, dim=1)
        ar_logits = logits

        query_mask = query_mask.clone()
        query_mask[torch.triu(torch.ones(num_steps, num_steps, dtype=torch.bool), 2)] = 0
        bos = torch.full((bs, 1), self.BOS, dtype=torch.long)
        tgt_in = torch.cat([bos, logits[:, :-1].argmax(-1)], dim=1)
        tgt_padding_mask = (tgt_in == self.EOS).int().cumsum(-1) > 0
        tgt_out = self.decode(tgt_in, memory, tgt_mask[: tgt_in.shape[1], : tgt_in.shape[1]], tgt_padding_mask,
                              tgt_query=pos_queries, tgt_query_mask=query_mask[:, : tgt_in.shape[1]])
        logits = self.head(tgt_out)
        if return_ar:
            return logits, ar_logits
        return logits
