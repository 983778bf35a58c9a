"""CPU oracle — the whole of ``image_to_data`` (tuatara.cpp:314-512) on CPU.

TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (see models.py / post.c headers): the
model arithmetic is the upstream architecture in fp32 eager PyTorch, the OpenCV
steps are the C restatement in post.c, glued together here in the reference's order.
"""
from __future__ import annotations

from typing import Dict, List

import numpy as np
import torch

from . import post
from .models import CRAFT, PARSeq


def load_models(craft_state: Dict[str, np.ndarray], parseq_state: Dict[str, np.ndarray]):
    c, p = CRAFT().eval(), PARSeq().eval()
    c.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in craft_state.items()}, strict=False)
    p.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in parseq_state.items()}, strict=True)
    return c, p


@torch.no_grad()
def craft_heatmap(craft: CRAFT, canvas_u8: np.ndarray) -> np.ndarray:
    """tuatara.cpp:363-394: u8 [H,W,3] -> f32 [1,3,H,W] / 255 -> forward -> element 0 [H/2,W/2,2]."""
    x = torch.from_numpy(np.ascontiguousarray(canvas_u8)).unsqueeze(0).permute(0, 3, 1, 2).to(torch.float32).div(255.0)
    y, _ = craft(x)
    return y[0].contiguous().numpy()


@torch.no_grad()
def parseq_logits(parseq: PARSeq, crops_u8: np.ndarray, batch: int = 64) -> np.ndarray:
    """tuatara.cpp:443-446 + :307: u8 [N,32,128,3] -> f32 [N,3,32,128] / 255 -> logits [N,26,95].
    (The reference chunks by 4 over 6 threads, :450-485; results are batch-invariant.)"""
    outs = []
    for i in range(0, len(crops_u8), batch):
        x = torch.from_numpy(np.ascontiguousarray(crops_u8[i:i + batch])).permute(0, 3, 1, 2).to(torch.float32).div(255.0)
        outs.append(parseq(x).numpy())
    return np.concatenate(outs, 0) if outs else np.zeros((0, 26, 95), np.float32)


def detect(craft: CRAFT, image_u8: np.ndarray, canvas_size: int = 1024, mag_ratio: float = 1.0,
           text_threshold: float = 0.7, link_threshold: float = 0.4, low_text: float = 0.4):
    """tuatara.cpp:349-406.  image_u8 is the caller's array (any channel order).
    Returns dict with the swapped image, canvas, heatmap, det rects (heatmap units), boxes (image units)."""
    swapped = np.ascontiguousarray(image_u8[:, :, ::-1])                      # :349 cvtColor BGR2RGB in place
    canvas, ratio = post.resize_aspect_ratio(swapped, canvas_size, mag_ratio)  # :358
    ratio_h = ratio_w = np.float32(1) / np.float32(ratio)                      # :360-361
    heat = craft_heatmap(craft, canvas)                                        # :363-394
    det, labels, tn = post.get_detected_boxes(heat[:, :, 0], heat[:, :, 1], text_threshold, link_threshold, low_text)  # :400
    boxes = post.adjust_result_coordinates(det, float(ratio_w), float(ratio_h))  # :406
    return dict(swapped=swapped, canvas=canvas, ratio=ratio, heat=heat, det=det, boxes=boxes, labels=labels, textnorm=tn)


def image_to_data(craft: CRAFT, parseq: PARSeq, image_u8: np.ndarray, clamp: bool = True, debug: bool = False, **kw):
    d = detect(craft, image_u8, **kw)
    crops, keep = [], []
    for i, b in enumerate(d["boxes"]):                  # :408-418 + :436-448
        c = post.crop_resize(d["swapped"], b, clamp)
        if c is None:
            if not clamp:
                raise RuntimeError("crop leaves the image (the reference throws cv::Exception here, tuatara.cpp:416)")
            continue
        crops.append(c)
        keep.append(i)
    crops = np.stack(crops) if crops else np.zeros((0, 32, 128, 3), np.uint8)
    logits = parseq_logits(parseq, crops)               # :450-485
    texts, _ = post.decode_logits(logits)               # :486-505
    out = [{"text": t, "bbox": post.tesseract_bbox(d["boxes"][i])} for t, i in zip(texts, keep)]  # :511
    d.update(crops=crops, logits=logits, result=out)
    return d if debug else out
