"""Result / parameter containers with the reference's names and field meaning
(head_detector/head_info.py:9-109): Bbox, RPY, FLAME_CONSTS, HeadMetadata, FlameParams."""
from __future__ import annotations

from collections import namedtuple
from dataclasses import dataclass
from typing import Dict, Optional

import numpy as np
import torch
from torch import Tensor

Bbox = namedtuple("Bbox", ["x", "y", "w", "h"])
RPY = namedtuple("RPY", ["roll", "pitch", "yaw"])

# widths of the 413-vector (head_detector/head_info.py:12-21)
FLAME_CONSTS: Dict[str, int] = {"shape": 300, "expression": 100, "rotation": 6, "jaw": 3, "eyeballs": 0, "neck": 0, "translation": 3, "scale": 1}

# The reference READS the vector as [shape, expression, jaw, rotation, eyeballs, neck, translation, scale]
# (from_3dmm, head_info.py:54-78) but WRITES it as [shape, expression, rotation, jaw, ...] (to_3dmm_tensor,
# head_info.py:95-106).  Both orders are part of the contract (SURVEY.md 8a row a6').
_READ_ORDER = ("shape", "expression", "jaw", "rotation", "eyeballs", "neck", "translation", "scale")
_WRITE_ORDER = ("shape", "expression", "rotation", "jaw", "eyeballs", "neck", "translation", "scale")


@dataclass
class HeadMetadata:
    bbox: Bbox
    score: float
    flame_params: object
    vertices_3d: np.ndarray
    head_pose: RPY


@dataclass
class FlameParams:
    shape: Tensor
    expression: Tensor
    rotation: Tensor
    translation: Tensor
    scale: Tensor
    jaw: Tensor
    eyeballs: Tensor
    neck: Tensor

    @classmethod
    def from_3dmm(cls, tensor_3dmm: Tensor, constants: Optional[Dict[str, int]] = None, zero_expr: bool = False) -> "FlameParams":
        """tensor_3dmm: [B, num_params, ...] -> views into it, sliced in the reference's READ order."""
        widths = FLAME_CONSTS if constants is None else constants
        total = sum(widths.values())
        if tensor_3dmm.size(1) != total:
            raise ValueError(f"Invalid number of parameters. Expected: {total}. Got: {tensor_3dmm.size(1)}.")
        parts, start = {}, 0
        for key in _READ_ORDER:
            parts[key] = tensor_3dmm[:, start : start + widths[key]]
            start += widths[key]
        if zero_expr:
            parts["expression"] = torch.zeros_like(parts["expression"])
        return cls(**parts)

    def to_3dmm_tensor(self) -> Tensor:
        """[B, C, ...] in the reference's WRITE order."""
        return torch.cat([getattr(self, key) for key in _WRITE_ORDER], dim=1)
