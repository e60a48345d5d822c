"""PredictionResult with the reference's surface (head_detector/detection_result.py:38-81).  ``heads`` is the
accelerated product.  ``get_pncc`` runs the HIP z-buffer rasteriser (csrc/raster.hip = the reference's Sim3DR kernel,
SURVEY.md 8(f) N3) and needs the reference's mesh assets (user-supplied, see ``head_detector_amd.pncc.MeshAssets``);
``save_meshes`` is pure file IO.  ``draw`` / ``get_aligned_heads`` are cv2 visualisation helpers outside the scope
(SURVEY.md 2 rows 5-8): they raise a clear error instead of silently doing something else."""
from __future__ import annotations

import os
from typing import List, Optional

import numpy as np

from .head_info import HeadMetadata


class PredictionResult:
    def __init__(self, original_image: np.ndarray, heads: List[HeadMetadata], faces: Optional[np.ndarray] = None, pncc_processor=None):
        self.original_image = original_image
        self.heads = heads
        self._faces = faces  # [F,3] 0-based triangle indices of the FLAME mesh
        self.pncc_processor = pncc_processor  # head_detector_amd.pncc.PNCCProcessor or None (no mesh assets supplied)

    def _unsupported(self, what: str):
        raise NotImplementedError(f"PredictionResult.{what} is a cv2/Sim3DR visualisation helper of the reference and is outside the accelerated forward path; "
                                  "use `.heads` (bbox, score, flame_params, vertices_3d, head_pose).")

    def draw(self, method: str = "full"):
        self._unsupported("draw")

    def get_pncc(self):
        """detection_result.py:58-59: PNCC image of all heads (uint8 [H,W,3]); like the reference it negates z of every
        head's ``vertices_3d`` in place."""
        if self.pncc_processor is None:
            raise FileNotFoundError("get_pncc needs the reference's mesh assets (full_faces.npy, v_template.npy, flame_indices/head_w_ears.npy): "
                                    "construct HeadDetector(..., assets_dir=<reference>/head_detector/assets)")
        return self.pncc_processor(self.original_image, self.heads)

    def get_aligned_heads(self):
        self._unsupported("get_aligned_heads")

    def save_meshes(self, save_folder: str):
        """One Wavefront OBJ per head, 'v x y z' / 'f a b c' with 1-based faces (detection_result.py:22-35,73-78)."""
        if self._faces is None:
            raise ValueError("no triangle list available (FLAME model without faces)")
        os.makedirs(save_folder, exist_ok=True)
        tri = np.asarray(self._faces).astype(np.int64) + 1
        for i, head in enumerate(self.heads):
            with open(os.path.join(save_folder, f"head_{i}.obj"), "w") as f:
                for v in head.vertices_3d:
                    f.write("v %.8f %.8f %.8f\n" % tuple(v))
                for t in tri:
                    f.write("f %d %d %d\n" % tuple(t))

    def __repr__(self):
        return f"PredictionResult(original_image={self.original_image.shape}, num heads={len(self.heads)})"
