"""Pinhole camera parameters (host side).

Mirrors the init-time part of the reference's ``tinycarlo/camera.py``: config defaults
(``camera.py:16-21``), extrinsic matrix (``145-156``), intrinsic matrix (``158-178``) and
``update_params()`` (``48-50``).  The per-step work of ``Camera.capture_frame`` (``52-110``) --
transform, near-plane / max-range fix-ups, projection, visibility, rasterise -- runs in the HIP
kernels; this class only owns E (3x4) and K (3x3) and pushes them to the device when they change.
"""
from __future__ import annotations

import math
from typing import Any, Callable, Dict, List, Optional

import numpy as np


def rodrigues(rvec) -> np.ndarray:
    """Rotation vector -> rotation matrix, the closed form ``cv2.Rodrigues`` evaluates:
    ``R = cos(t) I + (1-cos(t)) k k^T + sin(t) [k]x`` with ``t = |rvec|``, ``k = rvec/t``."""
    r = np.asarray(rvec, dtype=np.float64).reshape(3)
    theta = math.sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2])
    if theta < np.finfo(np.float64).eps:
        return np.eye(3)
    c, s = math.cos(theta), math.sin(theta)
    k = r * (1.0 / theta)
    kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return c * np.eye(3) + (1.0 - c) * np.outer(k, k) + s * kx


class Camera:
    def __init__(self, camera_config: Dict[str, Any], on_update: Optional[Callable[["Camera"], None]] = None):
        self.resolution: List[int] = list(camera_config.get("resolution", [128, 160]))  # [H, W]
        self.position = list(camera_config.get("position", [0, 0, 0]))
        self.orientation = list(camera_config.get("orientation", [0, 0, 0]))
        self.fov = camera_config.get("fov", 90)
        self.max_range = camera_config.get("max_range", None)
        self.line_thickness: int = int(camera_config.get("line_thickness", 1))
        if not self.max_range:
            # the reference's capture_frame evaluates `-self.max_range` (camera.py:82,85) and breaks
            # on None; the HIP path requires a finite positive range.
            raise ValueError("camera.max_range must be a positive number (the reference fails on None at camera.py:82)")
        if self.line_thickness < 1:
            raise ValueError("camera.line_thickness must be >= 1")
        self._on_update = on_update
        self.E = self._extrinsic()
        self.K = self._intrinsic()

    def update_params(self) -> None:
        """Re-derive E and K after ``orientation``/``fov``/``position`` were changed (camera.py:48-50)."""
        self.E = self._extrinsic()
        self.K = self._intrinsic()
        if self._on_update is not None:
            self._on_update(self)

    def _extrinsic(self) -> np.ndarray:  # camera.py:145-156
        angles = np.radians(np.asarray(self.orientation) + np.array([-90, 0, 90]))
        r_pr = rodrigues(np.array([1, 1, 0]) * angles)
        r_y = rodrigues(np.array([0, 0, 1]) * angles)
        t = np.column_stack((np.eye(3), -np.array(self.position)))
        return np.ascontiguousarray(r_pr @ r_y @ t, dtype=np.float64)

    def _intrinsic(self) -> np.ndarray:  # camera.py:158-178
        fov = np.radians(self.fov)
        fx = self.resolution[1] / (2 * np.tan(fov / 2))
        fy = self.resolution[0] / (2 * np.tan(fov / 2))
        cx = self.resolution[1] / 2
        cy = self.resolution[0] / 2
        return np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1]], dtype=np.float64)
