"""``NoiseObservationWrapper`` of the reference (``tinycarlo/wrapper/observation.py``): blob noise on class-mask
observations.  Per class plane ``n_blobs`` filled circles, each erasing the plane inside the circle or -- with
probability 0.3 -- OR-ing in the circle-masked content of a random plane.

* single ``TinyCarloEnv``: the reference's code path -- the draws come from the global ``np.random`` generator in the
  reference's order (pinned by ``tests/golden/noise_draws.json``), the circles are painted on the host frame with a
  restatement of OpenCV's filled ``Circle`` (pixels unpinned: ``cv2`` is not available);
* batched ``TinyCarloVecEnv``: the engine applies the noise after every step in its own kernel (``tc_env_set_noise``)
  with blobs drawn on the device (same distributions, not numpy's stream)."""
from typing import List

import numpy as np

from .. import gym


def circle_half_widths(radius: int) -> List[int]:
    """Half width of each row offset 0..radius of cv2.circle(..., thickness=-1): the midpoint loop of OpenCV's
    ``Circle()`` (drawing.cpp) paints, per iteration, rows +-dy with half width dx and rows +-dx with half width dy."""
    hw = [0] * (radius + 1)
    err, dx, dy, plus, minus = 0, radius, 0, 1, (radius << 1) - 1
    while dx >= dy:
        hw[dy] = max(hw[dy], dx)
        hw[dx] = max(hw[dx], dy)
        dy += 1
        err += plus
        plus += 2
        mask = -1 if err > 0 else 0
        err -= minus & mask
        dx += mask
        minus -= mask & 2
    return hw


def circle_mask(shape, x: int, y: int, radius: int) -> np.ndarray:
    """boolean [H, W] mask of cv2.circle(img, (x, y), radius, ..., -1) for a centre inside the image"""
    H, W = shape
    m = np.zeros((H, W), dtype=bool)
    hw = circle_half_widths(radius)
    for t in range(-radius, radius + 1):
        row = y + t
        if 0 <= row < H:
            w = hw[abs(t)]
            m[row, max(x - w, 0):min(x + w, W - 1) + 1] = True
    return m


def draw_blobs(n_planes: int, height: int, width: int, n_blobs: int, max_radius: int) -> List[List[int]]:
    """The random draws of add_blob_noise_classes (observation.py:16-23) from the global numpy generator, in the
    reference's order: x, y, radius, the 0.3 / 0.7 choice and -- only on its True branch -- the source plane.
    Rows (x, y, radius, mode, src); src is -1 on the erase branch (nothing is drawn for it)."""
    out = []
    for _c in range(n_planes):
        for _ in range(n_blobs):
            x, y = np.random.randint(0, width), np.random.randint(0, height)
            radius = np.random.randint(1, max_radius)
            if np.random.choice([True, False], p=[0.3, 0.7]):
                out.append([int(x), int(y), int(radius), 1, int(np.random.randint(0, n_planes))])
            else:
                out.append([int(x), int(y), int(radius), 0, -1])
    return out


def apply_blobs(observation: np.ndarray, blobs, n_blobs: int) -> np.ndarray:
    """observation.py:21-26 for a [C, H, W] uint8 frame and the rows of ``draw_blobs``, in place"""
    C, H, W = observation.shape
    for k, (x, y, radius, mode, src) in enumerate(blobs):
        c = k // n_blobs
        m = circle_mask((H, W), x, y, radius)
        if mode:
            observation[c] = observation[c] | np.where(m, observation[src], 0).astype(np.uint8)
        else:
            observation[c][m] = 0
    return observation


class NoiseObservationWrapper(gym.Wrapper):
    def __init__(self, env, blob_max_radius=100, n_blobs=10, *, seed: int = 0):
        super().__init__(env)
        u = self.unwrapped
        u.wrapped = True  # observation.py:12
        self.max_radius = blob_max_radius
        self.n_blobs = n_blobs
        self.engine_side = hasattr(u, "set_noise")
        if self.engine_side:  # batched: the engine's own kernel, every step
            if u.observation_space_format == "classes":
                u.set_noise(n_blobs, blob_max_radius, seed)

    def add_blob_noise_classes(self, observation: np.ndarray) -> np.ndarray:
        C, H, W = observation.shape
        return apply_blobs(observation, draw_blobs(C, H, W, self.n_blobs, self.max_radius), self.n_blobs)

    def step(self, action):
        observation, reward, terminated, truncated, info = self.env.step(action)
        u = self.unwrapped
        if not self.engine_side and u.observation_space_format == "classes" and not u.no_observation:
            observation = self.add_blob_noise_classes(observation)
        return observation, reward, terminated, truncated, info
