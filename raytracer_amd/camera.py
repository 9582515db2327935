"""Camera: host-side mirror of the reference camera (reference src/camera/camera.py).

The kernel consumes two things from the camera: the 4x4 inverse
view-projection matrix (16 f32 in memory order, handed over untransposed,
reference src/renderer/renderer.py:186-191) and the eye position
(renderer.py:193-198).  The reference builds the matrix with pyrr 0.10.3
(camera.py:60,81-91,153-192), which is not importable here; the pyrr
arithmetic is restated below in float64 numpy from pyrr's published
definitions:

* ``Matrix44.perspective_projection(fovy, aspect, near, far)`` ->
  ``create_perspective_projection_from_bounds`` (row-major, row-vector layout)
* object product ``A * B`` == ``numpy.dot(B, A)``
* ``~M`` == ``numpy.linalg.inv(M)``

Parity of this restatement with pyrr itself is unpinned: the known-answer
uniform words in tests/test_scene_pipeline.py (test_camera_known_answer) come
from an independent numpy evaluation of the same published formulas, not from
pyrr.  Kernel parity never depends on it, because both sides consume the same
16 floats.
"""

from __future__ import annotations

import math

import numpy as np


def perspective_projection(fovy: float, aspect: float, near: float, far: float) -> np.ndarray:
    """pyrr ``matrix44.create_perspective_projection`` (float64, row-vector layout)."""
    ymax = near * math.tan(fovy * math.pi / 360.0)
    xmax = ymax * aspect
    left, right, bottom, top = -xmax, xmax, -ymax, ymax
    a = (right + left) / (right - left)
    b = (top + bottom) / (top - bottom)
    c = -(far + near) / (far - near)
    d = -2.0 * far * near / (far - near)
    e = 2.0 * near / (right - left)
    f = 2.0 * near / (top - bottom)
    return np.array(
        ((e, 0.0, 0.0, 0.0), (0.0, f, 0.0, 0.0), (a, b, c, -1.0), (0.0, 0.0, d, 0.0)),
        dtype=np.float64,
    )


def _normalise(v: np.ndarray) -> np.ndarray:
    return v / np.sqrt(np.sum(v * v))


class Camera:
    """Same attributes and movement methods as the reference Camera (camera.py:37-151).

    Unlike the reference this is not a process-wide singleton that the
    renderer fetches implicitly: ``Renderer`` owns one (``renderer.camera``),
    and ``Camera.get_instance()`` returns a shared default for callers that
    use the reference idiom.
    """

    _instance: "Camera | None" = None

    @staticmethod
    def get_instance() -> "Camera":
        if Camera._instance is None:
            Camera._instance = Camera()
        return Camera._instance

    def __init__(self) -> None:
        # defaults: camera.py:51-60
        self.position = np.array([8.0, 1.5, 0.0])
        self.world_up = np.array([0.0, 1.0, 0.0])
        self.up = np.array([0.0, 0.0, 0.0])
        self.front = np.array([0.0, 0.0, -1.0])
        self.right = np.array([0.0, 0.0, 0.0])
        self.yaw = 180.0
        self.pitch = 0
        # the aspect ratio is fixed at construction and never follows a resize (camera.py:60)
        self.projection_matrix = perspective_projection(45, 128 / 72, 0.1, 1)
        self._update_camera_vectors()
        self.sensitivity = 0.1
        self.changed = False

    # --- movement: camera.py:93-141 -------------------------------------------------
    def forward(self, amount: float) -> None:
        self.position += self.front * amount
        self._update_camera_vectors()

    def strafe(self, amount: float) -> None:
        self.position += self.right * amount
        self._update_camera_vectors()

    def rise(self, amount: float) -> None:
        self.position += self.world_up * amount
        self._update_camera_vectors()

    def turn(self, x: float, y: float) -> None:
        self.yaw += x * self.sensitivity
        self.pitch += y * self.sensitivity
        self.pitch = min(self.pitch, 89.0)
        self.pitch = max(self.pitch, -89.0)
        self._update_camera_vectors()

    def _update_camera_vectors(self) -> None:
        # camera.py:143-151
        fx = math.cos(math.radians(self.yaw)) * math.cos(math.radians(self.pitch))
        fy = math.sin(math.radians(self.pitch))
        fz = math.sin(math.radians(self.yaw)) * math.cos(math.radians(self.pitch))
        front = np.array([fx, fy, fz])
        self.front = front / np.linalg.norm(front)
        right = np.cross(self.front, self.world_up)
        self.right = right / np.linalg.norm(right)
        self.up = np.cross(self.right, self.front)
        self.changed = True

    # --- matrices: camera.py:68-91,153-192 ------------------------------------------
    def look_at(self, position, target, world_up) -> np.ndarray:
        p = np.array([position[0], position[1], position[2]], dtype=np.float64)
        t = np.array([target[0], target[1], target[2]], dtype=np.float64)
        w = np.array([world_up[0], world_up[1], world_up[2]], dtype=np.float64)
        zaxis = _normalise(p - t)
        xaxis = _normalise(np.cross(_normalise(w), zaxis))
        yaxis = np.cross(zaxis, xaxis)
        translation = np.identity(4)
        translation[3][0] = -p[0]
        translation[3][1] = -p[1]
        translation[3][2] = -p[2]
        rotation = np.identity(4)
        for i in range(3):
            rotation[i][0] = xaxis[i]
            rotation[i][1] = yaxis[i]
            rotation[i][2] = zaxis[i]
        # pyrr object product: translation * rotation == dot(rotation, translation)
        return np.dot(rotation, translation)

    def get_view_matrix(self) -> np.ndarray:
        self.view_matrix = self.look_at(self.position, self.position + self.front, np.array([0, 1, 0]))
        return self.view_matrix

    def get_inv_view_proj_matrix(self) -> np.ndarray:
        # projection * view == dot(view, projection); ~m == inverse
        view_projection = np.dot(self.get_view_matrix(), self.projection_matrix)
        return np.linalg.inv(view_projection)

    def uniforms_f32(self) -> tuple[np.ndarray, np.ndarray]:
        """(inv_vp[16] in memory order, eye[3]) as float32, as glUniform* would receive them."""
        m = np.ascontiguousarray(self.get_inv_view_proj_matrix(), dtype=np.float64).astype(np.float32)
        eye = np.asarray(self.position, dtype=np.float64).astype(np.float32)
        return m.reshape(16), eye
