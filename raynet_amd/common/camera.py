"""Finite pinhole camera -- the input contract of the hot path.

Same quantities as raynet/common/camera.py:4-65: P = K [R t] (3x4),
P_pinv = pinv(P) (4x3), center = [-R^-1 t; 1] (4x1 float32).
"""
import numpy as np


class Camera(object):
    def __init__(self, K, R, t):
        assert K.shape == (3, 3)
        assert R.shape == (3, 3)
        assert t.shape == (3, 1)
        self._K, self._R, self._t = K, R, t
        self._P = self._P_pinv = self._center = None

    K = property(lambda self: self._K)
    R = property(lambda self: self._R)
    t = property(lambda self: self._t)

    @property
    def center(self):
        if self._center is None:
            self._center = np.vstack(
                [(-np.linalg.inv(self.R)).dot(self.t), [1]]).astype(np.float32)
        return self._center

    @property
    def P(self):
        if self._P is None:
            self._P = self._K.dot(np.hstack([self._R, self._t]))
        return self._P

    @property
    def P_pinv(self):
        if self._P_pinv is None:
            self._P_pinv = np.linalg.pinv(self.P)
        return self._P_pinv

    @classmethod
    def look_at(cls, position, target, focal, height, width, up=(0.0, 0.0, 1.0)):
        """Camera at `position` looking at `target` (synthetic scenes)."""
        pos = np.asarray(position, np.float64)
        z = np.asarray(target, np.float64) - pos
        z /= np.linalg.norm(z)
        x = np.cross(z, np.asarray(up, np.float64))
        x /= np.linalg.norm(x)
        y = np.cross(z, x)
        R = np.stack([x, y, z])
        t = -R.dot(pos).reshape(3, 1)
        K = np.array([[focal, 0, width / 2.0], [0, focal, height / 2.0], [0, 0, 1.0]])
        return cls(K, R, t)
