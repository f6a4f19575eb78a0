"""SO(3) / SE(3) elements for the host side of the MI355X backend.

Host-side mirror of the reference's `So3` / `Se3` public surface
(utils/lie_algebra/special_orthogonal_group.py:13-244, special_euclidean_group.py:13-128):
quaternion-backed rotation (w, x, y, z; float32 storage), `exp()` = matrix, `log()` = tangent
vector, `hat()`, `inverse()`, `*`, `identity()`, `copy()`, `from_se3()`; xi = (upsilon, omega).
Numerics follow the reference where it matters for pose chaining: float32 quaternion/matrix
storage, `_LIE_EPSILON = 1e-6` small-angle switch, `wrap_angle` applied as the reference does.
Only poses cross into / out of the HIP kernels, so this stays NumPy (a handful of flops per frame).
"""
from __future__ import annotations

import math

import numpy as np

_LIE_EPSILON = 1e-6      # utils/lie_algebra/base_special_group.py:8


def wrap_angle(angle):
    """[-pi, pi) wrap (utils/lie_algebra/common.py:32-46)."""
    return (angle + np.pi) % (2 * np.pi) - np.pi


def is_rotation_matrix(m: np.ndarray) -> bool:
    """Orthogonality + unit determinant, atol 1e-3 (utils/lie_algebra/common.py:4-29)."""
    assert m.shape == (3, 3), f"Expected matrix shape to be (3,3), got {m.shape} instead"
    eye = np.eye(3)
    ortho = np.allclose(m @ m.T, eye, atol=1e-3) and np.allclose(m.T @ m, eye, atol=1e-3)
    return bool(ortho and np.isclose(np.linalg.det(m), 1.0, atol=1e-3))


def quat_mult(qa: np.ndarray, qb: np.ndarray) -> np.ndarray:
    """Hamilton product of two (4,1) wxyz quaternions, float32 result (common.py:51-73)."""
    assert qa.shape == qb.shape == (4, 1), "Both quaternions should have shape (4, 1) got '{}' and '{}".format(
        qa.shape, qb.shape)
    a0, av = qa[0, 0], qa[1:, 0]
    b0, bv = qb[0, 0], qb[1:, 0]
    out = np.empty((4, 1), dtype=np.float32)
    out[0, 0] = a0 * b0 - np.dot(av, bv)
    out[1:, 0] = a0 * bv + b0 * av + np.cross(av, bv)
    return out


def _skew(v) -> np.ndarray:
    x, y, z = (float(t) for t in np.asarray(v).reshape(3))
    return np.array([[0.0, -z, y], [z, 0.0, -x], [-y, x, 0.0]], dtype=np.float32)


class So3:
    """Rotation stored as a unit quaternion.  Construct from a (4,1) quaternion, a (3,1) rotation
    vector or a (3,3) rotation matrix, like the reference class."""

    _QUAT_REPR_VALUES = ["wxyz", "xyzw"]

    def __init__(self, rot: np.ndarray, quat_repr: str = "wxyz"):
        assert isinstance(rot, np.ndarray), "Expected 'rot' to be a numpy array, got {} instead".format(type(rot))
        assert quat_repr in self._QUAT_REPR_VALUES, "Invalid 'quat_repr': '{}'. Options are: {}".format(
            quat_repr, self._QUAT_REPR_VALUES)
        self._quat_repr = quat_repr
        self._matrix = None
        self._phi = None
        self._hat = None
        if rot.shape == (4, 1):
            self._quat = rot.copy()
        elif rot.shape == (3, 1):
            theta = np.linalg.norm(rot)
            if theta < _LIE_EPSILON:
                self._phi = np.zeros((3, 1), dtype=np.float32)
                ident = [0.0, 0.0, 0.0, 1.0] if quat_repr == "xyzw" else [1.0, 0.0, 0.0, 0.0]
                self._quat = np.array(ident, dtype=np.float32).reshape(4, 1)
            else:
                self._phi = wrap_angle(theta) * (rot / theta)
                self._quat = self._se3_to_quat(self._phi, quat_repr)
        elif rot.shape == (3, 3):
            assert is_rotation_matrix(rot), "Got invalid rotation matrix '{}'".format(rot.tolist())
            self._matrix = rot.copy()
            self._quat = self._SE3_to_quat(self._matrix, quat_repr)
        else:
            raise ValueError("Expected 'rot' to have shape (4, 1), (3, 1) or (3, 3), got '{}' instead".format(rot.shape))

    # -- conversions ------------------------------------------------------------------------------
    @staticmethod
    def _se3_to_quat(phi: np.ndarray, quat_repr: str = "wxyz") -> np.ndarray:
        assert phi.shape == (3, 1), "Expected 'phi' shape to be (3, 1), got '{}' instead".format(phi.shape)
        theta = np.linalg.norm(phi)
        axis = (phi / theta).flatten()
        s, c = math.sin(theta / 2), math.cos(theta / 2)
        vec = [s * axis[0], s * axis[1], s * axis[2]]
        vals = [c] + vec if quat_repr == "wxyz" else vec + [c]
        return np.array(vals, dtype=np.float32).reshape(4, 1)

    @staticmethod
    def _SE3_to_quat(R: np.ndarray, quat_repr: str = "wxyz") -> np.ndarray:
        assert R.shape == (3, 3), "Expected 'R' shape to be (3, 3), got '{}' instead".format(R.shape)
        q = np.zeros(4)
        tr = R[0, 0] + R[1, 1] + R[2, 2]
        if tr > 0:
            s = np.sqrt(1 + tr)
            q[0] = 0.5 * s
            s = 0.5 / s
            q[1] = (R[2, 1] - R[1, 2]) * s
            q[2] = (R[0, 2] - R[2, 0]) * s
            q[3] = (R[1, 0] - R[0, 1]) * s
        else:
            # tie-breaking as in the reference: prefer 0, then 1 if strictly larger, then 2 if strictly larger
            i = 0
            if R[1, 1] > R[0, 0]:
                i = 1
            if R[2, 2] > R[i, i]:
                i = 2
            j, k = (i + 1) % 3, (i + 2) % 3
            s = np.sqrt(R[i, i] - R[j, j] - R[k, k] + 1)
            q[1 + i] = 0.5 * s
            s = 0.5 / s
            q[0] = (R[k, j] - R[j, k]) * s
            q[1 + j] = (R[j, i] + R[i, j]) * s
            q[1 + k] = (R[k, i] + R[i, k]) * s
        q = q.reshape(4, 1)
        if quat_repr == "xyzw":
            q = np.roll(q, 1, axis=0)
        return q

    @classmethod
    def from_se3(cls, phi: np.ndarray):
        assert phi.shape == (3, 1), "Expected 'phi' shape to be (3, 1), got '{}' instead".format(phi.shape)
        return cls(cls._se3_to_quat(phi))

    @classmethod
    def from_SE3(cls, R: np.ndarray):
        assert R.shape == (3, 3), "Expected 'R' shape to be (3, 3), got '{}' instead".format(R.shape)
        return cls(cls._SE3_to_quat(R))

    def _wxyz(self):
        q = self._quat.flatten()
        return (q[0], q[1], q[2], q[3]) if self._quat_repr == "wxyz" else (q[3], q[0], q[1], q[2])

    # -- group operations -------------------------------------------------------------------------
    def exp(self) -> np.ndarray:
        """3x3 float32 rotation matrix from the stored quaternion."""
        if self._matrix is None:
            w, x, y, z = self._wxyz()
            self._matrix = np.array([
                [2 * (w * w + x * x) - 1, 2 * (x * y - w * z), 2 * (x * z + w * y)],
                [2 * (x * y + w * z), 2 * (w * w + y * y) - 1, 2 * (y * z - w * x)],
                [2 * (x * z - w * y), 2 * (y * z + w * x), 2 * (w * w + z * z) - 1]], dtype=np.float32)
        return self._matrix

    def log(self) -> np.ndarray:
        """(3,1) rotation vector."""
        if self._phi is None:
            w, x, y, z = self._wxyz()
            vec = np.array([x, y, z])
            n = np.linalg.norm(vec)
            if n < _LIE_EPSILON:
                self._phi = np.zeros((3, 1), dtype=np.float32)
            else:
                self._phi = wrap_angle(2 * math.atan2(n, w) / n) * vec.reshape(3, 1)
        return self._phi

    def hat(self) -> np.ndarray:
        if self._hat is None:
            self._hat = _skew(self.log())
        return self._hat

    @property
    def quat(self) -> np.ndarray:
        return self._quat

    @property
    def quat_repr(self) -> str:
        return self._quat_repr

    @property
    def theta(self):
        return np.linalg.norm(self.log())

    def __mul__(self, right: "So3") -> "So3":
        assert isinstance(right, So3), "Got invalid type '{}', expected So3".format(type(right))
        qa = self._quat if self._quat_repr == "wxyz" else np.roll(self._quat, 1, axis=0)
        qb = right.quat if right.quat_repr == "wxyz" else np.roll(right.quat, 1, axis=0)
        return So3(quat_mult(qa, qb), quat_repr="wxyz")

    def inverse(self) -> "So3":
        return So3(self.exp().T.copy())

    def copy(self) -> "So3":
        return So3(self.quat.copy())

    @classmethod
    def identity(cls) -> "So3":
        return cls(np.zeros((3, 1), dtype=np.float32))

    def __eq__(self, other) -> bool:
        assert isinstance(other, So3), "Expected 'other' to be of type 'So3', got '{}' instead".format(type(other))
        return np.allclose(self.log(), other.log(), atol=_LIE_EPSILON)


class Se3:
    """Rigid transform = (So3, (3,1) translation)."""

    def __init__(self, so3: So3, tvec: np.ndarray):
        assert isinstance(so3, So3), "Expected 'so3' to be of type 'So3', got '{}' instead".format(type(so3))
        assert tvec.shape == (3, 1), "Expected 'tvec' to have shape '(3, 1)', got '{}' instead".format(tvec.shape)
        self._so3 = so3
        self._tvec = tvec.copy()
        self._matrix = None
        self._xi = None
        self._hat = None

    @property
    def so3(self) -> So3:
        return self._so3

    @property
    def tvec(self) -> np.ndarray:
        return self._tvec

    def exp(self) -> np.ndarray:
        """4x4 float32 homogeneous matrix (rotation block left at identity below the epsilon angle)."""
        if self._matrix is None:
            T = np.eye(4, dtype=np.float32)
            if abs(np.linalg.norm(self._so3.log())) >= _LIE_EPSILON:
                T[:3, :3] = self._so3.exp()
            T[:3, 3] = self._tvec.flatten()
            self._matrix = T
        return self._matrix

    def log(self) -> np.ndarray:
        """(6,1) float32 twist (upsilon, omega)."""
        if self._xi is None:
            xi = np.zeros((6, 1), dtype=np.float32)
            phi = self._so3.log()
            theta = np.linalg.norm(phi)
            if abs(theta) < _LIE_EPSILON:
                xi[:3, 0] = self._tvec.flatten()
            else:
                axis = So3(phi.reshape(3, 1) / theta)
                a = axis.log()
                half = theta / 2
                A = half * np.cos(half) / np.sin(half)
                V_inv = A * np.eye(3, dtype=np.float32) + (1 - A) * np.dot(a, a.T) - half * axis.hat()
                xi[:3, 0] = np.dot(V_inv, self._tvec).flatten()
                xi[3:, 0] = phi.flatten()
            self._xi = xi
        return self._xi

    def hat(self) -> np.ndarray:
        if self._hat is None:
            h = np.zeros((4, 4), dtype=np.float32)
            h[:3, :3] = self._so3.hat()
            h[:3, 3] = self.log()[:3, 0]
            self._hat = h
        return self._hat

    def inverse(self) -> "Se3":
        r_inv = self._so3.inverse()
        return Se3(r_inv, -np.dot(r_inv.exp(), self._tvec))

    def __mul__(self, right: "Se3") -> "Se3":
        assert isinstance(right, Se3), "Expected 'right' to be of type 'Se3', got '{}' instead.".format(type(right))
        return Se3(self._so3 * right.so3, self._tvec + np.dot(self._so3.exp(), right.tvec))

    @classmethod
    def identity(cls) -> "Se3":
        return cls(So3.identity(), np.zeros((3, 1), dtype=np.float32))

    def copy(self) -> "Se3":
        return Se3(self.so3.copy(), self.tvec.copy())

    @classmethod
    def from_se3(cls, xi: np.ndarray) -> "Se3":
        assert xi.shape == (6, 1), "Expected 'xi' to have shape '(6, 1)', got '{}' instead".format(xi.shape)
        upsilon = xi[:3]
        rot = So3(xi[3:])
        theta = rot.theta
        if theta < _LIE_EPSILON:
            return cls(So3.identity(), upsilon)
        ph = rot.hat()
        V = (np.eye(3, dtype=np.float32) + ((1 - math.cos(theta)) / (theta ** 2)) * ph +
             ((theta - math.sin(theta)) / (theta ** 3)) * np.dot(ph, ph))
        return cls(So3(xi[3:]), np.dot(V, upsilon))

    @classmethod
    def from_matrix(cls, T: np.ndarray) -> "Se3":
        """Convenience (not in the reference): 4x4 or 3x4 [R|t] -> Se3."""
        T = np.asarray(T)
        return cls(So3(np.array(T[:3, :3], dtype=np.float64)), np.array(T[:3, 3], dtype=np.float64).reshape(3, 1))

    def __eq__(self, other) -> bool:
        assert isinstance(other, Se3), "Expected 'other' to be of type 'Se3', got '{}' instead".format(type(other))
        return np.allclose(self.log(), other.log(), atol=_LIE_EPSILON)
