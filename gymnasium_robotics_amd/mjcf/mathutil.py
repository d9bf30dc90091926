"""Small fp64 rotation helpers used by the MJCF model compiler (host, cold path).

Quaternions are (w, x, y, z), rotation matrices are row-major 3x3, the same
conventions MuJoCo uses and that the reference's ``utils/rotations.py`` assumes
(/root/reference/gymnasium_robotics/utils/rotations.py:245 ``quat2mat``).
"""
import numpy as np


def quat_normalize(q):
    q = np.asarray(q, dtype=np.float64)
    n = np.linalg.norm(q)
    if n < 1e-15:
        return np.array([1.0, 0.0, 0.0, 0.0])
    return q / n


def quat_mul(a, b):
    aw, ax, ay, az = a
    bw, bx, by, bz = b
    return np.array(
        [
            aw * bw - ax * bx - ay * by - az * bz,
            aw * bx + ax * bw + ay * bz - az * by,
            aw * by - ax * bz + ay * bw + az * bx,
            aw * bz + ax * by - ay * bx + az * bw,
        ]
    )


def quat_conj(q):
    return np.array([q[0], -q[1], -q[2], -q[3]])


def quat2mat(q):
    w, x, y, z = q
    return np.array(
        [
            [w * w + x * x - y * y - z * z, 2 * (x * y - w * z), 2 * (x * z + w * y)],
            [2 * (x * y + w * z), w * w - x * x + y * y - z * z, 2 * (y * z - w * x)],
            [2 * (x * z - w * y), 2 * (y * z + w * x), w * w - x * x - y * y + z * z],
        ]
    )


def mat2quat(m):
    """Rotation matrix -> unit quaternion (w>=0 branch selection by largest diagonal)."""
    m = np.asarray(m, dtype=np.float64)
    tr = m[0, 0] + m[1, 1] + m[2, 2]
    if tr > 0:
        s = np.sqrt(tr + 1.0) * 2
        q = np.array([0.25 * s, (m[2, 1] - m[1, 2]) / s, (m[0, 2] - m[2, 0]) / s, (m[1, 0] - m[0, 1]) / s])
    elif m[0, 0] > m[1, 1] and m[0, 0] > m[2, 2]:
        s = np.sqrt(1.0 + m[0, 0] - m[1, 1] - m[2, 2]) * 2
        q = np.array([(m[2, 1] - m[1, 2]) / s, 0.25 * s, (m[0, 1] + m[1, 0]) / s, (m[0, 2] + m[2, 0]) / s])
    elif m[1, 1] > m[2, 2]:
        s = np.sqrt(1.0 + m[1, 1] - m[0, 0] - m[2, 2]) * 2
        q = np.array([(m[0, 2] - m[2, 0]) / s, (m[0, 1] + m[1, 0]) / s, 0.25 * s, (m[1, 2] + m[2, 1]) / s])
    else:
        s = np.sqrt(1.0 + m[2, 2] - m[0, 0] - m[1, 1]) * 2
        q = np.array([(m[1, 0] - m[0, 1]) / s, (m[0, 2] + m[2, 0]) / s, (m[1, 2] + m[2, 1]) / s, 0.25 * s])
    return quat_normalize(q)


def axisangle2quat(axis, angle):
    axis = np.asarray(axis, dtype=np.float64)
    n = np.linalg.norm(axis)
    if n < 1e-15:
        return np.array([1.0, 0.0, 0.0, 0.0])
    axis = axis / n
    s = np.sin(0.5 * angle)
    return np.array([np.cos(0.5 * angle), axis[0] * s, axis[1] * s, axis[2] * s])


def euler2quat(e, seq="xyz"):
    """MJCF ``euler`` attribute: successive rotations about the axes named in
    ``eulerseq`` (lower case = intrinsic / body-fixed, upper case = extrinsic)."""
    q = np.array([1.0, 0.0, 0.0, 0.0])
    for ang, ax in zip(e, seq):
        axis = {"x": [1, 0, 0], "y": [0, 1, 0], "z": [0, 0, 1]}[ax.lower()]
        r = axisangle2quat(axis, ang)
        if ax.islower():
            q = quat_mul(q, r)
        else:
            q = quat_mul(r, q)
    return quat_normalize(q)


def zaxis2quat(z):
    """Minimal rotation taking (0,0,1) to z (MJCF ``zaxis`` / ``fromto``)."""
    z = np.asarray(z, dtype=np.float64)
    n = np.linalg.norm(z)
    if n < 1e-15:
        return np.array([1.0, 0.0, 0.0, 0.0])
    z = z / n
    a = np.cross([0.0, 0.0, 1.0], z)
    s = np.linalg.norm(a)
    c = z[2]
    if s < 1e-12:
        return np.array([1.0, 0.0, 0.0, 0.0]) if c > 0 else np.array([0.0, 1.0, 0.0, 0.0])
    return axisangle2quat(a / s, np.arctan2(s, c))


def xyaxes2quat(xy):
    x = np.asarray(xy[:3], dtype=np.float64)
    y = np.asarray(xy[3:], dtype=np.float64)
    x = x / np.linalg.norm(x)
    y = y - x * np.dot(x, y)
    y = y / np.linalg.norm(y)
    z = np.cross(x, y)
    return mat2quat(np.stack([x, y, z], axis=1))


def rot_vec(q, v):
    return quat2mat(q) @ np.asarray(v, dtype=np.float64)
