"""Functional restatement of the pytorch3d.transforms symbols the reference's physics path calls.

TEST INFRASTRUCTURE ONLY (see refshim/__init__.py).  pytorch3d==0.7.5 is pinned by the
reference (`environment.yaml:13`) but not installed here and not vendored, so the
semantics are restated from its published documentation:

* quaternions are real-first (w, x, y, z);
* ``so3_exponential_map(v, eps=1e-4)`` clamps ``|v|^2`` from below by ``eps`` before the
  square root, so tiny rotations are *not* exactly Rodrigues;
* ``quaternion_multiply`` standardises the result to a non-negative real part;
* ``matrix_to_quaternion`` picks the best-conditioned of four candidates.
"""
import torch
import torch.nn.functional as F


def _hat(v):
    x, y, z = v.unbind(-1)
    o = torch.zeros_like(x)
    return torch.stack([o, -z, y, z, o, -x, -y, x, o], dim=-1).reshape(v.shape[:-1] + (3, 3))


def so3_exponential_map(log_rot, eps=1e-4):
    nrms = (log_rot * log_rot).sum(1)
    ang = torch.clamp(nrms, eps).sqrt()
    inv = 1.0 / ang
    fac1 = inv * ang.sin()
    fac2 = inv * inv * (1.0 - ang.cos())
    K = _hat(log_rot)
    K2 = torch.bmm(K, K)
    eye = torch.eye(3, dtype=log_rot.dtype, device=log_rot.device)[None]
    return fac1[:, None, None] * K + fac2[:, None, None] * K2 + eye


def quaternion_to_matrix(q):
    r, i, j, k = torch.unbind(q, -1)
    two_s = 2.0 / (q * q).sum(-1)
    o = torch.stack(
        (
            1 - two_s * (j * j + k * k), two_s * (i * j - k * r), two_s * (i * k + j * r),
            two_s * (i * j + k * r), 1 - two_s * (i * i + k * k), two_s * (j * k - i * r),
            two_s * (i * k - j * r), two_s * (j * k + i * r), 1 - two_s * (i * i + j * j),
        ),
        -1,
    )
    return o.reshape(q.shape[:-1] + (3, 3))


def _sqrt_positive_part(x):
    ret = torch.zeros_like(x)
    pos = x > 0
    if torch.is_grad_enabled():
        ret[pos] = torch.sqrt(x[pos])
    else:
        ret = torch.where(pos, torch.sqrt(x), ret)
    return ret


def standardize_quaternion(q):
    return torch.where(q[..., 0:1] < 0, -q, q)


def matrix_to_quaternion(matrix):
    batch = matrix.shape[:-2]
    m00, m01, m02, m10, m11, m12, m20, m21, m22 = torch.unbind(matrix.reshape(batch + (9,)), dim=-1)
    q_abs = _sqrt_positive_part(
        torch.stack(
            [1.0 + m00 + m11 + m22, 1.0 + m00 - m11 - m22, 1.0 - m00 + m11 - m22, 1.0 - m00 - m11 + m22],
            dim=-1,
        )
    )
    cand = torch.stack(
        [
            torch.stack([q_abs[..., 0] ** 2, m21 - m12, m02 - m20, m10 - m01], dim=-1),
            torch.stack([m21 - m12, q_abs[..., 1] ** 2, m10 + m01, m02 + m20], dim=-1),
            torch.stack([m02 - m20, m10 + m01, q_abs[..., 2] ** 2, m12 + m21], dim=-1),
            torch.stack([m10 - m01, m20 + m02, m21 + m12, q_abs[..., 3] ** 2], dim=-1),
        ],
        dim=-2,
    )
    floor = torch.tensor(0.1, dtype=q_abs.dtype, device=q_abs.device)
    cand = cand / (2.0 * q_abs[..., None].max(floor))
    out = cand[F.one_hot(q_abs.argmax(dim=-1), num_classes=4) > 0.5, :].reshape(batch + (4,))
    return standardize_quaternion(out)


def quaternion_raw_multiply(a, b):
    aw, ax, ay, az = torch.unbind(a, -1)
    bw, bx, by, bz = torch.unbind(b, -1)
    ow = aw * bw - ax * bx - ay * by - az * bz
    ox = aw * bx + ax * bw + ay * bz - az * by
    oy = aw * by - ax * bz + ay * bw + az * bx
    oz = aw * bz + ax * by - ay * bx + az * bw
    return torch.stack((ow, ox, oy, oz), -1)


def quaternion_multiply(a, b):
    return standardize_quaternion(quaternion_raw_multiply(a, b))


def quaternion_invert(q):
    return q * q.new_tensor([1, -1, -1, -1])


def quaternion_apply(q, point):
    real = point.new_zeros(point.shape[:-1] + (1,))
    pq = torch.cat((real, point), -1)
    out = quaternion_raw_multiply(quaternion_raw_multiply(q, pq), quaternion_invert(q))
    return out[..., 1:]


def axis_angle_to_matrix(aa):
    return so3_exponential_map(aa.reshape(-1, 3), eps=1e-12).reshape(aa.shape[:-1] + (3, 3))


def random_quaternions(n, dtype=None, device=None):
    o = torch.randn((n, 4), dtype=dtype, device=device)
    s = (o * o).sum(1)
    return o / torch.copysign(torch.sqrt(s), o[:, 0])[:, None]


def so3_relative_angle(R1, R2, cos_angle=False):
    R12 = torch.bmm(R1, R2.permute(0, 2, 1))
    c = ((R12[:, 0, 0] + R12[:, 1, 1] + R12[:, 2, 2]) - 1.0) * 0.5
    return c if cos_angle else torch.acos(c.clamp(-1, 1))
