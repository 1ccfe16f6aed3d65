"""Rotation algebra used around the sampling loop (device-agnostic torch, no pytorch3d dependency).

The reference calls pytorch3d.transforms (third-party, absent from /root/reference, version unpinned —
SURVEY.md §8c) at transformer_cond_diffusion_model.py:375-376, 450-464, 493-507 and
amass_diffusion_dataset.py:109-143, 265-293.  These are restatements of the published definitions:
quaternions are real-first (w, x, y, z); the 6D representation is the first two ROWS of the matrix
(Zhou et al. 2019).  Results agree with any pytorch3d release up to the sign of a quaternion (q and -q
are the same rotation); axis-angle outputs use the rotation angle in [0, pi].
"""
import torch
import torch.nn.functional as F


def quaternion_to_matrix(q):
    r, i, j, k = torch.unbind(q, -1)
    two_s = 2.0 / (q * q).sum(-1)
    o = torch.stack((1 - two_s * (j * j + k * k), two_s * (i * j - k * r), two_s * (i * k + j * r),
                     two_s * (i * j + k * r), 1 - two_s * (i * i + k * k), two_s * (j * k - i * r),
                     two_s * (i * k - j * r), two_s * (j * k + i * r), 1 - two_s * (i * i + j * j)), -1)
    return o.reshape(q.shape[:-1] + (3, 3))


def standardize_quaternion(q):
    return torch.where(q[..., 0:1] < 0, -q, q)


def matrix_to_quaternion(m):
    """Numerically safe branch selection on the largest of (w, x, y, z); result has w >= 0."""
    m00, m01, m02 = m[..., 0, 0], m[..., 0, 1], m[..., 0, 2]
    m10, m11, m12 = m[..., 1, 0], m[..., 1, 1], m[..., 1, 2]
    m20, m21, m22 = m[..., 2, 0], m[..., 2, 1], m[..., 2, 2]
    q_abs = torch.sqrt(torch.clamp(torch.stack((1.0 + m00 + m11 + m22, 1.0 + m00 - m11 - m22,
                                                1.0 - m00 + m11 - m22, 1.0 - m00 - m11 + m22), -1), min=0.0))
    cand = torch.stack((
        torch.stack((q_abs[..., 0] ** 2, m21 - m12, m02 - m20, m10 - m01), -1),
        torch.stack((m21 - m12, q_abs[..., 1] ** 2, m10 + m01, m02 + m20), -1),
        torch.stack((m02 - m20, m10 + m01, q_abs[..., 2] ** 2, m12 + m21), -1),
        torch.stack((m10 - m01, m20 + m02, m21 + m12, q_abs[..., 3] ** 2), -1)), -2)
    cand = cand / (2.0 * q_abs[..., None].clamp(min=0.1))
    best = F.one_hot(q_abs.argmax(-1), 4) > 0.5
    return standardize_quaternion(cand[best, :].reshape(m.shape[:-2] + (4,)))


def quaternion_raw_multiply(a, b):
    aw, ax, ay, az = torch.unbind(a, -1)
    bw, bx, by, bz = torch.unbind(b, -1)
    return torch.stack((aw * bw - ax * bx - ay * by - az * bz, aw * bx + ax * bw + ay * bz - az * by,
                        aw * by - ax * bz + ay * bw + az * bx, aw * bz + ax * by - ay * bx + az * bw), -1)


def quaternion_multiply(a, b):
    return standardize_quaternion(quaternion_raw_multiply(a, b))


def quaternion_invert(q):
    return q * q.new_tensor([1, -1, -1, -1])


def quaternion_apply(q, p):
    pq = torch.cat((p.new_zeros(p.shape[:-1] + (1,)), p), -1)
    return quaternion_raw_multiply(quaternion_raw_multiply(q, pq), quaternion_invert(q))[..., 1:]


def matrix_to_rotation_6d(m):
    return m[..., :2, :].clone().reshape(m.shape[:-2] + (6,))


def rotation_6d_to_matrix(d6):
    """On a ROCm tensor this runs the HIP kernel (egoego_rot6d_to_matrix); the torch expression below is
    for CPU tensors (tests, tiny host-side uses)."""
    if d6.is_cuda:
        import ctypes as C
        from . import _lib
        lib = _lib.load()
        x = d6.to(torch.float32).contiguous()
        out = torch.empty(x.shape[:-1] + (3, 3), device=x.device, dtype=torch.float32)
        _lib.check(lib.egoego_rot6d_to_matrix(x.data_ptr(), out.data_ptr(), x.numel() // 6,
                                              C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)))
        return out
    a1, a2 = d6[..., :3], d6[..., 3:]
    b1 = F.normalize(a1, dim=-1)
    b2 = F.normalize(a2 - (b1 * a2).sum(-1, keepdim=True) * b1, dim=-1)
    return torch.stack((b1, b2, torch.cross(b1, b2, dim=-1)), -2)


def quaternion_to_axis_angle(q):
    q = standardize_quaternion(q)
    n = torch.norm(q[..., 1:], p=2, dim=-1, keepdim=True)
    half = torch.atan2(n, q[..., :1])
    ang = 2 * half
    small = ang.abs() < 1e-6
    s = torch.where(small, 0.5 - ang * ang / 48, torch.sin(half) / torch.where(small, torch.ones_like(ang), ang))
    return q[..., 1:] / s


def matrix_to_axis_angle(m):
    return quaternion_to_axis_angle(matrix_to_quaternion(m))


def axis_angle_to_quaternion(aa):
    ang = torch.norm(aa, p=2, dim=-1, keepdim=True)
    half = 0.5 * ang
    small = ang.abs() < 1e-6
    s = torch.where(small, 0.5 - ang * ang / 48, torch.sin(half) / torch.where(small, torch.ones_like(ang), ang))
    return torch.cat((torch.cos(half), aa * s), -1)


def axis_angle_to_matrix(aa):
    return quaternion_to_matrix(axis_angle_to_quaternion(aa))
