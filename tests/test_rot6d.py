"""6D -> rotation matrix (pytorch3d restatement; third-party, unpinned -> anchored on known answers)."""
import math

import pytest
import torch

from oracle import egoego_oracle as O


def _rand_rot(n, seed=0):
    g = torch.Generator().manual_seed(seed)
    q = torch.nn.functional.normalize(torch.randn(n, 4, generator=g), dim=-1)
    w, x, y, z = q.unbind(-1)
    return torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w),
                        2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
                        2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], -1).view(n, 3, 3)


def test_oracle_known_answers():
    eye = O.rotation_6d_to_matrix(torch.tensor([[1.0, 0, 0, 0, 1, 0]]))
    assert torch.allclose(eye[0], torch.eye(3))
    # non-orthonormal input: Gram-Schmidt by hand
    r = O.rotation_6d_to_matrix(torch.tensor([[2.0, 0, 0, 1, 3, 0]]))
    assert torch.allclose(r[0], torch.eye(3), atol=1e-6)
    c, s = math.cos(0.3), math.sin(0.3)
    rz = O.rotation_6d_to_matrix(torch.tensor([[c, -s, 0, s, c, 0]]))
    assert torch.allclose(rz[0], torch.tensor([[c, -s, 0], [s, c, 0], [0, 0, 1.0]]), atol=1e-6)


def test_oracle_roundtrip_and_orthonormality():
    R = _rand_rot(512)
    back = O.rotation_6d_to_matrix(R[:, :2, :].reshape(-1, 6))  # 6D = first two ROWS
    assert (back - R).abs().max() < 1e-5
    g = torch.Generator().manual_seed(3)
    M = O.rotation_6d_to_matrix(torch.randn(512, 6, generator=g))
    assert (M @ M.transpose(1, 2) - torch.eye(3)).abs().max() < 1e-5
    assert (torch.linalg.det(M) - 1).abs().max() < 1e-5


@pytest.mark.gpu
def test_hip_rot6d_matches_oracle():
    from egoego_release_amd import _lib
    import ctypes as C
    lib = _lib.load()
    g = torch.Generator().manual_seed(4)
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def run(d6):
        x = d6.cuda().contiguous()
        out = torch.empty(*d6.shape[:-1], 3, 3, device="cuda")
        _lib.check(lib.egoego_rot6d_to_matrix(x.data_ptr(), out.data_ptr(), x.numel() // 6, stream))
        torch.cuda.synchronize()
        return out

    assert run(torch.zeros(0, 6)).shape == (0, 3, 3)  # empty input
    # well-conditioned inputs (rows of rotations, scaled and slightly sheared): tight tolerance
    R = _rand_rot(4096, seed=6)
    d6 = (R[:, :2, :] * torch.tensor([[[1.7], [0.6]]])).reshape(-1, 6)
    d6[:, 3:] += 0.3 * d6[:, :3]
    assert (run(d6).cpu() - O.rotation_6d_to_matrix(d6)).abs().max().item() < 5e-6
    assert (run(d6).cpu() - R).abs().max().item() < 5e-6
    # raw gaussian 6D at the pose-tensor size (B=256, T=120, 22 joints): Gram-Schmidt conditioning varies
    d6 = torch.randn(256, 120, 22, 6, generator=g)
    out = run(d6)
    err = (out.cpu() - O.rotation_6d_to_matrix(d6)).abs()
    assert err.max().item() < 2e-3 and err.mean().item() < 1e-6
    # full-size property: orthonormal, det +1
    m = out.view(-1, 3, 3)
    assert (m @ m.transpose(1, 2) - torch.eye(3, device="cuda")).abs().max() < 1e-4
    assert (torch.linalg.det(m.cpu()) - 1).abs().max() < 1e-4
