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
    for shape in ((1, 6), (256, 120, 22, 6), (0, 6)):
        d6 = torch.randn(*shape, generator=g)
        want = O.rotation_6d_to_matrix(d6) if d6.numel() else torch.zeros(0, 3, 3)
        x = d6.cuda().contiguous()
        out = torch.empty(*shape[:-1], 3, 3, device="cuda")
        _lib.check(lib.egoego_rot6d_to_matrix(x.data_ptr(), out.data_ptr(), x.numel() // 6,
                                              C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        torch.cuda.synchronize()
        assert (out.cpu() - want).abs().max().item() < 2e-6 if d6.numel() else True
    # full-size property: orthonormal, det +1
    m = out.view(-1, 3, 3)
    assert (m @ m.transpose(1, 2) - torch.eye(3, device="cuda")).abs().max() < 1e-5
