"""Rotation conversions the reference takes from pytorch3d 0.6.0 (absent third-party dependency; published
definitions restated). Composed of differentiable torch ops: used only OUTSIDE the fused path
(API compatibility for ``rotation_output_format='matrix'``, teacher forcing targets); the training hot path feeds the
raw 6-D tensor to the HIP pose head, which orthonormalises in-kernel.
"""
import torch
from torch import Tensor


def rotation_6d_to_matrix(d6: Tensor) -> Tensor:
    """Zhou et al. 2019: Gram-Schmidt of the two 3-vectors, third row = cross product; rows (b1, b2, b3)."""
    a1, a2 = d6[..., :3], d6[..., 3:]
    b1 = torch.nn.functional.normalize(a1, dim=-1)
    b2 = torch.nn.functional.normalize(a2 - (b1 * a2).sum(-1, keepdim=True) * b1, dim=-1)
    return torch.stack((b1, b2, torch.cross(b1, b2, dim=-1)), dim=-2)


def matrix_to_rotation_6d(matrix: Tensor) -> Tensor:
    return matrix[..., :2, :].clone().reshape(matrix.shape[:-2] + (6,))


def euler_angles_to_matrix(euler_angles: Tensor, convention: str = 'XYZ') -> Tensor:
    def axis(k, a):
        c, s, o, z = torch.cos(a), torch.sin(a), torch.ones_like(a), torch.zeros_like(a)
        rows = {'X': (o, z, z, z, c, -s, z, s, c), 'Y': (c, z, s, z, o, z, -s, z, c), 'Z': (c, -s, z, s, c, z, z, z, o)}[k]
        return torch.stack(rows, -1).reshape(a.shape + (3, 3))
    m = [axis(k, euler_angles[..., i]) for i, k in enumerate(convention)]

    def mm3(a, b):       # 3x3 products as element-wise work: the synthetic data module composes (B, T, 26) of them per batch,
        return (a.unsqueeze(-1) * b.unsqueeze(-3)).sum(-2)      # which as a batched library GEMM took ~1 ms per call
    return mm3(mm3(m[0], m[1]), m[2])
