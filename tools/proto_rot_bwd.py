"""fp64 check of the extra torque from a gradient on the ABSOLUTE ROTATIONS (rot_3d-type losses), in the convention of the
tangent-space backward (tools/proto_bwd_math.py): tau_j += Sub(t)_j, t_m = (P_zy - P_yz, P_xz - P_zx, P_yx - P_xy), P = A_m^T G_m."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import pose_head as O

torch.manual_seed(1)
dt = torch.float64
B, T, J = 3, 5, 26
par = list(O.parents())
end = list(range(J))
for j in reversed(range(J)):
    if par[j] >= 0:
        end[par[j]] = max(end[par[j]], end[j])
y = torch.randn(B, T, J, 6, dtype=dt)
y[..., 0] += 2; y[..., 4] += 2
y.requires_grad_(True)
st = torch.tensor([0, 2, 3])
F = torch.randn(B, T, J, 3, dtype=dt)
G = torch.randn(B, T, J, 3, 3, dtype=dt)          # upstream grad wrt abs_rot
o = O.pose_head(y, 'pose_changes_6d', st, transform='none')
((o['absolute_pose_loc'] * F).sum() + (o['absolute_pose_rot'] * G).sum()).backward()
g_ref = y.grad.clone()

with torch.no_grad():
    c = O.rotation_6d_to_matrix(y.detach())
    A, x, R = o['absolute_pose_rot'].detach(), o['absolute_pose_loc'].detach(), o['relative_pose_rot'].detach()
    Rref = O.relative_tensors()[1].to(dt)[st] if hasattr(O, 'relative_tensors') else None
    if Rref is None:
        raise SystemExit('oracle API changed')
    gy2 = torch.zeros_like(y)
    S = torch.zeros(B, J, 3, dtype=dt)
    Rt = R[:, T - 1].clone()
    for t in reversed(range(T)):
        At, xt, Ft = A[:, t], x[:, t], F[:, t]
        P_ = At.transpose(-1, -2) @ G[:, t]
        tm = torch.stack((P_[..., 2, 1] - P_[..., 1, 2], P_[..., 0, 2] - P_[..., 2, 0], P_[..., 1, 0] - P_[..., 0, 1]), -1)
        FX = torch.cross(Ft, xt, dim=-1)
        Pc = torch.cumsum(torch.cat((Ft, FX, tm), -1), 1)
        Pm1 = torch.cat((torch.zeros(B, 1, 9, dtype=dt), Pc[:, :-1]), 1)
        Sub = Pc[:, end] - Pm1
        SubF, SubFX, SubT = Sub[..., :3], Sub[..., 3:6], Sub[..., 6:]
        tau = SubFX - torch.cross(SubF, xt, dim=-1) + SubT
        ct = c[:, t]
        Rprev = ct.transpose(-1, -2) @ Rt if t > 0 else Rref
        taup = ((tau[..., None, :] @ At.transpose(-1, -2)) @ Rt)[..., 0, :]
        S = S + taup
        g = (S[..., None, :] @ Rprev.transpose(-1, -2))[..., 0, :]
        Gm = 0.5 * torch.cross(ct, g[..., None, :].expand_as(ct), dim=-1)
        Rt = Rprev
        a1, a2 = y.detach()[:, t, :, :3], y.detach()[:, t, :, 3:]
        n1 = a1.norm(dim=-1, keepdim=True); b1 = a1 / n1
        d = (b1 * a2).sum(-1, keepdim=True); u2 = a2 - d * b1
        n2 = u2.norm(dim=-1, keepdim=True); b2 = u2 / n2
        g1, g2, g3 = Gm[..., 0, :], Gm[..., 1, :], Gm[..., 2, :]
        gb1 = g1 + torch.cross(b2, g3, dim=-1)
        gb2 = g2 + torch.cross(g3, b1, dim=-1)
        gu2 = (gb2 - b2 * (b2 * gb2).sum(-1, keepdim=True)) / n2
        ga2 = gu2 - b1 * (b1 * gu2).sum(-1, keepdim=True)
        gb1 = gb1 - d * gu2 - (gu2 * b1).sum(-1, keepdim=True) * a2
        ga1 = (gb1 - b1 * (b1 * gb1).sum(-1, keepdim=True)) / n1
        gy2[:, t] = torch.cat((ga1, ga2), -1)
print('with rotation torque, max rel err vs autograd: %.3e' % ((gy2 - g_ref).abs().max() / g_ref.abs().max()))
