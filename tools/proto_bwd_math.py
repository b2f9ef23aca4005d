"""Prototype (CPU, fp64) of the backward math the HIP kernel uses, checked against autograd of the oracle.

Lane model: one lane per (clip, joint); prefix sums over the DFS-ordered joints give subtree sums.
"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import pose_head as O

torch.manual_seed(0)
dt = torch.float64
B, T, J = 3, 5, 26
par = list(O.parents())
end = list(range(J))
for j in reversed(range(J)):
    if par[j] >= 0:
        end[par[j]] = max(end[par[j]], end[j])
print('subtree end', end)

y = torch.randn(B, T, J, 6, dtype=dt)
y[..., 0] += 2; y[..., 4] += 2
y.requires_grad_(True)
st = torch.tensor([0, 2, 3])
F = torch.randn(B, T, J, 3, dtype=dt)            # upstream grad wrt abs_loc
o = O.pose_head(y, 'pose_changes_6d', st, transform='none')
(o['absolute_pose_loc'] * F).sum().backward()
g_ref = y.grad.clone()

# ---- manual backward ----
with torch.no_grad():
    c = O.rotation_6d_to_matrix(y.detach())                      # (B,T,J,3,3)
    rel_loc, rel_rot = O.relative_tensors(dt)
    l = rel_loc[st]                                              # (B,J,3)
    Rref = rel_rot[st]
    R = o['relative_pose_rot'].detach()
    A = o['absolute_pose_rot'].detach()
    x = o['absolute_pose_loc'].detach()
    gy = torch.zeros_like(y)
    carry = torch.zeros(B, J, 3, 3, dtype=dt)
    Rt = R[:, T - 1].clone()                                     # final relative rotation (saved by fwd)
    for t in reversed(range(T)):
        At, xt, Ft = A[:, t], x[:, t], F[:, t]
        P = torch.cumsum(Ft, 1)                                  # inclusive prefix over joints
        Pm1 = torch.cat((torch.zeros(B, 1, 3, dtype=dt), P[:, :-1]), 1)
        SubF = P[:, end] - Pm1
        xp = torch.stack([xt[:, p] if p >= 0 else torch.zeros(B, 3, dtype=dt) for p in par], 1)
        Ap = torch.stack([At[:, p] if p >= 0 else torch.eye(3, dtype=dt).expand(B, 3, 3) for p in par], 1)
        r = xt - xp
        Y = r[..., :, None] * SubF[..., None, :]
        PY = torch.cumsum(Y, 1)
        Z = PY[:, end] - PY
        GA = At @ Z
        gRd = GA @ Ap.transpose(-1, -2)
        GR = gRd + carry
        ct = c[:, t]
        Rprev = ct.transpose(-1, -2) @ Rt if t > 0 else Rref     # inversion (orthonormal c)
        if t > 0:
            print('  inversion err t=%d: %.2e' % (t, (Rprev - R[:, t - 1]).abs().max()))
        gc = GR @ Rprev.transpose(-1, -2)
        carry = ct.transpose(-1, -2) @ GR
        Rt = Rprev
        # 6D -> R backward
        a1, a2 = y.detach()[:, t, :, :3], y.detach()[:, t, :, 3:]
        n1 = a1.norm(dim=-1, keepdim=True); b1 = a1 / n1
        d = (b1 * a2).sum(-1, keepdim=True); u2 = a2 - d * b1
        n2 = u2.norm(dim=-1, keepdim=True); b2 = u2 / n2
        g1, g2, g3 = gc[..., 0, :], gc[..., 1, :], gc[..., 2, :]
        gb1 = g1 + torch.cross(b2, g3, dim=-1)
        gb2 = g2 + torch.cross(g3, b1, dim=-1)
        gu2 = (gb2 - b2 * (b2 * gb2).sum(-1, keepdim=True)) / n2
        ga2 = gu2 - b1 * (b1 * gu2).sum(-1, keepdim=True)
        gb1 = gb1 - d * gu2 - (gu2 * b1).sum(-1, keepdim=True) * a2
        ga1 = (gb1 - b1 * (b1 * gb1).sum(-1, keepdim=True)) / n1
        gy[:, t] = torch.cat((ga1, ga2), -1)
print('max rel err vs autograd: %.3e' % ((gy - g_ref).abs().max() / g_ref.abs().max()))

# ---- pointer doubling FK check ----
with torch.no_grad():
    Rt = R[:, 2]
    IDL = J                                                    # identity lane index
    Mr = torch.cat((Rt, torch.eye(3, dtype=dt).expand(B, 1, 3, 3)), 1)
    Ml = torch.cat((l, torch.zeros(B, 1, 3, dtype=dt)), 1)
    anc = [p if p >= 0 else IDL for p in par] + [IDL]
    for k in range(3):
        Ar, Al = Mr[:, anc], Ml[:, anc]
        Ml = (Ml[..., None, :] @ Ar)[..., 0, :] + Al
        Mr = Mr @ Ar
        anc = [anc[a] for a in anc]
    print('doubling FK err rot %.2e loc %.2e' % ((Mr[:, :J] - A[:, 2]).abs().max(), (Ml[:, :J] - x[:, 2]).abs().max()))

# ---- tangent-space ("rigid body") backward: torques instead of 3x3 matrix adjoints ----
with torch.no_grad():
    gy2 = torch.zeros_like(y)
    S = torch.zeros(B, J, 3, dtype=dt)                      # suffix sum over time of parent-frame torques
    Rt = R[:, T - 1].clone()
    for t in reversed(range(T)):
        At, xt, Ft = A[:, t], x[:, t], F[:, t]
        FX = torch.cross(Ft, xt, dim=-1)
        P = torch.cumsum(torch.cat((Ft, FX), -1), 1)
        Pm1 = torch.cat((torch.zeros(B, 1, 6, dtype=dt), P[:, :-1]), 1)
        Sub = P[:, end] - Pm1
        SubF, SubFX = Sub[..., :3], Sub[..., 3:]
        tau = SubFX - torch.cross(SubF, xt, dim=-1)           # world-frame torque about joint j
        ct = c[:, t]
        Rprev = ct.transpose(-1, -2) @ Rt if t > 0 else Rref
        # parent-frame torque: tau @ A_p^T with A_p^T = A^T R   (A = R A_p, R orthonormal)
        taup = ((tau[..., None, :] @ At.transpose(-1, -2)) @ Rt)[..., 0, :]
        S = S + taup
        g = (S[..., None, :] @ Rprev.transpose(-1, -2))[..., 0, :]      # gradient in the right-tangent of c_t
        G = 0.5 * torch.cross(ct, g[..., None, :].expand_as(ct), dim=-1)  # rows: (c_i x g) / 2
        Rt = Rprev
        a1, a2 = y.detach()[:, t, :, :3], y.detach()[:, t, :, 3:]
        n1 = a1.norm(dim=-1, keepdim=True); b1 = a1 / n1
        d = (b1 * a2).sum(-1, keepdim=True); u2 = a2 - d * b1
        n2 = u2.norm(dim=-1, keepdim=True); b2 = u2 / n2
        g1, g2, g3 = G[..., 0, :], G[..., 1, :], G[..., 2, :]
        gb1 = g1 + torch.cross(b2, g3, dim=-1)
        gb2 = g2 + torch.cross(g3, b1, dim=-1)
        gu2 = (gb2 - b2 * (b2 * gb2).sum(-1, keepdim=True)) / n2
        ga2 = gu2 - b1 * (b1 * gu2).sum(-1, keepdim=True)
        gb1 = gb1 - d * gu2 - (gu2 * b1).sum(-1, keepdim=True) * a2
        ga1 = (gb1 - b1 * (b1 * gb1).sum(-1, keepdim=True)) / n1
        gy2[:, t] = torch.cat((ga1, ga2), -1)
print('tangent-space backward, max rel err vs autograd: %.3e' % ((gy2 - g_ref).abs().max() / g_ref.abs().max()))
