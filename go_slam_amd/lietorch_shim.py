"""Minimal pure-PyTorch `lietorch` surface -- exactly the ops GO-SLAM uses:
SE3(data), .data, indexing, .to, .inv(), `a * b` (group product and action on [...,4] homogeneous
points, with broadcasting), .matrix(), .log(), SE3.exp, SE3.Identity, .adjT, cat.

The reference's lietorch submodule is an empty directory (`thirdparty/lietorch`, commit not
recorded), so this follows lietorch's published SE3 conventions: data = [tx,ty,tz,qx,qy,qz,qw],
(q1,t1)*(q2,t2) = (q1 q2, t1 + q1 t2), tangent order [tau, phi].  No autograd is needed through it
at inference (tracking runs under torch.no_grad).  The hot path itself does not use it any more
(`droid_backends.reproject` is a fused HIP kernel); it exists so the reference's remaining call
sites (`depth_video.py:162-164`, `trajectory_filler.py:43-55`, `slam.py:315-316`) resolve.
"""
import torch


def _cross(a, b):
    return torch.stack([a[..., 1] * b[..., 2] - a[..., 2] * b[..., 1],
                        a[..., 2] * b[..., 0] - a[..., 0] * b[..., 2],
                        a[..., 0] * b[..., 1] - a[..., 1] * b[..., 0]], dim=-1)


def _qmul(a, b):
    ax, ay, az, aw = a.unbind(-1)
    bx, by, bz, bw = b.unbind(-1)
    return torch.stack([aw * bx + ax * bw + ay * bz - az * by,
                        aw * by + ay * bw + az * bx - ax * bz,
                        aw * bz + az * bw + ax * by - ay * bx,
                        aw * bw - ax * bx - ay * by - az * bz], dim=-1)


def _qrot(q, p):
    uv = _cross(q[..., :3], p)
    uv = uv + uv
    return p + q[..., 3:4] * uv + _cross(q[..., :3], uv)


def _qinv(q):
    return torch.cat([-q[..., :3], q[..., 3:4]], dim=-1)


class SE3:
    manifold_dim = 6
    embedded_dim = 7

    def __init__(self, data):
        self.data = data

    # ---- tensor-like plumbing
    @property
    def device(self):
        return self.data.device

    @property
    def shape(self):
        return self.data.shape[:-1]

    def __getitem__(self, idx):
        return SE3(self.data[idx])

    def to(self, *a, **k):
        return SE3(self.data.to(*a, **k))

    def view(self, *dims):
        return SE3(self.data.view(*dims, 7))

    def clone(self):
        return SE3(self.data.clone())

    def vec(self):
        return self.data

    @staticmethod
    def Identity(*batch, device="cpu", dtype=torch.float32):
        d = torch.zeros(*batch, 7, device=device, dtype=dtype)
        d[..., 6] = 1.0
        return SE3(d)

    # ---- group ops
    def inv(self):
        t, q = self.data[..., :3], self.data[..., 3:]
        qi = _qinv(q)
        return SE3(torch.cat([-_qrot(qi, t), qi], dim=-1))

    def __mul__(self, other):
        t, q = self.data[..., :3], self.data[..., 3:]
        if isinstance(other, SE3):
            t2, q2 = other.data[..., :3], other.data[..., 3:]
            return SE3(torch.cat([t + _qrot(q, t2), _qmul(q, q2)], dim=-1))
        p = other
        if p.shape[-1] == 4:     # homogeneous [X,Y,Z,d]: R X + t d
            return torch.cat([_qrot(q, p[..., :3]) + t * p[..., 3:4], p[..., 3:4]], dim=-1)
        return _qrot(q, p) + t

    def act(self, p):
        return self * p

    def retr(self, a):
        """lietorch retraction: exp(a) * X (left perturbation), as used by src/geom/ba.py pose_retr."""
        return self.__class__.exp(a) * self

    def adjT(self, a):
        """Ad(X)^T a for a covector a = [a_tau, a_phi]."""
        t, q = self.data[..., :3], self.data[..., 3:]
        qi = _qinv(q)
        a_tau, a_phi = a[..., :3], a[..., 3:]
        return torch.cat([_qrot(qi, a_tau), _qrot(qi, a_phi - _cross(t, a_tau))], dim=-1)

    def matrix(self):
        t, q = self.data[..., :3], self.data[..., 3:]
        eye = torch.eye(3, device=self.data.device, dtype=self.data.dtype).expand(q.shape[:-1] + (3, 3))
        R = torch.stack([_qrot(q, eye[..., :, k]) for k in range(3)], dim=-1)
        top = torch.cat([R, t[..., None]], dim=-1)
        bot = torch.zeros(q.shape[:-1] + (1, 4), device=self.data.device, dtype=self.data.dtype)
        bot[..., 0, 3] = 1.0
        return torch.cat([top, bot], dim=-2)

    @staticmethod
    def exp(xi):
        tau, phi = xi[..., :3], xi[..., 3:]
        th2 = (phi * phi).sum(-1, keepdim=True)
        th = torch.sqrt(th2)
        small = th2 < 1e-8
        ths = torch.where(small, torch.ones_like(th), th)
        imag = torch.where(small, 0.5 - th2 / 48.0, torch.sin(0.5 * ths) / ths)
        real = torch.where(small, 1.0 - th2 / 8.0, torch.cos(0.5 * ths))
        q = torch.cat([imag * phi, real], dim=-1)
        a = torch.where(small, 0.5 - th2 / 24.0, (1 - torch.cos(ths)) / torch.where(small, torch.ones_like(th2), th2))
        b = torch.where(small, 1.0 / 6.0 - th2 / 120.0, (ths - torch.sin(ths)) / torch.where(small, torch.ones_like(th2), th2 * ths))
        c1 = _cross(phi, tau)
        t = tau + a * c1 + b * _cross(phi, c1)
        return SE3(torch.cat([t, q], dim=-1))

    def log(self):
        t, q = self.data[..., :3], self.data[..., 3:]
        q = torch.where(q[..., 3:4] < 0, -q, q)
        v, w = q[..., :3], q[..., 3:4]
        n = v.norm(dim=-1, keepdim=True)
        small = n < 1e-6
        ns = torch.where(small, torch.ones_like(n), n)
        th = 2.0 * torch.atan2(n, w)
        phi = torch.where(small, 2.0 * v / w, th * v / ns)
        th2 = (phi * phi).sum(-1, keepdim=True)
        thn = torch.sqrt(th2)
        smallt = th2 < 1e-8
        tt = torch.where(smallt, torch.ones_like(thn), thn)
        # V^-1 = I - 1/2 [phi]x + c [phi]x^2 ,  c = (1 - th cos(th/2) / (2 sin(th/2))) / th^2
        c = torch.where(smallt, torch.full_like(th2, 1.0 / 12.0),
                        (1.0 - 0.5 * tt * torch.cos(0.5 * tt) / torch.sin(0.5 * tt)) / torch.where(smallt, torch.ones_like(th2), th2))
        c1 = _cross(phi, t)
        tau = t - 0.5 * c1 + c * _cross(phi, c1)
        return torch.cat([tau, phi], dim=-1)


class Sim3(SE3):
    """Only referenced by dead code paths of the reference (projective_ops.py:73-82,137)."""


def cat(group_objects, dim=0):
    return SE3(torch.cat([g.data for g in group_objects], dim=dim))
