"""SE3 helpers of the CPU oracle (TEST INFRASTRUCTURE, see oracle/__init__.py).

Pose convention (reference src/depth_video.py:43): pose = [tx,ty,tz, qx,qy,qz,qw], world->camera.
Each function cites the reference lines it restates; `droid_kernels.cu` means
/root/reference/src/lib/droid_kernels.cu.
"""
import torch


def cross(a, b):
    return torch.stack([
        a[..., 1] * b[..., 2] - a[..., 2] * b[..., 1],
        a[..., 2] * b[..., 0] - a[..., 0] * b[..., 2],
        a[..., 0] * b[..., 1] - a[..., 1] * b[..., 0],
    ], dim=-1)


def act_so3(q, X):
    """droid_kernels.cu:58-68 (actSO3): rotate X[...,3] by unit quaternion q[...,4] (xyzw)."""
    uv = 2.0 * cross(q[..., :3], X)
    return X + q[..., 3:4] * uv + cross(q[..., :3], uv)


def act_se3(t, q, X):
    """droid_kernels.cu:70-77 (actSE3): homogeneous action, X[...,4] = [X,Y,Z,d]."""
    Y3 = act_so3(q, X[..., :3]) + X[..., 3:4] * t
    return torch.cat([Y3, X[..., 3:4]], dim=-1)


def adj_se3(t, q, X):
    """droid_kernels.cu:79-94 (adjSE3): dual adjoint applied to a 6-covector X[...,6]."""
    qinv = torch.cat([-q[..., :3], q[..., 3:4]], dim=-1)
    Y0 = act_so3(qinv, X[..., 0:3])
    Y1 = act_so3(qinv, X[..., 3:6])
    u = torch.stack([
        t[..., 2] * X[..., 1] - t[..., 1] * X[..., 2],
        t[..., 0] * X[..., 2] - t[..., 2] * X[..., 0],
        t[..., 1] * X[..., 0] - t[..., 0] * X[..., 1],
    ], dim=-1)
    v = act_so3(qinv, u)
    return torch.cat([Y0, Y1 + v], dim=-1)


def rel_se3(ti, qi, tj, qj):
    """droid_kernels.cu:96-107 (relSE3): Gij = Gj * Gi^-1 -> (tij, qij)."""
    qij = torch.stack([
        -qj[..., 3] * qi[..., 0] + qj[..., 0] * qi[..., 3] - qj[..., 1] * qi[..., 2] + qj[..., 2] * qi[..., 1],
        -qj[..., 3] * qi[..., 1] + qj[..., 1] * qi[..., 3] - qj[..., 2] * qi[..., 0] + qj[..., 0] * qi[..., 2],
        -qj[..., 3] * qi[..., 2] + qj[..., 2] * qi[..., 3] - qj[..., 0] * qi[..., 1] + qj[..., 1] * qi[..., 0],
        qj[..., 3] * qi[..., 3] + qj[..., 0] * qi[..., 0] + qj[..., 1] * qi[..., 1] + qj[..., 2] * qi[..., 2],
    ], dim=-1)
    tij = tj - act_so3(qij, ti)
    return tij, qij


def quat_mul(a, b):
    """Hamilton product a (x) b, xyzw (lietorch SO3 group product, used by SE3.__mul__)."""
    ax, ay, az, aw = a.unbind(-1)
    bx, by, bz, bw = b.unbind(-1)
    return torch.stack([
        aw * bx + ax * bw + ay * bz - az * by,
        aw * by + ay * bw + az * bx - ax * bz,
        aw * bz + az * bw + ax * by - ay * bx,
        aw * bw - ax * bx - ay * by - az * bz,
    ], dim=-1)


def se3_inv(t, q):
    """lietorch SE3.inv(): (q^-1, -(q^-1 * t))."""
    qinv = torch.cat([-q[..., :3], q[..., 3:4]], dim=-1)
    return -act_so3(qinv, t), qinv


def se3_mul(t1, q1, t2, q2):
    """lietorch SE3 product: (q1 q2, t1 + q1 * t2)."""
    return t1 + act_so3(q1, t2), quat_mul(q1, q2)


def exp_so3(phi):
    """droid_kernels.cu:110-132 (expSO3)."""
    theta_sq = (phi * phi).sum(-1)
    theta_p4 = theta_sq * theta_sq
    theta = torch.sqrt(theta_sq)
    small = theta_sq < 1e-8
    safe = torch.where(small, torch.ones_like(theta), theta)
    imag = torch.where(small, 0.5 - (1.0 / 48.0) * theta_sq + (1.0 / 3840.0) * theta_p4,
                       torch.sin(0.5 * safe) / safe)
    real = torch.where(small, 1.0 - (1.0 / 8.0) * theta_sq + (1.0 / 384.0) * theta_p4,
                       torch.cos(0.5 * safe))
    return torch.cat([imag[..., None] * phi, real[..., None]], dim=-1)


def exp_se3(xi):
    """droid_kernels.cu:147-175 (expSE3): xi = [tau, phi] -> (t, q)."""
    tau, phi = xi[..., :3], xi[..., 3:]
    q = exp_so3(phi)
    theta_sq = (phi * phi).sum(-1)
    theta = torch.sqrt(theta_sq)
    big = theta > 1e-4
    safe_sq = torch.where(big, theta_sq, torch.ones_like(theta_sq))
    safe = torch.where(big, theta, torch.ones_like(theta))
    a = (1 - torch.cos(safe)) / safe_sq
    b = (safe - torch.sin(safe)) / (safe * safe_sq)
    c1 = cross(phi, tau)
    c2 = cross(phi, c1)
    t = tau + torch.where(big[..., None], a[..., None] * c1, torch.zeros_like(tau))
    t = t + torch.where(big[..., None], b[..., None] * c2, torch.zeros_like(tau))
    return t, q


def retr_se3(xi, t, q):
    """droid_kernels.cu:877-895 (retrSE3): left retraction exp(xi) * (t, q)."""
    dt, dq = exp_se3(xi)
    q1 = quat_mul(dq, q)
    t1 = act_so3(dq, t) + dt
    return t1, q1
