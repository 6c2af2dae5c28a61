"""Absolute trajectory error with Sim(3) alignment -- the number the reference reports at the end of a run
(src/slam.py:343-360: evo `main_ape.ape(traj_ref, traj_est, pose_relation=translation_part, align=True,
correct_scale=True)`).  evo is not in this image; this is the published Umeyama (1991) least-squares similarity
alignment followed by the RMSE of the translation residuals, i.e. the same mathematics (evo.core.geometry.umeyama_alignment
+ APE translation part)."""
import numpy as np


def umeyama_alignment(x, y, with_scale=True):
    """Least-squares similarity transform y ~ c R x + t for point sets x, y [3, n].  Returns (R [3,3], t [3], c)."""
    x, y = np.asarray(x, dtype=np.float64), np.asarray(y, dtype=np.float64)
    assert x.shape == y.shape and x.shape[0] == 3
    n = x.shape[1]
    mx, my = x.mean(axis=1), y.mean(axis=1)
    xc, yc = x - mx[:, None], y - my[:, None]
    sx = (xc * xc).sum() / n                                   # variance of x
    cov = yc @ xc.T / n
    U, D, Vt = np.linalg.svd(cov)
    if np.count_nonzero(D > np.finfo(D.dtype).eps) < 2:
        raise ValueError("degenerate covariance rank, Umeyama alignment is not possible")
    S = np.eye(3)
    if np.linalg.det(U) * np.linalg.det(Vt) < 0.0:             # keep a proper rotation
        S[2, 2] = -1.0
    R = U @ S @ Vt
    c = float(np.trace(np.diag(D) @ S) / sx) if with_scale else 1.0
    t = my - c * (R @ mx)
    return R, t, c


def ate_rmse(est_xyz, ref_xyz, align=True, correct_scale=True):
    """RMSE of |ref - aligned(est)| over positions [n, 3]; returns (rmse, dict with the alignment and statistics)."""
    est, ref = np.asarray(est_xyz, dtype=np.float64), np.asarray(ref_xyz, dtype=np.float64)
    assert est.shape == ref.shape and est.ndim == 2 and est.shape[1] == 3
    R, t, c = np.eye(3), np.zeros(3), 1.0
    if align:
        R, t, c = umeyama_alignment(est.T, ref.T, with_scale=correct_scale)
    err = np.linalg.norm(ref - (c * (R @ est.T).T + t), axis=1)
    T = np.eye(4)
    T[:3, :3], T[:3, 3] = c * R, t
    return float(np.sqrt((err ** 2).mean())), {"rotation": R, "translation": t, "scale": c, "mean": float(err.mean()),
                                               "median": float(np.median(err)), "max": float(err.max()),
                                               "alignment_transformation_sim3": T}
