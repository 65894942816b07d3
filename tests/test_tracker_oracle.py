"""CPU: the dense-coarse-tracker oracle (oracle/tracker_oracle.c).  The reference ships neither tests nor the fixtures
its own driver reads (cuda_coarse_tracker/src/main.cu wants cct_data/*.npy), so the restatement is pinned by
hand-derived known answers of the reference's formulas plus domain properties."""
import numpy as np

from synth import scene
from oracle.tracker_oracle import TrackerOracle

F = np.float32


def ramp_image(w, h, a, b, c):
    """I(u, v) = a*u + b*v + c with exact gradients: bilinear interpolation reproduces it exactly (up to fp32)."""
    v, u = np.meshgrid(np.arange(h, dtype=np.float32), np.arange(w, dtype=np.float32), indexing="ij")
    I = (F(a) * u + F(b) * v + F(c)).astype(np.float32)
    return np.ascontiguousarray(np.stack([I, np.full_like(I, a), np.full_like(I, b)], axis=2))


def test_known_answer_single_point_translation():
    """One point, R = I, t = (tx, 0, 0): p = Ki*(x,y,1) + t*id, (u, v) = p.xy / p.z, pixel = K*(u,v,1)
    (cuda_coarse_tracker_private.cu:96-116); residual / Huber weight / energy per :181-207; J per :309-318."""
    w, h, fx, fy, cx, cy = 64, 48, 50.0, 40.0, 31.5, 23.5
    a_, b_, c_ = 0.5, -0.25, 100.0
    o = TrackerOracle(w, h, 9.0, 20.0)
    o.setK(fx, fy, cx, cy)
    x, y, idp, col = 20.0, 30.0, 0.5, 117.0
    o.setReference([x], [y], [idp], [col], 1.0, [0.0, 0.0])
    o.setNew(ramp_image(w, h, a_, b_, c_))
    T = np.eye(4)
    T[0, 3] = 0.2
    out, sums = o.calcRes(T, 1.0, [0.0, 0.0], 20.0)
    # by hand, in double (the fp32 path agrees to ~1e-6)
    px, py = (x - cx) / fx + 0.2 * idp, (y - cy) / fy
    Ku, Kv = fx * px + cx, fy * py + cy
    I = a_ * Ku + b_ * Kv + c_
    r = I - col
    assert abs(r) > 9.0 and abs(r) < 20.0  # Huber branch, not saturated
    hw = 9.0 / abs(r)
    E = hw * r * r * (2 - hw)
    assert np.isclose(out[0], E, rtol=1e-5) and out[1] == 1 and out[5] == 0
    # point index 0 contributes to the shift statistics (i % 32 == 0): translation-only shift, +t and -t
    sT = 2 * (fx * 0.2 * idp) ** 2
    assert np.isclose(out[2], sT / 2, rtol=1e-5) and np.isclose(out[4], sT / 2, rtol=1e-5) and sums[6] == 2
    wu, wv, wdx, wdy, wid, wres, ww = [float(k[0]) for k in o.warped()]
    assert np.isclose(wu, px, rtol=1e-6) and np.isclose(wv, py, rtol=1e-6) and np.isclose(wid, idp, rtol=1e-6)
    assert wdx == F(a_) and wdy == F(b_) and np.isclose(wres, r, rtol=1e-5) and np.isclose(ww, hw, rtol=1e-5)
    H, b, raw = o.calcG(1.0, [0.0, 0.0])
    dx, dy = a_ * fx, b_ * fy
    J = np.array([idp * dx, idp * dy, -idp * (px * dx + py * dy), -(px * py * dx + dy + dy * py * py),
                  px * py * dy + dx + dx * px * px, px * dy - py * dx, 1.0 * (0.0 - col), -1.0, r])
    full = hw * np.outer(J, J)
    s = np.array([1, 1, 1, 0.5, 0.5, 0.5, 10, 1000.0])
    assert np.allclose(H, full[:8, :8] * np.outer(s, s), rtol=2e-5)
    assert np.allclose(b, full[:8, 8] * s, rtol=2e-5)
    assert np.allclose(raw, full[np.triu_indices(9)], rtol=2e-5)


def test_saturated_out_of_bounds_and_exposure():
    w, h = 64, 48
    o = TrackerOracle(w, h, 9.0, 20.0)
    o.setK(50.0, 40.0, 31.5, 23.5)
    # point 0: residual above the cutoff -> maxEnergy, saturated; point 1: projects outside [2, w-3] -> no term;
    # point 2: non-positive new idepth cannot happen with id > 0 and z > 0, so use an exact-match point (residual 0)
    img = ramp_image(w, h, 0.0, 0.0, 100.0)
    o.setReference([20, 1, 30], [20, 20, 25], [0.5, 0.5, 0.25], [10.0, 100.0, 100.0], 1.0, [0.0, 0.0])
    o.setNew(img)
    out, sums = o.calcRes(np.eye(4), 1.0, [0.0, 0.0], 20.0)
    maxE = 2 * 9.0 * 20.0 - 81.0
    assert sums[1] == 2 and sums[2] == 1 and sums[3] == 1 and np.isclose(sums[0], maxE)
    assert np.isclose(out[5], 0.5)
    wres = o.warped()[5]
    assert wres[0] == 0 and wres[1] == 0 and wres[2] == 0  # saturated / skipped points leave zeros; exact match has r = 0
    assert o.warped()[6][2] == 1.0  # weight 1 below the Huber threshold
    # AffLight::fromToVecExposure (cuda_coarse_tracker.cpp:40-49): a = exp(aT - aF) * eT / eF, b = bT - a * bF
    o.setReference([30], [25], [0.25], [40.0], 2.0, [0.1, 3.0])
    out, _ = o.calcRes(np.eye(4), 4.0, [0.3, 7.0], 1000.0)
    a = np.exp(0.3 - 0.1) * 4.0 / 2.0
    r = 100.0 - (a * 40.0 + (7.0 - a * 3.0))
    assert np.isclose(o.warped()[5][0], r, rtol=1e-5)


def test_true_motion_has_lower_energy_and_gauss_newton_descends():
    p = scene.make_tracking_pair(120, 160, seed=3, sparse_fraction=0.2)
    o = TrackerOracle(160, 120, 9.0, 20.0)
    o.setK(p["fx"], p["fy"], p["cx"], p["cy"])
    o.setReference(p["pc_u"], p["pc_v"], p["pc_idepth"], p["pc_color"], 1.0, [0.0, 0.0])
    o.setNew(p["dI_new"])
    e_true = o.calcRes(p["refToNew"], 1.0, [0.0, 0.0], 20.0)[0]
    e_id = o.calcRes(np.eye(4), 1.0, [0.0, 0.0], 20.0)[0]
    assert e_true[0] / e_true[1] < 0.5 * e_id[0] / e_id[1] and e_true[5] < 0.2 < e_id[5]
    # one damped Gauss-Newton step from a small perturbation of the true pose lowers the mean energy
    T0 = p["refToNew"].copy()
    T0[:3, 3] += [0.004, -0.003, 0.002]
    e0 = o.calcRes(T0, 1.0, [0.0, 0.0], 20.0)[0]
    H, b, _ = o.calcG(1.0, [0.0, 0.0])
    s = np.array([1, 1, 1, 0.5, 0.5, 0.5, 10, 1000.0])
    inc = np.linalg.solve(H[:6, :6] + 1e-3 * np.diag(np.diag(H[:6, :6])), -b[:6]) * s[:6]  # CoarseTracker.cpp:820-850 (pose part)
    v_, w_ = inc[:3], inc[3:6]  # Sophus tangent order: translation first, then rotation; left-multiplied (SE3::exp(inc) * T)
    W = np.array([[0, -w_[2], w_[1]], [w_[2], 0, -w_[0]], [-w_[1], w_[0], 0]])
    dT = np.eye(4)
    dT[:3, :3] += W
    dT[:3, 3] = v_
    e1 = o.calcRes(dT @ T0, 1.0, [0.0, 0.0], 20.0)[0]
    assert e1[0] / e1[1] < e0[0] / e0[1]


def test_append_dense_identity_occlusion_and_mask():
    w, h = 40, 30
    o = TrackerOracle(w, h, 9.0, 20.0, n_max=w * h)
    o.setK(30.0, 30.0, 19.5, 14.5)
    depth = np.full((h, w), 2.0, np.float32)
    depth[10, 12] = 0.0   # invalid source pixel
    dIp = np.zeros((h, w, 3), np.float32)
    dIp[..., 0] = np.arange(h * w, dtype=np.float32).reshape(h, w)
    o.setReference([5.0], [5.0], [0.5], [1.0], 1.0, [0, 0])
    n, proj = o.appendDenseReference(depth, np.eye(3), np.zeros(3), 1, True, None, dIp)
    # identity warp: pixel (x, y) lands on itself; targets outside [3, w-4] x [3, h-4] are rejected (CoarseTracker.cpp:691-692)
    exp = np.full((h, w), -1.0, np.float32)
    exp[3:h - 3, 3:w - 3] = 2.0
    exp[10, 12] = -1.0
    assert np.array_equal(proj, exp)
    assert n == 1 + (w - 6) * (h - 6) - 1
    u, v, idp, col = o.points()
    assert (u[0], v[0]) == (5.0, 5.0) and (u[1], v[1]) == (3.0, 3.0) and idp[1] == 0.5 and col[1] == dIp[3, 3, 0]
    assert np.all(np.diff(v[1:] * w + u[1:]) > 0)  # row-major order
    # occlusion: x' = x/2 -> pixels 2k and 2k+1 of a row collide; the nearer one wins
    o.setReference([], [], [], [], 1.0, [0, 0])
    depth2 = np.full((h, w), 3.0, np.float32)
    depth2[:, 1::2] = 1.5
    Kh = np.diag([0.5, 1.0, 1.0]).astype(np.float32)
    _, proj2 = o.appendDenseReference(depth2, Kh, np.zeros(3), 1, True, None, dIp)
    assert set(np.unique(proj2[proj2 > 0])) == {1.5}
    # sparse mask: pixels that already have a sparse idepth are not appended unless dense_only
    id0 = np.zeros((h, w), np.float32)
    id0[5:10, 5:10] = 0.7
    o.setReference([], [], [], [], 1.0, [0, 0])
    n3, _ = o.appendDenseReference(depth, np.eye(3), np.zeros(3), 1, False, id0, dIp)
    assert n3 == (w - 6) * (h - 6) - 1 - 25
    # lattice step 2: only even source pixels are warped
    o.setReference([], [], [], [], 1.0, [0, 0])
    n4, proj4 = o.appendDenseReference(depth, np.eye(3), np.zeros(3), 2, True, None, dIp)
    assert (proj4[3::2, 3::2] == -1).all() and n4 == np.count_nonzero(proj4[2:h - 2, 2:w - 2] > 0)
