"""Seeded synthetic inputs for the DrMvsnet / DrFusion hot path (TEST INFRASTRUCTURE).

No dataset is reachable (no network, Replica absent), so every test, golden
fixture and bench run draws its input from here:

* `make_window`  -- a keyframe window exactly as TANDEM hands it to
  `DrMvsnet::CallAsync` (dr_mvsnet.h:42-52): `view_num` u8 BGR images (H,W,3),
  one full-resolution 3x3 K, `view_num` row-major 4x4 cam_to_world, with the
  reference view at index view_num-2 (FullSystem.cpp:1127).
* `make_scans`   -- a sequence of (bgr, depth, pose) scans of an analytic room
  exactly as `DrFusion::IntegrateScanAsync` consumes them (dr_fusion.h:46).

Pure numpy; deterministic for a given seed.
"""
import numpy as np


def _texture(X, Y, Z, seed, n=20):
    """Band-limited procedural RGB texture evaluated at world points (values in [0,1])."""
    rng = np.random.RandomState(seed)
    out = np.zeros(X.shape + (3,), np.float64)
    for c in range(3):
        f = rng.uniform(4.0, 110.0, size=(n, 3)) * rng.choice([-1, 1], size=(n, 3))
        ph = rng.uniform(0, 2 * np.pi, size=n)
        a = rng.uniform(0.3, 1.0, size=n)
        acc = np.zeros(X.shape, np.float64)
        for i in range(n):
            acc += a[i] * np.sin(f[i, 0] * X + f[i, 1] * Y + f[i, 2] * Z + ph[i])
        out[..., c] = 0.5 + 0.5 * acc / np.sqrt((a ** 2).sum() * 0.5) / 2.5
    return np.clip(out, 0.0, 1.0)


def _pose(rx, ry, rz, t):
    cx, sx, cy, sy, cz, sz = np.cos(rx), np.sin(rx), np.cos(ry), np.sin(ry), np.cos(rz), np.sin(rz)
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    T = np.eye(4)
    T[:3, :3] = Rz @ Ry @ Rx
    T[:3, 3] = t
    return T


def _render(K, c2w, H, W, seed, terms=20):
    """Ray-cast a tilted back plane plus a nearer rectangular slab; returns (rgb float HxWx3, z-depth HxW)."""
    v, u = np.meshgrid(np.arange(H, dtype=np.float64), np.arange(W, dtype=np.float64), indexing="ij")
    d_cam = np.stack([(u - K[0, 2]) / K[0, 0], (v - K[1, 2]) / K[1, 1], np.ones_like(u)], -1)
    R, o = c2w[:3, :3], c2w[:3, 3]
    d_w = d_cam @ R.T
    # plane 0: n.(X) = c  (tilted back wall around z ~ 2.4)
    n0, c0 = np.array([0.18, -0.10, 1.0]), 2.4
    t0 = (c0 - o @ n0) / (d_w @ n0)
    # plane 1: z = 1.3 slab, |x|<0.45, |y|<0.3
    t1 = (1.3 - o[2]) / d_w[..., 2]
    P1 = o + d_w * t1[..., None]
    hit1 = (np.abs(P1[..., 0] - 0.05) < 0.45) & (np.abs(P1[..., 1] + 0.02) < 0.30) & (t1 > 0)
    t = np.where(hit1, t1, t0)
    P = o + d_w * t[..., None]
    rgb = np.where(hit1[..., None], _texture(P[..., 0], P[..., 1], P[..., 2], seed + 1, terms),
                   _texture(P[..., 0], P[..., 1], P[..., 2], seed, terms))
    return rgb, t  # d_cam.z == 1 so t is z-depth in the camera


def make_window(height=480, width=640, view_num=7, seed=0):
    """Returns dict(bgrs [V] uint8 (H,W,3), K float32 (3,3), c2ws float32 (V,4,4), ref_index,
    depth_min, depth_max, gt_depth float32 (H,W) of the reference view)."""
    f = 0.78125 * width  # 500 px at 640
    K = np.array([[f, 0, (width - 1) / 2.0], [0, f, (height - 1) / 2.0], [0, 0, 1]], np.float64)
    rng = np.random.RandomState(seed + 1234)
    ref_index = view_num - 2
    bgrs, c2ws, gt = [], [], None
    for i in range(view_num):
        k = i - ref_index
        if k == 0:
            T = _pose(0, 0, 0, [0, 0, 0])
        else:
            T = _pose(*(rng.uniform(-0.02, 0.02, 3)),
                      [0.06 * k + rng.uniform(-0.01, 0.01), 0.025 * ((-1) ** i) * abs(k), rng.uniform(-0.02, 0.02)])
        rgb, z = _render(K, T, height, width, seed)
        if k == 0:
            gt = z.astype(np.float32)
        img = np.floor(rgb[..., ::-1] * 255.0 + 0.5).astype(np.uint8)  # BGR, u8 as TANDEM feeds it
        bgrs.append(np.ascontiguousarray(img))
        c2ws.append(T.astype(np.float32))
    return dict(bgrs=bgrs, K=K.astype(np.float32), c2ws=np.stack(c2ws), ref_index=ref_index,
                depth_min=0.5, depth_max=5.0, gt_depth=gt, height=height, width=width, view_num=view_num)


def make_scans(n, height=480, width=640, seed=0, drop_fraction=0.025, texture_terms=20):
    """n scans of the analytic scene from a smooth seeded camera loop.
    Returns dict(K (fx,fy,cx,cy), scans=[(bgr u8 HxWx3, depth f32 HxW, pose f32 4x4 row-major c2w)])."""
    f = 0.78125 * width
    K = np.array([[f, 0, (width - 1) / 2.0], [0, f, (height - 1) / 2.0], [0, 0, 1]], np.float64)
    rng = np.random.RandomState(seed + 99)
    scans = []
    for i in range(n):
        a = 2 * np.pi * i / max(n, 8)
        T = _pose(0.05 * np.sin(a), 0.12 * np.sin(a * 0.5), 0.02 * np.cos(a),
                  [0.25 * np.sin(a), 0.10 * np.cos(a), 0.15 * np.sin(2 * a)])
        rgb, z = _render(K, T, height, width, seed, texture_terms)
        depth = z.astype(np.float32)
        drop = rng.rand(height, width) < drop_fraction  # mimics the MVSNet edge filter's zeros
        depth[drop] = 0.0
        bgr = np.floor(rgb[..., ::-1] * 255.0 + 0.5).astype(np.uint8)
        scans.append((np.ascontiguousarray(bgr), np.ascontiguousarray(depth), T.astype(np.float32)))
    return dict(fx=float(f), fy=float(f), cx=float(K[0, 2]), cy=float(K[1, 2]), height=height, width=width,
                scans=scans)


def _dI(rgb):
    """DSO-style image triple (I, dx, dy) per pixel, I in [0, 255] (FrameHessian::makeImages): central differences."""
    I = (255.0 * rgb.mean(axis=2)).astype(np.float32)
    dx, dy = np.zeros_like(I), np.zeros_like(I)
    dx[:, 1:-1] = 0.5 * (I[:, 2:] - I[:, :-2])
    dy[1:-1, :] = 0.5 * (I[2:, :] - I[:-2, :])
    return np.ascontiguousarray(np.stack([I, dx, dy], axis=2))


def make_tracking_pair(height=480, width=640, seed=0, sparse_fraction=0.03, motion=1.0):
    """A reference frame and a new frame of the analytic scene for the dense coarse tracker:
    dict(K (fx,fy,cx,cy), dI_ref / dI_new (H,W,3 f32), depth_ref / depth_new (H,W f32), c2w_ref / c2w_new (4x4 f64),
    refToNew (4x4 f64), sparse points pc_u/pc_v/pc_idepth/pc_color of the reference, idepth0 (H,W; > 0 at sparse points))."""
    f = 0.78125 * width
    K = np.array([[f, 0, (width - 1) / 2.0], [0, f, (height - 1) / 2.0], [0, 0, 1]], np.float64)
    Tr = _pose(0.01, -0.02, 0.005, [0.02, 0.01, 0.0])
    Tn = _pose(0.01 + 0.012 * motion, -0.02 + 0.02 * motion, 0.005 - 0.008 * motion,
               [0.02 + 0.04 * motion, 0.01 - 0.025 * motion, 0.03 * motion])
    rgb_r, z_r = _render(K, Tr, height, width, seed, 12)
    rgb_n, z_n = _render(K, Tn, height, width, seed, 12)
    dIr, dIn = _dI(rgb_r), _dI(rgb_n)
    rng = np.random.RandomState(seed + 7)
    pick = (rng.rand(height, width) < sparse_fraction)
    pick[:3, :] = pick[-3:, :] = False
    pick[:, :3] = pick[:, -3:] = False
    ys, xs = np.nonzero(pick)
    idepth = (1.0 / z_r).astype(np.float32)
    idepth0 = np.zeros((height, width), np.float32)
    idepth0[ys, xs] = idepth[ys, xs]
    return dict(fx=float(f), fy=float(f), cx=float(K[0, 2]), cy=float(K[1, 2]), height=height, width=width,
                dI_ref=dIr, dI_new=dIn, depth_ref=z_r.astype(np.float32), depth_new=z_n.astype(np.float32),
                c2w_ref=Tr, c2w_new=Tn, refToNew=np.linalg.inv(Tn) @ Tr,
                pc_u=xs.astype(np.float32), pc_v=ys.astype(np.float32), pc_idepth=idepth[ys, xs].copy(),
                pc_color=dIr[ys, xs, 0].copy(), idepth0=idepth0)
