"""Synthetic fusion input of SURVEY.md section 8(d) (TEST INFRASTRUCTURE / bench input, not product code):
`frames` depth + colour images of an analytic room -- an axis-aligned 6 x 4 x 3 m box with a sphere in it -- seen from a
camera that moves on a seeded smooth closed loop INSIDE the room, so that a DrFusion map keeps growing while the loop is
walked (BASELINE configs[3]: "TSDF fusion of 1000 predicted depth maps ... + raycast").  Depth is z-depth in metres with
`drop_fraction` of the pixels zeroed (the MVSNet edge filter's invalid pixels), colour is a procedural BGR texture of the
hit point.  Generated with torch on whatever device is asked for (on the GPU box: straight into HBM, 2.1 MB per frame)."""
import numpy as np


def loop_poses(n, seed=0):
    """(n, 4, 4) float32 row-major cam-to-world on a closed loop inside the room, looking mostly outwards."""
    rng = np.random.RandomState(seed)
    ph = rng.uniform(0, 2 * np.pi, 3)
    out = np.zeros((n, 4, 4), np.float32)
    for i in range(n):
        a = 2 * np.pi * i / n
        pos = np.array([1.6 * np.cos(a), 0.9 * np.sin(a), 0.35 * np.sin(2 * a + ph[0])])
        yaw = a + 0.5 * np.sin(3 * a + ph[1])          # look roughly along the outward radius, sweeping
        pitch = 0.25 * np.sin(2 * a + ph[2])
        f = np.array([np.cos(yaw) * np.cos(pitch), np.sin(yaw) * np.cos(pitch), np.sin(pitch)])   # camera +z
        up = np.array([0.0, 0.0, 1.0])
        r = np.cross(f, up); r /= np.linalg.norm(r)      # camera +x
        d = np.cross(f, r)                               # camera +y (down)
        T = np.eye(4)
        T[:3, 0], T[:3, 1], T[:3, 2], T[:3, 3] = r, d, f, pos
        out[i] = T.astype(np.float32)
    return out


def render_frames(poses, height=480, width=640, device="cpu", seed=0, drop_fraction=0.025, chunk=50):
    """-> dict(bgr uint8 (n,H,W,3), depth float32 (n,H,W), fx, fy, cx, cy) as torch tensors on `device`."""
    import torch
    n = len(poses)
    f = 0.78125 * width
    cx, cy = (width - 1) / 2.0, (height - 1) / 2.0
    dev = torch.device(device)
    v, u = torch.meshgrid(torch.arange(height, dtype=torch.float32, device=dev), torch.arange(width, dtype=torch.float32, device=dev), indexing="ij")
    dcam = torch.stack([(u - cx) / f, (v - cy) / f, torch.ones_like(u)], -1)           # z = 1: ray parameter == z-depth
    lo = torch.tensor([-3.0, -2.0, -1.5], device=dev)
    hi = torch.tensor([3.0, 2.0, 1.5], device=dev)
    sc, sr = torch.tensor([2.0, 0.8, -0.6], device=dev), 0.7
    gen = torch.Generator(device=dev); gen.manual_seed(seed)
    bgr = torch.empty((n, height, width, 3), dtype=torch.uint8, device=dev)
    depth = torch.empty((n, height, width), dtype=torch.float32, device=dev)
    P = torch.as_tensor(np.asarray(poses, np.float32), device=dev)
    for i0 in range(0, n, chunk):
        T = P[i0:i0 + chunk]
        R, o = T[:, :3, :3], T[:, :3, 3]
        d = torch.einsum("nij,hwj->nhwi", R, dcam)
        oo = o[:, None, None, :]
        # box interior: exit distance along every axis
        tx = torch.where(d > 0, (hi - oo) / d.clamp_min(1e-9), (lo - oo) / d.clamp_max(-1e-9))
        t = tx.min(dim=-1).values
        # sphere
        oc = oo - sc
        b = (oc * d).sum(-1)
        a = (d * d).sum(-1)
        c = (oc * oc).sum(-1) - sr * sr
        disc = b * b - a * c
        ts = (-b - disc.clamp_min(0).sqrt()) / a
        hit = (disc > 0) & (ts > 0.05) & (ts < t)
        t = torch.where(hit, ts, t)
        X = oo + d * t[..., None]
        tex = [0.5 + 0.5 * torch.sin(7.0 * X[..., 0] * (k + 1) + 5.0 * X[..., 1] + 3.0 * (2 - k) * X[..., 2] + k) *
               torch.cos(4.0 * X[..., 1] * (k + 1) - 2.0 * X[..., 2]) for k in range(3)]
        bgr[i0:i0 + chunk] = (torch.stack(tex, -1).clamp(0, 1) * 255.0 + 0.5).to(torch.uint8)
        drop = torch.rand(t.shape, generator=gen, device=dev) < drop_fraction
        depth[i0:i0 + chunk] = torch.where(drop, torch.zeros_like(t), t)
    return dict(bgr=bgr, depth=depth, fx=float(f), fy=float(f), cx=float(cx), cy=float(cy), height=height, width=width)
