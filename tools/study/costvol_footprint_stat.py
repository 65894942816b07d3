"""Go / no-go statistic for a view-outer / plane-inner cost-volume sweep.

From the geometry of the committed headline fixture (tests/golden/mvsnet_v7_480x640_headline.npz: 640x480, 7 views, planes 48/32/8,
depth range 0.01 .. 10) count, per stage, over all (pixel, source view, plane d -> d + 1) steps, how the upper-left tap (ix, iy) of the
bilinear footprint moves: unchanged / one texel along one axis / anything else; and from that how many of the four 16-byte gathers of
plane d + 1 a lane that still holds plane d's 2 x 2 footprint would have to issue.  Needs no GPU and nothing under /root/reference.
"""
import json, os, sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def up2(prev):  # F.interpolate(scale 2, bilinear, align_corners=False)
    hp, wp = prev.shape
    y = np.arange(2 * hp, dtype=np.float32); x = np.arange(2 * wp, dtype=np.float32)
    sy = np.maximum(0.5 * (y + 0.5) - 0.5, 0).astype(np.float32); sx = np.maximum(0.5 * (x + 0.5) - 0.5, 0).astype(np.float32)
    y0 = sy.astype(int); x0 = sx.astype(int); y1 = np.minimum(y0 + 1, hp - 1); x1 = np.minimum(x0 + 1, wp - 1)
    ly = (sy - y0)[:, None]; lx = (sx - x0)[None, :]
    return ((1 - ly) * ((1 - lx) * prev[y0][:, x0] + lx * prev[y0][:, x1]) + ly * ((1 - lx) * prev[y1][:, x0] + lx * prev[y1][:, x1])).astype(np.float32)


def main():
    z = np.load(os.path.join(ROOT, "tests", "golden", "mvsnet_v7_480x640_headline.npz"))
    from tandem_amd import weights as Wt
    meta, _ = Wt.read_blob(os.path.join(ROOT, "weights", "tandem_va.tdmw"))
    ratio = meta["interval_ratio"]
    K, c2ws, ref = z["K"].astype(np.float64), z["c2ws"].astype(np.float64), int(z["ref_index"])
    order = [ref] + [i for i in range(len(c2ws)) if i != ref]
    dmin, dmax, planes = float(z["depth_min"]), float(z["depth_max"]), [int(p) for p in z["planes"]]
    H, W = z["gt_depth"].shape
    base = (dmax - dmin) / (planes[0] - 1)
    out = {}
    for s in range(3):
        sc = 4 >> s
        h, w, D = H // sc, W // sc, planes[s]
        Ks = K.copy(); Ks[:2] /= sc
        if s == 0:
            dep = (dmin + base * np.arange(D))[:, None, None] * np.ones((1, h, w))
        else:
            cur = up2(z["ref_s%d_depth_dense" % s])
            delta = ratio[s - 1] * base
            lo = np.maximum(cur - D / 2 * delta, 1e-3)
            dep = lo[None] + (lo + D * delta - lo)[None] * (np.arange(D) / D)[:, None, None]
        P = lambda c2w: np.vstack([Ks @ np.linalg.inv(c2w)[:3], [0, 0, 0, 1]])
        Pref_inv = np.linalg.inv(P(c2ws[order[0]]))
        yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
        tot = dict(steps=0, same=0, one=0, far=0, loads_needed=0, inside_steps=0, outside_both=0)
        for v in order[1:]:
            M = P(c2ws[v]) @ Pref_inv
            r = M[:3, :3] @ np.stack([xx.ravel(), yy.ravel(), np.ones(h * w)])
            p = r[None] * dep.reshape(D, 1, -1) + M[:3, 3][None, :, None]
            with np.errstate(all="ignore"):
                u, vv = p[:, 0] / p[:, 2], p[:, 1] / p[:, 2]
            inside = (p[:, 2] >= 1e-3) & (u > -1) & (u < w) & (vv > -1) & (vv < h)
            ix = np.where(inside, np.floor(np.where(inside, u, -1)), -1).astype(int)
            iy = np.where(inside, np.floor(np.where(inside, vv, -1)), -1).astype(int)
            dx, dy = ix[1:] - ix[:-1], iy[1:] - iy[:-1]
            same = (dx == 0) & (dy == 0)
            one = ((np.abs(dx) == 1) & (dy == 0)) | ((dx == 0) & (np.abs(dy) == 1))
            diag = (np.abs(dx) == 1) & (np.abs(dy) == 1)
            n = dx.size
            tot["steps"] += n; tot["same"] += int(same.sum()); tot["one"] += int(one.sum()); tot["far"] += int(n - same.sum() - one.sum())
            # gathers a lane must issue for plane d+1 when it keeps plane d's footprint: 0 / 2 / 3 (diagonal) / 4
            tot["loads_needed"] += int(2 * one.sum() + 3 * diag.sum() + 4 * (n - same.sum() - one.sum() - diag.sum()))
            tot["outside_both"] += int((~inside[1:] & ~inside[:-1]).sum())
        t = tot
        out["stage%d" % (s + 1)] = dict(D=D, h=h, w=w, steps=t["steps"], frac_same=t["same"] / t["steps"], frac_one_texel=t["one"] / t["steps"],
                                        frac_far=t["far"] / t["steps"], frac_outside_both=t["outside_both"] / t["steps"],
                                        gathers_left=(t["loads_needed"] + 4 * t["steps"] / (D - 1)) / (4 * t["steps"] * D / (D - 1)))
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
