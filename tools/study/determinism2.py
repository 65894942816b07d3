import os, sys, numpy as np
sys.path.insert(0, '/root/repo')
from tandem_amd import _lib
from tandem_amd.dr_mvsnet import DrMvsnet
g = np.load('/root/repo/tests/golden/mvsnet_v7_480x640_headline.npz')
bgrs = [np.ascontiguousarray(b) for b in g["bgrs"]]
H, W = bgrs[0].shape[:2]
args = (H, W, len(bgrs), int(g["ref_index"]), bgrs, g["K"], list(g["c2ws"]), float(g["depth_min"]), float(g["depth_max"]), float(g["discard"]))
blob = '/root/repo/weights/tandem_va.tdmw'
_lib.switch(_lib.HOOKS_LIB_PATH)
os.environ["DR_CONV_BF16X3"] = "1"
vols = []
for rep in range(3):
    m = DrMvsnet(blob)
    m.upload(*args)
    m.forward(1)
    v = [m.tensor("volume%d" % s).copy() for s in (1, 2, 3)]
    # the same engine, a second forward of the same staged window
    m.forward(1)
    v2 = [m.tensor("volume%d" % s).copy() for s in (1, 2, 3)]
    print("rep", rep, "second forward equals first:", [bool(np.array_equal(a, b)) for a, b in zip(v, v2)])
    vols.append(v)
    m.close()
a, b = vols[0][0], vols[1][0]
print("volume1 shape", a.shape)
d = np.argwhere(a != b)
print("differing elements", len(d), "of", a.size)
if len(d):
    print("planes", np.unique(d[:, 0])[:20], "rows", np.unique(d[:, 1])[:20], "cols", np.unique(d[:, 2])[:30], "chans", np.unique(d[:, 3]))
    print("first diffs", d[:10].tolist(), [(float(a[tuple(i)]), float(b[tuple(i)])) for i in d[:5]])
