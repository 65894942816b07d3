"""Bit stability of the fp32 product path with SEVERAL engines in flight (bench.py's configuration): E engines run the same fixture concurrently from E threads, R rounds;
every engine's stage-1..3 volumes and depth map must equal the single-engine result bit for bit."""
import os, sys, threading
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tandem_amd.dr_mvsnet import DrMvsnet
g = np.load(os.path.join(ROOT, "tests", "golden", "mvsnet_v7_480x640_headline.npz"))
bgrs = [np.ascontiguousarray(b) for b in g["bgrs"]]
H, W = bgrs[0].shape[:2]
args = (H, W, len(bgrs), int(g["ref_index"]), bgrs, g["K"], list(g["c2ws"]), float(g["depth_min"]), float(g["depth_max"]), float(g["discard"]))
blob = os.path.join(ROOT, "weights", "tandem_va.tdmw")
E, R = 4, 6
ref = DrMvsnet(blob)
ref.upload(*args); ref.forward(1)
want = {n: ref.tensor(n).copy() for n in ("volume1", "volume2", "volume3", "depth3")}
engines = [DrMvsnet(blob) for _ in range(E)]
for m in engines:
    m.upload(*args)
bad = 0
for r in range(R):
    th = [threading.Thread(target=lambda m=m: m.forward(5)) for m in engines]
    for t in th: t.start()
    for t in th: t.join()
    for e, m in enumerate(engines):
        for n, w in want.items():
            got = m.tensor(n)
            if not np.array_equal(got.view(np.uint32), w.view(np.uint32)):
                bad += 1
                d = np.argwhere(got != w)
                print("round", r, "engine", e, n, "differs in", len(d), "elements; first", d[:3].tolist())
print("engines", E, "rounds", R, "x 5 forwards each:", "ALL BIT-IDENTICAL to the single-engine run" if bad == 0 else "%d tensors differed" % bad)
