"""Per-op hipEvent profile of one 640x480x7 forward (no CPU oracle): python tools/profile_ops.py [regex]"""
import os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from synth import scene
from tandem_amd.dr_mvsnet import DrMvsnet
H, W, V = 480, 640, 7
m = DrMvsnet(os.path.join(ROOT, "weights", "tandem_va.tdmw"))
win = scene.make_window(H, W, V, seed=0)
dmin, dmax = (float(t) for t in os.environ["DR_OPS_RANGE"].split(",")) if os.environ.get("DR_OPS_RANGE") else (win["depth_min"], win["depth_max"])  # the headline runs 0.01,10
m.upload(H, W, V, win["ref_index"], win["bgrs"], win["K"], list(win["c2ws"]), dmin, dmax, 10.0)
m.forward(5)
pat = re.compile(sys.argv[1] if len(sys.argv) > 1 else ".")
rows = [m.profile() for _ in range(3)]
best = {}
for rr in rows:
    for r in rr:
        best[r["op"]] = min(best.get(r["op"], 1e9), r["ms"])
print(" ".join("%s=%.3f" % (k, v) for k, v in best.items() if pat.search(k)), "| forward %.3f ms" % (m.forward(30) / 30))
