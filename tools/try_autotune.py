"""Experiment: how much does on-device autotuning of the conv plans buy at 640x480x7?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from synth import scene
from tandem_amd.dr_mvsnet import DrMvsnet
# usage: try_autotune.py [candidates] [H W d1,d2,d3]   (default: the headline 480 x 640, planes 48,32,8)
H, W, V = 480, 640, 7
blob = os.path.join(ROOT, "weights", "tandem_va.tdmw")
if len(sys.argv) > 4:
    import tempfile
    from tandem_amd import weights as Wt
    H, W = int(sys.argv[2]), int(sys.argv[3])
    planes = tuple(int(v) for v in sys.argv[4].split(","))
    _, tens = Wt.read_blob(blob)
    blob = os.path.join(tempfile.mkdtemp(), "w.tdmw")
    Wt.write_blob(blob, tens, depth_num=planes)
m = DrMvsnet(blob)
win = scene.make_window(H, W, V, seed=0)
m.upload(H, W, V, win["ref_index"], win["bgrs"], win["K"], list(win["c2ws"]), win["depth_min"], win["depth_max"], 10.0)
m.forward(5)
print("before: %.3f ms per depth map" % (m.forward(30) / 30))
k = int(sys.argv[1]) if len(sys.argv) > 1 else 12
a, b = m.autotune(k)
print("autotune(%d): conv layers %.3f -> %.3f ms" % (k, a, b))
print("after:  %.3f ms per depth map" % (m.forward(30) / 30))
