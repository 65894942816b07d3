"""Experiment: how much does on-device autotuning of the conv plans buy at 640x480x7?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import scene
from tandem_amd.dr_mvsnet import DrMvsnet
H, W, V = 480, 640, 7
m = DrMvsnet(os.path.join(ROOT, "weights", "tandem_va.tdmw"))
win = scene.make_window(H, W, V, seed=0)
m.upload(H, W, V, win["ref_index"], win["bgrs"], win["K"], list(win["c2ws"]), win["depth_min"], win["depth_max"], 10.0)
m.forward(5)
print("before: %.3f ms per depth map" % (m.forward(30) / 30))
k = int(sys.argv[1]) if len(sys.argv) > 1 else 12
a, b = m.autotune(k)
print("autotune(%d): conv layers %.3f -> %.3f ms" % (k, a, b))
print("after:  %.3f ms per depth map" % (m.forward(30) / 30))
