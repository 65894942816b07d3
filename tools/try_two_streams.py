"""Experiment: depth-maps/s with 1, 2, 3 DrMvsnet engines (each its own stream and window) in flight on one GPU."""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from synth import scene
from tandem_amd.dr_mvsnet import DrMvsnet
H, W, V = 480, 640, 7
blob = os.path.join(ROOT, "weights", "tandem_va.tdmw")
for n in (1, 2, 3, 1):
    ms = []
    for i in range(n):
        win = scene.make_window(H, W, V, seed=i)
        m = DrMvsnet(blob)
        m.upload(H, W, V, win["ref_index"], win["bgrs"], win["K"], list(win["c2ws"]), win["depth_min"], win["depth_max"], 10.0)
        m.forward(3)
        ms.append(m)
    K = 40
    th = [threading.Thread(target=m.forward, args=(K,)) for m in ms]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    dt = time.perf_counter() - t0
    print("engines %d: %.1f depth-maps/s (%.3f ms per map)" % (n, n * K / dt, 1e3 * dt / (n * K)))
    for m in ms: m.close()
