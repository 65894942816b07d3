#!/usr/bin/env python
"""VERDICT r4 item 3 (batched forward vs engines in flight), decided by a proxy that costs no engine change: a window
B times as TALL has exactly the launch count of one window and B times the grid of every kernel -- what a batched forward
of B windows would launch (minus B - 1 internal halos).  Reports ms per 480x640-window-equivalent for
  E engines in flight at 480x640   against   one / two engines at (B*480)x640,
each shape with the autotuner run on it (the committed plan table is keyed by the layer dims)."""
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from synth import scene  # noqa: E402
from tandem_amd.dr_mvsnet import DrMvsnet  # noqa: E402

BLOB = os.path.join(ROOT, "weights", "tandem_va.tdmw")
V, N = 7, int(os.environ.get("DR_PROXY_FORWARDS", "30"))


def run(H, W, E, tune):
    engines = []
    for e in range(E):
        win = scene.make_window(H, W, V, seed=e)
        m = DrMvsnet(BLOB)
        m.upload(H, W, V, win["ref_index"], win["bgrs"], win["K"], list(win["c2ws"]), 0.01, 10.0, 10.0)
        m.forward(3)
        if tune:
            m.autotune(tune)
        m.forward(3)
        engines.append(m)
    lat = engines[0].forward(N) / N
    best = 1e9
    for _ in range(3):
        th = [threading.Thread(target=lambda m=m: m.forward(N)) for m in engines]
        t0 = time.perf_counter()
        for t in th:
            t.start()
        for t in th:
            t.join()
        best = min(best, time.perf_counter() - t0)
    for m in engines:
        m.close()
    B = (H * W) / (480.0 * 640.0)
    return dict(H=H, W=W, engines=E, tuned=tune, single_engine_ms=lat, ms_per_window_equivalent=1e3 * best / (N * E * B),
                window_equivalents_per_s=N * E * B / best)


if __name__ == "__main__":
    rows = []
    for H, W, E, tune in ((480, 640, 1, 0), (480, 640, 3, 0), (480, 640, 4, 0), (960, 640, 1, 6), (960, 640, 2, 6), (1440, 640, 1, 6), (1440, 640, 2, 6),
                          (1920, 640, 1, 6)):
        r = run(H, W, E, tune)
        rows.append(r)
        print(json.dumps(r), flush=True)
