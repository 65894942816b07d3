"""Device time of one 640x480x7 forward with the key-frame feature cache answering six of the seven views (drm_set_feature_cache), beside the same
engine without the cache: python tools/time_feature_cache.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from synth import scene
from tandem_amd.dr_mvsnet import DrMvsnet
H, W = 480, 640
big = scene.make_window(H, W, 9, seed=3)
wins = [dict(bgrs=[np.ascontiguousarray(b) for b in big["bgrs"][k:k + 7]], c2ws=list(big["c2ws"][k:k + 7])) for k in range(3)]
res = {}
for cache in (0, 16):
    m = DrMvsnet(os.path.join(ROOT, "weights", "tandem_va.tdmw"))
    if cache:
        m.set_feature_cache(cache)
    for w in wins:  # the last one is staged with six of its images cached
        m.upload(H, W, 7, 5, w["bgrs"], big["K"], w["c2ws"], 0.01, 10.0, 10.0)
        m.forward(2)
    res[cache] = min(m.forward(30) / 30 for _ in range(3))
    prof = {r["op"]: r["ms"] for r in m.profile()}
    print("cache %2d: forward %.3f ms  FeatureNet ops in the profile %.3f ms  stats %s" % (cache, res[cache], sum(v for k, v in prof.items() if k.startswith("fn.")), m.feature_cache_stats()))
    m.close()
print("saved per window: %.3f ms (%.1f %%)" % (res[0] - res[16], 100 * (res[0] - res[16]) / res[0]))
