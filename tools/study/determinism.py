"""Run-to-run bit stability of the engine on the headline fixture: N fresh engines per configuration, the final depth map and the intermediate tensors compared with the first run's.
Round 6: the fp32 product path is bit-stable (16 of 16); the parity-build-only bf16x3 mode is not when its unfused FeatureNet head launches overlap the stage-1 plane sweep."""
import os, sys, numpy as np
sys.path.insert(0, '/root/repo')
from tandem_amd import _lib
from tandem_amd.dr_mvsnet import DrMvsnet
g = np.load('/root/repo/tests/golden/mvsnet_v7_480x640_headline.npz')
bgrs = [np.ascontiguousarray(b) for b in g["bgrs"]]
H, W = bgrs[0].shape[:2]
args = (H, W, len(bgrs), int(g["ref_index"]), bgrs, g["K"], list(g["c2ws"]), float(g["depth_min"]), float(g["depth_max"]), float(g["discard"]))
blob = '/root/repo/weights/tandem_va.tdmw'
CASES = [(None, {}), (_lib.HOOKS_LIB_PATH, {"DR_COSTVOL_V3": "1"}), (_lib.HOOKS_LIB_PATH, {"DR_VOL_NO_SPLIT": "1"}),
         (_lib.HOOKS_LIB_PATH, {"DR_CONV_BF16X3": "1"}), (_lib.HOOKS_LIB_PATH, {"DR_CONV_BF16X3": "1", "DR_MVS_NO_SIDE_STREAM": "1"}),
         (_lib.HOOKS_LIB_PATH, {"DR_CONV_BF16X3": "1", "DR_COSTVOL_V3": "1"})]
for lib, env in CASES:
    _lib.switch(lib)
    for k in ("DR_CONV_BF16X3", "DR_VOL_NO_SPLIT", "DR_COSTVOL_V3", "DR_CV5_REUSE", "DR_MVS_NO_SIDE_STREAM", "DR_CV_DCHUNK1", "DR_FN_FRONT", "DR_FN_HEAD3"):
        os.environ.pop(k, None)
    os.environ.update(env)
    mode = str(env)
    for _ in (0,):
        outs = []
        tens = []
        for rep in range(8):
            m = DrMvsnet(blob)
            m.CallAsync(*args)
            o = m.GetResult()
            outs.append(o.depth_dense.copy())
            tens.append({n: m.tensor(n).copy() for n in ("feat1", "feat2", "feat3", "volume1", "depth1", "depth2")})
            m.close()
        print("lib", os.path.basename(lib) if lib else "product", mode, "runs equal:", [bool(np.array_equal(outs[0], x)) for x in outs[1:]],
              "max diff", [float(np.abs(outs[0] - x).max()) for x in outs[1:]])
        for n in tens[0]:
            print("   ", n, [bool(np.array_equal(tens[0][n], t[n])) for t in tens[1:]])
