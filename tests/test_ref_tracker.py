"""CPU: oracle/tracker_oracle.c (the restatement the -m gpu tracker tests check the HIP path against) PINNED to the
reference's own calcResKernelNew / calcGKernel, compiled for the host from
ref:tandem/libdr/cuda_coarse_tracker/src/cuda_coarse_tracker_private.cu by oracle/Makefile.ref.

* the seven warped buffers (u, v, dx, dy, idepth, residual, weight): BIT-EXACT per point;
* the 7 calcRes sums and 45 calcG sums: the reference adds floats per 128-thread block and atomicAdds the block sums
  (order unspecified on a GPU, descending thread order in the host build); the restatement adds the same fp32 terms in
  double.  Integer-valued sums (term counts) must be equal; the others agree to 2e-5 relative (fp32 accumulation noise of
  ~1e4 terms); calcG against the reference kernel's own double-accumulator instantiation with one point per thread
  (loops = 1, no float partial sums): 1e-12.

Skipped when neither oracle/_ref/libcoarse_tracker_ref.so nor /root/reference exists."""
import numpy as np
import pytest

from oracle import ref_tracker
from synth import scene
from oracle.tracker_oracle import TrackerOracle

pytestmark = pytest.mark.skipif(not ref_tracker.available(), reason="reference build (oracle/_ref) not available")


def _setup(H, W, seed, frac, huber=9.0, exposure=(1.0, 1.0), ref_aff=(0.0, 0.0), motion=1.0):
    p = scene.make_tracking_pair(H, W, seed=seed, sparse_fraction=frac, motion=motion)
    o = TrackerOracle(W, H, huber, 20.0)
    o.setK(p["fx"], p["fy"], p["cx"], p["cy"])
    o.setReference(p["pc_u"], p["pc_v"], p["pc_idepth"], p["pc_color"], exposure[0], list(ref_aff))
    o.setNew(p["dI_new"])
    return p, o


@pytest.mark.parametrize("H,W,seed,frac,cutoff,aff,expo", [
    (120, 160, 3, 0.2, 20.0, (0.0, 0.0), 1.0),
    (96, 128, 1, 0.5, 6.0, (0.05, -3.0), 1.3),      # tight cut-off: many saturated terms; affine brightness + exposure ratio
    (240, 320, 7, 0.05, 1000.0, (0.0, 0.0), 1.0),
    (64, 64, 2, 1.0, 20.0, (-0.1, 12.0), 0.7),
])
def test_reference_kernels_equal_restatement(H, W, seed, frac, cutoff, aff, expo):
    huber = 9.0
    p, o = _setup(H, W, seed, frac, huber=huber, ref_aff=(0.02, 1.5))
    T = p["refToNew"]
    out, sums = o.calcRes(T, expo, list(aff), cutoff)
    ow = o.warped()
    r2n, Ki, a2, maxE, rb = o.kernel_inputs(T, expo, list(aff), cutoff)
    rw, rout = ref_tracker.calc_res(huber, W, H, p["fx"], p["fy"], p["cx"], p["cy"], r2n, Ki, a2, maxE, cutoff,
                                    p["pc_u"], p["pc_v"], p["pc_idepth"], p["pc_color"], p["dI_new"])
    n = len(p["pc_u"])
    assert n > 500
    for k, (a, b) in enumerate(zip(rw, ow)):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), f"warped buffer {k}: {(a != b).sum()} of {n} points differ"
    # counts are exact in both (small integers in fp32), the rest to fp32 accumulation noise
    for k in (1, 2, 3, 6):
        assert rout[k] == sums[k], (k, rout[k], sums[k])
    assert sums[2] > 100
    np.testing.assert_allclose(rout[[0, 4, 5]], sums[[0, 4, 5]], rtol=2e-5)

    H8, b8, raw = o.calcG(expo, list(aff))
    rd = ref_tracker.calc_g(p["fx"], p["fy"], a2, rb, p["pc_color"], rw, loops=16, double=True)
    # even the reference's Accum = double instantiation adds each thread's `loops` = 16 points in a float result_private[45]
    np.testing.assert_allclose(rd, raw, rtol=2e-5, atol=1e-7 * np.abs(raw).max())
    rd1 = ref_tracker.calc_g(p["fx"], p["fy"], a2, rb, p["pc_color"], rw, loops=1, double=True)   # one point per thread: no float partials
    np.testing.assert_allclose(rd1, raw, rtol=1e-12, atol=1e-12 * np.abs(raw).max())
    rf = ref_tracker.calc_g(p["fx"], p["fy"], a2, rb, p["pc_color"], rw, loops=16, double=False)
    np.testing.assert_allclose(rf, raw, rtol=3e-4, atol=1e-6 * np.abs(raw).max())


def test_out_of_image_and_empty():
    """identity warp of border points (none survive the 2 / w-3 margin), and n = 0."""
    H, W = 48, 64
    p, o = _setup(H, W, 4, 0.3)
    u = np.array([0.0, 1.0, 2.0, W - 3.0, W - 1.0, 30.0], np.float32)
    v = np.array([10.0, 0.5, 2.0, 20.0, H - 1.0, H - 3.0], np.float32)
    idp = np.full(6, 0.5, np.float32)
    col = np.full(6, 100.0, np.float32)
    o.setReference(u, v, idp, col, 1.0, [0.0, 0.0])
    out, sums = o.calcRes(np.eye(4), 1.0, [0.0, 0.0], 20.0)
    r2n, Ki, a2, maxE, rb = o.kernel_inputs(np.eye(4), 1.0, [0.0, 0.0], 20.0)
    rw, rout = ref_tracker.calc_res(9.0, W, H, p["fx"], p["fy"], p["cx"], p["cy"], r2n, Ki, a2, maxE, 20.0, u, v, idp, col, p["dI_new"])
    for a, b in zip(rw, o.warped()):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    assert rout[1] == sums[1] == 0
    rw0, rout0 = ref_tracker.calc_res(9.0, W, H, p["fx"], p["fy"], p["cx"], p["cy"], r2n, Ki, a2, maxE, 20.0,
                                      u[:0], v[:0], idp[:0], col[:0], p["dI_new"])
    assert not rout0.any()
