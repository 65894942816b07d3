"""-m gpu: the HIP DrFusion path through the C ABI (IntegrateScanAsync -> RenderAsync -> GetRenderResult,
dr_fusion.h:44-66) against the CPU oracle: BIT-EXACT voxel state keyed by block coordinate (sdf bits, BGR,
weight), bit-exact ray-cast depth and colour, identical update counts."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def options(sc, H, W, vs, **kw):
    d = dict(voxel_size=vs, num_buckets=40000, bucket_size=10, num_blocks=40000, block_size=8, max_sdf_weight=64,
             truncation_distance=4 * vs, max_sensor_depth=10.0, min_sensor_depth=0.1, num_render_streams=1,
             fx=sc["fx"], fy=sc["fy"], cx=sc["cx"], cy=sc["cy"], height=H, width=W)
    d.update(kw)
    return d


def assert_same_volume(f, o):
    a, b = f.export_blocks(), o.export_blocks()
    assert a.keys() == b.keys(), f"allocated sets differ: {len(a)} vs {len(b)}"
    bad = [k for k in a if not np.array_equal(a[k], b[k])]
    assert not bad, f"{len(bad)} of {len(a)} blocks differ, e.g. {bad[:3]}"
    sa, sb = f.stats(), o.stats()
    assert sa == sb, (sa, sb)
    assert sa["mismatches"] == 0


@pytest.mark.parametrize("H,W,vs,n", [(96, 128, 0.02, 4), (120, 160, 0.01, 3), (64, 64, 0.04, 6)])
def test_integrate_and_raycast_bit_exact(H, W, vs, n):
    from oracle import scene
    from oracle.tsdf_oracle import TsdfOracle
    from tandem_amd.dr_fusion import DrFusion, DrFusionOptions
    sc = scene.make_scans(n, H, W, seed=H + n)
    opt = options(sc, H, W, vs)
    f, o = DrFusion(DrFusionOptions(**opt)), TsdfOracle(**opt)
    for i, (bgr, depth, pose) in enumerate(sc["scans"]):
        f.IntegrateScanAsync(bgr, depth, pose)
        view = sc["scans"][(i + 1) % n][2]
        f.RenderAsync([view])
        rb, rd = f.GetRenderResult()
        assert o.integrate(bgr, depth, pose) == 0
        ob, od = o.render(view)
        assert np.array_equal(rd[0].view(np.uint32), od.view(np.uint32)), f"scan {i}: ray-cast depth differs at {(rd[0] != od).sum()} px"
        assert np.array_equal(rb[0], ob), f"scan {i}: ray-cast colour differs"
        assert f.stats()["updated_last"] == o.stats()["updated_last"]
    assert_same_volume(f, o)
    f.close()


def test_edge_cases_empty_invalid_and_rotated():
    from oracle import scene
    from oracle.tsdf_oracle import TsdfOracle
    from tandem_amd.dr_fusion import DrFusion, DrFusionOptions
    H, W = 64, 96
    sc = scene.make_scans(2, H, W, seed=9)
    opt = options(sc, H, W, 0.02, num_render_streams=2)
    f, o = DrFusion(DrFusionOptions(**opt)), TsdfOracle(**opt)
    eye = np.eye(4, dtype=np.float32)
    # (1) all-invalid depth: nothing allocated, render is empty
    z = np.zeros((H, W), np.float32)
    z[0, 0], z[1, 1] = 0.05, 20.0
    f.IntegrateScanAsync(sc["scans"][0][0], z, eye)
    f.RenderAsync([eye, eye])
    rb, rd = f.GetRenderResult()
    o.integrate(sc["scans"][0][0], z, eye)
    assert f.stats()["blocks"] == 0 and not rd[0].any() and not rd[1].any()
    # (2) strongly rotated + translated pose with negative world coordinates (negative block indices)
    c, s = np.cos(2.4), np.sin(2.4)
    T = np.array([[c, 0, s, -1.7], [0, 1, 0, -0.9], [-s, 0, c, -2.2], [0, 0, 0, 1]], np.float32)
    bgr, depth, _ = sc["scans"][1]
    f.IntegrateScanAsync(bgr, depth, T)
    f.RenderAsync([T, eye])
    rb, rd = f.GetRenderResult()
    o.integrate(bgr, depth, T)
    for k, pose in enumerate((T, eye)):
        ob, od = o.render(pose)
        assert np.array_equal(rd[k].view(np.uint32), od.view(np.uint32)) and np.array_equal(rb[k], ob)
    assert min(k[0] for k in f.export_blocks()) < 0
    assert_same_volume(f, o)
    f.close()


def test_call_order_state_machine():
    """tsdf_volume.cu:520-524,635-653,703-713: wrong order is a protocol error (reference: exit(1))."""
    from oracle import scene
    from tandem_amd import _lib
    from tandem_amd.dr_fusion import DrFusion, DrFusionOptions
    H, W = 64, 96
    sc = scene.make_scans(1, H, W, seed=2)
    f = DrFusion(DrFusionOptions(**options(sc, H, W, 0.02)))
    bgr, depth, pose = sc["scans"][0]
    with pytest.raises(_lib.DrError) as e:
        f.RenderAsync([pose])
    assert e.value.code == 2
    f.IntegrateScanAsync(bgr, depth, pose)
    with pytest.raises(_lib.DrError):
        f.IntegrateScanAsync(bgr, depth, pose)
    with pytest.raises(_lib.DrError):
        f.GetRenderResult()
    with pytest.raises(_lib.DrError):  # one pose per render stream
        f.RenderAsync([pose, pose])
    f.close()
    f = DrFusion(DrFusionOptions(**options(sc, H, W, 0.02)))
    f.IntegrateScanAsync(bgr, depth, pose)
    f.RenderAsync([pose])
    f.GetRenderResult()
    f.ExtractMeshAsync([-1, -1, -1], [1, 1, 1])  # legal here (tsdf_volume.cu:760); covered by tests/test_mesh_gpu.py
    f.GetMeshSync()
    f.close()


def test_pool_exhaustion_is_reported():
    from oracle import scene
    from tandem_amd import _lib
    from tandem_amd.dr_fusion import DrFusion, DrFusionOptions
    H, W = 64, 96
    sc = scene.make_scans(1, H, W, seed=2)
    f = DrFusion(DrFusionOptions(**options(sc, H, W, 0.02, num_blocks=64, num_buckets=64)))
    bgr, depth, pose = sc["scans"][0]
    f.IntegrateScanAsync(bgr, depth, pose)
    with pytest.raises(_lib.DrError) as e:
        f.Synchronize()
    assert e.value.code == 5
    f.close()


def test_full_size_properties():
    """BASELINE config 4 shape (640x480 scans, 5 mm voxels) at a bounded scan count: size-independent properties
    the domain offers -- weights count observations and saturate, re-integrating a scan allocates nothing new,
    update counts repeat, the volume ray-casts back to the scanned depth."""
    from oracle import scene
    from tandem_amd.dr_fusion import DrFusion, DrFusionOptions
    H, W, vs = 480, 640, 0.005
    sc = scene.make_scans(3, H, W, seed=0)
    f = DrFusion(DrFusionOptions(**options(sc, H, W, vs, num_blocks=1500000, num_buckets=400000, max_sdf_weight=4)))
    bgr, depth, pose = sc["scans"][0]
    counts = []
    for _ in range(6):
        f.IntegrateScanAsync(bgr, depth, pose)
        f.RenderAsync([pose])
        rb, rd = f.GetRenderResult()
        counts.append(f.stats())
    assert len({c["blocks"] for c in counts}) == 1            # allocation is idempotent
    assert len({c["updated_last"] for c in counts}) == 1      # same voxels updated every time
    assert counts[-1]["updated_total"] == 6 * counts[0]["updated_last"] and counts[-1]["mismatches"] == 0
    blocks = f.export_blocks()
    w = np.stack(list(blocks.values())).reshape(-1, 512, 8)[:, :, 7]
    assert w.max() == 4 and set(np.unique(w)) <= {0, 4}      # min(6 scans, max_sdf_weight)
    m = (rd[0] > 0) & (depth > 0)
    assert m.mean() > 0.9 and np.abs(rd[0][m] - depth[m]).mean() < vs
    f.close()
