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
    from synth import scene
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
    from synth import scene
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
    from synth import scene
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
    from synth import scene
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
    from synth import scene
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


def test_combine_exhaustive_against_reference_and_restatement():
    """Voxel::Combine (voxel.h:21-50) as the integration kernel evaluates it -- v_rcp_f32 + half-gap bias instead of three
    IEEE divisions for the colour blend -- over EVERY (c, vc in 0..255, w in 0..64) [4.26 M cases], plus 1 M random
    sdf / weight / cap cases, against the reference's own Combine compiled for the host (oracle/_ref) when present and
    against the restatement always."""
    import ctypes as C
    from oracle import ref_fusion, tsdf_oracle
    from synth import scene
    from tandem_amd.dr_fusion import DrFusion, DrFusionOptions
    sc = scene.make_scans(1, 8, 8)
    f = DrFusion(DrFusionOptions(**options(sc, 8, 8, 0.02, num_blocks=64, num_buckets=64)))
    O = tsdf_oracle.lib()
    O.tsdf_pin_combine_colour_table.argtypes = [C.c_ubyte, C.c_void_p]
    R = ref_fusion.lib() if ref_fusion.available() else None
    c, vc = np.meshgrid(np.arange(256, dtype=np.uint8), np.arange(256, dtype=np.uint8), indexing="ij")
    a = np.zeros((65536, 8), np.uint8)
    b = np.zeros((65536, 8), np.uint8)
    a[:, 4] = a[:, 5] = a[:, 6] = c.ravel()
    b[:, 4] = b[:, 5] = b[:, 6] = vc.ravel()
    b[:, 7] = 1
    want = np.empty(65536, np.uint8)
    for w in range(0, 65):
        a[:, 7] = w
        got = f.test_combine(a, b, 255)
        O.tsdf_pin_combine_colour_table(w, want.ctypes.data)
        for ch in (4, 5, 6):
            assert np.array_equal(got[:, ch], want), f"w={w}: {np.count_nonzero(got[:, ch] != want)} colour blends differ from the restatement"
        assert (got[:, 7] == w + 1).all()
        if R is not None:
            R.ref_combine_colour_table(w, want.ctypes.data)
            assert np.array_equal(got[:, 4], want), f"w={w}: differs from the reference's Voxel::Combine"
    # sdf running mean, weight increment and cap, vw = 1 (what IntegrateScanKernel passes) and vw = 2 (general branch)
    rng = np.random.RandomState(0)
    n = 1 << 20
    a = rng.randint(0, 256, (n, 8)).astype(np.uint8)
    b = rng.randint(0, 256, (n, 8)).astype(np.uint8)
    a[:, :4] = rng.uniform(-0.05, 0.05, n).astype(np.float32).view(np.uint8).reshape(n, 4)
    b[:, :4] = rng.uniform(-0.05, 0.05, n).astype(np.float32).view(np.uint8).reshape(n, 4)
    a[:, 7] = rng.randint(0, 65, n)
    b[:, 7] = 1 + (rng.rand(n) < 0.1)
    got = f.test_combine(a, b, 64)
    O.tsdf_pin_combine.argtypes = [C.c_float, C.c_void_p, C.c_ubyte, C.c_float, C.c_void_p, C.c_ubyte, C.c_ubyte, C.c_void_p, C.c_void_p, C.c_void_p]
    # vectorised restatement of voxel.h:21-50 in numpy fp32 (division by IEEE), spot-checked against the C oracle below
    w, vw = a[:, 7].astype(np.float32), b[:, 7].astype(np.float32)
    col = ((a[:, 4:7].astype(np.float32) * w[:, None] + b[:, 4:7].astype(np.float32) * vw[:, None]) / (w + vw)[:, None]).astype(np.uint8)
    sdf = (a[:, :4].copy().view(np.float32)[:, 0] * w + b[:, :4].copy().view(np.float32)[:, 0] * vw) / (w + vw)
    nw = np.minimum(a[:, 7].astype(np.int32) + b[:, 7], 64).astype(np.uint8)
    assert np.array_equal(got[:, 4:7], col) and np.array_equal(got[:, 7], nw)
    assert np.array_equal(got[:, :4].copy().view(np.uint32)[:, 0], sdf.astype(np.float32).view(np.uint32))
    s, co, wo = np.zeros(1, np.float32), np.zeros(3, np.uint8), np.zeros(1, np.uint8)
    for i in range(0, n, 4099):
        ca, cb = a[i, 4:7].copy(), b[i, 4:7].copy()   # keep the temporaries alive across the call
        O.tsdf_pin_combine(float(a[i, :4].copy().view(np.float32)[0]), ca.ctypes.data, int(a[i, 7]),
                           float(b[i, :4].copy().view(np.float32)[0]), cb.ctypes.data, int(b[i, 7]), 64,
                           s.ctypes.data, co.ctypes.data, wo.ctypes.data)
        assert s.view(np.uint32)[0] == got[i, :4].copy().view(np.uint32)[0] and tuple(co) == tuple(got[i, 4:7]) and wo[0] == got[i, 7]
    f.close()


@pytest.mark.parametrize("H,W,vs,n", [(96, 128, 0.02, 4), (60, 80, 0.01, 3)])
def test_hip_path_equals_the_reference_build(H, W, vs, n):
    """The HIP engine against the REFERENCE's own DrFusion compiled for the host (oracle/_ref/libdr_fusion_ref.so, built by
    oracle/Makefile.ref from the sources under /root/reference; the prebuilt library travels to the GPU box): allocated
    set, every voxel, ray-cast depth and colour bit-exact.  Scenes keep block (0,0,0) out of the frustum (the reference's
    free-entry alias, tests/test_ref_fusion.py::test_origin_block_alias_is_the_only_deviation)."""
    from oracle import ref_fusion
    from synth import scene
    from tandem_amd.dr_fusion import DrFusion, DrFusionOptions
    if not ref_fusion.available():
        pytest.skip("oracle/_ref/libdr_fusion_ref.so not present")
    sc = scene.make_scans(n, H, W, seed=H + n)
    opt = options(sc, H, W, vs)
    f, r = DrFusion(DrFusionOptions(**opt)), ref_fusion.RefFusion(**opt)
    for i, (bgr, depth, pose) in enumerate(sc["scans"]):
        f.IntegrateScanAsync(bgr, depth, pose)
        view = sc["scans"][(i + 1) % n][2]
        f.RenderAsync([view])
        rb, rd = f.GetRenderResult()
        r.integrate(bgr, depth, pose)
        (ob, od), = r.render([view])
        assert np.array_equal(rd[0].view(np.uint32), od.view(np.uint32)), f"scan {i}: ray-cast depth differs at {(rd[0] != od).sum()} px"
        assert np.array_equal(rb[0], ob), f"scan {i}: ray-cast colour differs"
    a, b = f.export_blocks(), r.export_blocks()
    assert a.keys() == b.keys(), f"allocated sets differ: {len(a)} vs {len(b)}"
    bad = [k for k in a if not np.array_equal(a[k], b[k])]
    assert not bad, f"{len(bad)} of {len(a)} blocks differ, e.g. {bad[:3]}"
    assert f.stats()["mismatches"] == 0
    lo, hi = (-1.0, -1.0, 0.5), (1.0, 1.0, 3.0)
    fv, fc = f.GetMesh(lo, hi)
    rv, rc = r.extract_mesh(lo, hi)

    def canon(v, c):
        t = np.concatenate([v.reshape(-1, 9), c.reshape(-1, 9)], axis=1).view(np.uint32)
        return t[np.lexsort(t.T[::-1])]
    assert len(fv) == len(rv) > 0 and np.array_equal(canon(fv, fc), canon(rv, rc)), "mesh triangle sets differ"
    f.close(); r.close()


def test_full_size_bit_exact_against_the_oracle():
    """BASELINE config 4's shape -- 640x480 scans into 5 mm voxels (truncation 20 mm) -- two scans, voxel state, update
    counts and the ray-cast of the second view BIT-EXACT against oracle/tsdf_oracle.c (which tests/test_ref_fusion.py
    pins to the reference build)."""
    from synth import scene
    from oracle.tsdf_oracle import TsdfOracle
    from tandem_amd.dr_fusion import DrFusion, DrFusionOptions
    H, W, vs = 480, 640, 0.005
    sc = scene.make_scans(2, H, W, seed=0)
    opt = options(sc, H, W, vs, num_blocks=600000, num_buckets=200000)
    f, o = DrFusion(DrFusionOptions(**opt)), TsdfOracle(**opt)
    for i, (bgr, depth, pose) in enumerate(sc["scans"]):
        f.IntegrateScanAsync(bgr, depth, pose)
        f.RenderAsync([pose])
        rb, rd = f.GetRenderResult()
        assert o.integrate(bgr, depth, pose) == 0
        assert f.stats()["updated_last"] == o.stats()["updated_last"] > 10_000_000
    ob, od = o.render(sc["scans"][-1][2])
    assert np.array_equal(rd[0].view(np.uint32), od.view(np.uint32)) and np.array_equal(rb[0], ob)
    assert_same_volume(f, o)
    assert f.stats()["blocks"] > 80_000
    f.close()


@pytest.mark.parametrize("vs,trunc,nframes", [(0.005, 0.02, 5), (0.01, 0.04, 2)])
def test_bench_workload_bit_exact_against_the_oracle(vs, trunc, nframes):
    """bench.py's own TSDF workload (BASELINE configs[3]): consecutive frames of the camera loop through the analytic room
    (synth/room.py), 640x480, fused at 5 mm / 20 mm and at TANDEM's native 1 cm / 4 cm -- voxel state keyed by block
    coordinate, update counts and the ray-cast of every frame BIT-EXACT against oracle/tsdf_oracle.c.  Consecutive loop
    poses overlap almost completely, so weights climb 1, 2, 3 ... on the same voxels (the running-average path the
    single-scan tests do not reach)."""
    import torch
    from synth import room
    from oracle.tsdf_oracle import TsdfOracle
    from tandem_amd.dr_fusion import DrFusion, DrFusionOptions
    H, W = 480, 640
    poses = room.loop_poses(1000, seed=7)[:nframes]
    fr = room.render_frames(poses, H, W, device="cuda:0", seed=0)
    opt = dict(voxel_size=vs, num_buckets=300000, bucket_size=10, num_blocks=700000, block_size=8, max_sdf_weight=64,
               truncation_distance=trunc, max_sensor_depth=10.0, min_sensor_depth=0.1, num_render_streams=1,
               fx=fr["fx"], fy=fr["fy"], cx=fr["cx"], cy=fr["cy"], height=H, width=W)
    f, o = DrFusion(DrFusionOptions(**opt)), TsdfOracle(**opt)
    for i in range(nframes):
        bgr, depth = fr["bgr"][i].cpu().numpy(), fr["depth"][i].cpu().numpy()
        f.IntegrateScanAsync(bgr, depth, poses[i])
        f.RenderAsync([poses[i]])
        rb, rd = f.GetRenderResult()
        assert o.integrate(bgr, depth, poses[i]) == 0
        assert f.stats()["updated_last"] == o.stats()["updated_last"] > 1_000_000
        if i in (0, nframes - 1):  # (the oracle's ray-caster is the slow part: first and last frame)
            ob, od = o.render(poses[i])
            assert np.array_equal(rd[0].view(np.uint32), od.view(np.uint32)), f"frame {i}: ray-cast depth differs at {(rd[0] != od).sum()} px"
            assert np.array_equal(rb[0], ob), f"frame {i}: ray-cast colour differs"
    assert_same_volume(f, o)
    w = np.stack(list(f.export_blocks().values())).reshape(-1, 512, 8)[:, :, 7]
    assert w.max() == nframes  # the frames really overlap: some voxels were combined every time
    f.close()
    del fr
    torch.cuda.empty_cache()


def test_hip_path_equals_the_reference_build_at_qvga():
    """The direct HIP-vs-reference comparison at 240x320 (the other cases stop at 96x128): the reference's integration kernel
    walks EVERY hash entry (free ones alias block (0,0,0)), so the table is kept small to keep its serial host build
    affordable.  Allocated set, every voxel, ray-cast depth and colour bit-exact."""
    from oracle import ref_fusion
    from synth import scene
    from tandem_amd.dr_fusion import DrFusion, DrFusionOptions
    if not ref_fusion.available():
        pytest.skip("oracle/_ref/libdr_fusion_ref.so not present")
    H, W, vs, n = 240, 320, 0.02, 2
    sc = scene.make_scans(n, H, W, seed=21)
    opt = options(sc, H, W, vs, num_buckets=6000, bucket_size=10, num_blocks=50000)
    f, r = DrFusion(DrFusionOptions(**opt)), ref_fusion.RefFusion(**opt)
    for i, (bgr, depth, pose) in enumerate(sc["scans"]):
        f.IntegrateScanAsync(bgr, depth, pose)
        view = sc["scans"][(i + 1) % n][2]
        f.RenderAsync([view])
        rb, rd = f.GetRenderResult()
        r.integrate(bgr, depth, pose)
        (ob, od), = r.render([view])
        assert np.array_equal(rd[0].view(np.uint32), od.view(np.uint32)), f"scan {i}: ray-cast depth differs at {(rd[0] != od).sum()} px"
        assert np.array_equal(rb[0], ob), f"scan {i}: ray-cast colour differs"
    a, b = f.export_blocks(), r.export_blocks()
    assert a.keys() == b.keys(), f"allocated sets differ: {len(a)} vs {len(b)}"
    assert len(a) > 2000  # measured: 2150 blocks
    bad = [k for k in a if not np.array_equal(a[k], b[k])]
    assert not bad, f"{len(bad)} of {len(a)} blocks differ, e.g. {bad[:3]}"
    assert f.stats()["mismatches"] == 0
    f.close(); r.close()


@pytest.mark.parametrize("vs,f", [(0.005, 500.0), (0.01, 481.2), (0.02, 500.0), (0.04, 250.0), (0.0123, 617.3)])
def test_exact_fast_division_is_verified_at_construction(vs, f):
    """div_exact (reciprocal + FMA correction, 3 instructions) replaces the IEEE division by voxel_size / fx / fy in the
    ray-caster: the engine checks it against a / b for ALL 2^32 dividends per divisor when it is created."""
    from synth import scene
    from tandem_amd.dr_fusion import DrFusion, DrFusionOptions
    sc = scene.make_scans(1, 8, 8)
    fu = DrFusion(DrFusionOptions(**options(sc, 8, 8, vs, num_blocks=64, num_buckets=64, fx=f, fy=f * 0.997)))
    assert fu.fast_div_status() == (True, 0)
    fu.close()


def test_raycast_generations_agree_and_ieee_fallback(monkeypatch, parity_hooks):
    """k_raycast2 (dense-grid look-ups, exact fast division, shared corner coordinates, empty-superblock skip, two-round-trip
    sampler) against the literal k_raycast, against k_raycast2 with IEEE division, without the skip and with round 2's four-stage
    sampler (DR_RAYCAST_SAMPLER=0), on the same volume: bit-identical depth and colour."""
    from synth import scene
    from tandem_amd.dr_fusion import DrFusion, DrFusionOptions
    H, W = 120, 160
    sc = scene.make_scans(3, H, W, seed=4)
    outs = []
    for env in ({}, {"DR_RAYCAST_V1": "1"}, {"DR_FUSION_IEEE_DIV": "1"}, {"DR_RAYCAST_NO_SKIP": "1"}, {"DR_RAYCAST_SAMPLER": "0"},
                {"DR_RAYCAST_SAMPLER": "0", "DR_FUSION_IEEE_DIV": "1"}):
        for k in ("DR_RAYCAST_V1", "DR_FUSION_IEEE_DIV", "DR_RAYCAST_NO_SKIP", "DR_RAYCAST_SAMPLER"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        fu = DrFusion(DrFusionOptions(**options(sc, H, W, 0.01, num_render_streams=2)))
        assert fu.fast_div_status()[0] == ("DR_FUSION_IEEE_DIV" not in env)
        res = []
        for i, (bgr, depth, pose) in enumerate(sc["scans"]):
            fu.IntegrateScanAsync(bgr, depth, pose)
            far = pose.copy(); far[:3, 3] += (0.3, -0.2, -0.4)       # a view from somewhere else: rays through unobserved space
            fu.RenderAsync([sc["scans"][(i + 1) % 3][2], far])
            rb, rd = fu.GetRenderResult()
            res.append((rb[0].copy(), rd[0].copy(), rb[1].copy(), rd[1].copy()))
        outs.append(res)
        fu.close()
    for other in outs[1:]:
        for a, b in zip(outs[0], other):
            for x, y in zip(a, b):
                assert np.array_equal(x.view(np.uint8), y.view(np.uint8))
    assert (outs[0][-1][1] > 0).mean() > 0.5


def test_blocks_outside_the_dense_grid():
    """The dense block grid covers block coordinates [-256, 256)^3; beyond it the open-addressing table takes over.  A scene
    pushed 40.9 m along +x at 2 cm voxels (block edge 16 cm) straddles the border: allocation, integration, the
    ray-caster's hand-over to the literal pass and the mesh all have to agree with the oracle bit for bit."""
    from synth import scene
    from oracle.tsdf_oracle import TsdfOracle
    from tandem_amd.dr_fusion import DrFusion, DrFusionOptions
    H, W, vs = 96, 128, 0.02
    sc = scene.make_scans(3, H, W, seed=6)
    opt = options(sc, H, W, vs)
    f, o = DrFusion(DrFusionOptions(**opt)), TsdfOracle(**opt)
    S = np.eye(4, dtype=np.float32)
    c, s = np.cos(1.45), np.sin(1.45)
    S[:3, :3] = [[c, 0, s], [0, 1, 0], [-s, 0, c]]      # look along +x
    S[:3, 3] = (40.2, 0.3, -0.2)
    for i, (bgr, depth, pose) in enumerate(sc["scans"]):
        pose = (S @ pose).astype(np.float32)
        f.IntegrateScanAsync(bgr, depth, pose)
        f.RenderAsync([pose])
        rb, rd = f.GetRenderResult()
        assert o.integrate(bgr, depth, pose) == 0
        ob, od = o.render(pose)
        assert np.array_equal(rd[0].view(np.uint32), od.view(np.uint32)), f"scan {i}: {(rd[0] != od).sum()} px differ"
        assert np.array_equal(rb[0], ob)
    xs = [k[0] for k in f.export_blocks()]
    assert min(xs) < 256 <= max(xs), (min(xs), max(xs))
    assert_same_volume(f, o)
    assert (od > 0).mean() > 0.3
    f.close()


def test_far_blocks_through_the_device_resident_sequence():
    """in drf_bench_sequence the allocation of scan k + 1 runs BESIDE the ray-cast of scan k.  For blocks outside the
    dense grid the insert publishes its key before the pool index, so the concurrent ray-cast must never read an unpublished
    value (vals starts at -1 = absent; find_block_table).  The far-block scene of test_blocks_outside_the_dense_grid goes
    through the sequence hook: the ray-cast of the second-to-last scan (the one that overlapped an allocation), the last one
    and the voxel state have to equal the oracle's bit for bit -- repeatedly, since a race is a matter of timing."""
    import torch
    from synth import scene
    from oracle.tsdf_oracle import TsdfOracle
    from tandem_amd.dr_fusion import DrFusion, DrFusionOptions
    H, W, vs, n = 96, 128, 0.02, 4
    sc = scene.make_scans(n, H, W, seed=6)
    opt = options(sc, H, W, vs)
    S = np.eye(4, dtype=np.float32)
    c, s = np.cos(1.45), np.sin(1.45)
    S[:3, :3] = [[c, 0, s], [0, 1, 0], [-s, 0, c]]
    S[:3, 3] = (40.2, 0.3, -0.2)
    scans = [(bgr, depth, (S @ pose).astype(np.float32)) for bgr, depth, pose in sc["scans"]]
    o = TsdfOracle(**opt)
    renders = []
    for bgr, depth, pose in scans:
        assert o.integrate(bgr, depth, pose) == 0
        renders.append(o.render(pose))
    d_bgr = torch.from_numpy(np.stack([b for b, _, _ in scans])).cuda()
    d_depth = torch.from_numpy(np.stack([d for _, d, _ in scans])).cuda()
    poses = np.stack([p.reshape(16) for _, _, p in scans])
    for rep in range(5):
        f = DrFusion(DrFusionOptions(**opt))
        f.bench_sequence(d_bgr.data_ptr(), d_depth.data_ptr(), poses, render=True)
        for back in (0, 1):
            rb, rd = f.bench_last_render(back)
            ob, od = renders[n - 1 - back]
            assert np.array_equal(rd.view(np.uint32), od.view(np.uint32)), f"rep {rep}, scan {n - 1 - back}: {(rd != od).sum()} px differ"
            assert np.array_equal(rb, ob)
        xs = [k[0] for k in f.export_blocks()]
        assert min(xs) < 256 <= max(xs)
        assert_same_volume(f, o)
        f.close()
