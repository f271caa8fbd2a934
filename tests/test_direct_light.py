"""Direct light sampling (`direct_light = 1`): next-event
estimation built on the reference's own light samplers getRandomPointOnCube / getRandomPointOnSphere
(/root/reference/src/intersections.h:220-286, no call sites there).  The reference defines nothing
beyond those two functions, so the estimator is pinned the way the other build-defined pieces are:
physical invariants of the oracle on the CPU (same expectation as plain path tracing, correct
sampling densities), and bit-exact agreement of the HIP path with the oracle on the GPU."""
import ctypes as C
import math

import numpy as np
import pytest

import orc
from gpu_common import make_tracer, oracle_config, to_product


def _scene(name="sampleScene", w=32, h=32, keep=None):
    sc = orc.load_golden_scene(name).with_resolution(w, h)
    if keep is not None:
        sc = orc.Scene([sc.geoms[i] for i in keep], sc.materials, sc.camera, sc.iterations, sc.image_name)
    return sc


def _sphere_light_scene(w=32, h=32):
    """The Cornell walls with the cube light replaced by an emitting sphere hanging in the room."""
    sc = _scene(w=w, h=h, keep=[0, 1, 2, 3, 4, 5])
    g = orc.Geom()
    C.memmove(C.byref(g), C.byref(sc.geoms[5]), C.sizeof(orc.Geom))
    T, Ti = np.zeros(16, np.float32), np.zeros(16, np.float32)
    orc.lib().orc_build_transform(orc.vec3(0, 7.5, 0), orc.vec3(0, 0, 0), orc.vec3(1.5, 1.5, 1.5), orc.fptr(T), orc.fptr(Ti))
    for k in range(16):
        g.transform[k], g.inverseTransform[k] = T[k], Ti[k]
    g.type, g.materialid = 0, 8                                        # material 8 = the emitter of sampleScene
    geoms = list(sc.geoms[:5]) + [g]
    return orc.Scene(geoms, sc.materials, sc.camera, sc.iterations, sc.image_name)


# ---------------------------------------------------------------- CPU: sampling densities ----

def test_cube_light_sample_density_is_one_over_area():
    sc = _scene()
    light = sc.geoms[8]
    L = orc.lib()
    rad = (C.c_float * 3)()
    L.orc_get_radiuses(C.byref(light), rad)
    sx, sy, sz = 2 * rad[0], 2 * rad[1], 2 * rad[2]
    area = 2 * (sx * sy + sy * sz + sx * sz)
    Q, inv = (C.c_float * 3)(), (C.c_float * 1)()
    inv_m = np.array([light.inverseTransform[k] for k in range(12)], np.float64).reshape(3, 4)
    faces = np.zeros(3)
    for seed in range(4000):
        assert L.orc_sample_light(C.byref(light), float(seed), Q, inv) == 1
        assert abs(inv[0] - area) < 1e-4 * area
        p = inv_m[:, :3] @ np.array(list(Q), np.float64) + inv_m[:, 3]
        assert abs(np.abs(p).max() - 0.5) < 1e-5                      # on the surface of the unit cube
        faces[int(np.argmax(np.abs(p)))] += 1
    want = np.array([sy * sz, sx * sz, sx * sy]) * 2 / area            # faces normal to x, y, z
    assert np.abs(faces / faces.sum() - want).max() < 0.03
    # the same point as the reference-shaped sampler
    P = (C.c_float * 3)()
    L.orc_random_point_on_cube(C.byref(light), 77.0, P)
    L.orc_sample_light(C.byref(light), 77.0, Q, inv)
    assert list(P) == list(Q)


def test_sphere_light_sample_density_integrates_to_the_area():
    """E[valid / pdf] = area of the sphere; samples outside the disk are flagged unusable."""
    sc = _sphere_light_scene()
    light = sc.geoms[5]
    L = orc.lib()
    Q, inv = (C.c_float * 3)(), (C.c_float * 1)()
    total, valid, n = 0.0, 0, 20000
    centre = np.array([light.transform[3], light.transform[7], light.transform[11]], np.float64)
    for seed in range(n):
        ok = L.orc_sample_light(C.byref(light), float(seed), Q, inv)
        if ok:
            valid += 1
            total += inv[0]
            assert abs(np.linalg.norm(np.array(list(Q), np.float64) - centre) - 0.75) < 1e-4
    assert abs(valid / n - math.pi / 4) < 0.02                         # the disk inside the unit square
    area = 4 * math.pi * 0.75 ** 2
    assert abs(total / n - area) < 0.03 * area


# ---------------------------------------------------------------- CPU: estimator ------------

def _row_means(sc, depth, direct, first, n):
    img, _ = orc.render(sc, orc.default_config(depth, direct_light=direct), first, n)
    return (img / n).mean(axis=(1, 2)), img / n


@pytest.mark.parametrize("maker,depth", [(lambda: _scene(keep=[0, 1, 2, 3, 4, 8]), 3), (lambda: _scene("cornell_mirror"), 4),
                                         (_sphere_light_scene, 3)])
def test_same_expectation_as_plain_path_tracing(maker, depth):
    """Unbiasedness: with shadow rays at bounces < depth-1 and emitter hits counted only for camera
    rays / after specular events, every light path of the plain estimator is counted exactly once.
    (Rows that look at the ceiling right beside the light are left out: 1/d^2 there makes the direct
    estimator heavy-tailed -- unbiased, but not within 2 % after a few thousand samples.)"""
    sc = maker()
    n = 6000
    plain, _ = _row_means(sc, depth, 0, 1, n)
    direct, _ = _row_means(sc, depth, 1, 1, n)
    rows = slice(8, 32)
    assert abs(direct[rows].sum() / plain[rows].sum() - 1) < 0.02
    for band in (slice(8, 16), slice(16, 24), slice(24, 32)):         # and band by band (fireflies allow no per-row bound)
        assert abs(direct[band].sum() / plain[band].sum() - 1) < 0.05


def test_less_noise_than_plain_on_the_floor():
    sc = _scene(keep=[0, 1, 2, 3, 4, 8])
    _, a1 = _row_means(sc, 3, 0, 1, 300)
    _, a2 = _row_means(sc, 3, 0, 301, 300)
    _, b1 = _row_means(sc, 3, 1, 1, 300)
    _, b2 = _row_means(sc, 3, 1, 301, 300)
    floor = slice(22, 32)
    # medians: the area-sampled estimator is heavy-tailed (one firefly of 26 among 19 200 pixel values moves the mean
    # by more than the whole plain-estimator noise), its typical pixel is ~7x quieter
    assert np.median(np.abs(b1[floor] - b2[floor])) < 0.5 * np.median(np.abs(a1[floor] - a2[floor]))


def test_directly_visible_emitter_pixels_are_unchanged():
    """Camera rays that hit the light add thr*Le in both modes -- bit for bit."""
    sc = _scene(w=48, h=48)
    a, _ = orc.render(sc, orc.default_config(1), 1, 1)                 # depth 1: only primary emitter hits
    b, _ = orc.render(sc, orc.default_config(1, direct_light=1), 1, 1)
    assert a.max() > 0 and np.array_equal(a, b)
    c, _ = orc.render(sc, orc.default_config(4, direct_light=1), 1, 1)
    lit = a.sum(axis=2) > 0
    assert np.array_equal(c[lit], a[lit])                              # the path ended there


def test_pool_carries_the_count_emission_flag():
    sc = _scene("cornell_mirror", 40, 30)
    n0, _, pix0 = orc.trace_pool(sc, orc.default_config(6, direct_light=1), 2, 0)
    assert n0 == 1200 and np.all(pix0 >> 31 == 1)                      # camera rays count emission
    n1, _, pix1 = orc.trace_pool(sc, orc.default_config(6, direct_light=1), 2, 1)
    flags = pix1 >> 31
    assert 0 < flags.sum() < n1                                        # mirror hits set it, diffuse hits clear it
    _, _, plain = orc.trace_pool(sc, orc.default_config(6), 2, 1)
    assert np.array_equal(pix1 & 0x7FFFFFFF, plain)                    # same survivors, same order


def test_no_emitters_means_no_contribution():
    sc = _scene(keep=[0, 1, 2, 3, 4, 5])
    img, live = orc.render(sc, orc.default_config(4, direct_light=1), 1, 2)
    assert not img.any() and live[0] == 2 * 32 * 32


# ---------------------------------------------------------------- GPU: parity with the oracle --

@pytest.mark.gpu
@pytest.mark.parametrize("name,depth,iters,kw", [
    ("sampleScene", 8, 5, dict()),
    ("cornell_mirror", 8, 4, dict(batch=1)),
    ("cornell_mirror", 6, 5, dict(batch=2, chunk_rays=100)),
    ("cornell_glass_4k", 12, 3, dict(camera_mode=1, antialias=1, aperture=0.25, focal_distance=12.0)),
    ("random256", 8, 2, dict()),
    ("sampleScene", 2, 3, dict(ordering=1)),                           # ordering 1 is ignored, not an error
    # ordering = 2 on <= 32 primitives: whole paths in one launch, the shadow rays as records of the typed queues (k_path_q<NEE>)
    ("sampleScene", 3, 2, dict(ordering=2)),
    ("sampleScene", 8, 5, dict(ordering=2)),
    ("cornell_mirror", 8, 4, dict(ordering=2, batch=1)),
    ("cornell_mirror", 6, 5, dict(ordering=2, batch=2, chunk_rays=128)),
    ("cornell_mirror", 8, 3, dict(ordering=2, streams=2)),
    ("cornell_glass_4k", 12, 3, dict(ordering=2, camera_mode=1, antialias=1, aperture=0.25, focal_distance=12.0)),
    ("random256", 8, 2, dict(ordering=2)),                             # more than 32 primitives: the per-bounce kernels, as before
])
def test_gpu_image_and_live_counts_match_oracle(pt, name, depth, iters, kw):
    sc = orc.load_golden_scene(name).with_resolution(160, 120)
    okw = {k: v for k, v in kw.items() if k in ("camera_mode", "antialias", "aperture", "focal_distance")}
    tr = make_tracer(sc, depth=depth, direct_light=1, **kw)
    tr.set_image(None)
    tr.render(1, iters)
    img, st = tr.image(), tr.stats()
    want, live = orc.render(sc, oracle_config(depth, direct_light=1, **okw), 1, iters)
    assert [st.live[k] for k in range(depth + 1)] == [int(v) for v in live]
    assert np.array_equal(img, want)
    assert img.max() > 0 and not np.isnan(img).any()


@pytest.mark.gpu
@pytest.mark.parametrize("ordering", [0, 2])
def test_gpu_sphere_emitter_and_host_image_round_trip(pt, ordering):
    sc = _sphere_light_scene(96, 64)
    tr = make_tracer(sc, depth=5, direct_light=1, ordering=ordering)
    start = np.random.default_rng(3).random((64, 96, 3)).astype(np.float32)
    tr.set_image(start)
    tr.render(1, 2)
    mid = tr.image()
    tr.set_image(mid)
    tr.render(3, 2)
    want, _ = orc.render(sc, oracle_config(5, direct_light=1), 1, 4, image=start.copy())
    assert np.array_equal(tr.image(), want)


@pytest.mark.gpu
@pytest.mark.parametrize("bounces", [0, 1, 3])
def test_gpu_pool_and_flag_bit_exact(pt, bounces):
    sc = orc.load_golden_scene("cornell_mirror").with_resolution(120, 90)
    tr = make_tracer(sc, depth=6, direct_light=1)
    n, arrs, pix = tr.trace_pool(2, bounces)
    on, oarrs, opix = orc.trace_pool(sc, oracle_config(6, direct_light=1), 2, bounces)
    assert n == on and np.array_equal(pix, opix)
    for a, b in zip(arrs, oarrs):
        assert np.array_equal(a, b)
    # the hook leaves image, planes and statistics untouched
    tr.set_image(None)
    tr.render(1, 2)
    want, _ = orc.render(sc, oracle_config(6, direct_light=1), 1, 2)
    assert np.array_equal(tr.image(), want)


@pytest.mark.gpu
@pytest.mark.parametrize("ordering", [0, 2])
def test_gpu_row_shards_sum_to_the_full_frame(pt, ordering):
    sc = orc.load_golden_scene("sampleScene").with_resolution(128, 96)
    want, _ = orc.render(sc, oracle_config(6, direct_light=1), 1, 3)
    total = np.zeros_like(want)
    for r in range(3):
        tr = make_tracer(sc, depth=6, direct_light=1, row_offset=r, row_stride=3, ordering=ordering)
        tr.set_image(None)
        tr.render(1, 3)
        part = tr.image()
        mine, _ = orc.render(sc, oracle_config(6, direct_light=1, row_offset=r, row_stride=3), 1, 3)
        assert np.array_equal(part, mine)
        total += part
    assert np.array_equal(total, want)


@pytest.mark.gpu
def test_gpu_rejects_unsupported_combinations(pt):
    sc = orc.load_golden_scene("sampleScene").with_resolution(64, 48)
    for kw in (dict(culling=1),):
        cfg = pt.default_config(max_depth=4, direct_light=1, **kw)
        tr = pt.PathTracer(cfg)
        with pytest.raises(pt.PtError, match="direct_light"):
            tr.upload(*to_product(sc))
        tr.close()


@pytest.mark.gpu
@pytest.mark.parametrize("ordering", [0, 2])
def test_gpu_full_size_1080p_one_iteration(pt, ordering):
    """BASELINE configs[2] geometry at full size with shadow rays: image and live counts equal the oracle."""
    sc = orc.load_golden_scene("cornell_mirror")
    tr = make_tracer(sc, depth=8, direct_light=1, ordering=ordering)
    tr.set_image(None)
    tr.render(1, 1)
    want, live = orc.render(sc, oracle_config(8, direct_light=1), 1, 1)
    st = tr.stats()
    assert [st.live[k] for k in range(9)] == [int(v) for v in live]
    assert np.array_equal(tr.image(), want)
