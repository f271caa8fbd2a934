"""-m gpu: the HIP path (through the C ABI) against the CPU oracle on identical seeds.

Bar: every kernel here is integer/IEEE-binary32 work with a fixed expression order on both
sides, so the comparison is BIT-EXACT (numpy == on float32, i.e. -0 == +0) unless a test states
a tolerance.  The north-star image tolerance (L2 error < 1e-4 per channel) is asserted too."""
import ctypes as C
import hashlib
import json
import os

import numpy as np
import pytest

import orc
from gpu_common import make_tracer, oracle_config, rel_l2, to_product

pytestmark = pytest.mark.gpu

K = json.load(open(os.path.join(orc.GOLD, "survey_kats.json")))


@pytest.fixture(scope="module")
def cornell200():
    return orc.load_golden_scene("sampleScene").with_resolution(200, 200)


def test_device_present_and_native_library_loaded(pt):
    assert pt.lib().pt_device_count() >= 1
    tr = pt.PathTracer()
    tr.close()


def test_rng_from_thread_kat_and_sweep(pt, cornell200):
    tr = make_tracer(cornell200)
    r = K["rng_from_thread"]
    out = tr.rng_from_thread(800, 800, r["time"], [[r["x"], r["y"]]])
    for got, want in zip(out[0], r["out"]):
        assert float("%.9g" % got) == want
    xy = np.stack(np.meshgrid(np.arange(0, 800, 37), np.arange(0, 800, 41)), -1).reshape(-1, 2)
    got = tr.rng_from_thread(800, 800, 5.0, xy)
    want = np.zeros_like(got)
    o = (C.c_float * 3)()
    for i, (x, y) in enumerate(xy):
        orc.lib().orc_rng_from_thread(800, 800, 5.0, int(x), int(y), o)
        want[i] = list(o)
    assert np.array_equal(got, want)


def test_sincos_polynomial_bit_exact(pt, cornell200):
    tr = make_tracer(cornell200)
    rng = np.random.default_rng(1)
    a = np.concatenate([rng.random(200000).astype(np.float32) * np.float32(6.2831855),
                        np.array([0, 6.2831855, 3.1415927, 1.5707964, 4.712389, 1e-30, 0.7853982], np.float32)])
    s, c = tr.sincos(a)
    ws, wc = np.zeros_like(a), np.zeros_like(a)
    L = orc.lib()
    sv, cv = C.c_float(), C.c_float()
    for i in range(0, len(a), 97):                      # sample (python loop), bit-exact
        L.orc_sincos(a[i], C.byref(sv), C.byref(cv))
        assert s[i] == sv.value and c[i] == cv.value
    # and accurate: within 2e-7 of the true values everywhere
    assert np.abs(s - np.sin(a.astype(np.float64))).max() < 2e-7
    assert np.abs(c - np.cos(a.astype(np.float64))).max() < 2e-7


def test_hemisphere_bit_exact(pt, cornell200):
    tr = make_tracer(cornell200)
    rng = np.random.default_rng(2)
    n = rng.normal(size=(5000, 3)).astype(np.float32)
    n /= np.linalg.norm(n, axis=1, keepdims=True).astype(np.float32)
    n[:6] = [[0, 1, 0], [1, 0, 0], [0, 0, 1], [0, -1, 0], [-1, 0, 0], [0, 0, -1]]
    xi = rng.random((5000, 2)).astype(np.float32)
    xi[0] = [0.25, 0.5]
    got = tr.hemisphere(n, xi)
    want = np.zeros_like(got)
    o = (C.c_float * 3)()
    for i in range(len(n)):
        orc.lib().orc_hemisphere(orc.vec3(*n[i]), xi[i, 0], xi[i, 1], o)
        want[i] = list(o)
    assert np.array_equal(got, want)
    for g, w in zip(got[0], K["hemisphere"]["out"]):
        assert float("%.9g" % g) == w


def test_primary_hits_bit_exact_and_reference_kats(pt):
    sc = orc.load_golden_scene("sampleScene")          # 800x800, the survey's KAT configuration
    tr = make_tracer(sc)
    d, hit, t, P, N = tr.primary_hits()
    for h in K["primary_hits_800"]:
        idx = h["pixel"][0] + h["pixel"][1] * 800
        assert hit[idx] == h["obj"]
        for g, w in zip(d[idx], h["dir"]):
            assert float("%.9g" % g) == w
        if sc.geoms[h["obj"]].type == 1:
            assert float("%.9g" % t[idx]) == h["t"]
    # every pixel against the oracle, bit for bit
    L = orc.lib()
    cb = orc.CameraBasis()
    L.orc_camera_setup(C.byref(sc.camera), C.byref(cb))
    ga = sc.geom_array()
    o, dd, PP, NN, tt = (C.c_float * 3)(), (C.c_float * 3)(), (C.c_float * 3)(), (C.c_float * 3)(), C.c_float()
    rng = np.random.default_rng(3)
    for idx in np.concatenate([rng.integers(0, 640000, 4000), [0, 799, 639999, 320400]]):
        x, y = int(idx % 800), int(idx // 800)
        L.orc_camera_ray(C.byref(cb), None, x, y, 0, 0, 0, 0, o, dd)
        h = L.orc_nearest_hit(ga, sc.G, None, o, dd, C.byref(tt), PP, NN)
        assert h == hit[idx]
        assert np.array_equal(np.array(list(dd), np.float32), d[idx])
        assert tt.value == t[idx]
        assert np.array_equal(np.array(list(PP), np.float32), P[idx]) and np.array_equal(np.array(list(NN), np.float32), N[idx])
    _, ohit = orc.raycast_flat(sc)
    assert np.array_equal(ohit.reshape(-1), hit)


def test_config1_reference_kernel_mode_matches_reference_image(pt):
    """BASELINE config 1: 400x400, one hit, one iteration -- the reference kernel as shipped."""
    sc = orc.load_golden_scene("cornell_c1")
    tr = make_tracer(sc, depth=1, mode=1)
    tr.set_image(None)
    tr.render(1, 1)
    img = tr.image()
    want, _ = orc.raycast_flat(sc)
    assert np.array_equal(img, want)
    np.testing.assert_allclose(img.reshape(-1, 3).mean(0, dtype=np.float64), K["flat_image_mean_rgb"]["400"], atol=1e-6)
    u8 = pt.image_to_u8(img, 1, np.float32(1.0 / 2.2))
    assert hashlib.sha256(u8.tobytes()).hexdigest() == K["c1_bmp"]["raster_sha256"]
    # sendImageToPBO
    disp = tr.display(1.0)
    o = (C.c_uint8 * 4)()
    for idx in (0, 12345, 159999):
        orc.lib().orc_display_pixel(orc.vec3(*img.reshape(-1, 3)[idx]), o)
        assert list(disp.reshape(-1, 4)[idx]) == list(o)


@pytest.mark.parametrize("bounces", [0, 1, 2, 5])
def test_ray_pool_bit_exact_after_k_bounces(pt, cornell200, bounces):
    """Stable compaction: the pool after k bounces equals the oracle's survivors in generation order."""
    tr = make_tracer(cornell200)
    n, arrs, pix = tr.trace_pool(3, bounces)
    on, oarrs, opix = orc.trace_pool(cornell200, oracle_config(8), 3, bounces)
    assert n == on
    assert np.array_equal(pix, opix)
    assert np.all(np.diff(pix.astype(np.int64)) > 0)    # sortedness: generation (pixel-major) order kept
    for a, b in zip(arrs, oarrs):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("scene_name,depth,iters", [("sampleScene", 8, 6), ("cornell_mirror", 8, 4), ("cornell_glass_4k", 12, 4), ("random256", 8, 3)])
def test_image_and_live_counts_match_oracle(pt, scene_name, depth, iters):
    sc = orc.load_golden_scene(scene_name).with_resolution(160, 120)
    tr = make_tracer(sc, depth=depth)
    tr.set_image(None)
    tr.render(1, iters)
    img = tr.image()
    st = tr.stats()
    want, live = orc.render(sc, oracle_config(depth), 1, iters)
    assert [st.live[k] for k in range(depth + 1)] == [int(v) for v in live]
    assert np.array_equal(img, want)                                   # bit-exact
    for ch in range(3):                                                # the north-star tolerance
        assert rel_l2(img[..., ch] / iters, want[..., ch] / iters) < 1e-4
    assert not np.isnan(img).any()


@pytest.mark.parametrize("kw", [dict(), dict(ordering=1), dict(ordering=2), dict(direct_light=1)])
def test_long_launch_groups_use_every_slot_bit(pt, kw):
    """Small frames batch up to 128 iterations into one launch group (slot = bits 24..30 of the pixel
    word, bit 31 = the direct-light flag): 150 iterations = one full group + a partial one."""
    sc = orc.load_golden_scene("cornell_mirror").with_resolution(64, 48)
    tr = make_tracer(sc, depth=4, **kw)
    tr.set_image(None)
    tr.render(1, 150)
    okw = {k: v for k, v in kw.items() if k == "direct_light"}
    want, live = orc.render(sc, oracle_config(4, **okw), 1, 150)
    st = tr.stats()
    assert [st.live[k] for k in range(5)] == [int(v) for v in live]
    assert np.array_equal(tr.image(), want)


def test_accumulation_continues_from_host_image(pt, cornell200):
    """camera::image is in/out: rendering 2+3 iterations with a host round trip == 5 in one go."""
    tr = make_tracer(cornell200)
    tr.set_image(None)
    tr.render(1, 2)
    part = tr.image()
    tr.set_image(part)
    tr.render(3, 3)
    a = tr.image()
    want, _ = orc.render(cornell200, oracle_config(8), 1, 5)
    assert np.array_equal(a, want)


@pytest.mark.parametrize("kw", [dict(chunk_rays=64), dict(batch=1), dict(batch=2), dict(batch=3, chunk_rays=100), dict(chunk_rays=1000), dict(chunk_rays=16), dict(blocks_per_cu=1),
                                dict(culling=1), dict(ordering=1), dict(ordering=1, batch=2), dict(ordering=1, batch=1, chunk_rays=64), dict(ordering=1, chunk_rays=128, blocks_per_cu=1), dict(ordering=1, batch=5, blocks_per_cu=2),
                                dict(ordering=2), dict(ordering=2, batch=2), dict(ordering=2, batch=1, chunk_rays=64), dict(ordering=2, chunk_rays=192, blocks_per_cu=1), dict(ordering=2, batch=5, blocks_per_cu=2)])
def test_launch_variants_are_bit_identical(pt, cornell200, kw):
    ref = make_tracer(cornell200)
    ref.set_image(None); ref.render(1, 3)
    var = make_tracer(cornell200, **kw)
    var.set_image(None); var.render(1, 3)
    assert np.array_equal(ref.image(), var.image())
    n1, a1, p1 = ref.trace_pool(2, 3)
    n2, a2, p2 = var.trace_pool(2, 3)
    if kw.get("ordering"):            # the sparse-work queue keeps every ray in its segment, not its order in it
        order = np.argsort(p2, kind="stable")
        p2, a2 = p2[order], [x[order] for x in a2]
    assert n1 == n2 and np.array_equal(p1, p2) and all(np.array_equal(x, y) for x, y in zip(a1, a2))


def test_antialias_and_thin_lens_match_oracle(pt):
    sc = orc.load_golden_scene("cornell_glass_4k").with_resolution(128, 72)
    kw = dict(camera_mode=1, antialias=1, aperture=0.25, focal_distance=12.0)
    tr = make_tracer(sc, depth=6, **kw)
    tr.set_image(None); tr.render(1, 4)
    want, live = orc.render(sc, oracle_config(6, **kw), 1, 4)
    assert np.array_equal(tr.image(), want)
    n, arrs, pix = tr.trace_pool(1, 0)
    on, oarrs, opix = orc.trace_pool(sc, oracle_config(6, **kw), 1, 0)
    assert n == on and all(np.array_equal(a, b) for a, b in zip(arrs, oarrs))


def test_row_sharded_contexts_sum_to_the_full_frame(pt, cornell200):
    """Multi-GPU decomposition on one device: shards own interleaved rows; summing the per-shard
    accumulators (zeros elsewhere) is bit-identical to the unsharded render."""
    full = make_tracer(cornell200)
    full.set_image(None); full.render(1, 3)
    want = full.image()
    total = np.zeros_like(want)
    owned = 0
    for r in range(3):
        sh = make_tracer(cornell200, row_offset=r, row_stride=3)
        sh.set_image(None); sh.render(1, 3)
        part = sh.image()
        rows = np.arange(part.shape[0]) % 3 == r
        assert not part[~rows].any()
        total += part
        owned += sh.owned
    assert owned == 200 * 200
    assert np.array_equal(total, want)


@pytest.mark.parametrize("contexts,kw", [(2, dict(ordering=1)), (3, dict()), (2, dict(direct_light=1)), (3, dict(ordering=2))])
def test_concurrent_contexts_share_one_device_image(pt, contexts, kw):
    """What bench.py --streams does: several contexts on ONE device, each on its own stream and owning every
    n-th row, all bound to the same device accumulator and enqueued before any is awaited.  Their launches
    overlap on the GPU; the frame is the single-context frame bit for bit."""
    # the shared accumulator comes from the HIP runtime the library itself is linked against (no torch here:
    # a second HIP runtime in the process could not open the device)
    hip = C.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipFree.argtypes = [C.c_void_p]
    sc = orc.load_golden_scene("cornell_mirror").with_resolution(192, 108)
    nbytes = 108 * 192 * 3 * 4
    trs = [make_tracer(sc, depth=6, row_offset=r, row_stride=contexts, **kw) for r in range(contexts)]
    dptr = C.c_void_p()
    assert hip.hipMalloc(C.byref(dptr), nbytes) == 0 and hip.hipMemset(dptr, 0, nbytes) == 0 and hip.hipDeviceSynchronize() == 0
    for tr in trs:
        assert pt.lib().pt_bind_device_image(tr._h, dptr) == 0
    for first in (1, 21):
        for tr in trs:
            tr.render(first, 20)
    for tr in trs:
        tr.sync()
    got = np.zeros((108, 192, 3), np.float32)
    assert hip.hipMemcpy(got.ctypes.data_as(C.c_void_p), dptr, nbytes, 2) == 0          # hipMemcpyDeviceToHost
    okw = {k: v for k, v in kw.items() if k == "direct_light"}
    want, live = orc.render(sc, oracle_config(6, **okw), 1, 40)
    assert np.array_equal(got, want)
    total = [sum(int(tr.stats().live[k]) for tr in trs) for k in range(7)]
    assert total == [int(v) for v in live]
    for tr in trs:
        tr.close()
    hip.hipFree(dptr)


@pytest.mark.parametrize("streams,kw,depth", [(2, dict(ordering=1), 8), (3, dict(), 6), (2, dict(direct_light=1), 6), (2, dict(ordering=2), 8), (3, dict(row_offset=2, row_stride=3, ordering=2, batch=2), 5),
                                               (2, dict(row_offset=1, row_stride=2, ordering=1), 6), (4, dict(batch=3), 5)])
def test_streams_inside_one_context_are_bit_identical(pt, streams, kw, depth):
    """pt_config.streams: the context shards its rows over internal contexts on separate streams sharing one image.
    Image, live counts, display bytes and the host-image round trip equal the oracle / the one-stream context."""
    sc = orc.load_golden_scene("cornell_mirror").with_resolution(160, 90)
    tr = make_tracer(sc, depth=depth, streams=streams, **kw)
    start = np.random.default_rng(9).random((90, 160, 3)).astype(np.float32)
    tr.set_image(start)
    tr.render(1, 7)
    mid = tr.image()
    tr.set_image(mid)
    tr.render(8, 5)
    okw = {k: v for k, v in kw.items() if k in ("direct_light", "row_offset", "row_stride")}
    want, live = orc.render(sc, oracle_config(depth, **okw), 1, 12, image=start.copy())
    got = tr.image()
    assert np.array_equal(got, want)
    st = tr.stats()
    assert [st.live[k] for k in range(depth + 1)] == [int(v) for v in live]
    assert tr.owned == sum(1 for y in range(90) if y % kw.get("row_stride", 1) == kw.get("row_offset", 0)) * 160
    # sendImageToPBO over the shared frame
    disp = tr.display(1.0 / 12)
    ref = np.zeros((90 * 160, 4), np.uint8)
    px = (C.c_uint8 * 4)()
    flat = (want.reshape(-1, 3) * np.float32(1.0 / 12)).astype(np.float32)
    for i in range(0, 90 * 160, 97):
        orc.lib().orc_display_pixel(orc.vec3(*flat[i]), px)
        assert list(disp.reshape(-1, 4)[i]) == list(px)
    with pytest.raises(pt.PtError, match="streams = 1"):
        tr.trace_pool(1, 1)
    tr.close()


def test_streams_survive_a_scene_change(pt):
    """Re-uploading another scene at another resolution into a streams = 2 context (what the adaptor does when the
    caller's scene content changes): buffers are rebuilt, the shared image is re-created and re-bound."""
    a = orc.load_golden_scene("cornell_mirror").with_resolution(64, 48)
    b = orc.load_golden_scene("sampleScene").with_resolution(96, 64)
    tr = make_tracer(a, depth=5, streams=2, ordering=1)
    tr.set_image(None)
    tr.render(1, 3)
    wa, _ = orc.render(a, oracle_config(5), 1, 3)
    assert np.array_equal(tr.image(), wa)
    tr.upload(*to_product(b))
    tr.set_image(None)
    tr.render(1, 4)
    wb, live = orc.render(b, oracle_config(5), 1, 4)
    assert tr.image().shape == (64, 96, 3) and np.array_equal(tr.image(), wb)
    tr.close()


def test_empty_and_tiny_inputs(pt):
    """Edge cases: a scene whose rays all miss (live count drops to 0 after the first bounce), a
    2x2 frame, a frame narrower than one wave."""
    sc = orc.load_golden_scene("sampleScene").with_resolution(2, 2)
    tr = make_tracer(sc)
    tr.set_image(None); tr.render(1, 2)
    want, live = orc.render(sc, oracle_config(8), 1, 2)
    assert np.array_equal(tr.image(), want)
    # camera looking away from everything: every primary ray misses
    away = orc.load_golden_scene("sampleScene").with_resolution(37, 5)
    away.camera.view[2] = 1.0
    tr2 = make_tracer(away)
    tr2.set_image(None); tr2.render(1, 2)
    st = tr2.stats()
    want2, live2 = orc.render(away, oracle_config(8), 1, 2)
    assert [st.live[k] for k in range(9)] == [int(v) for v in live2]
    assert st.live[1] == 0 and not tr2.image().any()


def test_full_size_properties_1080p(pt):
    """BASELINE config 3 size (1920x1080, 8 bounces, mirrors): properties that need no oracle run --
    monotone live counts, sorted compacted pool, run-to-run determinism, shard sum == full."""
    sc = orc.load_golden_scene("cornell_mirror")
    assert (sc.W, sc.H) == (1920, 1080)
    tr = make_tracer(sc)
    tr.set_image(None); tr.render(1, 2)
    a = tr.image()
    st = tr.stats()
    live = [st.live[k] for k in range(9)]
    assert live[0] == 2 * 1920 * 1080 and all(live[k] >= live[k + 1] for k in range(8)) and live[8] > 0
    tr.set_image(None); tr.reset_stats(); tr.render(1, 2)
    assert np.array_equal(a, tr.image())                           # deterministic
    n, arrs, pix = tr.trace_pool(1, 4)
    assert n == len(pix) and np.all(np.diff(pix.astype(np.int64)) > 0)
    assert np.isfinite(a).all() and a.max() > 0
    # oracle on a thin slice of rows of the same frame (row interleave 135 -> 8 rows), bit-exact
    sh = make_tracer(sc, row_offset=7, row_stride=135)
    sh.set_image(None); sh.render(1, 2)
    want, _ = orc.render(sc, oracle_config(8, row_offset=7, row_stride=135), 1, 2)
    got = sh.image()
    assert np.array_equal(got, want)
    rows = np.arange(1080) % 135 == 7
    assert np.array_equal(got[rows], a[rows])


@pytest.mark.parametrize("kw", [dict(), dict(batch=2, chunk_rays=100), dict(culling=1), dict(streams=2), dict(direct_light=1),
                                dict(ordering=1), dict(ordering=1, batch=3, chunk_rays=128), dict(ordering=1, streams=2),
                                dict(ordering=2), dict(ordering=2, batch=3, chunk_rays=128), dict(ordering=2, streams=2), dict(ordering=2, grid_density=1), dict(ordering=2, grid_density=16),
                                dict(ordering=2, chunk_rays=64, batch=1), dict(ordering=2, blocks_per_cu=1, batch=2)])
def test_many_primitives_scene_matches_oracle(pt, kw):
    """BASELINE config 4's scene (256 spheres+cubes incl. rotated cubes, mirrors, glass): the two-level
    candidate culling must never change the nearest hit."""
    sc = orc.load_golden_scene("random256").with_resolution(192, 108)
    tr = make_tracer(sc, depth=8, **kw)
    tr.set_image(None); tr.render(1, 2)
    okw = {k: v for k, v in kw.items() if k == "direct_light"}
    want, live = orc.render(sc, oracle_config(8, **okw), 1, 2)
    st = tr.stats()
    assert [st.live[k] for k in range(9)] == [int(v) for v in live]
    assert np.array_equal(tr.image(), want)
    if kw.get("streams", 1) == 1:
        n, arrs, pix = tr.trace_pool(2, 3)
        on, oarrs, opix = orc.trace_pool(sc, oracle_config(8, **okw), 2, 3)
        if kw.get("ordering"):             # the typed work queues keep the set of rays, not their order
            order = np.argsort(pix, kind="stable")
            pix, arrs = pix[order], [x[order] for x in arrs]
        assert n == on and np.array_equal(pix, opix) and all(np.array_equal(a, b) for a, b in zip(arrs, oarrs))


def test_stats_survive_the_parity_hook_and_repeated_iterations(pt, cornell200):
    """live[] must count each rendered iteration once, whatever is interleaved: the pool hook, the
    same iteration index rendered twice, odd/even iteration numbers."""
    tr = make_tracer(cornell200)
    tr.set_image(None)
    _, live1 = orc.render(cornell200, oracle_config(8), 1, 1)
    _, live3 = orc.render(cornell200, oracle_config(8), 3, 1)
    tr.render(1, 1)
    tr.trace_pool(5, 3)
    tr.render(3, 1)
    tr.trace_pool(2, 0)
    tr.render(3, 1)
    st = tr.stats()
    assert [st.live[k] for k in range(9)] == [int(a) + 2 * int(b) for a, b in zip(live1, live3)]


def test_config2_full_size_against_oracle(pt):
    """BASELINE configs[1]: the sampleScene-equivalent Cornell box at its own 800x800, 8 bounces,
    diffuse only (a few iterations of the 1000; the comparison is per iteration-exact anyway)."""
    sc = orc.load_golden_scene("cornell")
    assert (sc.W, sc.H) == (800, 800)
    tr = make_tracer(sc)
    tr.set_image(None); tr.render(1, 4)
    want, live = orc.render(sc, oracle_config(8), 1, 4)
    st = tr.stats()
    assert [st.live[k] for k in range(9)] == [int(v) for v in live]
    got = tr.image()
    assert np.array_equal(got, want)
    for ch in range(3):
        assert rel_l2(got[..., ch] / 4, want[..., ch] / 4) < 1e-4


def test_config4_and_config5_full_size_slices(pt):
    """configs[3] (1080p, 256 primitives) and configs[4] (3840x2160, 16 bounces, Fresnel glass + thin
    lens + jittered AA): full-size renders checked on a thin interleave of rows against the oracle,
    plus size-independent properties (sorted compacted stream, monotone live counts)."""
    c4 = orc.load_golden_scene("random256")
    assert (c4.W, c4.H, c4.G) == (1920, 1080, 256)
    tr = make_tracer(c4)
    tr.set_image(None); tr.render(1, 1)
    full = tr.image()
    st = tr.stats()
    assert all(st.live[k] >= st.live[k + 1] for k in range(8)) and st.live[0] == 1920 * 1080
    sh_cfg = dict(row_offset=11, row_stride=270)
    want, _ = orc.render(c4, oracle_config(8, **sh_cfg), 1, 1)
    rows = np.arange(1080) % 270 == 11
    assert np.array_equal(full[rows], want[rows])
    # what bench.py --workload c4 times: the whole-path kernel for 33..256 primitives (k_path_w), two contexts
    for kwq in (dict(ordering=2, streams=2), dict(ordering=2, streams=1), dict(ordering=1, streams=2)):
        trq = make_tracer(c4, **kwq)
        trq.set_image(None); trq.render(1, 1)
        stq = trq.stats()
        assert np.array_equal(trq.image(), full) and [stq.live[k] for k in range(9)] == [st.live[k] for k in range(9)], kwq
        assert stq.emitted == st.emitted
        trq.close()
    n, arrs, pix = tr.trace_pool(1, 3)
    assert np.all(np.diff(pix.astype(np.int64)) > 0)

    c5 = orc.load_golden_scene("cornell_glass_4k")
    assert (c5.W, c5.H) == (3840, 2160)
    kw = dict(camera_mode=1, antialias=1, aperture=0.25, focal_distance=12.0)
    tr5 = make_tracer(c5, depth=16, **kw)
    tr5.set_image(None); tr5.render(1, 1)
    full5 = tr5.image()
    st5 = tr5.stats()
    assert st5.live[0] == 3840 * 2160 and all(st5.live[k] >= st5.live[k + 1] for k in range(16))
    want5, _ = orc.render(c5, oracle_config(16, row_offset=5, row_stride=540, **kw), 1, 1)
    rows5 = np.arange(2160) % 540 == 5
    assert np.array_equal(full5[rows5], want5[rows5])
    assert np.isfinite(full5).all()


def _scaled_scene(name, factor, w, h):
    """The same scene with every length multiplied by `factor` (object transforms rebuilt with the
    oracle's transform builder = the reference's maths), camera moved accordingly."""
    import ctypes as C
    gold = json.load(open(os.path.join(orc.GOLD, "ref_scene_%s.json" % name)))
    base = orc.load_golden_scene(name).with_resolution(w, h)
    geoms = []
    xf, inv = np.zeros(16, np.float32), np.zeros(16, np.float32)
    for o, g0 in zip(gold["objects"], base.geoms):
        fr = o["frames"][0]
        t = [orc.f32_from_bits(v) * factor for v in fr["translation"]]
        r = [orc.f32_from_bits(v) for v in fr["rotation"]]
        s = [orc.f32_from_bits(v) * factor for v in fr["scale"]]
        orc.lib().orc_build_transform(orc.vec3(*t), orc.vec3(*r), orc.vec3(*s), orc.fptr(xf), orc.fptr(inv))
        g = orc.Geom()
        g.type, g.materialid = g0.type, g0.materialid
        for k in range(16):
            g.transform[k] = float(xf[k]); g.inverseTransform[k] = float(inv[k])
        geoms.append(g)
    cam = orc.Camera()
    C.memmove(C.byref(cam), C.byref(base.camera), C.sizeof(orc.Camera))
    for k in range(3):
        cam.position[k] = base.camera.position[k] * factor
    return orc.Scene(geoms, base.materials, cam)


@pytest.mark.parametrize("factor", [0.05, 1.0, 37.0, 1000.0])
@pytest.mark.parametrize("name,ordering", [("sampleScene", 0), ("random256", 0), ("random256", 1), ("random256", 2)])
def test_culling_is_conservative_at_other_scene_scales(pt, name, ordering, factor):
    """The culling margins are absolute + relative; RAY_BIAS / the sphere pull-back are absolute in the
    reference's own spec.  Whatever that does to the picture at odd scales, the culled nearest hit
    must still equal the brute-force (oracle) one bit for bit."""
    sc = _scaled_scene(name, factor, 128, 96)
    tr = make_tracer(sc, depth=6, ordering=ordering)
    tr.set_image(None); tr.render(1, 3)
    want, live = orc.render(sc, oracle_config(6), 1, 3)
    st = tr.stats()
    assert [st.live[k] for k in range(7)] == [int(v) for v in live]
    assert np.array_equal(tr.image(), want)


def test_light_sampling_helpers_bit_exact(pt):
    sc = orc.load_golden_scene("sampleScene").with_resolution(64, 64)
    tr = make_tracer(sc)
    seeds = np.arange(1, 4001, dtype=np.float32) * np.float32(7.25)
    out = (C.c_float * 3)()
    for gi, fn in ((8, orc.lib().orc_random_point_on_cube), (0, orc.lib().orc_random_point_on_cube), (5, orc.lib().orc_random_point_on_sphere)):
        got = tr.light_points(gi, seeds)
        want = np.zeros_like(got)
        for i, sd in enumerate(seeds):
            fn(C.byref(sc.geoms[gi]), float(sd), out)
            want[i] = list(out)
        assert np.array_equal(got, want, equal_nan=True)


@pytest.mark.parametrize("scene_name,depth,iters,kw", [
    ("sampleScene", 8, 5, dict()), ("cornell_mirror", 8, 4, dict()), ("cornell_glass_4k", 12, 3, dict()),
    ("cornell_glass_4k", 6, 3, dict(camera_mode=1, antialias=1, aperture=0.25, focal_distance=12.0))])
@pytest.mark.parametrize("ordering", [1, 2])
def test_sparse_work_queue_ordering_is_bit_identical(pt, scene_name, depth, iters, kw, ordering):
    """ordering=1 (typed work queues: one exact test per stage on full waves): same image, same live counts as the
    oracle; the pool holds the same set of rays."""
    sc = orc.load_golden_scene(scene_name).with_resolution(200, 150)
    tr = make_tracer(sc, depth=depth, ordering=ordering, **kw)
    tr.set_image(None); tr.render(1, iters)
    want, live = orc.render(sc, oracle_config(depth, **kw), 1, iters)
    st = tr.stats()
    assert [st.live[k] for k in range(depth + 1)] == [int(v) for v in live]
    assert np.array_equal(tr.image(), want)
    n, arrs, pix = tr.trace_pool(2, 3)
    on, oarrs, opix = orc.trace_pool(sc, oracle_config(depth, **kw), 2, 3)
    order = np.argsort(pix, kind="stable")
    assert n == on and np.array_equal(pix[order], opix) and all(np.array_equal(a[order], b) for a, b in zip(arrs, oarrs))
