"""-m gpu: scenes of MORE than 256 primitives.  The reference's nearest-hit loop takes any `numberOfGeoms`
(/root/reference/src/raytraceKernel.cu:137-153, cudaMalloc(numberOfGeoms * sizeof(staticGeom)) at :192-194); so must
every kernel family here: 257 (one beyond the byte ids), 600, and 1 500 primitives (the geometry table no longer fits
in LDS).  Image, live counts, emitter hits and the ray pool of one bounce, bit-exact against the oracle."""
import numpy as np
import pytest

import orc
from gpu_common import make_tracer, oracle_config

pytestmark = pytest.mark.gpu

_oracle_cache = {}


def _oracle(extra, depth, iters):
    key = (extra, depth, iters)
    if key not in _oracle_cache:
        sc = orc.many_primitives_scene(extra)
        want, live = orc.render(sc, oracle_config(depth), 1, iters)
        pool = orc.trace_pool(sc, oracle_config(depth), 2, 3)
        _oracle_cache[key] = (sc, want, live, pool)
    return _oracle_cache[key]


@pytest.mark.parametrize("kw", [dict(ordering=0, streams=1), dict(ordering=0, streams=2), dict(ordering=2, streams=1), dict(ordering=2, streams=2),
                                dict(ordering=2, streams=1, batch=3)])
@pytest.mark.parametrize("extra", [251, 594, 1494])
def test_more_than_256_primitives_match_the_oracle(pt, extra, kw):
    depth, iters = 6, 2
    sc, want, live, (on, oarrs, opix) = _oracle(extra, depth, iters)
    assert sc.G == extra + 6 and sc.G > 256
    tr = make_tracer(sc, depth=depth, **kw)
    tr.set_image(None)
    tr.render(1, iters)
    st = tr.stats()
    assert [st.live[k] for k in range(depth + 1)] == [int(v) for v in live], (sc.G, kw)
    assert np.array_equal(tr.image(), want), (sc.G, kw)
    if kw["ordering"] == 2:
        # whole paths on k_path_w with wide ids, whatever the count: the grid staged in LDS while it fits beside the waves' regions
        # (257 and 600 primitives), read from global memory beyond that (1 500: its blob alone exceeds what is left of the CU's LDS)
        shape = tr.path_shape()
        assert shape["family"] == "k_path_w" and shape["waves_per_block"] == 16, shape
        if sc.G == 1500:
            assert shape["lds_bytes"] < 140 * 1024 and shape["records_per_wave"] == 96, shape
        else:
            assert shape["lds_bytes"] > 140 * 1024, shape
    if kw.get("streams", 1) == 1:
        n, arrs, pix = tr.trace_pool(2, 3)
        order = np.argsort(pix, kind="stable")
        pix, arrs = pix[order], [x[order] for x in arrs]
        assert n == on and np.array_equal(pix, opix) and all(np.array_equal(a, b) for a, b in zip(arrs, oarrs)), (sc.G, kw)
    tr.close()


def test_more_than_256_primitives_with_direct_light_and_flat_mode(pt):
    """the other launch paths a big scene can take: direct light (per-bounce kernels) and mode 1 (the reference kernel as shipped)"""
    sc = orc.many_primitives_scene(594)
    tr = make_tracer(sc, depth=5, direct_light=1)
    tr.set_image(None); tr.render(1, 2)
    want, live = orc.render(sc, oracle_config(5, direct_light=1), 1, 2)
    st = tr.stats()
    assert [st.live[k] for k in range(6)] == [int(v) for v in live]
    assert np.array_equal(tr.image(), want)
    tr.close()
    flat, hit = orc.raycast_flat(sc)
    trf = make_tracer(sc, mode=1)
    trf.set_image(None); trf.render(1, 1)
    assert np.array_equal(trf.image(), flat)
    trf.close()


@pytest.mark.parametrize("extra,w,h,kw", [(250, 1920, 12, dict()), (250, 1920, 12, dict(antialias=1)), (250, 1280, 10, dict(camera_mode=1)),
                                          (594, 1920, 8, dict()), (250, 1920, 12, dict(camera_mode=1, aperture=0.2, focal_distance=11.0)),
                                          (250, 1000, 10, dict(streams=2))])
def test_camera_groups_on_wide_frames_match_the_oracle(pt, extra, w, h, kw):
    """k_path_w gives a group of 64 camera rays ONE cone test per primitive instead of 64 grid walks when the group spans a few
    degrees -- frames a thousand pixels wide and more.  Thin strips of such frames (1 920 / 1 280 / 1 000 pixels: groups that
    wrap around a row end included) against the oracle: reference camera, jitter, corrected pinhole, thin lens (no cone: walks)."""
    if extra == 250:
        sc = orc.load_golden_scene("random256").with_resolution(w, h)
    else:
        sc = orc.many_primitives_scene(extra, w=w, h=h)
    depth, iters = 5, 3
    okw = {k: v for k, v in kw.items() if k != "streams"}
    tr = make_tracer(sc, depth=depth, ordering=2, **kw)
    tr.set_image(None)
    tr.render(1, iters)
    want, live = orc.render(sc, oracle_config(depth, **okw), 1, iters)
    st = tr.stats()
    assert [st.live[k] for k in range(depth + 1)] == [int(v) for v in live], (extra, w, kw)
    assert np.array_equal(tr.image(), want), (extra, w, kw)
    if kw.get("streams", 1) == 1:
        n, arrs, pix = tr.trace_pool(2, 1)
        on, oarrs, opix = orc.trace_pool(sc, oracle_config(depth, **okw), 2, 1)
        order = np.argsort(pix, kind="stable")
        pix, arrs = pix[order], [x[order] for x in arrs]
        assert n == on and np.array_equal(pix, opix) and all(np.array_equal(a, b) for a, b in zip(arrs, oarrs)), (extra, w, kw)
    tr.close()


def test_1024_primitive_bench_scene_full_size_slices(pt):
    """bench.py --workload c1k (scenes/random1024.txt at its own 1920 x 1080, 8 bounces): the whole-path kernel with wide ids on two
    streams against the oracle on a thin interleave of rows, and against the stable per-bounce kernels' live counts."""
    sc = orc.load_golden_scene("random1024")
    assert (sc.W, sc.H, sc.G) == (1920, 1080, 1024)
    tr = make_tracer(sc, ordering=2, streams=2)
    tr.set_image(None); tr.render(1, 2)
    full = tr.image()
    st = tr.stats()
    assert st.live[0] == 2 * 1920 * 1080 and all(st.live[k] >= st.live[k + 1] for k in range(8))
    want, _ = orc.render(sc, oracle_config(8, row_offset=7, row_stride=360), 1, 2)
    rows = np.arange(1080) % 360 == 7
    assert np.array_equal(full[rows], want[rows])
    tr.close()


def test_5006_primitives_match_the_oracle(pt):
    """far beyond the byte ids: 5 006 primitives (17 x 17 x 17 cells and tens of thousands of references in global memory)"""
    sc = orc.many_primitives_scene(5000, w=96, h=54, size=(0.05, 0.25))
    want, live = orc.render(sc, oracle_config(5), 1, 2)
    for kw in (dict(ordering=2), dict(ordering=2, streams=2, batch=1), dict(ordering=0)):
        tr = make_tracer(sc, depth=5, **kw)
        tr.set_image(None); tr.render(1, 2)
        st = tr.stats()
        assert [st.live[k] for k in range(6)] == [int(v) for v in live], kw
        assert np.array_equal(tr.image(), want), kw
        tr.close()


@pytest.mark.parametrize("kind", ["cubes only", "spheres only, no room", "scaled x 37", "scaled x 0.05", "all in one spot", "glass and mirrors only"])
def test_wide_ids_on_other_kinds_of_scenes(pt, kind):
    """The wide-id form of k_path_w beyond the random room: one primitive type only (an empty typed stack), no enclosing room
    (most rays leave without a candidate), other scene scales (the grid's margins and the keys' distance steps are relative),
    hundreds of primitives piled into a few cells (long reference lists, lists that overflow), specular materials only (long
    paths: every bounce survives)."""
    import ctypes as C
    base = orc.many_primitives_scene(594, w=128, h=72)
    geoms, cam = list(base.geoms), base.camera
    if kind == "cubes only":
        geoms = [g for g in geoms if g.type == 1]
    elif kind == "spheres only, no room":
        geoms = [g for g in geoms[6:] if g.type == 0]
    elif kind.startswith("scaled"):
        f = float(kind.split("x")[1])
        sc2 = orc.many_primitives_scene(594, w=128, h=72)
        # rebuild every transform at the new scale with the oracle's transform builder (translation and scale multiplied)
        rng = np.random.default_rng(565)
        geoms = []
        xf, inv = np.zeros(16, np.float32), np.zeros(16, np.float32)
        gold = __import__("json").load(open(__import__("os").path.join(orc.GOLD, "ref_scene_random256.json")))
        for o, g0 in zip(gold["objects"][:6], sc2.geoms[:6]):
            fr = o["frames"][0]
            t = [orc.f32_from_bits(v) * f for v in fr["translation"]]; r = [orc.f32_from_bits(v) for v in fr["rotation"]]; sc_ = [orc.f32_from_bits(v) * f for v in fr["scale"]]
            orc.lib().orc_build_transform(orc.vec3(*t), orc.vec3(*r), orc.vec3(*sc_), orc.fptr(xf), orc.fptr(inv))
            g = orc.Geom(); g.type, g.materialid = g0.type, g0.materialid
            for k in range(16):
                g.transform[k] = float(xf[k]); g.inverseTransform[k] = float(inv[k])
            geoms.append(g)
        for i in range(594):
            c = [float(np.float32(v)) * f for v in (rng.uniform(-4.6, 4.6), rng.uniform(0.4, 9.2), rng.uniform(-4.6, 4.6))]
            sz = float(np.float32(rng.uniform(0.12, 0.5))) * f
            rot = [float(np.float32(v)) for v in rng.uniform(0, 360, 3)] if i & 1 else [0.0, 0.0, 0.0]
            orc.lib().orc_build_transform(orc.vec3(*c), orc.vec3(*rot), orc.vec3(sz, sz, sz), orc.fptr(xf), orc.fptr(inv))
            g = orc.Geom(); g.type, g.materialid = (1 if i & 1 else 0), i % 7
            for k in range(16):
                g.transform[k] = float(xf[k]); g.inverseTransform[k] = float(inv[k])
            geoms.append(g)
        cam = orc.Camera()
        C.memmove(C.byref(cam), C.byref(base.camera), C.sizeof(orc.Camera))
        for k in range(3):
            cam.position[k] = base.camera.position[k] * f
    elif kind == "all in one spot":
        rng = np.random.default_rng(7)
        xf, inv = np.zeros(16, np.float32), np.zeros(16, np.float32)
        geoms = list(base.geoms[:6])
        for i in range(400):
            c = [float(np.float32(v)) for v in (rng.uniform(-0.6, 0.6), rng.uniform(4.0, 5.2), rng.uniform(-0.6, 0.6))]
            sz = float(np.float32(rng.uniform(0.2, 0.6)))
            orc.lib().orc_build_transform(orc.vec3(*c), orc.vec3(0.0, 0.0, 0.0), orc.vec3(sz, sz, sz), orc.fptr(xf), orc.fptr(inv))
            g = orc.Geom(); g.type, g.materialid = (1 if i % 3 == 0 else 0), i % 7
            for k in range(16):
                g.transform[k] = float(xf[k]); g.inverseTransform[k] = float(inv[k])
            geoms.append(g)
    elif kind == "glass and mirrors only":
        geoms = [g for g in geoms]
        for g in geoms[6:]:
            g.materialid = 3 if g.materialid % 2 else 6             # (random256's materials: 3 mirror, 6 glass)
    sc = orc.Scene(geoms, base.materials, cam)
    assert sc.G > 256
    depth, iters = 6, 2
    want, live = orc.render(sc, oracle_config(depth), 1, iters)
    for kw in (dict(ordering=2), dict(ordering=2, streams=2, batch=1)):
        tr = make_tracer(sc, depth=depth, **kw)
        tr.set_image(None); tr.render(1, iters)
        st = tr.stats()
        assert tr.path_shape()["family"] == "k_path_w"
        assert [st.live[k] for k in range(depth + 1)] == [int(v) for v in live], (kind, kw)
        assert np.array_equal(tr.image(), want), (kind, kw)
        tr.close()


def test_tables_that_leave_no_room_for_k_path_w_fall_back_to_the_per_bounce_kernels(pt):
    """256 primitives with 1 400 materials: the tables alone take 105 KB of a CU's LDS, no block shape of k_path_w fits beside them,
    and pt_upload_scene says so BEFORE it sizes launch groups and pools -- the scene renders on the per-bounce kernels, bit-exact,
    instead of failing the upload (advisor, round 3)."""
    import copy
    base = orc.load_golden_scene("random256").with_resolution(160, 90)
    mats = [copy.copy(base.materials[i % base.M]) for i in range(1400)]
    geoms = list(base.geoms)
    for i, g in enumerate(geoms):
        g.materialid = g.materialid + base.M * (i % 150)          # the same material, 150 copies further on
    sc = orc.Scene(geoms, mats, base.camera)
    want, live = orc.render(sc, oracle_config(6), 1, 2)
    for kw in (dict(ordering=2), dict(ordering=2, streams=2)):
        tr = make_tracer(sc, depth=6, **kw)
        assert tr.path_shape()["family"] == "per-bounce"
        tr.set_image(None); tr.render(1, 2)
        st = tr.stats()
        assert [st.live[k] for k in range(7)] == [int(v) for v in live], kw
        assert np.array_equal(tr.image(), want), kw
        tr.close()
