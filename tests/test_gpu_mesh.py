"""-m gpu: GEOMTYPE MESH on the HIP path (OBJ triangles behind a per-mesh threaded BVH) against
the oracle's brute-force loop over the triangles.  Bit-exact like every other parity test: same nearest triangle (ties to
the earlier one), same hit point / normal / depth, hence same images, live counts and ray pools."""
import ctypes as C

import numpy as np
import pytest

import orc
from gpu_common import make_tracer, oracle_config

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mesh_scene():
    return orc.load_golden_scene("cornell_mesh").with_resolution(200, 150)


def test_mesh_primary_hits_match_brute_force(pt, mesh_scene):
    tr = make_tracer(mesh_scene)
    d, hit, t, P, N = tr.primary_hits()
    _, ohit = orc.raycast_flat(mesh_scene)
    assert np.array_equal(ohit.reshape(-1), hit)
    assert {6, 7, 8} <= set(np.unique(hit))
    L = orc.lib()
    cb = orc.CameraBasis()
    L.orc_camera_setup(C.byref(mesh_scene.camera), C.byref(cb))
    ga = mesh_scene.geom_array()
    o, dd, PP, NN, tt = (C.c_float * 3)(), (C.c_float * 3)(), (C.c_float * 3)(), (C.c_float * 3)(), C.c_float()
    on_mesh = np.flatnonzero((hit >= 6) & (hit <= 8))
    rng = np.random.default_rng(5)
    with orc.registered_meshes(mesh_scene):
        for idx in np.concatenate([rng.choice(on_mesh, 1500), rng.integers(0, 200 * 150, 500)]):
            x, y = int(idx % 200), int(idx // 200)
            L.orc_camera_ray(C.byref(cb), None, x, y, 0, 0, 0, 0, o, dd)
            h = L.orc_nearest_hit(ga, mesh_scene.G, None, o, dd, C.byref(tt), PP, NN)
            assert h == hit[idx] and tt.value == t[idx]
            if h >= 0:
                assert np.array_equal(np.array(list(PP), np.float32), P[idx]) and np.array_equal(np.array(list(NN), np.float32), N[idx])


@pytest.mark.parametrize("kw", [dict(), dict(culling=1), dict(batch=2, chunk_rays=100),
                                dict(streams=2), dict(ordering=1), dict(ordering=1, batch=2, streams=2), dict(ordering=2), dict(direct_light=1)])
def test_mesh_scene_matches_oracle(pt, mesh_scene, kw):
    """mirror torus, glass tetrahedron (rays start inside it), diffuse icosphere, next to a sphere and a rotated cube"""
    depth, iters = 6, 3
    tr = make_tracer(mesh_scene, depth=depth, **kw)
    tr.set_image(None)
    tr.render(1, iters)
    okw = {k: v for k, v in kw.items() if k == "direct_light"}
    want, live = orc.render(mesh_scene, oracle_config(depth, **okw), 1, iters)
    st = tr.stats()
    assert [st.live[k] for k in range(depth + 1)] == [int(v) for v in live]
    assert np.array_equal(tr.image(), want)
    if kw.get("streams", 1) == 1:
        n, arrs, pix = tr.trace_pool(2, 3)
        on, oarrs, opix = orc.trace_pool(mesh_scene, oracle_config(depth, **okw), 2, 3)
        if kw.get("ordering"):                 # the typed work queues keep the set of rays, not their order
            order = np.argsort(pix, kind="stable")
            pix, arrs = pix[order], [a[order] for a in arrs]
        assert n == on and np.array_equal(pix, opix) and all(np.array_equal(a, b) for a, b in zip(arrs, oarrs))
    tr.close()


def test_mesh_without_data_is_skipped_and_meshes_can_be_cleared(pt, mesh_scene):
    bare = orc.Scene(mesh_scene.geoms, mesh_scene.materials, mesh_scene.camera)
    tr = make_tracer(bare, depth=5)
    tr.set_image(None); tr.render(1, 2)
    want, _ = orc.render(bare, oracle_config(5), 1, 2)
    assert np.array_equal(tr.image(), want)
    tr.close()
    from gpu_common import to_product
    tr = make_tracer(mesh_scene, depth=5)
    tr.set_meshes([])                                      # cleared: the next upload sees plain MESH tags again
    tr.upload(*to_product(mesh_scene))
    tr.set_image(None); tr.render(1, 2)
    assert np.array_equal(tr.image(), want)
    with pytest.raises(pt.PtError, match="not a MESH"):
        tr.set_meshes([(0, mesh_scene.meshes[0][1], mesh_scene.meshes[0][2])])
        tr.upload(*to_product(mesh_scene))
    tr.close()


def test_larger_mesh_and_odd_scales(pt):
    """5 120-triangle icosphere (deeper BVH), scaled 0.05x and 40x with the whole scene: the BVH's margins are relative"""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    from make_scenes import icosphere
    v, f = icosphere(4)
    v, f = np.array(v, np.float32), np.array(f, np.int32)
    base = orc.load_golden_scene("cornell_mesh").with_resolution(160, 120)
    for factor in (1.0, 0.05, 40.0):
        geoms = []
        for g0 in base.geoms:
            g = orc.Geom()
            C.memmove(C.byref(g), C.byref(g0), C.sizeof(orc.Geom))
            for r in range(3):
                for c in range(4):
                    g.transform[4 * r + c] = g0.transform[4 * r + c] * factor
                    g.inverseTransform[4 * r + c] = g0.inverseTransform[4 * r + c] / factor if c < 3 else g0.inverseTransform[4 * r + c]
            geoms.append(g)
        cam = orc.Camera()
        C.memmove(C.byref(cam), C.byref(base.camera), C.sizeof(orc.Camera))
        for k in range(3):
            cam.position[k] = base.camera.position[k] * factor
        sc = orc.Scene(geoms, base.materials, cam, meshes=[(6, v, f)] + base.meshes[1:])
        tr = make_tracer(sc, depth=5)
        tr.set_image(None); tr.render(1, 2)
        want, live = orc.render(sc, oracle_config(5), 1, 2)
        st = tr.stats()
        assert [st.live[k] for k in range(6)] == [int(x) for x in live], factor
        assert np.array_equal(tr.image(), want), factor
        tr.close()


@pytest.mark.parametrize("workload", ["mesh", "mesh5k"])
def test_mesh_stages_at_full_size_equal_the_per_bounce_kernels(pt, workload):
    """k_path_q<MESH> at 1920x1080: every wave's mesh stack fills, so MESH turns run with a reserve of waiting rays and a WALK that
    has thinned out is INTERRUPTED -- its rays go back on the stack with their node cursor and best triangle and are resumed in a
    later turn (small frames never get there).  Image and live counts must equal those of the per-bounce kernels, whose per-lane
    mesh_test is checked against the oracle above; twice, because a race would show as a difference between runs."""
    import os, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    sf = pt.SceneFile(os.path.join(root, bench.WORKLOADS[workload][0]))
    geoms, mats, cam = sf.flatten(0)
    cam.resolution[0], cam.resolution[1] = 1920.0, 1080.0
    res = []
    for ordering in (0, 2, 2):
        tr = pt.PathTracer(pt.default_config(max_depth=8, ordering=ordering))
        tr.set_meshes(sf.meshes())
        tr.upload(geoms, mats, cam)
        tr.set_image(None)
        tr.render(1, 2)
        st = tr.stats()
        res.append(([int(st.live[k]) for k in range(9)], tr.image().copy()))
        tr.close()
    for live, img in res[1:]:
        assert live == res[0][0]
        assert np.array_equal(img, res[0][1])


def test_interrupted_mesh_traversals_match_the_oracle(pt):
    """The same mechanism against the oracle itself: one block per CU and a mid-sized frame give every wave a few hundred rays, enough
    for its mesh stack to hold the reserve that lets WALKs be interrupted (tools/meshstats.py counts them on a stats build)."""
    sc = orc.load_golden_scene("cornell_mesh").with_resolution(640, 360)
    tr = make_tracer(sc, depth=5, ordering=2, blocks_per_cu=1)
    tr.set_image(None)
    tr.render(1, 1)
    want, live = orc.render(sc, oracle_config(5), 1, 1)
    st = tr.stats()
    assert [st.live[k] for k in range(6)] == [int(v) for v in live]
    assert np.array_equal(tr.image(), want)
    tr.close()


def _degenerate_meshes():
    """meshes that stress the BVH builder's cuts and the tie rule, in the unit cube around the origin (object space)"""
    rng = np.random.default_rng(12)
    out = {}
    # 200 copies of ONE triangle (all centroids equal: every cut is a tie) in front of 50 copies of a second one: the earliest
    # triangle of the file must win every tie on t
    a = np.array([[-0.4, -0.4, 0.1], [0.4, -0.4, 0.1], [0.0, 0.45, 0.1]], np.float32)
    b = a + np.float32([0.0, 0.0, -0.3])
    out["copies"] = (np.concatenate([a, b]), np.array([[0, 1, 2]] * 200 + [[3, 4, 5]] * 50, np.int32))
    # a strip of thin slivers along x: lopsided surface-area cuts, deep tree
    n = 300
    xs = np.linspace(-0.5, 0.5, n + 1, dtype=np.float32) ** 3 * 4.0
    v = np.stack([np.stack([xs, np.full(n + 1, -0.3, np.float32), np.zeros(n + 1, np.float32)], 1),
                  np.stack([xs, np.full(n + 1, 0.3, np.float32), 0.2 * np.sin(np.arange(n + 1, dtype=np.float32))], 1)], 1).reshape(-1, 3)
    f = np.array([[2 * i, 2 * i + 2, 2 * i + 1] for i in range(n)] + [[2 * i + 1, 2 * i + 2, 2 * i + 3] for i in range(n)], np.int32)
    out["strip"] = (v.astype(np.float32), f)
    # one triangle; five triangles (one cut above the leaf size); a cloud of random small triangles, some of zero area
    out["single"] = (a.copy(), np.array([[0, 1, 2]], np.int32))
    v5 = rng.uniform(-0.5, 0.5, (15, 3)).astype(np.float32)
    out["five"] = (v5, np.arange(15, dtype=np.int32).reshape(5, 3))
    c = rng.uniform(-0.45, 0.45, (120, 1, 3)).astype(np.float32)
    vc = (c + rng.uniform(-0.08, 0.08, (120, 3, 3)).astype(np.float32)).reshape(-1, 3)
    fc = np.arange(360, dtype=np.int32).reshape(120, 3)
    fc[::17, 2] = fc[::17, 1]                                # zero-area triangles
    out["cloud"] = (vc, fc)
    return out


@pytest.mark.parametrize("ordering", [0, 2])
def test_degenerate_meshes_match_the_oracle(pt, ordering):
    """BVH cuts by the surface-area heuristic on meshes that give it nothing to choose (identical centroids), lopsided cuts
    (slivers), trees of one node, zero-area triangles -- and the tie rule of the (t, index) keys: of coincident triangles
    the earliest in the file wins.  Each mesh takes the place of the icosphere of the mesh scene, one at a time."""
    base = orc.load_golden_scene("cornell_mesh").with_resolution(160, 120)
    gi = base.meshes[0][0]
    for name, (v, f) in _degenerate_meshes().items():
        sc = orc.Scene(base.geoms, base.materials, base.camera, meshes=[(gi, v, f)] + base.meshes[1:])
        tr = make_tracer(sc, depth=5, ordering=ordering)
        tr.set_image(None)
        tr.render(1, 2)
        want, live = orc.render(sc, oracle_config(5), 1, 2)
        st = tr.stats()
        assert [st.live[k] for k in range(6)] == [int(x) for x in live], (name, ordering)
        assert np.array_equal(tr.image(), want), (name, ordering)
        tr.close()


def test_many_overlapping_meshes_match_the_oracle(pt):
    """Ten meshes (icospheres of 20..320 triangles and tori, glass / mirror / diffuse) that overlap and sit inside one another in the
    Cornell room: rays leave one mesh for the next inside the same turn (a finished traversal whose next candidate is another
    mesh goes straight back on the mesh stack), carry a best hit from a cube into a mesh test and out again.  Whole paths against
    the oracle, and the pool of ray records at two bounces."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    from make_scenes import icosphere, torus
    base = orc.load_golden_scene("cornell_mesh").with_resolution(200, 150)
    rng = np.random.default_rng(77)
    xf, inv = np.zeros(16, np.float32), np.zeros(16, np.float32)
    geoms = [g for g in base.geoms[:6]] + [base.geoms[9], base.geoms[10]]          # the room, the light, a sphere and a cube
    meshes = []
    shapes = [icosphere(0), icosphere(1), icosphere(2), icosphere(1), icosphere(2)]
    tv, tf = torus(12, 8)
    tri = []
    for q in tf:                                             # the generator's quads, fanned like the loader does
        tri += [(q[0], q[k], q[k + 1]) for k in range(1, len(q) - 1)]
    shapes += [(tv, tri)] * 2 + [icosphere(0), icosphere(1), icosphere(2)]
    for k, (v, f) in enumerate(shapes):
        centre = (rng.uniform(-2.5, 2.5), rng.uniform(1.0, 6.0), rng.uniform(-2.0, 3.0)) if k % 3 else (0.3 * k - 1.0, 3.0 + 0.1 * k, 0.5)    # every third one nested around one spot
        scale = [float(s) for s in rng.uniform(1.2, 3.5, 3)]
        orc.lib().orc_build_transform(orc.vec3(*centre), orc.vec3(*[float(a) for a in rng.uniform(0, 180, 3)]), orc.vec3(*scale), orc.fptr(xf), orc.fptr(inv))
        g = orc.Geom()
        g.type, g.materialid = 2, [2, 4, 5, 1, 3][k % 5]
        for i in range(16):
            g.transform[i] = float(xf[i]); g.inverseTransform[i] = float(inv[i])
        meshes.append((len(geoms), np.array(v, np.float32), np.array(f, np.int32)))
        geoms.append(g)
    sc = orc.Scene(geoms, base.materials, base.camera, meshes=meshes)
    for ordering in (2, 0):
        tr = make_tracer(sc, depth=7, ordering=ordering)
        tr.set_image(None)
        tr.render(1, 2)
        want, live = orc.render(sc, oracle_config(7), 1, 2)
        st = tr.stats()
        assert [st.live[k] for k in range(8)] == [int(x) for x in live], ordering
        assert np.array_equal(tr.image(), want), ordering
        n, arrs, pix = tr.trace_pool(2, 2)
        on, oarrs, opix = orc.trace_pool(sc, oracle_config(7), 2, 2)
        if ordering:
            order = np.argsort(pix, kind="stable")
            pix, arrs = pix[order], [a[order] for a in arrs]
        assert n == on and np.array_equal(pix, opix) and all(np.array_equal(a, b) for a, b in zip(arrs, oarrs)), ordering
        tr.close()
