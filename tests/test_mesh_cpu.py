"""CPU: GEOMTYPE MESH -- the reference declares the type (src/sceneStructs.h:14), tags `*.obj` objects in its parser
(src/scene.cpp:55-64) and leaves the kernel branch empty (src/raytraceKernel.cu:144-145).  The build's definition is
DESIGN.md section 3.8; here the oracle's restatement of it is checked against geometry it must reproduce ("parity
unpinned" by the reference: there is nothing to compare with), and the product's OBJ loader against an independent
reader."""
import ctypes as C
import os

import numpy as np
import pytest

import orc
from conftest import ROOT

L = orc.lib()


def v3(*a):
    return (C.c_float * 3)(*[float(x) for x in a])


def test_obj_loader_matches_independent_reader(pt):
    sf = pt.SceneFile(os.path.join(ROOT, "scenes", "cornell_mesh.txt"))
    got = sf.meshes()
    want = orc.scene_meshes(os.path.join(ROOT, "scenes", "cornell_mesh.txt"))
    assert [g for g, _, _ in got] == [g for g, _, _ in want] == [6, 7, 8]
    for (_, v, i), (_, v2, i2) in zip(got, want):
        assert np.array_equal(v, v2) and np.array_equal(i, i2)
    assert got[0][2].shape == (320, 3)                  # icosphere, plain `f a b c`
    assert got[1][2].shape == (400, 3)                  # torus: 200 quads `f a/b/c ...`, fanned
    assert got[2][2].shape == (4, 3) and got[2][2].min() == 0 and got[2][2].max() == 3      # negative indices
    geoms, _, _ = sf.flatten(0)
    assert [g.type for g in geoms][6:9] == [2, 2, 2]


def test_missing_or_broken_obj(pt, tmp_path):
    text = open(os.path.join(ROOT, "scenes", "cornell_mesh.txt")).read()
    p = tmp_path / "scene.txt"
    p.write_text(text)                                   # no meshes/ directory beside it: like the reference, not an error
    sf = pt.SceneFile(str(p))
    assert sf.meshes() == [] and sf.ngeoms == 11
    (tmp_path / "meshes").mkdir()
    (tmp_path / "meshes" / "icosphere.obj").write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 7\n")
    with pytest.raises(pt.PtError, match="out of range"):
        pt.SceneFile(str(p))


def test_triangle_known_answers():
    v0, e1, e2 = v3(0, 0, 0), v3(1, 0, 0), v3(0, 1, 0)
    assert L.orc_triangle_test(v0, e1, e2, v3(.25, .25, 2), v3(0, 0, -1)) == 2.0
    assert L.orc_triangle_test(v0, e1, e2, v3(.25, .25, -3), v3(0, 0, 1)) == 3.0          # two-sided
    assert L.orc_triangle_test(v0, e1, e2, v3(.75, .75, 2), v3(0, 0, -1)) == -1.0         # u + v > 1
    assert L.orc_triangle_test(v0, e1, e2, v3(-.1, .2, 2), v3(0, 0, -1)) == -1.0
    assert L.orc_triangle_test(v0, e1, e2, v3(.25, .25, 2), v3(0, 0, 1)) == -1.0          # behind the origin
    assert L.orc_triangle_test(v0, e1, e2, v3(.25, .25, 2), v3(1, 0, 0)) == -1.0          # parallel: det = 0
    assert L.orc_triangle_test(v0, e1, e2, v3(0, 0, 1), v3(0, 0, -1)) == 1.0              # a corner counts (u = v = 0)


def test_icosphere_mesh_approximates_the_sphere_test():
    """A unit-diameter icosphere under the same transform as a sphere: same hit/miss away from the silhouette, depth
    within the faceting error, normals within a facet's tilt."""
    sc = orc.load_golden_scene("cornell_mesh")
    _, v, idx = sc.meshes[0]
    g = orc.Geom()
    C.memmove(C.byref(g), C.byref(sc.geoms[9]), C.sizeof(orc.Geom))       # the scene's sphere: uniform scale 2
    g.type = 2
    rng = np.random.default_rng(3)
    P, N, Ps, Ns = v3(0, 0, 0), v3(0, 0, 0), v3(0, 0, 0), v3(0, 0, 0)
    tri = C.c_int()
    vv, ii = np.ascontiguousarray(v, np.float32), np.ascontiguousarray(idx, np.int32)
    centre = np.array([g.transform[3], g.transform[7], g.transform[11]])
    hits = 0
    for _ in range(400):
        o = centre + rng.normal(size=3) * 4
        target = centre + rng.normal(size=3) * 0.6
        d = (target - o) / np.linalg.norm(target - o)
        tm = L.orc_mesh_test(C.byref(g), vv.ctypes.data_as(C.POINTER(C.c_float)), ii.ctypes.data_as(C.POINTER(C.c_int)), len(ii),
                             v3(*o), v3(*d), P, N, C.byref(tri))
        ts = L.orc_sphere_test(C.byref(g), v3(*o), v3(*d), Ps, Ns)
        miss_by = np.linalg.norm(np.cross(centre - o, d))                 # distance of the line from the centre
        if miss_by < 0.9:                                                 # radius 1, inscribed radius of the level-2 icosphere > 0.95
            assert tm > 0 and ts > 0
            assert abs(tm - ts) < 0.08 and 0 <= tri.value < 320
            assert np.dot(list(N), list(Ns)) > 0.9 and abs(np.linalg.norm(list(N)) - 1) < 1e-6
            hits += 1
        elif miss_by > 1.01:
            assert tm == -1.0 and ts == -1.0
    assert hits > 150


def test_mesh_without_registered_data_is_the_reference_empty_branch():
    sc = orc.load_golden_scene("cornell_mesh").with_resolution(64, 48)
    bare = orc.Scene(sc.geoms, sc.materials, sc.camera)                   # same PODs, no mesh data
    without = orc.Scene([g for g in sc.geoms if g.type != 2], sc.materials, sc.camera)
    a, la = orc.render(bare, orc.default_config(5), 1, 2)
    b, lb = orc.render(without, orc.default_config(5), 1, 2)
    assert np.array_equal(a, b) and np.array_equal(la, lb)
    c, lc = orc.render(sc, orc.default_config(5), 1, 2)
    assert not np.array_equal(a, c)                                       # with the data the meshes are there
    _, hit = orc.raycast_flat(sc)
    assert set(np.unique(hit)) >= {6, 7, 8}                               # every mesh is seen by some primary ray


def test_white_furnace_energy_is_conserved_with_meshes():
    """Closed white room, every surface (mesh included) albedo 1, no light: a path never terminates but by depth; make the
    mesh the only emitter instead and the mean radiance is bounded by its emittance."""
    sc = orc.load_golden_scene("cornell_mesh").with_resolution(48, 36)
    img, live = orc.render(sc, orc.default_config(6), 1, 4)
    assert np.isfinite(img).all() and img.min() >= 0 and img.max() > 0
    assert all(int(live[k]) >= int(live[k + 1]) for k in range(6))
