"""The spatial index of the whole-path kernel for 33..256 primitives (k_path_w), checked WITHOUT a device.

`pt_debug_grid_probe` builds the uniform grid exactly as `pt_upload_scene` does and walks rays on the host with the
kernel's own walk functions and reference flags (csrc/pt_kernels.hpp: grid_walk_begin / grid_walk_step).  What the
kernel needs from the index, and what is asserted here against a double-precision slab test of every primitive's
geometric bounding box:

  * complete: a primitive whose box the ray meets (t >= 0) is listed for the ray -- by a cell of the walk or by the
    list of big primitives (the exact test can only hit inside that box, so nothing the reference loop
    /root/reference/src/raytraceKernel.cu:134-153 would hit is lost);
  * once: no primitive is listed twice for a ray (a duplicate would waste a slot of the 8-entry candidate list);
  * bounded: a walk ends within n_x + n_y + n_z cells.

Ray families: random interior rays, camera rays, axis-parallel rays (zero direction components, origins ON cell
boundaries), rays aimed at the corners and edges of primitive boxes (grazing), rays from outside and far away, the
non-finite ones; at the scene's own scale and at 0.05 x, 37 x and 1000 x (as tests/test_gpu_parity.py renders them)."""
import numpy as np
import pytest

import orc
from conftest import load_package
from gpu_common import to_product


def _boxes(geoms):
    """geometric world AABB of every analytic primitive: cube = its 8 corners, sphere (ellipsoid) = centre +- 0.5 |row|"""
    lo, hi, ids = [], [], []
    for i, g in enumerate(geoms):
        if g.type not in (0, 1):
            continue
        m = np.array([g.transform[k] for k in range(12)], dtype=np.float64).reshape(3, 4)
        c = m[:, 3]
        if g.type == 1:
            ext = 0.5 * np.abs(m[:, :3]).sum(1)
        else:
            ext = 0.5 * np.sqrt((m[:, :3] ** 2).sum(1))
        lo.append(c - ext); hi.append(c + ext); ids.append(i)
    return np.array(lo), np.array(hi), np.array(ids)


def _meets(rays, lo, hi):
    """[n, P] bool: the ray (double precision) meets the box at some t >= 0; rays with a zero component handled by limits"""
    o = rays[:, None, :3].astype(np.float64); d = rays[:, None, 3:].astype(np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        t0 = (lo[None] - o) / d; t1 = (hi[None] - o) / d
    tn = np.minimum(t0, t1); tf = np.maximum(t0, t1)
    par = d == 0.0                                         # parallel to the slab: inside it or never
    inside = (o >= lo[None]) & (o <= hi[None])
    tn = np.where(par, np.where(inside, -np.inf, np.inf), tn)
    tf = np.where(par, np.where(inside, np.inf, -np.inf), tf)
    enter = tn.max(2); leave = tf.min(2)
    return (enter <= leave) & (leave >= 0.0)


def _scaled(name, factor):
    import json, os
    if factor == 1.0:
        return orc.load_golden_scene(name)
    gold = json.load(open(os.path.join(orc.GOLD, "ref_scene_%s.json" % name)))
    base = orc.load_golden_scene(name)
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
    return orc.Scene(geoms, base.materials, base.camera)


def _ray_families(lo, hi, rng, n):
    blo, bhi = lo.min(0), hi.max(0)
    size = (bhi - blo).max()
    fam = {}
    o = rng.uniform(blo, bhi, (n, 3)); d = rng.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    fam["interior"] = np.concatenate([o, d], 1)
    # from outside, towards a random point of the scene
    o = blo + (bhi - blo) * rng.uniform(-1.5, 2.5, (n, 3)); tgt = rng.uniform(blo, bhi, (n, 3)); d = tgt - o; d /= np.linalg.norm(d, axis=1, keepdims=True)
    fam["outside"] = np.concatenate([o, d], 1)
    # axis-parallel and planar directions (exact zeros), origins snapped to a coarse lattice so that many lie on cell boundaries
    o = rng.uniform(blo, bhi, (n, 3)); o = np.round((o - blo) / size * 32) / 32 * size + blo
    d = rng.normal(size=(n, 3)); z = rng.integers(0, 7, n)
    for k in range(3):
        d[(z >> k) & 1 == 1, k] = 0.0
    d[(d == 0).all(1)] = [1.0, 0.0, 0.0]
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    fam["axis"] = np.concatenate([o, d], 1)
    # grazing: aimed at box corners / points on box edges, from random origins, with perturbations of a few ulp
    pick = rng.integers(0, len(lo), n); cs = rng.integers(0, 2, (n, 3)).astype(bool)
    corner = np.where(cs, hi[pick], lo[pick])
    edge = rng.integers(0, 3, n); w = rng.uniform(0, 1, n)
    for k in range(3):
        sel = edge == k
        corner[sel, k] = lo[pick[sel], k] + w[sel] * (hi[pick[sel], k] - lo[pick[sel], k])
    o = rng.uniform(blo, bhi, (n, 3)); d = corner - o
    d *= 1.0 + rng.uniform(-3e-7, 3e-7, (n, 3))
    nrm = np.linalg.norm(d, axis=1, keepdims=True); keep = nrm[:, 0] > 1e-6 * size
    fam["grazing"] = np.concatenate([o, d / np.where(nrm > 0, nrm, 1)], 1)[keep]
    # far away (beyond the reach of the walk: the kernel takes the reference loop) and non-finite
    o = blo + (bhi - blo) * rng.uniform(-40, 40, (n // 4, 3)); tgt = rng.uniform(blo, bhi, (n // 4, 3)); d = tgt - o; d /= np.linalg.norm(d, axis=1, keepdims=True)
    far = np.concatenate([o, d], 1)
    bad = fam["interior"][:8].copy(); bad[0, 0] = np.nan; bad[1, 4] = np.nan; bad[2, 3] = np.inf; bad[3, 1] = -np.inf; bad[4, 3:] = 0.0
    fam["far_and_bad"] = np.concatenate([far, bad], 0)
    return {k: v.astype(np.float32) for k, v in fam.items()}


@pytest.mark.parametrize("density", [0, 1, 16])
@pytest.mark.parametrize("factor", [1.0, 0.05, 37.0, 1000.0])
def test_grid_walk_lists_every_primitive_a_ray_meets_exactly_once(pt, factor, density):
    sc = _scaled("random256", factor)
    geoms, _, _ = to_product(sc)
    lo, hi, ids = _boxes(sc.geoms)
    rng = np.random.default_rng(565 + density)
    total = 0
    for name, rays in _ray_families(lo, hi, rng, 6000).items():
        sets, info = pt.grid_probe(geoms, rays, density)
        assert info["duplicates"] == 0, (name, info)
        assert info["longest_walk"] <= info["nx"] + info["ny"] + info["nz"], (name, info)
        finite = np.isfinite(rays).all(1)
        need = _meets(rays[finite].astype(np.float64), lo, hi)
        have = sets[finite][:, ids]
        missing = need & ~have
        assert not missing.any(), (name, factor, density, np.argwhere(missing)[:5], info)
        total += int(need.sum())
        if name == "far_and_bad":
            assert info["unwalked"] >= 5                  # the non-finite rays and the zero direction... and whatever lies beyond the reach
    assert total > 20000


def test_grid_of_the_bench_scene_is_small_and_cheap(pt):
    """configs[3]: the figures DESIGN.md quotes -- the walls are the big primitives, the grid fits beside the tables in LDS,
    a ray tests about a third of the bounds the two-level clusters made it test (28.6)."""
    sc = orc.load_golden_scene("random256")
    geoms, _, _ = to_product(sc)
    lo, hi, ids = _boxes(sc.geoms)
    rng = np.random.default_rng(1)
    rays = _ray_families(lo, hi, rng, 20000)["interior"]
    sets, info = pt.grid_probe(geoms, rays, 0)
    assert info["big"] == 5 and info["unwalked"] == 0
    assert info["cells"] <= 2048 and info["lds_bytes"] <= 10 * 1024
    assert sets.sum(1).mean() < 12.0
    assert info["mean_walk"] <= 9
    # the walk-length estimate the survivors are sorted by (span of the ray inside the grid's box in cell units): within two cells
    assert info["length_estimate_worst"] <= 2 and info["length_estimate_mean_error_x100"] < 80
    assert 2 <= info["bin1"] < info["bin2"]


def test_grid_when_everything_is_big_or_flat(pt):
    """Degenerate scenes still give a valid index: co-located primitives (one crowded cell), a flat sheet of primitives
    (a one-cell-thick grid), and a scene whose primitives all span it (all of them big: an empty grid)."""
    base = orc.load_golden_scene("random256")
    rng = np.random.default_rng(7)
    xf, inv = np.zeros(16, np.float32), np.zeros(16, np.float32)

    def make(transforms):
        geoms = []
        for i, (t, r, s) in enumerate(transforms):
            orc.lib().orc_build_transform(orc.vec3(*t), orc.vec3(*r), orc.vec3(*s), orc.fptr(xf), orc.fptr(inv))
            g = orc.Geom()
            g.type, g.materialid = i & 1, 0
            for k in range(16):
                g.transform[k] = float(xf[k]); g.inverseTransform[k] = float(inv[k])
            geoms.append(g)
        return orc.Scene(geoms, base.materials, base.camera)

    scenes = {
        "colocated": make([((1.0, 2.0, 3.0), (10.0 * i, 0, 0), (0.5, 0.5, 0.5)) for i in range(40)]),
        "sheet": make([((float(i % 10), 0.0, float(i // 10)), (0, 0, 0), (0.6, 0.6, 0.6)) for i in range(100)]),
        "all_big": make([((0.1 * i, 0.0, 0.0), (0, 0, 7.0 * i), (8.0, 8.0, 8.0)) for i in range(40)]),
    }
    for name, sc in scenes.items():
        geoms, _, _ = to_product(sc)
        lo, hi, ids = _boxes(sc.geoms)
        for fam, rays in _ray_families(lo, hi, rng, 2000).items():
            sets, info = pt.grid_probe(geoms, rays, 0)
            assert info["duplicates"] == 0, (name, fam, info)
            finite = np.isfinite(rays).all(1)
            need = _meets(rays[finite].astype(np.float64), lo, hi)
            assert not (need & ~sets[finite][:, ids]).any(), (name, fam, info)


@pytest.mark.parametrize("extra,density", [(251, 0), (594, 0), (1494, 0), (1494, 2), (5000, 0)])
def test_wide_grid_lists_every_primitive_a_ray_meets_exactly_once(pt, extra, density):
    """More than 256 primitives: the same grid with 32-bit references and ids (k_path_w<BIG>), as pt_upload_scene builds it.
    Complete, listed once, bounded -- on the scenes tests/test_gpu_many_primitives.py renders and on a 5 006-primitive one."""
    sc = orc.many_primitives_scene(extra, size=(0.12, 0.5) if extra < 3000 else (0.05, 0.25))
    geoms, _, _ = to_product(sc)
    lo, hi, ids = _boxes(sc.geoms)
    rng = np.random.default_rng(extra + density)
    total = 0
    for name, rays in _ray_families(lo, hi, rng, 1500).items():
        sets, info = pt.grid_probe(geoms, rays, density)
        assert info["duplicates"] == 0, (name, info)
        assert info["longest_walk"] <= info["nx"] + info["ny"] + info["nz"], (name, info)
        assert info["big"] <= 16 and info["refs"] < (1 << 18)
        finite = np.isfinite(rays).all(1)
        need = _meets(rays[finite].astype(np.float64), lo, hi)
        have = sets[finite][:, ids]
        missing = need & ~have
        assert not missing.any(), (name, extra, density, np.argwhere(missing)[:5], info)
        total += int(need.sum())
    assert total > 5000


@pytest.mark.parametrize("scene,kw", [("random256", dict()), ("random256", dict(antialias=1)), ("random256", dict(camera_mode=1)), ("many1494", dict()),
                                      ("random256_x1000", dict()), ("random256_x0.05", dict(antialias=1))])
def test_camera_fan_cone_covers_every_primitive_its_rays_meet(pt, scene, kw):
    """k_path_w's camera groups: the cone of a group of 64 camera rays (the kernel's own fan_* functions, run on the host by
    pt_debug_fan_probe) meets every primitive whose geometry any of its rays can meet -- reference camera rays incl. the
    normalize(R) quirk, jittered ones, the corrected pinhole, scaled scenes; 1920 x 1080 (a group spans 2.6 degrees), a sample of
    the frame's 32 400 groups."""
    if scene == "many1494":
        sc = orc.many_primitives_scene(1494, w=1920, h=1080)
    elif scene.startswith("random256_x"):
        sc = _scaled("random256", float(scene.split("x")[1])).with_resolution(1920, 1080)
    else:
        sc = orc.load_golden_scene("random256")
    geoms, _, _ = to_product(sc)
    lo, hi, ids = _boxes(sc.geoms)
    n, arrs, pix = orc.trace_pool(sc, orc.default_config(4, **kw), 3, 0)                  # the pool before any bounce: the camera rays
    assert n == sc.W * sc.H and np.array_equal(pix & 0xFFFFFF, np.arange(n))
    pick = np.random.default_rng(len(scene)).choice(n // 64, 240, replace=False)
    rays = np.stack(arrs[:6], 1).reshape(-1, 64, 6)[pick]
    sets, cones = pt.fan_probe(geoms, rays)
    assert cones == len(rays)                                                             # 1920 = 30 x 64: no group wraps; every one gets its cone
    need = _meets(rays.reshape(-1, 6).astype(np.float64), lo, hi)
    # a sphere is met only where the ray comes within its (largest) radius of the centre -- its box has corners the sphere lacks
    flat = rays.reshape(-1, 6).astype(np.float64)
    sph = [col for col, i in enumerate(ids) if sc.geoms[i].type == 0]
    mats = np.array([[sc.geoms[ids[col]].transform[k] for k in range(12)] for col in sph], dtype=np.float64).reshape(-1, 3, 4)
    ctr, rad = mats[:, :, 3], 0.5 * np.linalg.norm(mats[:, :, :3], ord=2, axis=(1, 2))
    for lo_, hi_ in [(k, min(k + 4096, len(flat))) for k in range(0, len(flat), 4096)]:
        o_, d_ = flat[lo_:hi_, None, :3], flat[lo_:hi_, None, 3:]
        oc = ctr[None] - o_
        t = np.maximum((oc * d_).sum(2) / (d_ ** 2).sum(2), 0.0)
        need[lo_:hi_, sph] &= np.linalg.norm(oc - t[..., None] * d_, axis=2) <= rad[None] * (1 + 1e-9)
    need = need.reshape(len(rays), 64, -1).any(1)
    missing = need & ~sets[:, ids]
    assert not missing.any(), (scene, kw, np.argwhere(missing)[:5])
    assert sets[:, ids].sum(1).mean() < 40                                                # and it is worth having: a few dozen bounds per group, walls included


def test_camera_fan_refuses_what_it_cannot_bound(pt):
    """no common origin (thin lens), a fan too wide to be worth a cone, non-finite directions: no cone, every bit set (the kernel walks)"""
    sc = orc.load_golden_scene("random256").with_resolution(128, 64)
    geoms, _, _ = to_product(sc)
    n, arrs, pix = orc.trace_pool(sc, orc.default_config(4, camera_mode=1, aperture=0.3, focal_distance=10.0), 1, 0)
    lens = np.stack(arrs[:6], 1).reshape(-1, 64, 6)[:4]
    n, arrs, pix = orc.trace_pool(sc, orc.default_config(4), 1, 0)
    ok = np.stack(arrs[:6], 1).reshape(-1, 64, 6)
    wrap = np.concatenate([ok[1][32:], ok[2][:32]])[None]                                  # (128 pixels = two groups per row) half a row's end, half the next row's start
    bad = ok[:1].copy(); bad[0, 5, 3] = np.nan
    for fans in (lens, wrap, bad):
        sets, cones = pt.fan_probe(geoms, fans)
        assert cones == 0 and sets[:, :len(geoms)].all()
