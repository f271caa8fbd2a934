"""Helpers shared by the -m gpu parity tests: golden scene -> product PODs, oracle <-> GPU configs."""
import ctypes as C

import numpy as np

import orc
from conftest import load_package


def to_product(scene):
    """orc.Scene (reference-parser PODs) -> (Geom[], Material[], Camera) of the product ABI."""
    pkg = load_package()
    geoms = (pkg.Geom * scene.G)()
    for i, g in enumerate(scene.geoms):
        geoms[i].type, geoms[i].materialid = g.type, g.materialid
        for k in range(12):
            geoms[i].transform[k] = g.transform[k]
            geoms[i].inverseTransform[k] = g.inverseTransform[k]
    mats = (pkg.Material * scene.M)()
    for i, m in enumerate(scene.materials):
        C.memmove(C.byref(mats[i]), C.byref(m), 64)
    cam = pkg.Camera()
    C.memmove(C.byref(cam), C.byref(scene.camera), 52)
    return geoms, mats, cam


def make_tracer(scene, depth=8, **kw):
    """A context for the parity tests.  Unless a test says otherwise it is the STABLE kernel family on one stream (ordering = 0,
    streams = 1): what the parity hooks compare against -- pt_config_default itself gives the fast path (ordering = 2, two streams),
    which the tests ask for by name."""
    pkg = load_package()
    kw.setdefault("ordering", 0)
    kw.setdefault("streams", 1)
    cfg = pkg.default_config(max_depth=depth, **kw)
    tr = pkg.PathTracer(cfg)
    if getattr(scene, "meshes", None):
        tr.set_meshes(scene.meshes)
    tr.upload(*to_product(scene))
    return tr


def oracle_config(depth=8, **kw):
    return orc.default_config(depth, **kw)


def rel_l2(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.sqrt(np.mean((a - b) ** 2)))
