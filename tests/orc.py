"""ctypes binding of the CPU oracle (oracle/libpt_oracle.so) + fixture loaders.

TEST INFRASTRUCTURE: imported only by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg (oracle/pt_oracle.h explains the rule)."""
import ctypes as C
import json
import os
import struct
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
ORACLE_DIR = os.path.join(ROOT, "oracle")


class Material(C.Structure):
    _fields_ = [("color", C.c_float * 3), ("specularExponent", C.c_float), ("specularColor", C.c_float * 3),
                ("hasReflective", C.c_float), ("hasRefractive", C.c_float), ("indexOfRefraction", C.c_float),
                ("hasScatter", C.c_float), ("absorptionCoefficient", C.c_float * 3),
                ("reducedScatterCoefficient", C.c_float), ("emittance", C.c_float)]


class Geom(C.Structure):
    _fields_ = [("type", C.c_int), ("materialid", C.c_int), ("transform", C.c_float * 16),
                ("inverseTransform", C.c_float * 16)]


class Camera(C.Structure):
    _fields_ = [("resolution", C.c_float * 2), ("position", C.c_float * 3), ("view", C.c_float * 3),
                ("up", C.c_float * 3), ("fov", C.c_float * 2)]


class Config(C.Structure):
    _fields_ = [("max_depth", C.c_int), ("camera_mode", C.c_int), ("antialias", C.c_int),
                ("aperture", C.c_float), ("focal_distance", C.c_float), ("row_offset", C.c_int),
                ("row_stride", C.c_int), ("direct_light", C.c_int)]


class CameraBasis(C.Structure):
    _fields_ = [("E", C.c_float * 3), ("M", C.c_float * 3), ("H", C.c_float * 3), ("V", C.c_float * 3),
                ("Cn", C.c_float * 3), ("Ah", C.c_float * 3), ("Bh", C.c_float * 3),
                ("inv_wm1", C.c_float), ("inv_hm1", C.c_float), ("W", C.c_float), ("Hres", C.c_float)]


_lib = None


def build_oracle():
    subprocess.run(["make", "-s", "-C", ORACLE_DIR, "libpt_oracle.so"], check=True)


def lib():
    global _lib
    if _lib is not None:
        return _lib
    path = os.path.join(ORACLE_DIR, "libpt_oracle.so")
    if not os.path.exists(path):
        build_oracle()
    L = C.CDLL(path)
    f3 = C.POINTER(C.c_float)
    L.orc_hash.restype = C.c_uint32; L.orc_hash.argtypes = [C.c_uint32]
    L.orc_lcg_seed.restype = C.c_uint32; L.orc_lcg_seed.argtypes = [C.c_uint32]
    L.orc_lcg_next.restype = C.c_uint32; L.orc_lcg_next.argtypes = [C.c_uint32]
    L.orc_u01.restype = C.c_float; L.orc_u01.argtypes = [C.c_uint32]
    L.orc_stream_seed.restype = C.c_uint32; L.orc_stream_seed.argtypes = [C.c_uint32] * 3
    L.orc_rng_from_thread.restype = None
    L.orc_rng_from_thread.argtypes = [C.c_float, C.c_float, C.c_float, C.c_int, C.c_int, f3]
    L.orc_sincos.restype = None; L.orc_sincos.argtypes = [C.c_float, f3, f3]
    L.orc_camera_setup.restype = None; L.orc_camera_setup.argtypes = [C.POINTER(Camera), C.POINTER(CameraBasis)]
    L.orc_camera_ray.restype = None
    L.orc_camera_ray.argtypes = [C.POINTER(CameraBasis), C.POINTER(Config), C.c_int, C.c_int,
                                 C.c_float, C.c_float, C.c_float, C.c_float, f3, f3]
    L.orc_sphere_test.restype = C.c_float; L.orc_sphere_test.argtypes = [C.POINTER(Geom), f3, f3, f3, f3]
    L.orc_sphere_test_intminmax.restype = C.c_float; L.orc_sphere_test_intminmax.argtypes = [C.POINTER(Geom), f3, f3, f3, f3]
    L.orc_sphere_test_powdouble.restype = C.c_float; L.orc_sphere_test_powdouble.argtypes = [C.POINTER(Geom), f3, f3, f3, f3]
    L.orc_point_on_ray.restype = None; L.orc_point_on_ray.argtypes = [f3, f3, C.c_float, f3]
    L.orc_multiply_mv.restype = None; L.orc_multiply_mv.argtypes = [f3, f3, f3]
    L.orc_box_test.restype = C.c_float; L.orc_box_test.argtypes = [C.POINTER(Geom), C.c_int, f3, f3, f3, f3]
    L.orc_nearest_hit.restype = C.c_int
    L.orc_nearest_hit.argtypes = [C.POINTER(Geom), C.c_int, C.POINTER(Material), f3, f3, f3, f3, f3]
    L.orc_get_radiuses.restype = None; L.orc_get_radiuses.argtypes = [C.POINTER(Geom), f3]
    L.orc_random_point_on_cube.restype = None; L.orc_random_point_on_cube.argtypes = [C.POINTER(Geom), C.c_float, f3]
    L.orc_random_point_on_sphere.restype = None; L.orc_random_point_on_sphere.argtypes = [C.POINTER(Geom), C.c_float, f3]
    L.orc_sample_light.restype = C.c_int; L.orc_sample_light.argtypes = [C.POINTER(Geom), C.c_float, f3, f3]
    L.orc_hemisphere.restype = None; L.orc_hemisphere.argtypes = [f3, C.c_float, C.c_float, f3]
    L.orc_reflection_direction.restype = None; L.orc_reflection_direction.argtypes = [f3, f3, f3]
    L.orc_transmission_direction.restype = C.c_int
    L.orc_transmission_direction.argtypes = [f3, f3, C.c_float, C.c_float, f3]
    L.orc_fresnel.restype = None; L.orc_fresnel.argtypes = [f3, f3, C.c_float, C.c_float, f3, f3]
    L.orc_scatter.restype = C.c_int
    L.orc_scatter.argtypes = [C.POINTER(Material), f3, f3, C.c_float, C.c_float, C.c_float, f3, f3, f3, f3]
    L.orc_build_transform.restype = C.c_int; L.orc_build_transform.argtypes = [f3, f3, f3, f3, f3]
    L.orc_display_pixel.restype = None; L.orc_display_pixel.argtypes = [f3, C.POINTER(C.c_uint8)]
    L.orc_image_to_u8.restype = None
    L.orc_image_to_u8.argtypes = [f3, C.c_int, C.c_float, C.c_float, C.POINTER(C.c_uint8)]
    L.orc_raycast_flat.restype = C.c_int
    L.orc_raycast_flat.argtypes = [C.POINTER(Geom), C.c_int, C.POINTER(Material), C.c_int, C.POINTER(Camera),
                                   f3, C.POINTER(C.c_int), C.c_int]
    L.orc_render.restype = C.c_int
    L.orc_render.argtypes = [C.POINTER(Geom), C.c_int, C.POINTER(Material), C.c_int, C.POINTER(Camera),
                             C.POINTER(Config), C.c_int, C.c_int, f3, C.POINTER(C.c_uint64), C.c_int]
    L.orc_trace_pool.restype = C.c_int
    L.orc_trace_pool.argtypes = [C.POINTER(Geom), C.c_int, C.POINTER(Material), C.c_int, C.POINTER(Camera),
                                 C.POINTER(Config), C.c_int, C.c_int] + [f3] * 9 + [C.POINTER(C.c_uint32)]
    L.orc_max_threads.restype = C.c_int
    pf, pi = C.POINTER(C.c_float), C.POINTER(C.c_int)
    L.orc_set_meshes.restype = C.c_int
    L.orc_set_meshes.argtypes = [pi, C.POINTER(pf), pi, C.POINTER(pi), pi, C.c_int]
    L.orc_triangle_test.restype = C.c_float; L.orc_triangle_test.argtypes = [f3, f3, f3, f3, f3]
    L.orc_mesh_test.restype = C.c_float
    L.orc_mesh_test.argtypes = [C.POINTER(Geom), pf, pi, C.c_int, f3, f3, f3, f3, pi]
    _lib = L
    return L


def f32_from_bits(u):
    return struct.unpack("<f", struct.pack("<I", u & 0xFFFFFFFF))[0]


def bits_from_f32(f):
    return struct.unpack("<I", struct.pack("<f", f))[0]


def fptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def vec3(*v):
    return (C.c_float * 3)(*v)


class Scene:
    """Flattened scene for one frame: what cudaRaytraceCore builds at raytraceKernel.cu:179-206."""

    def __init__(self, geoms, materials, camera, iterations=1, image_name="", meshes=None):
        self.geoms, self.materials, self.camera = geoms, materials, camera
        self.iterations, self.image_name = iterations, image_name
        self.meshes = list(meshes or [])      # [(geom_index, vertices float32 [n,3], indices int32 [t,3])]

    @property
    def G(self):
        return len(self.geoms)

    @property
    def M(self):
        return len(self.materials)

    @property
    def W(self):
        return int(self.camera.resolution[0])

    @property
    def H(self):
        return int(self.camera.resolution[1])

    def geom_array(self):
        return (Geom * self.G)(*self.geoms)

    def material_array(self):
        return (Material * self.M)(*self.materials)

    @classmethod
    def from_product(cls, geoms, mats, cam, meshes=None):
        """The product ABI's PODs (pt_geom: rows x,y,z only) -> an oracle scene (full 4x4 rows, w = (0,0,0,1))."""
        og = []
        for g in geoms:
            o = Geom()
            o.type, o.materialid = g.type, g.materialid
            for k in range(12):
                o.transform[k] = g.transform[k]
                o.inverseTransform[k] = g.inverseTransform[k]
            o.transform[15] = o.inverseTransform[15] = 1.0
            og.append(o)
        om = []
        for m in mats:
            x = Material()
            C.memmove(C.byref(x), C.byref(m), 64)
            om.append(x)
        oc = Camera()
        C.memmove(C.byref(oc), C.byref(cam), 52)
        return cls(og, om, oc, meshes=meshes)

    def with_resolution(self, W, H):
        """Same scene at another resolution, fov recomputed the way scene.cpp:201-205 does."""
        import math
        cam = Camera()
        C.memmove(C.byref(cam), C.byref(self.camera), C.sizeof(Camera))
        cam.resolution[0], cam.resolution[1] = float(W), float(H)
        fovy = np.float32(self.camera.fov[1])
        pi = np.float32(3.1415926535897932384626422832795028841971)
        yscaled = np.float32(math.tan(float(np.float32(fovy * np.float32(pi / np.float32(180))))))
        xscaled = np.float32(np.float32(yscaled * np.float32(W)) / np.float32(H))
        fovx = np.float32(np.float32(np.float32(math.atan(float(xscaled))) * np.float32(180)) / pi)
        cam.fov[0], cam.fov[1] = float(fovx), float(fovy)
        return Scene(self.geoms, self.materials, cam, self.iterations, self.image_name, self.meshes)


class registered_meshes:
    """with registered_meshes(scene): ...  -- the oracle's MESH registry holds the scene's meshes inside the block"""

    def __init__(self, scene):
        self.scene = scene

    def __enter__(self):
        ms = self.scene.meshes
        n = len(ms)
        self.keep = [(np.ascontiguousarray(v, np.float32), np.ascontiguousarray(i, np.int32)) for _, v, i in ms]
        pf, pi = C.POINTER(C.c_float), C.POINTER(C.c_int)
        gi = (C.c_int * max(1, n))(*[int(m[0]) for m in ms])
        vp = (pf * max(1, n))(*[v.ctypes.data_as(pf) for v, _ in self.keep])
        nv = (C.c_int * max(1, n))(*[len(v) for v, _ in self.keep])
        ip = (pi * max(1, n))(*[i.ctypes.data_as(pi) for _, i in self.keep])
        nt = (C.c_int * max(1, n))(*[len(i) for _, i in self.keep])
        self.args = (gi, vp, nv, ip, nt)
        assert lib().orc_set_meshes(gi, vp, nv, ip, nt, n) == 0
        return self

    def __exit__(self, *exc):
        lib().orc_set_meshes(None, None, None, None, None, 0)
        return False


def read_obj(path):
    """Independent reader of the OBJ subset the product's loader accepts: (vertices float32 [n,3], triangles int32 [t,3])"""
    v, f = [], []
    for line in open(path):
        t = line.split()
        if not t:
            continue
        if t[0] == "v":
            v.append([float(x) for x in t[1:4]])
        elif t[0] == "f":
            poly = []
            for w in t[1:]:
                raw = int(w.split("/")[0])
                poly.append(raw - 1 if raw > 0 else len(v) + raw)
            for k in range(1, len(poly) - 1):
                f.append([poly[0], poly[k], poly[k + 1]])
    return np.array(v, np.float32), np.array(f, np.int32)


def scene_meshes(scene_txt):
    """[(geom_index, vertices, indices)] for the `*.obj` objects of a scene file (paths relative to the file)"""
    out, lines = [], open(scene_txt).read().splitlines()
    for k, line in enumerate(lines):
        if line.startswith("OBJECT ") and k + 1 < len(lines) and lines[k + 1].endswith(".obj"):
            v, f = read_obj(os.path.join(os.path.dirname(scene_txt), lines[k + 1]))
            out.append((int(line.split()[1]), v, f))
    return out


def load_golden_scene(name, frame=0):
    """Scene PODs from a tests/golden/ref_scene_<name>.json dump (produced by the REFERENCE's
    parser, oracle/gen_golden.py)."""
    d = json.load(open(os.path.join(GOLD, "ref_scene_%s.json" % name)))
    mats = []
    for m in d["materials"]:
        mm = Material()
        C.memmove(C.byref(mm), struct.pack("<16I", *m), 64)
        mats.append(mm)
    geoms = []
    for o in d["objects"]:
        g = Geom()
        g.type, g.materialid = o["type"], o["materialid"]
        fr = o["frames"][frame]
        for k in range(16):
            g.transform[k] = f32_from_bits(fr["transform"][k])
            g.inverseTransform[k] = f32_from_bits(fr["inverseTransform"][k])
        geoms.append(g)
    c = d["camera"]
    cam = Camera()
    cam.resolution[0], cam.resolution[1] = (f32_from_bits(v) for v in c["resolution"])
    for k in range(3):
        cam.position[k] = f32_from_bits(c["positions"][frame][k])
        cam.view[k] = f32_from_bits(c["views"][frame][k])
        cam.up[k] = f32_from_bits(c["ups"][frame][k])
    cam.fov[0], cam.fov[1] = (f32_from_bits(v) for v in c["fov"])
    meshes = []
    if any(g.type == 2 for g in geoms):       # the reference's parser drops the file name: take it from the scene text
        meshes = scene_meshes(os.path.join(ROOT, "scenes", name + ".txt"))
    return Scene(geoms, mats, cam, c["iterations"], c["imageName"], meshes)


def scene_from_pods(geoms, materials, camera):
    """Oracle scene from the product's pt_geom / pt_material / pt_camera PODs (SceneFile.flatten)."""
    sc = Scene([], [], Camera())
    for g in geoms:
        og = Geom()
        og.type, og.materialid = g.type, g.materialid
        for k in range(12):
            og.transform[k] = g.transform[k]
            og.inverseTransform[k] = g.inverseTransform[k]
        og.transform[15] = og.inverseTransform[15] = 1.0
        sc.geoms.append(og)
    for m in materials:
        om = Material()
        C.memmove(C.byref(om), C.byref(m), 64)
        sc.materials.append(om)
    C.memmove(C.byref(sc.camera), C.byref(camera), 52)
    return sc


def default_config(depth=8, **kw):
    cfg = Config(max_depth=depth, camera_mode=0, antialias=0, aperture=0.0, focal_distance=0.0,
                 row_offset=0, row_stride=1, direct_light=0)
    for k, v in kw.items():
        setattr(cfg, k, v)
    return cfg


def render(scene, cfg, first=1, count=1, image=None, nthreads=0):
    """Oracle render: returns (image float32 [H,W,3] holding the SUM, live counts uint64[depth+1])."""
    L = lib()
    if image is None:
        image = np.zeros((scene.H, scene.W, 3), np.float32)
    live = np.zeros(cfg.max_depth + 1, np.uint64)
    with registered_meshes(scene):
        rc = L.orc_render(scene.geom_array(), scene.G, scene.material_array(), scene.M, C.byref(scene.camera),
                          C.byref(cfg), first, count, fptr(image), live.ctypes.data_as(C.POINTER(C.c_uint64)), nthreads)
    assert rc == 0
    return image, live


def raycast_flat(scene, image=None, nthreads=0):
    L = lib()
    if image is None:
        image = np.zeros((scene.H, scene.W, 3), np.float32)
    hit = np.zeros((scene.H, scene.W), np.int32)
    with registered_meshes(scene):
        rc = L.orc_raycast_flat(scene.geom_array(), scene.G, scene.material_array(), scene.M, C.byref(scene.camera),
                                fptr(image), hit.ctypes.data_as(C.POINTER(C.c_int)), nthreads)
    assert rc == 0
    return image, hit


def trace_pool(scene, cfg, iteration, bounces):
    L = lib()
    n = scene.W * scene.H
    arrs = [np.zeros(n, np.float32) for _ in range(9)]
    pix = np.zeros(n, np.uint32)
    with registered_meshes(scene):
        cnt = L.orc_trace_pool(scene.geom_array(), scene.G, scene.material_array(), scene.M, C.byref(scene.camera),
                               C.byref(cfg), iteration, bounces, *[fptr(a) for a in arrs],
                               pix.ctypes.data_as(C.POINTER(C.c_uint32)))
    return cnt, [a[:cnt] for a in arrs], pix[:cnt]


def many_primitives_scene(extra, seed=565, w=160, h=90, size=(0.12, 0.5)):
    """random256's room (five walls + the light) with `extra` small spheres / rotated cubes of its seven materials: the
    scenes beyond 256 primitives that the reference's type- and count-agnostic loop (raytraceKernel.cu:134-153,192-194)
    takes like any other.  Transforms are built by the oracle's restatement of buildTransformationMatrix
    (orc_build_transform, pinned bit for bit on the reference's utilities.cpp by tests/test_oracle_kats.py)."""
    base = load_golden_scene("random256").with_resolution(w, h)
    rng = np.random.default_rng(seed)
    geoms = list(base.geoms[:6])
    xf, inv = np.zeros(16, np.float32), np.zeros(16, np.float32)
    for i in range(extra):
        c = [float(np.float32(v)) for v in (rng.uniform(-4.6, 4.6), rng.uniform(0.4, 9.2), rng.uniform(-4.6, 4.6))]
        s = float(np.float32(rng.uniform(*size)))
        rot = [float(np.float32(v)) for v in rng.uniform(0, 360, 3)] if i & 1 else [0.0, 0.0, 0.0]
        lib().orc_build_transform(vec3(*c), vec3(*rot), vec3(s, s, s), fptr(xf), fptr(inv))
        g = Geom()
        g.type, g.materialid = (1 if i & 1 else 0), i % 7
        for k in range(16):
            g.transform[k] = float(xf[k]); g.inverseTransform[k] = float(inv[k])
        geoms.append(g)
    return Scene(geoms, base.materials, base.camera)
