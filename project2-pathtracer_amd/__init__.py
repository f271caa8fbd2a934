"""project2-pathtracer_amd -- ctypes binding of libptmi355.so (include/ptmi355.h).

The product is the shared library: hand-written HIP kernels for gfx950 behind a C ABI that
replaces the reference's per-iteration render entry point `cudaRaytraceCore`
(/root/reference/src/raytraceKernel.h:18).  This module is only the thin Python face used by
tests/ and bench.py (device memory via torch, multi-GPU via torch.distributed).  It never
falls back to a CPU implementation: without the built library, or without a gfx950 device,
creating a renderer raises.

Import with ``importlib.import_module("project2-pathtracer_amd")`` (the directory name is not
a Python identifier) or through the helper ``load_package()`` in tests/conftest.py.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# PTMI355_LIB: load another build of the same library (compiler-flag experiments, tools/flag_sweep.sh)
LIB_PATH = os.environ.get("PTMI355_LIB") or os.path.join(_HERE, "libptmi355.so")


class PtError(RuntimeError):
    pass


class Material(C.Structure):   # == pt_material == reference `material` (sceneStructs.h:62-73)
    _fields_ = [("color", C.c_float * 3), ("specularExponent", C.c_float), ("specularColor", C.c_float * 3),
                ("hasReflective", C.c_float), ("hasRefractive", C.c_float), ("indexOfRefraction", C.c_float),
                ("hasScatter", C.c_float), ("absorptionCoefficient", C.c_float * 3),
                ("reducedScatterCoefficient", C.c_float), ("emittance", C.c_float)]


class Geom(C.Structure):       # == pt_geom: rows x,y,z of transform / inverseTransform
    _fields_ = [("type", C.c_int), ("materialid", C.c_int), ("transform", C.c_float * 12),
                ("inverseTransform", C.c_float * 12)]


class Camera(C.Structure):     # == pt_camera == reference `cameraData` (sceneStructs.h:42-48)
    _fields_ = [("resolution", C.c_float * 2), ("position", C.c_float * 3), ("view", C.c_float * 3),
                ("up", C.c_float * 3), ("fov", C.c_float * 2)]


class Config(C.Structure):     # == pt_config (ABI 9)
    _fields_ = [("device", C.c_int), ("mode", C.c_int), ("max_depth", C.c_int), ("camera_mode", C.c_int),
                ("antialias", C.c_int), ("aperture", C.c_float), ("focal_distance", C.c_float),
                ("row_offset", C.c_int), ("row_stride", C.c_int),
                ("chunk_rays", C.c_int), ("blocks_per_cu", C.c_int), ("profile", C.c_int),
                ("culling", C.c_int), ("batch", C.c_int), ("ordering", C.c_int), ("direct_light", C.c_int), ("streams", C.c_int),
                ("grid_density", C.c_int)]


class Mesh(C.Structure):       # == pt_mesh
    _fields_ = [("geom_index", C.c_int), ("vertices", C.POINTER(C.c_float)), ("nvertices", C.c_int),
                ("indices", C.POINTER(C.c_int)), ("ntriangles", C.c_int)]


class Stats(C.Structure):      # == pt_stats
    _fields_ = [("generate_ms", C.c_double), ("bounce_ms", C.c_double), ("display_ms", C.c_double),
                ("generate_launches", C.c_uint64), ("bounce_launches", C.c_uint64),
                ("display_launches", C.c_uint64), ("iterations", C.c_uint64), ("live", C.c_uint64 * 65),
                ("emitted", C.c_uint64)]


_lib = None


def build(verbose=False):
    """Compile libptmi355.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
    subprocess.run(["make", "-C", _HERE] + ([] if verbose else ["-s"]), check=True)


def lib():
    """The loaded C ABI.  Raises if the library has not been built -- there is no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PtError("libptmi355.so is not built (run `make -C %s` or __graft_entry__.build())" % _HERE)
    L = C.CDLL(LIB_PATH)
    fp, ip, vp = C.POINTER(C.c_float), C.POINTER(C.c_int), C.c_void_p
    L.pt_abi_version.restype = C.c_int
    L.pt_last_error.restype = C.c_char_p
    L.pt_config_default.argtypes = [C.POINTER(Config)]; L.pt_config_default.restype = None
    L.pt_device_count.restype = C.c_int
    L.pt_create.argtypes = [C.POINTER(Config), C.POINTER(vp)]
    L.pt_destroy.argtypes = [vp]; L.pt_destroy.restype = None
    L.pt_upload_scene.argtypes = [vp, C.POINTER(Geom), C.c_int, C.POINTER(Material), C.c_int, C.POINTER(Camera)]
    L.pt_set_meshes.argtypes = [vp, C.POINTER(Mesh), C.c_int]
    L.pt_scene_mesh_count.argtypes = [vp]
    L.pt_scene_mesh.argtypes = [vp, C.c_int, C.POINTER(Mesh)]
    L.pt_set_image.argtypes = [vp, fp]
    L.pt_bind_device_image.argtypes = [vp, vp]
    L.pt_get_image.argtypes = [vp, fp]
    L.pt_get_rows.argtypes = [vp, fp]
    L.pt_gather_rows_peer.argtypes = [vp, vp]
    L.pt_render.argtypes = [vp, C.c_int, C.c_int]
    L.pt_sync.argtypes = [vp]
    L.pt_display.argtypes = [vp, C.c_float, vp, C.c_int]
    L.pt_set_profiling.argtypes = [vp, C.c_int]
    L.pt_get_stats.argtypes = [vp, C.POINTER(Stats)]
    L.pt_reset_stats.argtypes = [vp]
    L.pt_get_resolution.argtypes = [vp, ip, ip, ip]
    L.pt_debug_primary_hits.argtypes = [vp, fp, ip, fp, fp, fp]
    L.pt_debug_trace_pool.argtypes = [vp, C.c_int, C.c_int, ip] + [fp] * 9 + [C.POINTER(C.c_uint32)]
    L.pt_debug_set_turn_limit.argtypes = [vp, C.c_uint]
    L.pt_debug_rng_from_thread.argtypes = [vp, C.c_float, C.c_float, C.c_float, C.c_int, ip, fp]
    L.pt_debug_hemisphere.argtypes = [vp, C.c_int, fp, fp, fp]
    L.pt_debug_sincos.argtypes = [vp, C.c_int, fp, fp, fp]
    L.pt_debug_light_points.argtypes = [vp, C.c_int, C.c_int, fp, fp]
    if hasattr(L, "pt_debug_path_shape"):
        L.pt_debug_path_shape.argtypes = [vp, C.POINTER(C.c_uint)]
    if hasattr(L, "pt_debug_grid_probe"):      # (absent from builds of older commits that tools/ab_lib.sh compares against)
        L.pt_debug_grid_probe.argtypes = [C.POINTER(Geom), C.c_int, C.c_int, fp, C.c_int, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    if hasattr(L, "pt_debug_fan_probe"):
        L.pt_debug_fan_probe.argtypes = [C.POINTER(Geom), C.c_int, fp, C.c_int, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    L.pt_scene_load.argtypes = [C.c_char_p, C.POINTER(vp)]
    L.pt_scene_free.argtypes = [vp]; L.pt_scene_free.restype = None
    L.pt_scene_counts.argtypes = [vp, ip, ip, ip, ip]
    L.pt_scene_image_name.argtypes = [vp]; L.pt_scene_image_name.restype = C.c_char_p
    L.pt_scene_flatten.argtypes = [vp, C.c_int, C.POINTER(Geom), C.POINTER(Material), C.POINTER(Camera)]
    L.pt_scene_object_matrices.argtypes = [vp, C.c_int, C.c_int, fp, fp]
    L.pt_build_transform.argtypes = [fp, fp, fp, fp, fp]
    L.pt_image_to_u8.argtypes = [fp, C.c_int, C.c_int, C.c_int, C.c_float, C.POINTER(C.c_uint8)]
    L.pt_image_save.argtypes = [C.c_char_p, fp, C.c_int, C.c_int, C.c_int, C.c_float]
    _lib = L
    return L


EXPORTS = [
    "pt_abi_version", "pt_last_error", "pt_config_default", "pt_device_count", "pt_create", "pt_destroy",
    "pt_upload_scene", "pt_set_meshes", "pt_scene_mesh_count", "pt_scene_mesh", "pt_set_image", "pt_bind_device_image", "pt_get_image", "pt_get_rows", "pt_gather_rows_peer", "pt_render", "pt_sync",
    "pt_display", "pt_set_profiling", "pt_get_stats", "pt_reset_stats", "pt_get_resolution", "pt_debug_primary_hits",
    "pt_debug_trace_pool", "pt_debug_set_turn_limit", "pt_debug_path_shape", "pt_debug_rng_from_thread", "pt_debug_hemisphere", "pt_debug_sincos", "pt_debug_light_points", "pt_debug_grid_probe", "pt_debug_fan_probe",
    "pt_scene_load", "pt_scene_free", "pt_scene_counts", "pt_scene_image_name", "pt_scene_flatten",
    "pt_scene_object_matrices", "pt_build_transform", "pt_image_to_u8", "pt_image_save",
]


def _check(rc):
    if rc != 0:
        raise PtError("ptmi355 error %d: %s" % (rc, lib().pt_last_error().decode(errors="replace")))


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def default_config(**kw):
    cfg = Config()
    lib().pt_config_default(C.byref(cfg))
    for k, v in kw.items():
        setattr(cfg, k, v)
    return cfg


class SceneFile:
    """A parsed scene file (reference grammar, src/scene.cpp); host-only, needs no GPU."""

    def __init__(self, path):
        self._h = C.c_void_p()
        _check(lib().pt_scene_load(os.fsencode(path), C.byref(self._h)))
        g, m, f, it = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        _check(lib().pt_scene_counts(self._h, C.byref(g), C.byref(m), C.byref(f), C.byref(it)))
        self.ngeoms, self.nmaterials, self.nframes, self.iterations = g.value, m.value, f.value, it.value
        self.image_name = lib().pt_scene_image_name(self._h).decode()

    def flatten(self, frame=0):
        geoms = (Geom * self.ngeoms)()
        mats = (Material * self.nmaterials)()
        cam = Camera()
        _check(lib().pt_scene_flatten(self._h, frame, geoms, mats, C.byref(cam)))
        return geoms, mats, cam

    def meshes(self):
        """[(geom_index, vertices float32 [n,3], indices int32 [t,3])] of the MESH objects whose .obj was found"""
        out = []
        for k in range(lib().pt_scene_mesh_count(self._h)):
            m = Mesh()
            _check(lib().pt_scene_mesh(self._h, k, C.byref(m)))
            v = np.ctypeslib.as_array(m.vertices, shape=(m.nvertices, 3)).astype(np.float32).copy()
            i = np.ctypeslib.as_array(m.indices, shape=(m.ntriangles, 3)).astype(np.int32).copy()
            out.append((m.geom_index, v, i))
        return out

    def object_matrices(self, obj, frame=0):
        xf, inv = np.zeros(16, np.float32), np.zeros(16, np.float32)
        _check(lib().pt_scene_object_matrices(self._h, obj, frame, _fp(xf), _fp(inv)))
        return xf, inv

    def close(self):
        if self._h:
            lib().pt_scene_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class PathTracer:
    """One render context on one GPU (pt_context)."""

    def __init__(self, cfg=None, **kw):
        self.cfg = cfg if cfg is not None else default_config(**kw)
        self._h = C.c_void_p()
        _check(lib().pt_create(C.byref(self.cfg), C.byref(self._h)))
        self.W = self.H = self.owned = 0
        self._bound = None

    def set_meshes(self, meshes):
        """meshes: [(geom_index, vertices [n,3] float32, indices [t,3] int32)]; used by the next upload()"""
        arr = (Mesh * max(1, len(meshes)))()
        keep = []
        for k, (gi, v, i) in enumerate(meshes):
            v = np.ascontiguousarray(v, np.float32); i = np.ascontiguousarray(i, np.int32)
            keep += [v, i]
            arr[k].geom_index = int(gi)
            arr[k].vertices = v.ctypes.data_as(C.POINTER(C.c_float)); arr[k].nvertices = len(v)
            arr[k].indices = i.ctypes.data_as(C.POINTER(C.c_int)); arr[k].ntriangles = len(i)
        _check(lib().pt_set_meshes(self._h, arr, len(meshes)))

    def upload(self, geoms, mats, cam):
        _check(lib().pt_upload_scene(self._h, geoms, len(geoms), mats, len(mats), C.byref(cam)))
        w, h, o = C.c_int(), C.c_int(), C.c_int()
        _check(lib().pt_get_resolution(self._h, C.byref(w), C.byref(h), C.byref(o)))
        self.W, self.H, self.owned = w.value, h.value, o.value

    def set_image(self, img=None):
        if img is None:
            _check(lib().pt_set_image(self._h, None))
        else:
            img = np.ascontiguousarray(img, np.float32)
            assert img.size == self.W * self.H * 3
            _check(lib().pt_set_image(self._h, _fp(img)))

    def bind_device_image(self, tensor):
        """Render into a torch CUDA tensor (float32, W*H*3).  Keeps a reference to it."""
        if tensor is None:
            _check(lib().pt_bind_device_image(self._h, None))
            self._bound = None
            return
        assert tensor.is_cuda and tensor.is_contiguous() and tensor.numel() == self.W * self.H * 3
        _check(lib().pt_bind_device_image(self._h, C.c_void_p(tensor.data_ptr())))
        self._bound = tensor

    def render(self, first, count=1):
        _check(lib().pt_render(self._h, first, count))

    def sync(self):
        _check(lib().pt_sync(self._h))

    def image(self):
        out = np.zeros((self.H, self.W, 3), np.float32)
        _check(lib().pt_get_image(self._h, _fp(out)))
        return out

    def get_rows(self, out):
        """Copy only the rows this context owns into the full-frame host array `out` (H, W, 3) float32."""
        assert out.dtype == np.float32 and out.flags["C_CONTIGUOUS"] and out.size == self.W * self.H * 3
        _check(lib().pt_get_rows(self._h, _fp(out)))
        return out

    def gather_rows_from(self, other):
        """Device-to-device: copy the rows `other` owns into this context's accumulator (peer copy)."""
        _check(lib().pt_gather_rows_peer(self._h, other._h))

    def display(self, scale=1.0):
        out = np.zeros((self.H, self.W, 4), np.uint8)
        _check(lib().pt_display(self._h, scale, out.ctypes.data_as(C.c_void_p), 0))
        return out

    def set_profiling(self, enabled):
        _check(lib().pt_set_profiling(self._h, int(bool(enabled))))

    def stats(self):
        s = Stats()
        _check(lib().pt_get_stats(self._h, C.byref(s)))
        return s

    def reset_stats(self):
        _check(lib().pt_reset_stats(self._h))

    def primary_hits(self):
        n = self.W * self.H
        d, P, N = (np.zeros((n, 3), np.float32) for _ in range(3))
        t, hit = np.zeros(n, np.float32), np.zeros(n, np.int32)
        _check(lib().pt_debug_primary_hits(self._h, _fp(d), hit.ctypes.data_as(C.POINTER(C.c_int)), _fp(t), _fp(P), _fp(N)))
        return d, hit, t, P, N

    def trace_pool(self, iteration, bounces):
        arrs = [np.zeros(self.owned, np.float32) for _ in range(9)]
        pix = np.zeros(self.owned, np.uint32)
        cnt = C.c_int()
        _check(lib().pt_debug_trace_pool(self._h, iteration, bounces, C.byref(cnt), *[_fp(a) for a in arrs],
                                         pix.ctypes.data_as(C.POINTER(C.c_uint32))))
        return cnt.value, [a[:cnt.value] for a in arrs], pix[:cnt.value]

    def path_shape(self):
        """how the uploaded scene runs on the whole-path kernels (pt_debug_path_shape)"""
        out = (C.c_uint * 8)()
        _check(lib().pt_debug_path_shape(self._h, out))
        return dict(family=("per-bounce", "k_path_q", "k_path_w")[out[0]], waves_per_block=out[1], blocks_per_cu=out[2], lds_bytes=out[3],
                    records_per_wave=out[4], arena_bytes=out[5], meshes=bool(out[6]), direct_light=bool(out[7]))

    def set_turn_limit(self, turns):
        _check(lib().pt_debug_set_turn_limit(self._h, int(turns)))

    def rng_from_thread(self, resx, resy, time, xy):
        xy = np.ascontiguousarray(xy, np.int32)
        out = np.zeros((len(xy), 3), np.float32)
        _check(lib().pt_debug_rng_from_thread(self._h, resx, resy, time, len(xy), xy.ctypes.data_as(C.POINTER(C.c_int)), _fp(out)))
        return out

    def hemisphere(self, normals, xi):
        normals = np.ascontiguousarray(normals, np.float32)
        xi = np.ascontiguousarray(xi, np.float32)
        out = np.zeros_like(normals)
        _check(lib().pt_debug_hemisphere(self._h, len(normals), _fp(normals), _fp(xi), _fp(out)))
        return out

    def light_points(self, geom, seeds):
        seeds = np.ascontiguousarray(seeds, np.float32)
        out = np.zeros((len(seeds), 3), np.float32)
        _check(lib().pt_debug_light_points(self._h, geom, len(seeds), _fp(seeds), _fp(out)))
        return out

    def sincos(self, a):
        a = np.ascontiguousarray(a, np.float32)
        s, c = np.zeros_like(a), np.zeros_like(a)
        _check(lib().pt_debug_sincos(self._h, len(a), _fp(a), _fp(s), _fp(c)))
        return s, c

    def close(self):
        if self._h:
            lib().pt_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def grid_probe(geoms, rays, density=0):
    """k_path_w's spatial index probed on the host (no device): (sets, info) for rays[n, 6] = origin + direction.
    sets[n, max(256, G rounded up to 32)] (bool): primitive p gets its bound tested for ray n; info: dict of the grid's figures.
    Up to 256 primitives the grid has narrow (16-bit) references, beyond that wide ones -- as pt_upload_scene builds it."""
    import numpy as np
    rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 6)
    n = rays.shape[0]
    nwords = max(8, (len(geoms) + 31) // 32)
    words = np.zeros((n, nwords), dtype=np.uint32)
    info = np.zeros(24, dtype=np.uint32)
    _check(lib().pt_debug_grid_probe(geoms, len(geoms), int(density), _fp(rays), n,
                                     words.ctypes.data_as(C.POINTER(C.c_uint32)), info.ctypes.data_as(C.POINTER(C.c_uint32))))
    sets = np.unpackbits(words.view(np.uint8), axis=1, bitorder="little").astype(bool)
    names = ["cells", "refs", "big", "duplicates", "unwalked", "nx", "ny", "nz", "mean_walk", "longest_walk",
             "cells_per_ray_x100", "listed_per_ray_x100", "lds_bytes", "bin1", "bin2", "length_estimate_worst", "length_estimate_mean_error_x100"]
    return sets, {k: int(v) for k, v in zip(names, info)}


def fan_probe(geoms, rays):
    """The cone test of k_path_w's camera groups on the host (no device): rays[nfans, 64, 6] -> (sets[nfans, >= 256] bool: the
    fan's rays test primitive p's bound, fans_with_a_cone)."""
    import numpy as np
    rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 64, 6)
    n = rays.shape[0]
    nwords = max(8, (len(geoms) + 31) // 32)
    words = np.zeros((n, nwords), dtype=np.uint32)
    info = np.zeros(4, dtype=np.uint32)
    _check(lib().pt_debug_fan_probe(geoms, len(geoms), _fp(rays), n, words.ctypes.data_as(C.POINTER(C.c_uint32)), info.ctypes.data_as(C.POINTER(C.c_uint32))))
    return np.unpackbits(words.view(np.uint8), axis=1, bitorder="little").astype(bool), int(info[0])


def build_transform(t, r, s):
    t, r, s = (np.asarray(v, np.float32) for v in (t, r, s))
    xf, inv = np.zeros(16, np.float32), np.zeros(16, np.float32)
    _check(lib().pt_build_transform(_fp(t), _fp(r), _fp(s), _fp(xf), _fp(inv)))
    return xf, inv


def image_to_u8(img_sum, divisor, gamma):
    img = np.ascontiguousarray(img_sum, np.float32)
    h, w = img.shape[:2]
    out = np.zeros((h, w, 3), np.uint8)
    _check(lib().pt_image_to_u8(_fp(img), w, h, int(divisor), float(gamma), out.ctypes.data_as(C.POINTER(C.c_uint8))))
    return out


def image_save(path, img_sum, divisor, gamma):
    img = np.ascontiguousarray(img_sum, np.float32)
    h, w = img.shape[:2]
    _check(lib().pt_image_save(os.fsencode(path), _fp(img), w, h, int(divisor), float(gamma)))
