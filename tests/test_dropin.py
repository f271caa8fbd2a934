"""The drop-in boundary: the reference's OWN host code -- main.cpp, glslUtility.cpp, scene.cpp, utilities.cpp,
image.cpp, compiled UNCHANGED from /root/reference where they lie against the product's headless shim include
directory (project2-pathtracer_amd/shim) -- driving the HIP library through the adaptor's `cudaRaytraceCore`
symbol.  oracle/_ref/main_dropin is built in the container by oracle/Makefile and travels to the GPU box as a binary
(the reference sources do not).  The viewer's own command line is used (`scene=... frame=N`); it saves
renders/<name>.<frame>.bmp relative to the working directory and needs shaders/passthrough{VS,FS}.glsl there
(src/main.cpp:75, src/glslUtility.cpp:19-37).  Test hooks of the adaptor / shim: PT_DUMP_IMAGE (raw accumulator),
PT_SHIM_PBO_DUMP (last display buffer), PT_SHIM_NO_PBO (no mapped display buffer)."""
import hashlib
import json
import os
import re
import subprocess

import numpy as np
import pytest

import orc
from conftest import ROOT, has_reference, load_package

DRIVER = os.path.join(ROOT, "oracle", "_ref", "main_dropin")
K = json.load(open(os.path.join(orc.GOLD, "survey_kats.json")))


@pytest.mark.skipif(not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "shim_adaptor.o")), reason="oracle/_ref not built")
def test_adaptor_defines_the_reference_symbol():
    """Same mangled name as the declaration in /root/reference/src/raytraceKernel.h:18 produces in
    main.cpp (types `uchar4`, `camera`, `material`, `geom` from the reference/CUDA headers)."""
    out = subprocess.run(["nm", os.path.join(ROOT, "oracle", "_ref", "shim_adaptor.o")], capture_output=True, text=True).stdout
    assert " T _Z16cudaRaytraceCoreP6uchar4P6cameraiiP8materialiP4geomi" in out
    assert "ptmi355_adaptor_reset" in out
    # ... and the unchanged main.cpp asks for exactly that symbol
    main_o = subprocess.run(["nm", os.path.join(ROOT, "oracle", "_ref", "shim_main.o")], capture_output=True, text=True).stdout
    assert " U _Z16cudaRaytraceCoreP6uchar4P6cameraiiP8materialiP4geomi" in main_o and " U cudaDeviceReset" in main_o


@pytest.mark.skipif(not os.path.exists(DRIVER), reason="oracle/_ref/main_dropin not built")
def test_error_convention_without_gpu(tmp_path):
    pkg = load_package()
    if pkg.lib().pt_device_count() > 0:
        pytest.skip("a GPU is present")
    r = run_viewer(str(tmp_path), os.path.join(ROOT, "scenes", "cornell_c1.txt"))
    assert r.returncode == 1                                        # exit(EXIT_FAILURE)
    assert r.stderr.startswith("Cuda error: ") and r.stderr.rstrip().endswith(".")   # raytraceKernel.cu:23


def run_viewer(workdir, scene, frame=0, **env):
    """the reference's unchanged main(): `scene=<file> frame=<n>` in a directory holding shaders/ and renders/"""
    os.makedirs(os.path.join(workdir, "shaders"), exist_ok=True)
    os.makedirs(os.path.join(workdir, "renders"), exist_ok=True)
    for name in ("passthroughVS.glsl", "passthroughFS.glsl"):
        with open(os.path.join(workdir, "shaders", name), "w") as f:
            f.write("void main() {}\n")
    e = dict(os.environ, PT_DUMP_IMAGE=os.path.join(workdir, "image.f32"), PT_SHIM_PBO_DUMP=os.path.join(workdir, "pbo.u8"))
    e.update({k: str(v) for k, v in env.items()})
    return subprocess.run([DRIVER, "scene=" + scene, "frame=%d" % frame], capture_output=True, text=True, cwd=workdir, env=e, timeout=600)


def _retarget(text, w, h, iters):
    text, n = re.subn(r"^RES\s+\d+\s+\d+$", "RES %d %d" % (w, h), text, flags=re.M)
    assert n == 1
    text, n = re.subn(r"^ITERATIONS\s+\d+$", "ITERATIONS %d" % iters, text, flags=re.M)
    assert n == 1
    return text


def _small_scene(tmp_path, w, h, iters):
    text = open(os.path.join(ROOT, "scenes", "cornell_mirror.txt")).read()
    text = _retarget(text, w, h, iters)
    p = tmp_path / "small.txt"
    p.write_text(text)
    return str(p)


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(DRIVER), reason="oracle/_ref/main_dropin not built")
def test_config1_through_reference_host_code(tmp_path):
    """BASELINE config 1 through the reference's scene parser, the adaptor, the HIP library in
    reference-kernel mode and the reference's image writer: the saved BMP's raster is the one the
    unchanged reference produced (sha256 from the survey)."""
    r = run_viewer(str(tmp_path), os.path.join(ROOT, "scenes", "cornell_c1.txt"), PT_MODE="reference")
    assert r.returncode == 0, r.stderr
    assert "Saved frame 0 to renders/sampleScene.0.bmp" in r.stdout
    bmp = (tmp_path / "renders" / "sampleScene.0.bmp").read_bytes()
    assert len(bmp) == K["c1_bmp"]["file_bytes"]
    rows = [bmp[54 + y * 1200: 54 + (y + 1) * 1200] for y in range(399, -1, -1)]
    raster = np.frombuffer(b"".join(rows), np.uint8).reshape(400, 400, 3)[:, :, ::-1]
    assert hashlib.sha256(raster.tobytes()).hexdigest() == K["c1_bmp"]["raster_sha256"]


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(DRIVER), reason="oracle/_ref/main_dropin not built")
def test_path_trace_through_reference_host_code(tmp_path):
    pkg = load_package()
    scene_path = _small_scene(tmp_path, 64, 48, 3)
    r = run_viewer(str(tmp_path), scene_path, PT_MODE="pathtrace", PT_MAX_DEPTH=5)
    assert r.returncode == 0, r.stderr
    got = np.fromfile(str(tmp_path / "image.f32"), np.float32).reshape(48, 64, 3)
    sf = pkg.SceneFile(scene_path)
    geoms, mats, cam = sf.flatten(0)
    import ctypes as C
    sc = orc.Scene([], [], orc.Camera())
    for g in geoms:
        og = orc.Geom(); og.type, og.materialid = g.type, g.materialid
        for k in range(12):
            og.transform[k] = g.transform[k]; og.inverseTransform[k] = g.inverseTransform[k]
        og.transform[15] = og.inverseTransform[15] = 1.0
        sc.geoms.append(og)
    for m in mats:
        om = orc.Material(); C.memmove(C.byref(om), C.byref(m), 64); sc.materials.append(om)
    C.memmove(C.byref(sc.camera), C.byref(cam), 52)
    want, _ = orc.render(sc, orc.default_config(5), 1, 3)
    assert np.array_equal(got, want)
    # the saved file = gamma/clamp/u8 of that sum with divisor = iterations, through the reference's image class
    ref_u8 = np.zeros(48 * 64 * 3, np.uint8)
    orc.lib().orc_image_to_u8(orc.fptr(want), 48 * 64, 3.0, float(np.float32(1 / 2.2)), ref_u8.ctypes.data_as(C.POINTER(C.c_uint8)))
    bmp = (tmp_path / "renders" / "cornell_mirror.0.bmp").read_bytes()
    rows = [bmp[54 + y * 192: 54 + (y + 1) * 192] for y in range(47, -1, -1)]
    assert np.array_equal(np.frombuffer(b"".join(rows), np.uint8).reshape(48, 64, 3)[:, :, ::-1].reshape(-1), ref_u8)
    # the PBO bytes of the last call: sendImageToPBO of sum/iterations
    pbo = np.fromfile(str(tmp_path / "pbo.u8"), np.uint8).reshape(-1, 4)
    o = (C.c_uint8 * 4)()
    scale = np.float32(1.0) / np.float32(3)
    for idx in (0, 1000, 3071):
        px = (want.reshape(-1, 3)[idx] * scale).astype(np.float32)
        orc.lib().orc_display_pixel(orc.vec3(*px), o)
        assert list(pbo[idx]) == list(o)


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(DRIVER), reason="oracle/_ref/main_dropin not built")
def test_lazy_batching_behind_the_per_iteration_api(tmp_path):
    """PT_LAZY_BATCH: the adaptor queues per-iteration calls and renders them as one launch group;
    what the caller can observe (camera::image after the last iteration) is bit-identical."""
    scene_path = _small_scene(tmp_path, 64, 48, 7)
    outs = []
    for lazy, sub in (("1", "a"), ("4", "b"), ("16", "c")):
        out = tmp_path / sub
        out.mkdir()
        r = run_viewer(str(out), scene_path, PT_MODE="pathtrace", PT_MAX_DEPTH=5, PT_LAZY_BATCH=lazy, PT_SHIM_NO_PBO=1)
        assert r.returncode == 0, r.stderr
        outs.append(np.fromfile(str(out / "image.f32"), np.float32))
    assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])
    assert outs[0].max() > 0


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(DRIVER), reason="oracle/_ref/main_dropin not built")
def test_frame_argument_selects_the_frame(tmp_path):
    """`frame=1` (src/main.cpp:39-42): the second frame of the scene file is flattened and rendered,
    the output is named X.1.bmp (src/main.cpp:148-154)."""
    text = open(os.path.join(ROOT, "scenes", "cornell_mirror.txt")).read()
    text = _retarget(text, 48, 32, 2)
    # move the camera in frame 1 only (the CAMERA block's second frame)
    cam_at = text.index("CAMERA")
    head, sep, tail = text[cam_at:].partition("frame 1\n")
    assert sep
    tail, n = re.subn(r"^EYE 0 4\.5 12$", "EYE 1 4.5 11", tail, count=1, flags=re.M)
    assert n == 1
    text = text[:cam_at] + head + sep + tail
    p = tmp_path / "two_frames.txt"
    p.write_text(text)
    outs = []
    for frame in (0, 1):
        r = run_viewer(str(tmp_path), str(p), frame=frame, PT_MODE="pathtrace", PT_MAX_DEPTH=4)
        assert r.returncode == 0, r.stderr
        assert os.path.exists(str(tmp_path / "renders" / ("cornell_mirror.%d.bmp" % frame)))
        outs.append(np.fromfile(str(tmp_path / "image.f32"), np.float32))
    assert not np.array_equal(outs[0], outs[1])
    pkg = load_package()
    sf = pkg.SceneFile(str(p))
    import ctypes as C
    for frame in (0, 1):
        geoms, mats, cam = sf.flatten(frame)
        sc = orc.Scene([], [], orc.Camera())
        for g in geoms:
            og = orc.Geom(); og.type, og.materialid = g.type, g.materialid
            for k in range(12):
                og.transform[k] = g.transform[k]; og.inverseTransform[k] = g.inverseTransform[k]
            og.transform[15] = og.inverseTransform[15] = 1.0
            sc.geoms.append(og)
        for m in mats:
            om = orc.Material(); C.memmove(C.byref(om), C.byref(m), 64); sc.materials.append(om)
        C.memmove(C.byref(sc.camera), C.byref(cam), 52)
        want, _ = orc.render(sc, orc.default_config(4), 1, 2)
        assert np.array_equal(outs[frame].reshape(want.shape), want)


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(DRIVER), reason="oracle/_ref/main_dropin not built")
@pytest.mark.parametrize("ngpu,pbo", [(2, "0"), (3, "1")])
def test_adaptor_shards_rows_over_several_contexts(tmp_path, ngpu, pbo):
    """PT_NGPU: the adaptor drives one context per GPU from the reference's single caller thread (rows
    interleaved, gathered at the observation points).  Exercised here with all contexts on device 0."""
    scene_path = _small_scene(tmp_path, 64, 48, 5)
    outs = {}
    for n in (1, ngpu):
        out = tmp_path / ("n%d" % n)
        out.mkdir()
        r = run_viewer(str(out), scene_path, PT_MODE="pathtrace", PT_MAX_DEPTH=5, PT_NGPU=n, PT_DEVICES=",".join(["0"] * n), PT_LAZY_BATCH=2,
                       PT_SHIM_NO_PBO=0 if pbo == "1" else 1)
        assert r.returncode == 0, r.stderr
        outs[n] = (np.fromfile(str(out / "image.f32"), np.float32), (out / "renders" / "cornell_mirror.0.bmp").read_bytes(),
                   np.fromfile(str(out / "pbo.u8"), np.uint8))
    assert np.array_equal(outs[1][0], outs[ngpu][0]) and outs[1][1] == outs[ngpu][1]
    if pbo == "1":
        assert np.array_equal(outs[1][2], outs[ngpu][2])


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "adaptor_probe")), reason="oracle/_ref/adaptor_probe not built")
@pytest.mark.parametrize("lazy,ngpu", [(1, 1), (8, 1), (3, 2)])
def test_scene_change_in_the_middle_of_a_frame_keeps_the_samples(tmp_path, lazy, ngpu):
    """The reference round-trips camera::image on every call, so a caller may change geometry or materials between two
    iterations of a frame and keeps what was rendered so far.  The adaptor holds the accumulator on the device between
    observation points: on a content change it must bring the samples home before it rebuilds its device state
    (ADVICE r1).  Expected = the oracle: iterations 1..2 of scene A, then 3..6 of scene B on top."""
    import ctypes as C
    pkg = load_package()
    scene_path = _small_scene(tmp_path, 64, 48, 6)
    out = str(tmp_path / "mid.f32")
    env = dict(os.environ, PT_MODE="pathtrace", PT_MAX_DEPTH="5", PT_LAZY_BATCH=str(lazy), PT_NGPU=str(ngpu), PT_DEVICES=",".join(["0"] * ngpu))
    r = subprocess.run([os.path.join(ROOT, "oracle", "_ref", "adaptor_probe"), scene_path, "6", "3", out], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr
    got = np.fromfile(out, np.float32).reshape(48, 64, 3)
    sf = pkg.SceneFile(scene_path)
    geoms, mats, cam = sf.flatten(0)
    a = orc.scene_from_pods(geoms, mats, cam)
    part, _ = orc.render(a, orc.default_config(5), 1, 2)
    b = orc.scene_from_pods(geoms, mats, cam)
    b.materials[0].color[0], b.materials[0].color[1], b.materials[0].color[2] = 0.2, 0.9, 0.4
    # object 5 moved up by 1: rebuild its matrices from the TRS triple the reference's parser read
    gold = json.load(open(os.path.join(orc.GOLD, "ref_scene_cornell_mirror.json")))["objects"][5]["frames"][0]
    tr = [orc.f32_from_bits(v) for v in gold["translation"]]
    ro = [orc.f32_from_bits(v) for v in gold["rotation"]]
    scl = [orc.f32_from_bits(v) for v in gold["scale"]]
    tr[1] = float(np.float32(tr[1]) + np.float32(1.0))
    xf, inv = np.zeros(16, np.float32), np.zeros(16, np.float32)
    orc.lib().orc_build_transform(orc.vec3(*tr), orc.vec3(*ro), orc.vec3(*scl), orc.fptr(xf), orc.fptr(inv))
    for k in range(16):
        b.geoms[5].transform[k] = float(xf[k]); b.geoms[5].inverseTransform[k] = float(inv[k])
    want, _ = orc.render(b, orc.default_config(5), 3, 4, image=part.copy())
    assert np.array_equal(got, want)
    assert not np.array_equal(got, orc.render(a, orc.default_config(5), 1, 6)[0])      # the change is visible
