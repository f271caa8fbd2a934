"""The product's own headless driver (project2-pathtracer_amd/driver/ptrender.cpp):
the command line, frame loop, file naming and messages of /root/reference/src/main.cpp on top of
the C ABI alone -- no reference sources involved at build or run time.  On the GPU its files are
compared byte for byte with the ones the reference's own scene/image code writes through the
adaptor (oracle/_ref/main_dropin = the unchanged main.cpp) and with the survey's sha256 of the unchanged reference's BMP."""
import hashlib
import json
import os
import re
import subprocess

import numpy as np
import pytest

import orc
from conftest import ROOT, load_package

PTRENDER = os.path.join(ROOT, "project2-pathtracer_amd", "ptrender")
DROPIN = os.path.join(ROOT, "oracle", "_ref", "main_dropin")
K = json.load(open(os.path.join(orc.GOLD, "survey_kats.json")))

pytestmark = pytest.mark.skipif(not os.path.exists(PTRENDER), reason="ptrender not built (make -C project2-pathtracer_amd)")


def _run(args, env=None):
    return subprocess.run([PTRENDER] + args, capture_output=True, text=True, cwd=ROOT, env=dict(os.environ, **(env or {})))


def _retarget(name, tmp_path, w, h, iters, camera_frame1_eye=None):
    text = open(os.path.join(ROOT, "scenes", name)).read()
    text, n = re.subn(r"^RES\s+\d+\s+\d+$", "RES %d %d" % (w, h), text, flags=re.M)
    assert n == 1
    text, n = re.subn(r"^ITERATIONS\s+\d+$", "ITERATIONS %d" % iters, text, flags=re.M)
    assert n == 1
    if camera_frame1_eye:
        cam_at = text.index("CAMERA")
        head, sep, tail = text[cam_at:].partition("frame 1\n")
        tail, n = re.subn(r"^EYE .*$", "EYE " + camera_frame1_eye, tail, count=1, flags=re.M)
        assert sep and n == 1
        text = text[:cam_at] + head + sep + tail
    p = tmp_path / "scene.txt"
    p.write_text(text)
    return str(p)


def test_no_scene_argument_is_the_reference_message():
    """src/main.cpp:44-47: prints the message and returns 0."""
    r = _run([])
    assert r.returncode == 0 and r.stdout == "Error: scene file needed!\n"
    r = _run(["frame=3", "bogus"])
    assert r.returncode == 0 and r.stdout == "Error: scene file needed!\n"


def test_unreadable_scene_fails_cleanly(tmp_path):
    r = _run(["scene=" + str(tmp_path / "missing.txt")])
    assert r.returncode == 1 and "missing.txt" in r.stderr


def test_no_gpu_is_a_loud_failure():
    """No CPU fallback: without a device the driver ends with the reference's error convention
    (src/raytraceKernel.cu:20-26) and a non-zero status."""
    pkg = load_package()
    if pkg.lib().pt_device_count() > 0:
        pytest.skip("a GPU is present")
    r = _run(["scene=" + os.path.join(ROOT, "scenes", "cornell_c1.txt"), "frame=0"])
    assert r.returncode == 1
    assert r.stderr.startswith("Cuda error: ") and r.stderr.rstrip().endswith(".")


def test_driver_links_only_the_c_abi():
    """The driver is plain C++ on include/ptmi355.h: its only non-system dependency is libptmi355.so."""
    out = subprocess.run(["ldd", PTRENDER], capture_output=True, text=True).stdout
    assert "libptmi355.so" in out
    assert "pt_oracle" not in out
    src = open(os.path.join(ROOT, "project2-pathtracer_amd", "driver", "ptrender.cpp")).read()
    includes = re.findall(r'#include\s+[<"]([^>"]+)[>"]', src)
    assert set(includes) == {"cstdio", "cstdlib", "cstring", "string", "vector", "ptmi355.h"}


@pytest.mark.gpu
def test_config1_bmp_is_the_reference_raster(tmp_path):
    """BASELINE config 1 (400x400, 1 iteration, the kernel as shipped) end to end through the product's
    own parser, kernels and BMP writer: the raster hashes to the unchanged reference's."""
    r = _run(["scene=" + os.path.join(ROOT, "scenes", "cornell_c1.txt"), "frame=0", "mode=reference", "out=" + str(tmp_path)])
    assert r.returncode == 0, r.stderr
    assert r.stdout == "Saved frame 0 to %s/sampleScene.0.bmp\n" % tmp_path
    bmp = (tmp_path / "sampleScene.0.bmp").read_bytes()
    assert len(bmp) == K["c1_bmp"]["file_bytes"]
    rows = [bmp[54 + y * 1200: 54 + (y + 1) * 1200] for y in range(399, -1, -1)]
    raster = np.frombuffer(b"".join(rows), np.uint8).reshape(400, 400, 3)[:, :, ::-1]
    assert hashlib.sha256(raster.tobytes()).hexdigest() == K["c1_bmp"]["raster_sha256"]
    if os.path.exists(DROPIN):
        d = tmp_path / "ref"
        d.mkdir()
        from test_dropin import run_viewer
        r2 = run_viewer(str(d), os.path.join(ROOT, "scenes", "cornell_c1.txt"), frame=0, PT_MODE="reference")
        assert r2.returncode == 0, r2.stderr
        assert (d / "renders" / "sampleScene.0.bmp").read_bytes() == bmp          # whole file, header included


@pytest.mark.gpu
def test_frame_loop_names_and_oracle_parity(tmp_path):
    """Without `frame=` every frame is rendered in turn into X.<frame>.bmp (src/main.cpp:148-173); each
    frame's float sum equals the oracle's and, where the reference-code driver is available, the BMP
    equals the one the reference's image class writes."""
    pkg = load_package()
    scene_path = _retarget("cornell_mirror.txt", tmp_path, 56, 40, 3, camera_frame1_eye="1 4.5 11")
    r = _run(["scene=" + scene_path, "depth=5", "raw=1", "out=" + str(tmp_path)])
    assert r.returncode == 0, r.stderr
    assert r.stdout == "".join("Saved frame %d to %s/cornell_mirror.%d.bmp\n" % (f, tmp_path, f) for f in (0, 1))
    sf = pkg.SceneFile(scene_path)
    images = []
    for f in (0, 1):
        got = np.fromfile(str(tmp_path / ("cornell_mirror.%d.bmp.f32" % f)), np.float32).reshape(40, 56, 3)
        want, _ = orc.render(orc.scene_from_pods(*sf.flatten(f)), orc.default_config(5), 1, 3)
        assert np.array_equal(got, want)
        images.append(got)
    assert not np.array_equal(images[0], images[1])                     # the camera moved in frame 1
    if os.path.exists(DROPIN):
        for f in (0, 1):
            d = tmp_path / ("ref%d" % f)
            d.mkdir()
            from test_dropin import run_viewer
            r2 = run_viewer(str(d), scene_path, frame=f, PT_MODE="pathtrace", PT_MAX_DEPTH=5)
            assert r2.returncode == 0, r2.stderr
            assert (d / "renders" / ("cornell_mirror.%d.bmp" % f)).read_bytes() == (tmp_path / ("cornell_mirror.%d.bmp" % f)).read_bytes()


@pytest.mark.gpu
def test_out_of_range_frame_and_two_contexts(tmp_path):
    """frame beyond the scene's frames -> the reference's warning and frame 0 (src/main.cpp:55-58);
    two contexts with interleaved rows (both on device 0 here) give the one-context picture bit for bit."""
    scene_path = _retarget("cornell_glass_4k.txt", tmp_path, 64, 36, 4)
    outs = []
    for extra in ([], ["gpus=2", "devices=0,0"]):
        d = tmp_path / ("o%d" % len(outs))
        d.mkdir()
        r = _run(["scene=" + scene_path, "frame=9", "depth=6", "camera=1", "aa=1", "aperture=0.25", "focal=12", "raw=1", "out=" + str(d)] + extra)
        assert r.returncode == 0, r.stderr
        assert r.stdout.startswith("Warning: Specified target frame is out of range, defaulting to frame 0.\nSaved frame 0 to ")
        f32 = [p for p in os.listdir(d) if p.endswith(".f32")]
        assert len(f32) == 1 and ".0." in f32[0]
        outs.append(np.fromfile(str(d / f32[0]), np.float32))
    assert outs[0].max() > 0 and np.array_equal(outs[0], outs[1])


@pytest.mark.gpu
def test_mesh_scene_through_ptrender(tmp_path):
    """scenes/cornell_mesh.txt: the loader finds the three .obj files beside the scene, ptrender registers them
    (pt_scene_mesh -> pt_set_meshes) and the float sum equals the oracle's brute-force triangles"""
    scene_path = os.path.join(ROOT, "scenes", "cornell_mesh.txt")
    text = open(scene_path).read()
    text, n = re.subn(r"^RES\s+\d+\s+\d+$", "RES 96 72", text, flags=re.M)
    text, m = re.subn(r"^ITERATIONS\s+\d+$", "ITERATIONS 2", text, flags=re.M)
    assert n == 1 and m == 1
    import shutil
    shutil.copytree(os.path.join(ROOT, "scenes", "meshes"), str(tmp_path / "meshes"))
    p = tmp_path / "mesh_small.txt"
    p.write_text(text)
    r = _run(["scene=" + str(p), "frame=0", "out=" + str(tmp_path), "depth=5", "raw=1"])
    assert r.returncode == 0, r.stderr
    assert "Loaded 3 mesh(es)" in r.stdout
    got = np.fromfile(str(tmp_path / "cornell_mesh.0.bmp.f32"), np.float32).reshape(72, 96, 3)
    sc = orc.load_golden_scene("cornell_mesh").with_resolution(96, 72)
    want, _ = orc.render(sc, orc.default_config(5), 1, 2)
    assert np.array_equal(got, want)
