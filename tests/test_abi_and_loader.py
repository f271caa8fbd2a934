"""CPU-side checks of the product library: it loads, exports every symbol include/ptmi355.h
declares, refuses to render without a GPU (no fallback), and its host-side scene loader /
transform builder / image writer agree with the REFERENCE's own code (tests/golden/ref_*.json,
produced by oracle/_ref = reference scene.cpp + utilities.cpp + image.cpp compiled where they lie)."""
import ctypes as C
import json
import os
import re
import struct

import numpy as np
import pytest

import orc
from conftest import ROOT, load_package


@pytest.fixture(scope="module")
def pkg(pt):
    return pt


def test_library_exports_every_declared_symbol(pkg):
    hdr = open(os.path.join(ROOT, "include", "ptmi355.h")).read()
    declared = sorted(set(re.findall(r"\b(pt_[a-z0-9_]+)\s*\(", hdr)))
    assert len(declared) >= 30
    L = C.CDLL(pkg.LIB_PATH)
    for name in declared:
        assert hasattr(L, name), "libptmi355.so does not export %s" % name
    assert sorted(pkg.EXPORTS) == declared
    version = int(re.search(r"#define\s+PTMI355_ABI_VERSION\s+(\d+)", hdr).group(1))
    assert pkg.lib().pt_abi_version() == version == 9


def test_pod_sizes_match_reference_structs(pkg):
    sizes = json.load(open(os.path.join(orc.GOLD, "ref_scene_sampleScene.json")))["sizeof"]
    assert C.sizeof(pkg.Material) == sizes["material"] == 64
    assert C.sizeof(pkg.Camera) == sizes["cameraData"] == 52
    assert C.sizeof(pkg.Geom) == 104
    assert sizes["staticGeom"] == 172 and sizes["cudaMat4"] == 64 and sizes["ray"] == 24


def test_no_cpu_fallback(pkg):
    """Without a gfx950 device the product must fail loudly, not render on the CPU."""
    if pkg.lib().pt_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(pkg.PtError) as e:
        pkg.PathTracer()
    assert "no HIP device" in str(e.value)
    assert pkg.lib().pt_render(None, 1, 1) != 0


def test_bad_options_are_refused_before_any_device_is_touched_and_the_default_is_the_fast_path(pkg):
    """Out-of-range options are an argument error -- with or without a GPU.  ABI 9: the fields that had to be zero and the A/B
    switches whose best setting is built in are gone (18 fields), and pt_config_default gives what bench.py, the adaptor
    and ptrender run: whole paths in one launch per group (ordering = 2) on two streams per GPU."""
    import ctypes as C
    for kw in (dict(grid_density=65), dict(grid_density=-1), dict(max_depth=0), dict(max_depth=65), dict(row_stride=0), dict(streams=0)):
        cfg = pkg.default_config(**kw)
        h = C.c_void_p()
        assert pkg.lib().pt_create(C.byref(cfg), C.byref(h)) == -3, kw          # PT_ERR_ARGUMENT
        assert not h.value
    d = pkg.default_config()
    assert (d.ordering, d.streams, d.max_depth, d.row_stride, d.grid_density, d.batch) == (2, 2, 8, 1, 0, 0)
    assert C.sizeof(pkg.Config) == 18 * 4
    for gone in ("geometry_path", "compaction", "merge_floor", "bvh", "cluster_size", "path_static_eighths", "wide_variant"):
        assert not hasattr(d, gone)
    header = open(os.path.join(ROOT, "include", "ptmi355.h")).read()
    body = header[header.index("typedef struct {\n    int   device;"):header.index("} pt_config;")]
    fields = re.findall(r"^    (?:int|float)\s+(\w+);", body, re.M)
    assert fields == [f for f, _ in pkg.Config._fields_], fields              # the ctypes mirror follows the header, field for field


def test_library_reads_no_environment_switches():
    """Behaviour switches travel in pt_config (ABI 7), not in the process environment (the adaptor's PT_* variables are
    the reference API's missing option channel and live in adaptor/ only)."""
    csrc = os.path.join(ROOT, "project2-pathtracer_amd", "csrc")
    for f in os.listdir(csrc):
        assert "getenv" not in open(os.path.join(csrc, f), errors="replace").read(), f


def test_product_tree_never_references_the_oracle():
    for base, _, files in os.walk(os.path.join(ROOT, "project2-pathtracer_amd")):
        for f in files:
            if f.endswith((".hip", ".hpp", ".cpp", ".h", ".py", "Makefile")):
                text = open(os.path.join(base, f), errors="replace").read()
                assert "pt_oracle" not in text and "libpt_oracle" not in text and "oracle/" not in text, f
    hdr = open(os.path.join(ROOT, "include", "ptmi355.h")).read()
    assert "oracle" not in hdr


SCENES = ["cornell", "cornell_c1", "cornell_mirror", "cornell_glass_4k", "random256", "random1024"]


@pytest.mark.parametrize("name", SCENES)
def test_scene_loader_matches_reference_parser(pkg, name):
    gold = json.load(open(os.path.join(orc.GOLD, "ref_scene_%s.json" % name)))
    sf = pkg.SceneFile(os.path.join(ROOT, "scenes", name + ".txt"))
    assert sf.ngeoms == len(gold["objects"]) and sf.nmaterials == len(gold["materials"])
    assert sf.nframes == gold["camera"]["frames"] and sf.iterations == gold["camera"]["iterations"]
    assert sf.image_name == gold["camera"]["imageName"]
    for frame in range(sf.nframes):
        geoms, mats, cam = sf.flatten(frame)
        for i, m in enumerate(gold["materials"]):
            assert bytes(mats[i]) == struct.pack("<16I", *m)
        for i, o in enumerate(gold["objects"]):
            fr = o["frames"][frame]
            assert geoms[i].type == o["type"] and geoms[i].materialid == o["materialid"]
            xf, inv = sf.object_matrices(i, frame)
            assert np.array_equal(xf, np.array(fr["transform"], np.uint32).view(np.float32))
            assert np.array_equal(inv, np.array(fr["inverseTransform"], np.uint32).view(np.float32))
            assert np.array_equal(np.array(list(geoms[i].transform), np.float32), xf[:12])
            assert np.array_equal(np.array(list(geoms[i].inverseTransform), np.float32), inv[:12])
            # the w row the C ABI drops is (0,0,0,1) for every matrix the parser builds
            assert list(xf[12:]) == [0, 0, 0, 1] and list(inv[12:]) == [0, 0, 0, 1]
        c = gold["camera"]
        assert [orc.bits_from_f32(v) for v in cam.resolution] == c["resolution"]
        assert [orc.bits_from_f32(v) for v in cam.fov] == c["fov"]
        assert [orc.bits_from_f32(v) for v in cam.position] == c["positions"][frame]
        assert [orc.bits_from_f32(v) for v in cam.view] == c["views"][frame]
        assert [orc.bits_from_f32(v) for v in cam.up] == c["ups"][frame]


def test_our_cornell_equals_reference_sample_scene():
    a = json.load(open(os.path.join(orc.GOLD, "ref_scene_sampleScene.json")))
    b = json.load(open(os.path.join(orc.GOLD, "ref_scene_cornell.json")))
    for k in ("materials", "objects", "camera", "sizeof"):
        assert a[k] == b[k]


def test_build_transform_matches_reference(pkg):
    T = json.load(open(os.path.join(orc.GOLD, "ref_transforms.json")))
    for case in T["cases"]:
        xf, inv = pkg.build_transform(case["t"], case["r"], case["s"])
        assert np.array_equal(xf, np.array(case["transform"], np.uint32).view(np.float32)), case
        assert np.array_equal(inv, np.array(case["inverseTransform"], np.uint32).view(np.float32)), case


def test_loader_error_behaviour(pkg, tmp_path):
    with pytest.raises(pkg.PtError):
        pkg.SceneFile(str(tmp_path / "missing.txt"))
    good = open(os.path.join(ROOT, "scenes", "cornell_c1.txt")).read()
    bad_id = tmp_path / "bad_id.txt"
    bad_id.write_text(good.replace("MATERIAL 3", "MATERIAL 7", 1))
    with pytest.raises(pkg.PtError) as e:
        pkg.SceneFile(str(bad_id))
    assert "MATERIAL ID does not match" in str(e.value)       # the reference's message (scene.cpp:223)
    bad_type = tmp_path / "bad_type.txt"
    bad_type.write_text(good.replace("\nsphere\n", "\nsphere \n", 1))   # whole-line strcmp (scene.cpp:48)
    with pytest.raises(pkg.PtError) as e:
        pkg.SceneFile(str(bad_type))
    assert "is not a valid object type" in str(e.value)
    bad_frame = tmp_path / "bad_frame.txt"
    bad_frame.write_text(good.replace("frame 1", "frame 5", 1))
    with pytest.raises(pkg.PtError) as e:
        pkg.SceneFile(str(bad_frame))
    assert "Incorrect frame count" in str(e.value)
    # trailing //comments are ignored tokens, keys may come in any order, CRLF breaks only type lines
    commented = tmp_path / "commented.txt"
    commented.write_text(good.replace("MATERIAL 0\n", "MATERIAL 0\t\t//white diffuse\n", 1))
    sf = pkg.SceneFile(str(commented))
    assert sf.nmaterials == 9


def test_image_writer_matches_reference_image_class(pkg, tmp_path):
    meta = json.load(open(os.path.join(orc.GOLD, "ref_image_meta.json")))
    W, H = meta["W"], meta["H"]
    src = np.fromfile(os.path.join(orc.GOLD, "ref_image_in.f32"), np.float32).reshape(H, W, 3)
    out = tmp_path / "out.bmp"
    pkg.image_save(str(out), src, meta["divisor"], np.float32(meta["gamma"]))
    assert out.read_bytes() == open(os.path.join(orc.GOLD, "ref_image_out.bmp"), "rb").read()   # byte-exact BMP
    from PIL import Image
    png = tmp_path / "out.png"
    pkg.image_save(str(png), src, meta["divisor"], np.float32(meta["gamma"]))
    ours = np.asarray(Image.open(str(png)).convert("RGB"))
    ref = np.asarray(Image.open(os.path.join(orc.GOLD, "ref_image_out.png")).convert("RGB"))
    assert np.array_equal(ours, ref)                    # same pixels (our PNG stores, stb deflates)
    assert np.array_equal(pkg.image_to_u8(src, meta["divisor"], np.float32(meta["gamma"])), ref)


def test_tokenizer_cases_from_reference(pkg):
    # the loader's tokenisation is whitespace splitting exactly like utilityCore::tokenizeString
    T = json.load(open(os.path.join(orc.GOLD, "ref_tokens.json")))
    for case in T["cases"]:
        assert case["line"].split() == case["tokens"]


def test_owned_index_magic_division_is_exact():
    """csrc/pt_scene.cpp camera_basis / csrc/pt_device.hpp owned_index: q = (n * m) >> sh with m = floor(2^sh / d) + 1,
    sh = 28 + ceil(log2 d) must equal n // d for every n < 2^28 (pixel indices) -- checked here on the formula itself
    at the edges of every quotient step for a spread of divisors, plus random n."""
    import random
    rnd = random.Random(7)
    for d in [1, 2, 3, 5, 7, 8, 64, 100, 640, 799, 800, 1000, 1920, 2048, 3840, 4095, 4096, 4100, 16384, 32767, 32768]:
        L = 0
        while (1 << L) < d:
            L += 1
        sh = 28 + L
        m = (1 << sh) // d + 1
        assert m < (1 << 32)
        ns = [rnd.randrange(1 << 28) for _ in range(20000)] + [(1 << 28) - 1, 0]
        for q in [rnd.randrange((1 << 28) // d) for _ in range(2000)] + [(1 << 28) // d - 1]:
            ns += [q * d, q * d + d - 1, max(0, q * d - 1)]
        for n in ns:
            if n < (1 << 28):
                assert (n * m) >> sh == n // d, (n, d)
