"""CPU: oracle/pt_oracle.c against the REFERENCE's own intersections.h / interactions.h.

tests/golden/ref_kernels.npz holds seeded inputs and the outputs of the reference functions compiled
unchanged as host code (oracle/ref_kernels_probe.cpp, recipe in oracle/Makefile, generator
oracle/gen_golden_kernels.py).  This pins rows (a)1, (a)6-(a)9 of SURVEY.md section 8 -- hash, multiplyMV,
getPointOnRay, the sphere and box tests, getRadiuses, getRandomPointOnCube/Sphere and
calculateRandomDirectionInHemisphere -- on the reference's code itself rather than on transcribed KATs.

Bar: bit patterns (numpy == on the uint32 views, so even the sign of zero and NaN payload classes count),
with two stated exceptions:
  * sphere test: the probe is a C++11 host build, where pow(radius,2) is the double overload; the oracle's
    pow-in-double variant must match bit for bit, the oracle proper (CUDA's float overload) to within the
    effect of that single rounding;
  * hemisphere sampler: cos/sin are libm's cosf/sinf in the reference's host build and a shared polynomial
    (<= 2e-7 off, tests/test_oracle_kats.py) in the oracle and the kernels: compared to 5e-7 absolute.
When /root/reference is present the fixture is also regenerated in memory and must equal the committed one.
"""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import orc
from conftest import ROOT, has_reference

G = np.load(os.path.join(orc.GOLD, "ref_kernels.npz"), allow_pickle=False)


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def same_bits(a, b):
    """bit-equal, treating every NaN as equal to every NaN (x86 and the oracle may differ in NaN sign)"""
    a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)
    return bool(np.all((bits(a) == bits(b)) | (np.isnan(a) & np.isnan(b))))


def geom_from(xf, inv, kind):
    g = orc.Geom()
    g.type, g.materialid = kind, 0
    for k in range(16):
        g.transform[k] = float(xf[k]); g.inverseTransform[k] = float(inv[k])
    return g


def f3(a):
    return (C.c_float * len(a))(*[float(v) for v in a])


def test_hash_matches_reference():
    L = orc.lib()
    got = np.array([L.orc_hash(int(v)) for v in G["hash_in"]], np.uint32)
    assert np.array_equal(got, G["hash_out"])


def test_multiply_mv_and_point_on_ray_match_reference():
    L = orc.lib()
    out = (C.c_float * 3)()
    got = np.zeros_like(G["multiplymv_out"])
    for i, r in enumerate(G["multiplymv_in"]):
        L.orc_multiply_mv(f3(r[:16]), f3(r[16:20]), out)
        got[i] = list(out)
    assert same_bits(got, G["multiplymv_out"])
    got = np.zeros_like(G["pointonray_out"])
    for i, r in enumerate(G["pointonray_in"]):
        L.orc_point_on_ray(f3(r[0:3]), f3(r[3:6]), float(r[6]), out)
        got[i] = list(out)
    assert same_bits(got, G["pointonray_out"])


def run_test(fn, rec, kind, *extra):
    P, N = (C.c_float * 3)(), (C.c_float * 3)()
    got = np.zeros((len(rec), 7), np.float32)
    for i, r in enumerate(rec):
        g = geom_from(r[:16], r[16:32], kind)
        P[:] = [0, 0, 0]; N[:] = [0, 0, 0]
        t = fn(C.byref(g), *extra, f3(r[32:35]), f3(r[35:38]), P, N)
        got[i] = [t] + list(P) + list(N)
    return got


def test_box_test_matches_reference_bit_for_bit():
    want = G["box_out"]
    got = run_test(orc.lib().orc_box_test, G["box_in"], 1, 0)
    hit = want[:, 0] >= 0
    assert hit.sum() > 1500 and (~hit).sum() > 1000
    assert same_bits(got[:, 0], want[:, 0])                     # -1 or the world distance, every record
    assert same_bits(got[hit], want[hit])                       # P and N of every hit


def test_sphere_test_matches_reference():
    want = G["sphere_out"]
    hit = want[:, 0] >= 0
    assert hit.sum() > 2000 and (~hit).sum() > 1000
    # (1) same overload set as the probe build (pow in double): bit for bit
    got_d = run_test(orc.lib().orc_sphere_test_powdouble, G["sphere_in"], 0)
    assert same_bits(got_d[:, 0], want[:, 0])
    assert same_bits(got_d[hit], want[hit])
    # (2) the oracle proper (CUDA's float pow): the radicand may differ by one rounding.  Hit/miss may flip
    # only where the radicand is at the rounding edge of zero (grazing rays); everything else stays within
    # a few ulps of t, P, N.
    got = run_test(orc.lib().orc_sphere_test, G["sphere_in"], 0)
    flips = (got[:, 0] >= 0) != hit
    assert flips.sum() <= 2
    both = hit & ~flips
    same = np.all(bits(got[both]) == bits(want[both]), axis=1)
    assert same.mean() > 0.90                                    # the single rounding rarely shows at all
    err = np.abs(got[both].astype(np.float64) - want[both].astype(np.float64))
    scale = np.maximum(1.0, np.abs(want[both].astype(np.float64)))
    # near-grazing hits amplify the radicand's rounding through sqrt: relative 2e-4 covers them, typical is 1e-7
    assert (err / scale).max() < 2e-4 and np.median(err / scale) < 1e-7


def test_radiuses_and_light_samplers_match_reference():
    L = orc.lib()
    out = (C.c_float * 3)()
    got = np.zeros_like(G["radiuses_out"])
    for i, xf in enumerate(G["radiuses_in"]):
        g = geom_from(xf, xf, 1)
        L.orc_get_radiuses(C.byref(g), out)
        got[i] = list(out)
    assert same_bits(got, G["radiuses_out"])
    for name, fn, kind in (("cubepoint", L.orc_random_point_on_cube, 1), ("spherepoint", L.orc_random_point_on_sphere, 0)):
        rec, want = G[name + "_in"], G[name + "_out"]
        got = np.zeros_like(want)
        for i, r in enumerate(rec):
            g = geom_from(r[:16], r[:16], kind)
            fn(C.byref(g), float(r[16]), out)
            got[i] = list(out)
        assert same_bits(got, want), name


def test_hemisphere_sampler_matches_reference():
    L = orc.lib()
    out = (C.c_float * 3)()
    rec, want = G["hemisphere_in"], G["hemisphere_out"]
    got = np.zeros_like(want)
    for i, r in enumerate(rec):
        L.orc_hemisphere(f3(r[0:3]), float(r[3]), float(r[4]), out)
        got[i] = list(out)
    # a non-unit normal along x below sqrt(1/3) (the reference's own box normals of thin walls, e.g. (0.01,0,0))
    # makes cross(normal, (1,0,0)) = 0 and normalize(0) = NaN in the reference: the oracle must do the same
    fin = np.isfinite(want).all(axis=1)
    assert (~fin).sum() >= 1 and fin.sum() > 4000
    assert np.array_equal(np.isfinite(got).all(axis=1), fin)
    # everything but cos/sin is shared arithmetic: the polynomial's 2e-7 times |over * tangent| <= 1
    assert np.abs(got[fin].astype(np.float64) - want[fin].astype(np.float64)).max() < 5e-7
    # where the azimuth is exactly representable in both (xi2 = 0.5 -> cos = -1 up to the libm's own last bit)
    assert ["%.9g" % v for v in want[0]] == ["7.57103464e-08", "0.5", "0.866025388"]      # the SURVEY's KAT


@pytest.mark.skipif(not has_reference(), reason="reference sources only exist in the build container")
def test_committed_fixture_is_what_the_reference_build_produces(tmp_path):
    """Regenerates a sample of the fixture with the freshly built probe: the committed file is not stale."""
    probe = os.path.join(ROOT, "oracle", "_ref", "ref_kernels_probe")
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "_ref/ref_kernels_probe"], check=True)
    for cmd, words in (("box", 7), ("sphere", 7), ("hemisphere", 3), ("cubepoint", 3)):
        rec = np.ascontiguousarray(G[cmd + "_in"][:512])
        out = subprocess.run([probe, cmd], input=rec.tobytes(), check=True, capture_output=True).stdout
        got = np.frombuffer(out, np.float32).reshape(len(rec), words)
        assert same_bits(got, G[cmd + "_out"][:512]), cmd
