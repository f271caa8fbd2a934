"""-m gpu: round-2 additions -- frames above 2^24 pixels, empty shards, the owned-row exchange
(pt_get_rows / pt_gather_rows_peer) and the headline launch variant checked against the oracle at
full size.  Same bar as tests/test_gpu_parity.py: bit-exact (numpy == on float32)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import orc
from gpu_common import make_tracer, oracle_config, to_product

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("ordering", [0, 1, 2])
def test_frame_above_2pow24_pixels_matches_oracle_rows(pt, ordering):
    """4100x4100 = 16.8 Mpx: the pool's pixel word has no room for an iteration slot any more; the
    library renders one iteration per launch with the raw 32-bit pixel index."""
    sc = orc.load_golden_scene("cornell_mirror").with_resolution(4100, 4100)
    tr = make_tracer(sc, depth=4, ordering=ordering)
    tr.set_image(None)
    tr.render(1, 2)
    got = tr.image()
    st = tr.stats()
    assert st.live[0] == 2 * 4100 * 4100 and all(st.live[k] >= st.live[k + 1] for k in range(4))
    stride, off = 1025, 1019                       # rows 1019, 2044, 3069, 4094: far beyond pixel 2^24
    want, _ = orc.render(sc, oracle_config(4, row_offset=off, row_stride=stride), 1, 2)
    rows = np.arange(4100) % stride == off
    assert rows.sum() == 4 and want[rows].any()
    assert np.array_equal(got[rows], want[rows])
    tr.close()


def test_resolution_limits_are_refused_not_truncated(pt):
    sc = orc.load_golden_scene("cornell_mirror").with_resolution(4100, 4100)
    with pytest.raises(pt.PtError, match="direct_light"):
        make_tracer(sc, depth=4, direct_light=1)
    huge = orc.load_golden_scene("cornell_mirror").with_resolution(32768, 8193)     # > 2^28 pixels
    with pytest.raises(pt.PtError, match="unsupported"):
        make_tracer(huge, depth=4)


def test_shards_without_rows_are_valid_and_empty(pt):
    """More shards than rows: row_offset >= H owns nothing; rendering is a no-op and the
    remaining shards still add up to the full frame.  Also through streams > rows."""
    sc = orc.load_golden_scene("sampleScene").with_resolution(64, 5)
    want, live = orc.render(sc, oracle_config(6), 1, 3)
    total = np.zeros_like(want)
    for r in range(8):
        tr = make_tracer(sc, depth=6, row_offset=r, row_stride=8)
        tr.set_image(None)
        tr.render(1, 3)
        img = tr.image()
        if r >= 5:
            assert tr.owned == 0 and not img.any()
        total += img
        tr.close()
    assert np.array_equal(total, want)
    tr = make_tracer(sc, depth=6, streams=8)
    tr.set_image(None)
    tr.render(1, 3)
    st = tr.stats()
    assert np.array_equal(tr.image(), want)
    assert [st.live[k] for k in range(7)] == [int(v) for v in live]
    tr.close()


@pytest.mark.parametrize("streams", [1, 2])
def test_owned_row_exchange_host_and_peer(pt, streams):
    """The per-frame exchange of a row-sharded render: pt_get_rows moves only the owned rows to the host,
    pt_gather_rows_peer moves them device to device into another context's accumulator."""
    sc = orc.load_golden_scene("cornell_mirror").with_resolution(160, 91)
    want, _ = orc.render(sc, oracle_config(5), 1, 3)
    N = 3
    trs = [make_tracer(sc, depth=5, row_offset=r, row_stride=N, streams=streams) for r in range(N)]
    for tr in trs:
        tr.set_image(None)
        tr.render(1, 3)
    host = np.full((91, 160, 3), -7.0, np.float32)
    for r, tr in enumerate(trs):
        tr.get_rows(host)
        seen = np.arange(91) % N <= r
        assert np.array_equal(host[seen], want[seen]) and (host[~seen] == -7.0).all()
    assert np.array_equal(host, want)
    for tr in trs[1:]:
        trs[0].gather_rows_from(tr)
    assert np.array_equal(trs[0].image(), want)
    for tr in trs:
        tr.close()


@pytest.mark.parametrize("kw", [dict(ordering=1, streams=2), dict(ordering=0, streams=2), dict(ordering=1, streams=1),
                                dict(direct_light=1, streams=2), dict(ordering=2, streams=1), dict(ordering=2, streams=2)])
def test_headline_launch_variant_full_size_against_oracle_rows(pt, kw):
    """What bench.py times (configs[2] at 1920x1080, ordering=2 -- and its predecessors ordering=1 / 0 --, streams=2, automatic batching over a
    20-iteration launch group) against the oracle on an interleave of rows."""
    sc = orc.load_golden_scene("cornell_mirror")
    assert (sc.W, sc.H) == (1920, 1080)
    tr = make_tracer(sc, depth=8, **kw)
    tr.set_image(None)
    tr.render(1, 20)
    got = tr.image()
    st = tr.stats()
    okw = dict(direct_light=1) if kw.get("direct_light") else {}
    want, _ = orc.render(sc, oracle_config(8, row_offset=77, row_stride=216, **okw), 1, 20)
    rows = np.arange(1080) % 216 == 77
    assert rows.sum() == 5
    assert np.array_equal(got[rows], want[rows])
    assert st.live[0] == 20 * 1920 * 1080
    tr.close()


def test_bench_two_ranks_started_plainly_gloo_rehearsal(pt):
    """`python3 bench.py --gpus 2` with no launcher around it: bench.py spawns its two ranks itself (both on this
    box's one GPU, exchange over gloo) and relays rank 0's JSON line."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env["PT_BENCH_BACKEND"] = "gloo"
    dump = os.path.join(root, "gpurun_out", "rehearsal_frame.npy")
    os.makedirs(os.path.dirname(dump), exist_ok=True)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--repeats", "2",
                        "--no-cpu-baseline", "--resolution", "640x360", "--dump-image", dump], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["steps"] == 4 and res["value"] > 0 and res["scaling"] == "strong"
    # self-consistent line: frac (4 decimals) = algorithmic bytes per step / ms_per_step / 8 TB/s
    assert res["roofline"]["frac"] == pytest.approx(res["roofline"]["algorithmic_bytes_per_step"] / (res["ms_per_step"] * 1e-3) / 8e12, abs=1e-4)
    # whole-job counters (both ranks' rows: bounce 0 alone has W*H live rays), and the per-frame exchange timed on its own
    assert res["config"]["live_ray_bounces_per_step"] >= 640 * 360
    ex = res["exchange"]
    assert ex["ms"] > 0 and ex["k_steps_only_ms"] > 0 and ex["value_if_exchanged_every_pass"] < res["value"]
    assert res["ms_per_step"] * 4 == pytest.approx(ex["k_steps_only_ms"] + ex["share_charged_to_value_ms"], rel=1e-3)
    # both accountings at the top level
    assert res["value_if_exchanged_every_pass"] == ex["value_if_exchanged_every_pass"] and "exchange_accounting" in res
    # the frame rank 0 holds after the exchange = the HIP path under two processes: bit-identical to the oracle (iterations 3..6)
    got = np.load(dump)
    sc = orc.load_golden_scene("cornell_mirror").with_resolution(640, 360)
    want, _ = orc.render(sc, oracle_config(8), 3, 4)
    assert got.shape == want.shape and np.array_equal(got, want)


def _subset_scene(name, keep, w, h):
    base = orc.load_golden_scene(name).with_resolution(w, h)
    return orc.Scene([base.geoms[i] for i in keep], base.materials, base.camera)


@pytest.mark.parametrize("label,keep", [
    ("32 mixed primitives (bit 31 of the mask, many rivals per ray)", list(range(6)) + list(range(6, 110, 4))),
    ("cubes only", [i for i in range(0, 64) if i < 6 or i % 2 == 1][:32]),
    ("spheres only (no enclosing room)", [i for i in range(6, 70) if i % 2 == 0][:32]),
    ("one sphere", [6]),
])
@pytest.mark.parametrize("kw", [dict(ordering=1), dict(ordering=1, streams=2, batch=3), dict(ordering=0), dict(ordering=2), dict(ordering=2, streams=2, batch=3)])
def test_queue_kernel_on_cluttered_scenes(pt, label, keep, kw):
    """The typed work queues on scenes that are nothing like the Cornell box: up to the 32 primitives the kernel takes, one
    type missing altogether, rays with many rival candidates (the in-place extra rounds), mirrors and glass."""
    sc = _subset_scene("random256", keep, 160, 120)
    assert sc.G == len(keep) <= 32
    depth, iters = 7, 3
    tr = make_tracer(sc, depth=depth, **kw)
    tr.set_image(None)
    tr.render(1, iters)
    want, live = orc.render(sc, oracle_config(depth), 1, iters)
    st = tr.stats()
    assert [st.live[k] for k in range(depth + 1)] == [int(v) for v in live], label
    assert np.array_equal(tr.image(), want), label
    if kw.get("streams", 1) == 1:
        n, arrs, pix = tr.trace_pool(2, 4)
        on, oarrs, opix = orc.trace_pool(sc, oracle_config(depth), 2, 4)
        order = np.argsort(pix, kind="stable")
        assert n == on and np.array_equal(pix[order], opix) and all(np.array_equal(a[order], b) for a, b in zip(arrs, oarrs))
    tr.close()


@pytest.mark.parametrize("ordering", [0, 1, 2])
def test_depth_64_and_rays_that_all_miss(pt, ordering):
    """max_depth = 64 (the ABI's limit) on the mirror box, and a camera that looks away from everything"""
    sc = orc.load_golden_scene("cornell_mirror").with_resolution(64, 48)
    tr = make_tracer(sc, depth=64, ordering=ordering)
    tr.set_image(None)
    tr.render(1, 2)
    want, live = orc.render(sc, oracle_config(64), 1, 2)
    st = tr.stats()
    assert [st.live[k] for k in range(65)] == [int(v) for v in live]
    assert np.array_equal(tr.image(), want)
    tr.close()
    away = orc.load_golden_scene("sampleScene").with_resolution(37, 5)
    away.camera.view[2] = 1.0
    tr = make_tracer(away, depth=8, ordering=ordering)
    tr.set_image(None)
    tr.render(1, 3)
    st = tr.stats()
    assert st.live[0] == 3 * 37 * 5 and st.live[1] == 0 and not tr.image().any()
    tr.close()


def _stacked_scene(w, h):
    """60 primitives lined up on the view axis (48 thin slabs, 12 concentric spheres) inside random256's room: nearly every
    ray has far more than 8 culling candidates."""
    base = orc.load_golden_scene("random256").with_resolution(w, h)
    cam = base.camera
    p = np.array([cam.position[k] for k in range(3)], np.float64)
    v = np.array([cam.view[k] for k in range(3)], np.float64)
    v /= np.linalg.norm(v)
    geoms = list(base.geoms[:6])
    xf, inv = np.zeros(16, np.float32), np.zeros(16, np.float32)

    def add(kind, centre, scale, rot, mat):
        orc.lib().orc_build_transform(orc.vec3(*centre), orc.vec3(*rot), orc.vec3(*scale), orc.fptr(xf), orc.fptr(inv))
        g = orc.Geom()
        g.type, g.materialid = kind, mat
        for k in range(16):
            g.transform[k] = float(xf[k]); g.inverseTransform[k] = float(inv[k])
        geoms.append(g)

    for k in range(48):
        add(1, p + v * (2.5 + 0.2 * k), (3.0 - 0.04 * k, 3.0 - 0.04 * k, 0.03), (0.0, 0.0, 7.0 * k), (6, 6, 3, 6, 1 + k % 5)[k % 5])
    for k in range(12):
        add(0, p + v * 7.0, (0.6 + 0.45 * k,) * 3, (0.0, 0.0, 0.0), (6, 6, 4)[k % 3])
    return orc.Scene(geoms, base.materials, cam)


@pytest.mark.parametrize("kw", [dict(ordering=0), dict(ordering=1), dict(ordering=1, streams=2, batch=2), dict(ordering=2), dict(ordering=2, streams=2, batch=2),
                                dict(ordering=2, grid_density=2, chunk_rays=64)])
def test_many_primitive_kernels_when_the_candidate_list_overflows(pt, kw):
    """The many-primitive kernels keep at most 8 candidates per ray in registers; a ray with more takes the reference's
    brute-force loop.  Here nearly every ray does."""
    sc = _stacked_scene(128, 96)
    assert sc.G == 66
    depth, iters = 6, 2
    tr = make_tracer(sc, depth=depth, **kw)
    tr.set_image(None)
    tr.render(1, iters)
    want, live = orc.render(sc, oracle_config(depth), 1, iters)
    st = tr.stats()
    assert [st.live[k] for k in range(depth + 1)] == [int(v) for v in live]
    assert np.array_equal(tr.image(), want)
    tr.close()


@pytest.mark.parametrize("w,h", [(126, 50), (127, 33), (128, 40)])
def test_fold_of_batched_iterations_at_odd_widths(pt, w, h):
    """Batched iterations at frame widths that are no multiple of anything: the per-slot planes hold the owned pixels only,
    the fold adds them in iteration order -- image bit-identical to the oracle's one-iteration-after-the-other accumulation."""
    sc = orc.load_golden_scene("cornell_mirror").with_resolution(w, h)
    for kw in (dict(batch=5), dict(batch=4, streams=2, ordering=1), dict(batch=7, ordering=1, row_offset=1, row_stride=3), dict(batch=6, ordering=2, row_offset=2, row_stride=3)):
        tr = make_tracer(sc, depth=5, **kw)
        tr.set_image(None)
        tr.render(1, 9)
        okw = {k: v for k, v in kw.items() if k in ("row_offset", "row_stride")}
        want, _ = orc.render(sc, oracle_config(5, **okw), 1, 9)
        assert np.array_equal(tr.image(), want), (w, h, kw)
        tr.close()


@pytest.mark.parametrize("job", [0, 64, 128, 1024])
def test_whole_path_kernel_job_hand_out_whatever_the_job_size(pt, job):
    """ordering = 2: the jobs of camera rays are half the waves' own contiguous ranges, half drawn from sixteen ticket
    counters.  Whatever the job size (pt_config.chunk_rays; 0 = by launch size), the launch-group size and the grid --
    jobs that do not divide the frame, more waves than jobs, one block per CU -- every camera ray is rendered exactly
    once: image and live counts are the oracle's."""
    sc = orc.load_golden_scene("cornell_mirror").with_resolution(213, 117)
    for kw in (dict(ordering=2, batch=4), dict(ordering=2, streams=2, batch=3), dict(ordering=2, batch=1, blocks_per_cu=1)):
        tr = make_tracer(sc, depth=6, chunk_rays=job, **kw)
        tr.set_image(None)
        tr.render(1, 7)
        want, live = orc.render(sc, oracle_config(6), 1, 7)
        st = tr.stats()
        assert [st.live[k] for k in range(7)] == [int(v) for v in live], (job, kw)
        assert np.array_equal(tr.image(), want), (job, kw)
        tr.close()


def test_whole_path_kernel_equals_the_stable_kernel_on_full_frames(pt):
    """64 iterations at 1920x1080 and 24 at 3840x2160 (glass, thin lens, jitter, 16 bounces): the whole-path kernel
    (ordering = 2, two streams, automatic batching) and the stable per-bounce kernel (ordering = 0) produce the same
    float image bit for bit, the same live counts and the same number of emitter hits."""
    for name, depth, iters, extra in (("cornell_mirror", 8, 64, {}),
                                      ("cornell_glass_4k", 16, 24, dict(camera_mode=1, antialias=1, aperture=0.25, focal_distance=12.0))):
        sc = orc.load_golden_scene(name)
        a = make_tracer(sc, depth=depth, ordering=0, streams=2, **extra)
        a.set_image(None); a.render(1, iters)
        ia, sa = a.image(), a.stats()
        a.close()
        b = make_tracer(sc, depth=depth, ordering=2, streams=2, **extra)
        b.set_image(None); b.render(1, iters)
        ib, sb = b.image(), b.stats()
        b.close()
        assert np.array_equal(ia, ib), name
        assert [sa.live[k] for k in range(depth + 1)] == [sb.live[k] for k in range(depth + 1)] and sa.emitted == sb.emitted, name


@pytest.mark.parametrize("depth", [1, 2, 3])
def test_whole_path_kernel_at_the_smallest_depths(pt, depth):
    """max_depth 1 (every ray is at its last level from the start: nothing ever goes on the stack), 2 and 3."""
    sc = orc.load_golden_scene("sampleScene").with_resolution(150, 100)
    for kw in (dict(ordering=2), dict(ordering=2, streams=2, batch=3)):
        tr = make_tracer(sc, depth=depth, **kw)
        tr.set_image(None)
        tr.render(1, 4)
        want, live = orc.render(sc, oracle_config(depth), 1, 4)
        st = tr.stats()
        assert [st.live[k] for k in range(depth + 1)] == [int(v) for v in live], (depth, kw)
        assert np.array_equal(tr.image(), want), (depth, kw)
        tr.close()
