"""-m gpu: round-3 additions -- the whole-path kernel for 33..256 primitives (k_path_w), the device-side guards of the
whole-path kernels surfaced through pt_sync, the roofline block of the bench line, and the MESH bench workloads.
Same bar as tests/test_gpu_parity.py: bit-exact (numpy == on float32)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import orc
from gpu_common import make_tracer, oracle_config, to_product

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _subset(name, keep, w, h):
    base = orc.load_golden_scene(name).with_resolution(w, h)
    return orc.Scene([base.geoms[i] for i in keep], base.materials, base.camera)


@pytest.mark.parametrize("label,keep", [
    ("33 primitives: the smallest scene the kernel takes", list(range(33))),
    ("cubes only (no sphere cluster at all)", [i for i in range(256) if i < 6 or i % 2 == 1][:100]),
    ("spheres only, no enclosing room: most rays leave without a candidate", [i for i in range(6, 256) if i % 2 == 0][:90]),
    ("all 256", list(range(256))),
])
@pytest.mark.parametrize("kw", [dict(), dict(grid_density=2, batch=3), dict(grid_density=12, streams=2), dict(chunk_rays=64)])
def test_whole_path_wide_kernel_on_subsets_of_config4(pt, label, keep, kw):
    """k_path_w (ordering = 2, 33..256 analytic primitives) against the oracle: image, live counts, emitter hits; and
    against the stable kernel's ray pool of the same context (the parity hook runs the per-bounce WIDE kernel)."""
    sc = _subset("random256", keep, 200, 112)
    depth, iters = 8, 3
    tr = make_tracer(sc, depth=depth, ordering=2, **kw)
    tr.set_image(None)
    tr.render(1, iters)
    want, live = orc.render(sc, oracle_config(depth), 1, iters)
    st = tr.stats()
    assert [st.live[k] for k in range(depth + 1)] == [int(v) for v in live], label
    assert np.array_equal(tr.image(), want), label
    if kw.get("streams", 1) == 1:
        n, arrs, pix = tr.trace_pool(2, 3)
        on, oarrs, opix = orc.trace_pool(sc, oracle_config(depth), 2, 3)
        assert n == on and np.array_equal(pix, opix) and all(np.array_equal(a, b) for a, b in zip(arrs, oarrs))
    tr.close()


def test_whole_path_wide_kernel_continues_a_host_image_and_long_groups(pt):
    """accumulation continues from an uploaded image; a 40-iteration launch group uses every slot bit"""
    sc = orc.load_golden_scene("random256").with_resolution(96, 54)
    want, _ = orc.render(sc, oracle_config(8), 1, 44)
    part, _ = orc.render(sc, oracle_config(8), 1, 4)
    tr = make_tracer(sc, depth=8, ordering=2)
    tr.set_image(part)
    tr.render(5, 40)
    assert np.array_equal(tr.image(), want)
    tr.close()


@pytest.mark.parametrize("name,kw", [("cornell_mirror", dict()), ("random256", dict()), ("random256", dict(grid_density=1)), ("cornell_mirror", dict(streams=2))])
def test_turn_limit_guard_surfaces_through_sync(pt, name, kw):
    """The whole-path kernels bound the scheduling turns of a wave (a broken build must end, not hang the device).  With
    the limit lowered to a few turns the guard trips: the launch ends at once and pt_sync reports PT_ERR_HIP."""
    sc = orc.load_golden_scene(name).with_resolution(160, 90)
    tr = make_tracer(sc, depth=8, ordering=2, **kw)
    tr.set_image(None)
    tr.set_turn_limit(3)
    tr.render(1, 2)
    with pytest.raises(pt.PtError, match="turn limit"):
        tr.sync()
    tr.close()
    # an untouched context right afterwards renders correctly (the guard left the device usable)
    tr = make_tracer(sc, depth=8, ordering=2, **kw)
    tr.set_image(None)
    tr.render(1, 2)
    want, _ = orc.render(sc, oracle_config(8), 1, 2)
    assert np.array_equal(tr.image(), want)
    tr.close()


def test_stack_overflow_guard_surfaces_through_sync(pt):
    """A build whose per-wave stack of survivors is too small for the depth-first bound (-DPT_STACK_SLOTS=64 instead of
    256: the bound is 63 + two pops of 64) must report 'a level ring overflowed' through pt_sync, not write past the
    stack.  The variant is built here (hipcc is on the box) and loaded beside the shipped library."""
    import ctypes as C
    out = os.path.join(ROOT, "project2-pathtracer_amd", "build", "variants", "smallstack.so")
    r = subprocess.run(["bash", os.path.join(ROOT, "tools", "build_variant.sh"), "smallstack", "-DPT_STACK_SLOTS=64"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and os.path.exists(out), (r.stdout + r.stderr)[-2000:]
    L = C.CDLL(out)
    L.pt_last_error.restype = C.c_char_p
    sc = orc.load_golden_scene("cornell_mirror").with_resolution(320, 180)
    geoms, mats, cam = to_product(sc)
    cfg = pt.default_config(max_depth=8, ordering=2, streams=1)        # (one stream: the whole frame's population on every wave)
    h = C.c_void_p()
    assert L.pt_create(C.byref(cfg), C.byref(h)) == 0
    assert L.pt_upload_scene(h, geoms, len(geoms), mats, len(mats), C.byref(cam)) == 0
    assert L.pt_set_image(h, None) == 0
    assert L.pt_render(h, 1, 8) == 0
    rc = L.pt_sync(h)
    assert rc == -2 and b"overflowed" in L.pt_last_error(), (rc, L.pt_last_error())
    L.pt_destroy(h)


def test_bench_line_states_what_bounds_the_kernel(pt):
    """The roofline block of bench.py's JSON line: `bound` is what the counters say, the survey-byte figure
    is labelled as such, the PMC traffic is per STEP with the per-launch figure beside it, and every derived number is
    consistent with the ones it is derived from."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "8", "--warmup", "4", "--repeats", "3", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    res = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    roof = res["roofline"]
    assert roof["bound"] in ("valu-issue", "hbm", "unmeasured") and roof["stated_roofline"].startswith("hbm")
    assert roof["frac"] == pytest.approx(roof["algorithmic_bytes_per_step"] / (res["ms_per_step"] * 1e-3) / 8e12, rel=2e-3)      # ms_per_step carries 4 decimals
    assert roof["frac"] == pytest.approx(roof["achieved"] / roof["peak"], abs=1e-4)
    assert roof["algorithmic_bytes_per_launch"] == pytest.approx(roof["algorithmic_bytes_per_step"] * res["steps"] / roof["launches"], rel=1e-6)
    src = roof["traffic_source"]
    assert src is not None and src["file"] == "profiles/traffic_latest.json"
    if src["stale"]:
        assert roof["traffic"] is None and roof["hbm_measured"] is None and roof["valu_issue"] is None
        assert roof["bound"] == "unmeasured"          # no counters of these kernel sources: no limiter claimed
    else:
        hm, vi = roof["hbm_measured"], roof["valu_issue"]
        assert roof["traffic"] == hm["bytes_per_step"]
        assert roof["traffic_per_launch"] == pytest.approx(roof["traffic"] * res["steps"] / roof["launches"], rel=1e-6)
        assert hm["GB_s"] == pytest.approx(hm["bytes_per_step"] / (res["ms_per_step"] * 1e-3) / 1e9, rel=2e-3)
        assert hm["frac_of_peak"] == pytest.approx(hm["GB_s"] / 8000.0, abs=1e-4)
        assert vi["frac"] == pytest.approx(vi["achieved_G_wave_inst_per_s"] / 1228.8, abs=1e-3) and "frac_of_4_cycle_issue" not in vi
        lo, hi = vi["measured_ceiling_ns_per_wave_instruction_per_simd"]
        assert vi["frac_of_measured_ceiling"] == [pytest.approx(lo / vi["ns_per_wave_instruction_per_simd"], abs=2e-3), pytest.approx(hi / vi["ns_per_wave_instruction_per_simd"], abs=2e-3)]
        assert roof["bound"] == ("hbm" if hm["frac_of_peak"] > vi["frac"] else "valu-issue")


@pytest.mark.parametrize("workload", ["mesh", "mesh5k"])
def test_mesh_bench_workloads_render_what_the_oracle_renders(pt, workload):
    """bench.py --workload mesh / mesh5k (scenes/cornell_mesh*.txt at 1920x1080 there): the same scene files through the
    library's loader at a small resolution, against the oracle's brute-force triangle loop."""
    sys.path.insert(0, ROOT)
    import bench
    path = os.path.join(ROOT, bench.WORKLOADS[workload][0])
    sf = pt.SceneFile(path)
    geoms, mats, cam = sf.flatten(0)
    cam.resolution[0], cam.resolution[1] = 128.0, 96.0
    meshes = sf.meshes()
    assert meshes and (workload != "mesh5k" or len(meshes[0][2]) == 5120)
    sc = orc.Scene.from_product(geoms, mats, cam, meshes)
    for kw in (dict(ordering=2), dict(ordering=0)):
        tr = pt.PathTracer(pt.default_config(max_depth=6, **kw))
        tr.set_meshes(meshes)
        tr.upload(geoms, mats, cam)
        tr.set_image(None)
        tr.render(1, 2)
        want, live = orc.render(sc, oracle_config(6), 1, 2)
        st = tr.stats()
        assert [st.live[k] for k in range(7)] == [int(v) for v in live], kw
        assert np.array_equal(tr.image(), want), kw
        tr.close()


@pytest.mark.parametrize("name,w,h,bounces", [("cornell_mirror", 200, 150, 1), ("cornell_mirror", 200, 150, 4), ("cornell_mirror", 200, 150, 7),
                                              ("cornell_glass_4k", 160, 90, 3), ("random256", 160, 120, 1), ("random256", 160, 120, 3), ("random256", 160, 120, 6)])
def test_whole_path_kernels_ray_records_match_the_oracle_pool(pt, name, w, h, bounces):
    """with ordering = 2 the pool hook probes the WHOLE-PATH kernels themselves -- the rays that survive
    bounce k - 1 leave k_path_q / k_path_w through a tap (in whatever order the waves meet them), the host sorts them by pixel --
    so origin, direction, throughput and pixel word of every live ray are compared with the oracle's pool, not only images."""
    sc = orc.load_golden_scene(name).with_resolution(w, h)
    tr = make_tracer(sc, depth=8, ordering=2)
    n, arrs, pix = tr.trace_pool(2, bounces)
    on, oarrs, opix = orc.trace_pool(sc, oracle_config(8), 2, bounces)
    assert n == on and n > 0
    assert np.array_equal(pix, opix)
    for a, b in zip(arrs, oarrs):
        assert np.array_equal(a, b)
    # the hook leaves image and statistics untouched, and the stable kernels' pool is the same
    tr.set_image(None); tr.render(1, 2)
    want, live = orc.render(sc, oracle_config(8), 1, 2)
    assert np.array_equal(tr.image(), want)
    assert [int(tr.stats().live[k]) for k in range(9)] == [int(v) for v in live]
    tr.close()


@pytest.mark.parametrize("ranks,workload,res,scene", [(4, "c3", "640x360", "cornell_mirror"), (3, "c4", "320x182", "random256")])
def test_bench_more_ranks_gloo_rehearsal_frames_match_the_oracle(pt, ranks, workload, res, scene):
    """The N > 1 path of bench.py with more ranks than the two of tests/test_gpu_round2.py, all on this box's one GPU over gloo
    (at most five processes on the card at once): rows interleaved over 4 ranks on k_path_q, and over 3 ranks with an uneven
    row count on k_path_w -- the frame rank 0 gathers is bit-identical to the oracle's."""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env["PT_BENCH_BACKEND"] = "gloo"
    dump = os.path.join(root, "gpurun_out", "rehearsal_frame_%d.npy" % ranks)
    os.makedirs(os.path.dirname(dump), exist_ok=True)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(ranks), "--steps", "4", "--warmup", "2", "--repeats", "2",
                        "--workload", workload, "--no-cpu-baseline", "--resolution", res, "--dump-image", dump],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == ranks and out["scaling"] == "strong" and out["value"] > 0
    w, h = [int(x) for x in res.split("x")]
    got = np.load(dump)
    want, _ = orc.render(orc.load_golden_scene(scene).with_resolution(w, h), oracle_config(8), 3, 4)
    assert got.shape == want.shape and np.array_equal(got, want)


def test_whole_path_kernel_shape_follows_the_scene(pt):
    """pt_upload_scene sizes the whole-path kernels to the scene: k_path_q takes the largest instantiated queue capacity that still
    leaves five blocks per CU beside the scene's tables, k_path_w one block per CU of as many waves as fit --
    reported by pt_debug_path_shape (and in bench.py's config)."""
    def shape(name, keep=None, **kw):
        sc = orc.load_golden_scene(name).with_resolution(64, 48)
        if keep is not None:
            sc = orc.Scene(sc.geoms[:keep], sc.materials, sc.camera)
        tr = make_tracer(sc, depth=4, **kw)
        out = tr.path_shape()
        tr.close()
        return out
    a = shape("cornell_mirror", ordering=2)
    assert a["family"] == "k_path_q" and a["blocks_per_cu"] == 5 and a["records_per_wave"] == 144 and a["waves_per_block"] == 4
    assert 5 * a["lds_bytes"] <= 160 * 1024 < 5 * (a["lds_bytes"] + 4 * 8 * 48 + 1279) // 1280 * 1280     # eight records more would cost the fifth block (LDS granule)
    b = shape("random256", keep=32, ordering=2)                                               # 32 primitives: bigger tables, fewer records, still five blocks
    assert b["family"] == "k_path_q" and b["blocks_per_cu"] == 5 and b["records_per_wave"] < 144
    c = shape("random256", ordering=2)
    assert c["family"] == "k_path_w" and c["blocks_per_cu"] == 1 and c["waves_per_block"] == 16 and c["records_per_wave"] == 112
    d = shape("cornell_mesh", ordering=2)
    assert d["family"] == "k_path_q" and d["meshes"] and d["blocks_per_cu"] == 5 and d["records_per_wave"] < 144    # (the mesh stages' scratch takes LDS)
    e = shape("cornell_mirror", ordering=2, direct_light=1)
    assert e["family"] == "k_path_q" and e["direct_light"] and e["blocks_per_cu"] == 5
    assert shape("cornell_mirror", ordering=0)["family"] == "per-bounce"



def test_exchange_code_runs_under_rccl_with_one_rank(pt, tmp_path):
    """bench.py's per-frame exchange (RowGather on device tensors, reduce, all-reduce, barrier) under backend "nccl" = RCCL with a
    world of ONE rank on this box's GPU: communicator creation and the device-tensor path execute, and the gathered frame is the
    oracle's.  (A child process: the test process keeps no process group.  The multi-GPU curve itself is the driver's to measure.)"""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / "frame.npy")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_worker.py"), str(port), out], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "rccl ok" in r.stdout, (r.stdout + r.stderr)[-3000:]
    sc = orc.load_golden_scene("cornell_mirror").with_resolution(96, 55)
    want, _ = orc.render(sc, oracle_config(6), 1, 3)
    assert np.array_equal(np.load(out), want)


@pytest.mark.parametrize("exchange", ["gather", "reduce"])
def test_bench_runs_its_own_multi_gpu_code_under_rccl_with_one_rank(pt, tmp_path, exchange):
    """bench.py itself with PT_BENCH_DIST=1: the N > 1 code of main() -- init_process_group(backend nccl = RCCL), the barriers, the
    all-reduce of the pass time and of the counters, the timed per-frame exchange on device tensors -- runs with a world of one rank
    on this box's GPU; the line carries the exchange block and the dumped frame is the oracle's."""
    out = str(tmp_path / "frame.npy")
    env = dict(os.environ, PT_BENCH_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("RANK", None); env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "2", "--repeats", "2", "--resolution", "320x180",
                        "--no-cpu-baseline", "--exchange", exchange, "--dump-image", out], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    res = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert res["n_gpus"] == 1 and "RCCL" in res["exchange"]["backend"] and "ONE rank" in res["exchange"]["backend"]
    assert res["exchange"]["ms"] > 0 and res["value_if_exchanged_every_pass"] < res["value"] * 1.0001
    sc = orc.load_golden_scene("cornell_mirror").with_resolution(320, 180)
    want, _ = orc.render(sc, oracle_config(8), 3, 4)              # the timed passes render iterations warmup + 1 .. warmup + steps
    assert np.array_equal(np.load(out), want)
