#!/usr/bin/env python3
"""bench.py -- Mray/s of the MI355X path tracer on BASELINE.json's headline configuration.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: either pre-launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`,
   or started plainly -- then this process spawns exactly that command as a child BEFORE touching the GPU and
   relays rank 0's JSON line and the exit code)

A "step" is ONE render iteration of the workload: one camera ray per pixel of the frame traced
for `depth` bounces through generate -> (trace + scatter + accumulate + compact)* on the GPU(s).
Workload at every N (BASELINE configs[2], the configuration the metric is quoted on):
scenes/cornell_mirror.txt = Cornell box, 1920x1080, 8 bounces, diffuse + perfect specular, stream
compaction on.  Inputs (scene tables, accumulator) are resident in HBM before the timed region.

Multi-GPU: the frame's rows are interleaved over the ranks (row y -> rank y % N) -> "scaling": "strong" (total work
fixed).  The timed region is exactly the K steps on every rank (barrier + synchronize on both sides, max over ranks).
The frame exchange (owned rows gathered on rank 0, or a full-frame RCCL reduce) happens once per FRAME -- once per the
scene's ITERATIONS (1000 here), not once per K-step pass: it is timed on its own, bracketed the same way, right after
the passes, reported as "exchange", and charged to `value` at its true share (t_K + t_exchange * K / ITERATIONS);
"value_if_exchanged_every_pass" is the figure with one exchange per K steps.

metric value = W*H*steps*depth / seconds / 1e6  ("rays launched x bounces / s", BASELINE.json).
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
RAY_BYTES = 40                 # SoA ray record: o(12) d(12) throughput(12) pixel(4)
ACCUM_BYTES = 24               # fp32 RGB accumulator read + write for a path that ends on an emitter

WORKLOADS = {
    "c3": ("scenes/cornell_mirror.txt", 8, "configs[2]: Cornell box 1920x1080, 8 bounces, diffuse + perfect specular, compaction on"),
    "c2": ("scenes/cornell.txt", 8, "configs[1]: sampleScene-equivalent Cornell box 800x800, 8 bounces, diffuse only"),
    "c4": ("scenes/random256.txt", 8, "configs[3]: 1920x1080, 8 bounces, 256 random spheres+cubes"),
    # beyond 256 primitives (the reference's loop takes any numberOfGeoms): not a BASELINE config
    "c1k": ("scenes/random1024.txt", 8, "1920x1080, 8 bounces, 1 024 random spheres+cubes (more than 256 primitives: k_path_w with wide ids)"),
    "c5": ("scenes/cornell_glass_4k.txt", 16, "configs[4]: 3840x2160, 16 bounces, Fresnel refraction + depth of field + jittered AA"),
    # GEOMTYPE MESH (the reference declares the type and leaves its kernel branch empty): not a BASELINE config
    "mesh": ("scenes/cornell_mesh.txt", 8, "MESH: Cornell box 1920x1080, 8 bounces, icosphere (80 triangles) + torus (400) + glass tetrahedron beside a sphere and a cube"),
    "mesh5k": ("scenes/cornell_mesh5k.txt", 8, "MESH: Cornell box 1920x1080, 8 bounces, one 5 120-triangle icosphere beside a sphere and a cube"),
}
# render options a workload needs beyond scene + depth (the reference has no channel for them)
WORKLOAD_OPTIONS = {"c5": dict(camera_mode=1, antialias=1, aperture=0.25, focal_distance=12.0)}
WORKLOAD_RESOLUTION = {"mesh": "1920x1080", "mesh5k": "1920x1080"}       # these scene files carry a small preview RES


def reduce_to_root(tensor, dst=0, force=False):
    """Sum the per-rank accumulators on rank `dst` (RCCL over xGMI when backend is nccl).
    Adding zeros is exact, so the row-sharded sum is bit-identical to a single-GPU render.
    `force`: also with a world of one rank (tests: the RCCL path on the one GPU a test box has)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or force):
        dist.reduce(tensor, dst=dst, op=dist.ReduceOp.SUM)
    return tensor


class RowGather:
    """Per-frame exchange, cheaper form: every rank sends only the rows it owns (y % world == rank,
    1/world of the frame) straight to rank `dst`, which copies them into its full-frame accumulator.
    On the point-to-point xGMI fabric the peers' sends use distinct links in parallel (24.9 MB / 8 =
    3.1 MB per link at 1080p) where a reduce moves the whole frame across every hop.
    No arithmetic at all, so the result is trivially bit-identical to a single-GPU render.

    Allocation-free in the timed region: the packed send buffer and rank `dst`'s [world, rows, W*3] receive
    buffer are allocated ONCE here; a call is one strided pack copy per rank, one gather, and on `dst` one
    strided copy into the frame (a permuted view [world, rows, W*3] of the accumulator when world divides H,
    else one copy per peer)."""

    def __init__(self, like, H, W, dst=0, force=False):
        import torch
        import torch.distributed as dist
        # (`force`: also with a world of one rank -- tests run the RCCL path on the one GPU a test box has)
        self.active = dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or force)
        if not self.active:
            return
        self.world, self.rank, self.dst, self.H, self.W = dist.get_world_size(), dist.get_rank(), dst, H, W
        self.max_rows = (H + self.world - 1) // self.world
        self.send = torch.zeros((self.max_rows, W * 3), dtype=like.dtype, device=like.device)
        self.recv = torch.empty((self.world, self.max_rows, W * 3), dtype=like.dtype, device=like.device) if self.rank == dst else None
        self.recv_list = list(self.recv.unbind(0)) if self.rank == dst else None

    def __call__(self, tensor):
        """`tensor`: this rank's flat float32 [H*W*3] accumulator (same device / dtype as at construction)."""
        import torch.distributed as dist
        if not self.active:
            return tensor
        frame = tensor.view(self.H, self.W * 3)
        mine = frame[self.rank::self.world]
        self.send[:mine.shape[0]].copy_(mine)
        dist.gather(self.send, self.recv_list, dst=self.dst)
        if self.rank == self.dst:
            if self.H % self.world == 0:
                # row y = k*world + r  ->  [k, r, :]; peers' rows in one strided copy, own rows written back unchanged
                self.recv[self.dst].copy_(mine)
                frame.view(self.max_rows, self.world, self.W * 3).permute(1, 0, 2).copy_(self.recv)
            else:
                for r in range(self.world):
                    if r != self.dst:
                        frame[r::self.world] = self.recv[r][:len(range(r, self.H, self.world))]
        return tensor


def gather_rows_to_root(tensor, H, W, dst=0):
    """One-off form of RowGather (tests): allocates its buffers per call."""
    return RowGather(tensor, H, W, dst)(tensor)


def kernel_source_id():
    """sha256[:16] of the kernel sources: ties borrowed counter figures (profiles/traffic_latest.json) to a build"""
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "project2-pathtracer_amd", "csrc")
    for name in sorted(n for n in os.listdir(csrc) if n.startswith("pt_k") or n == "pt_device.hpp"):     # kernel families + shared device code
        with open(os.path.join(csrc, name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def algorithmic_bytes(stats, depth, fused_generate=True):
    """Two byte counts for the K timed steps, from the device's live-ray counters:

    survey  -- SURVEY.md section 8(d)'s algorithmic figure, the one the roofline is quoted on:
               N_0*R (generate write) + sum_k [ N_k*R (read) + N_{k+1}*R (compacted write)
               + (N_k - N_{k+1})*24 (accumulator read+write of every terminated path) ], R = 40 B.
    design  -- what THIS implementation has to move: bounce 0 generates its rays in
               registers (no generate write, no bounce-0 read), the last bounce writes nothing back, and
               only paths that end on an emitter touch the accumulator.
    Geometry/material tables are LDS/L2 traffic and count as 0 in both."""
    live = [int(stats.live[k]) for k in range(depth + 1)]
    survey = live[0] * RAY_BYTES
    for k in range(depth):
        survey += live[k] * RAY_BYTES + live[k + 1] * RAY_BYTES + (live[k] - live[k + 1]) * ACCUM_BYTES
    read = sum(live[k] for k in range(0 if not fused_generate else 1, depth)) * RAY_BYTES
    written = sum(live[k] for k in range(1, depth)) * RAY_BYTES
    design = read + written + int(stats.emitted) * ACCUM_BYTES
    return survey, design, live


def cpu_baseline(scene_path, depth, budget_s=15.0, options=None):
    """The CPU oracle (oracle/pt_oracle.c, "port") timed on this host's cores on a bounded sample of
    the same workload: whole-frame iterations of the same scene/depth until ~budget_s is used."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import ctypes as C
    import numpy as np
    import orc
    pkg = importlib.import_module("project2-pathtracer_amd")
    sf = pkg.SceneFile(scene_path if os.path.isabs(scene_path) else os.path.join(ROOT, scene_path))
    geoms, mats, cam = sf.flatten(0)
    og = (orc.Geom * len(geoms))()
    for i, g in enumerate(geoms):
        og[i].type, og[i].materialid = g.type, g.materialid
        for r in range(3):
            for c in range(4):
                og[i].transform[4 * r + c] = g.transform[4 * r + c]
                og[i].inverseTransform[4 * r + c] = g.inverseTransform[4 * r + c]
        og[i].transform[15] = og[i].inverseTransform[15] = 1.0
    om = (orc.Material * len(mats))()
    C.memmove(om, mats, C.sizeof(om))
    oc = orc.Camera()
    C.memmove(C.byref(oc), C.byref(cam), 52)
    W, H = int(cam.resolution[0]), int(cam.resolution[1])
    L = orc.lib()
    cores = L.orc_max_threads()
    # threads actually usable here: the affinity mask and the cgroup CPU quota can be far below the core count
    try:
        cores = min(cores, len(os.sched_getaffinity(0)))
        quota = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota[0] != "max":
            cores = max(1, min(cores, int(-(-int(quota[0]) // int(quota[1])))))
    except Exception:
        pass
    cfg = orc.default_config(depth, **(options or {}))
    img = np.zeros((H, W, 3), np.float32)
    live = np.zeros(depth + 1, np.uint64)

    def run(first, count):
        t0 = time.perf_counter()
        rc = L.orc_render(og, len(og), om, len(om), C.byref(oc), C.byref(cfg), first, count, orc.fptr(img),
                          live.ctypes.data_as(C.POINTER(C.c_uint64)), cores)
        assert rc == 0
        return time.perf_counter() - t0

    t1 = run(1, 1)
    n = max(1, min(64, int(budget_s / max(t1, 1e-3)) - 1))
    t = run(2, n)
    out = {"value": round(W * H * n * depth / t / 1e6, 3), "unit": "Mray/s", "cores": int(cores), "kind": "port",
           "sample": "%d whole-frame iterations of the same workload (%dx%d, %d bounces) with oracle/pt_oracle.c, OpenMP over rows, %.1f s"
                     % (n, W, H, depth, t)}
    # SURVEY.md 8(d) also asks for the one-core figure: every 4th row of one iteration on a single thread
    # (rows are independent, so the sample scales to the frame), a few seconds of work
    one = orc.default_config(depth, **(options or {}))
    one.row_offset, one.row_stride = 0, 4
    img1 = np.zeros((H, W, 3), np.float32)
    t0 = time.perf_counter()
    rc = L.orc_render(og, len(og), om, len(om), C.byref(oc), C.byref(one), 1, 1, orc.fptr(img1),
                      live.ctypes.data_as(C.POINTER(C.c_uint64)), 1)
    assert rc == 0
    t_one = time.perf_counter() - t0
    rows = len(range(0, H, 4))
    out["one_core"] = {"value": round(W * rows * depth / t_one / 1e6, 3), "unit": "Mray/s",
                       "sample": "%d of %d rows of one iteration on one thread, %.1f s" % (rows, H, t_one)}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=192, help="timed iterations; the default is a whole number of launch groups (16 or 32 iterations each at 1080p)")
    ap.add_argument("--warmup", type=int, default=32)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-events", action="store_true", help="do not bracket launches with HIP events in the timed region")
    ap.add_argument("--chunk-rays", type=int, default=0)
    ap.add_argument("--blocks-per-cu", type=int, default=0)
    ap.add_argument("--ordering", type=int, default=None, help="0 = stable compaction, one launch per bounce, 1 = typed work queues, one launch per bounce (<= 32 primitives), "
                                                             "2 = whole paths, one launch per group (the library default): k_path_q (<= 32 primitives) / k_path_w (more than 32 analytic primitives). "
                                                             "Default: 2; with --direct-light 0 (the per-bounce kernels resolve a shadow ray inline and are the faster form there)")
    ap.add_argument("--grid-density", type=int, default=0, help="k_path_w: cells of its uniform grid per small primitive (0 = default 4)")
    ap.add_argument("--batch", type=int, default=0, help="iterations per launch group (0 = auto, 1 = off)")
    ap.add_argument("--resolution", default="", help="WxH override of the scene RES line (experiments only)")
    ap.add_argument("--culling", type=int, default=0, help="0 = AABB candidate culling (default), 1 = brute force")
    ap.add_argument("--exchange", default="gather", choices=["gather", "reduce"],
                    help="N > 1 frame exchange: gather = owned rows to rank 0 (default), reduce = full-frame sum")
    ap.add_argument("--streams", type=int, default=0,
                    help="contexts per GPU, each on its own HIP stream and owning every (streams*gpus)-th row: the tails of one "
                         "context's launches are filled by the other's (bit-identical, like the multi-GPU sharding); 1 = off; "
                         "0 = auto: 2 when a rank renders at least 30 M camera rays per timed pass (100 M on k_path_w's scenes), else 1 (measured: tools/shard_sim.py)")
    ap.add_argument("--direct-light", type=int, default=0, help="1 = next-event estimation (one shadow ray per diffuse hit); not the headline configuration")
    ap.add_argument("--warm-passes", type=int, default=0, help="untimed K-step passes before the timed ones (0 = until the pass time has settled; profiling runs fix it)")
    ap.add_argument("--dump-image", default="", help="rank 0 writes the frame it holds after the per-frame exchange (float32 .npy, H x W x 3): parity tests of the N > 1 path")
    ap.add_argument("--repeats", type=int, default=7, help="the exact K-step timed pass is repeated this many times; value = the median pass")
    args = ap.parse_args()
    if args.ordering is None:
        args.ordering = 0 if args.direct_light else 2

    if args.gpus > 1 and "RANK" not in os.environ:
        # not launched by torchrun: start the N ranks as a CHILD (nothing in this process has touched the GPU, and
        # nothing is exec'ed over a process that has); relay the child's output and exit code
        import socket
        import subprocess
        sock = socket.socket()
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
        sock.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        child = subprocess.run(cmd, env=env)
        raise SystemExit(child.returncode)

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # (the host driver only supports dmabuf IPC: RCCL needs it under any launcher)
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("WORLD_SIZE=%d does not match --gpus %d (start it plainly, or with torch.distributed.run --nproc-per-node %d)"
                         % (world, args.gpus, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the path tracer has no CPU fallback")
    backend = os.environ.get("PT_BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    device = local_rank % ndev
    torch.cuda.set_device(device)
    # PT_BENCH_DIST=1: run the N > 1 code -- process group, barriers, the all-reduce of the pass time, the per-frame exchange -- with
    # a world of ONE rank as well (a rehearsal of the RCCL calls on a box with one GPU: tests/; never the driver's command)
    dist_on = world > 1 or os.environ.get("PT_BENCH_DIST", "0") == "1"
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:
            import socket
            sk = socket.socket(); sk.bind(("127.0.0.1", 0)); os.environ.setdefault("MASTER_PORT", str(sk.getsockname()[1])); sk.close()
        dist.init_process_group(backend=backend, rank=rank, world_size=world)

    pkg = importlib.import_module("project2-pathtracer_amd")
    scene_path, depth, desc = WORKLOADS[args.workload]
    options = WORKLOAD_OPTIONS.get(args.workload, {})
    scene_file = os.path.join(ROOT, scene_path)
    args.resolution = args.resolution or WORKLOAD_RESOLUTION.get(args.workload, "")
    if args.resolution:
        import re
        import tempfile
        w_, h_ = args.resolution.lower().split("x")
        text = re.sub(r"RES\s+\d+\s+\d+", "RES         %d %d" % (int(w_), int(h_)), open(scene_file).read())
        # in a directory of its own (removed at exit) that mirrors the original's sub-directories, so that relative `*.obj`
        # names still resolve and nothing is written into scenes/
        import atexit
        import shutil
        tmpdir = tempfile.mkdtemp(prefix="ptbench_r%d_" % int(os.environ.get("RANK", "0")))
        atexit.register(shutil.rmtree, tmpdir, True)
        src_dir = os.path.dirname(scene_file)
        for entry in os.listdir(src_dir):
            if os.path.isdir(os.path.join(src_dir, entry)):
                os.symlink(os.path.join(src_dir, entry), os.path.join(tmpdir, entry))
        scene_file = os.path.join(tmpdir, os.path.basename(scene_file))
        with open(scene_file, "w") as fh:
            fh.write(text)
        if args.workload not in WORKLOAD_RESOLUTION:
            desc += " [RES overridden to %s]" % args.resolution
    sf = pkg.SceneFile(scene_file)
    geoms, mats, cam = sf.flatten(0)
    W, H = int(cam.resolution[0]), int(cam.resolution[1])

    # S streams per GPU (pt_config.streams): the context shards this rank's rows once more over S internal contexts,
    # each on its own HIP stream, all rendering straight into the same device accumulator.
    # k_path_w (more than 32 primitives: one block fills a CU's LDS, so a second context's blocks only start where the first one's have
    # finished) gains from the second stream only on long passes: configs[3] 0.865 with one stream against 0.879 with two at 20
    # steps, 0.846 against 0.840 at 192 (in-call pairs, round 4); k_path_q from 30 M camera rays per pass on
    S = args.streams if args.streams > 0 else (2 if (W * H // world) * args.steps >= (30_000_000 if len(geoms) <= 32 else 100_000_000) else 1)
    accum = torch.zeros(W * H * 3, dtype=torch.float32, device="cuda:%d" % device)
    tracer = pkg.PathTracer(pkg.default_config(device=device, max_depth=depth, row_offset=rank, row_stride=world, streams=S,
                                               chunk_rays=args.chunk_rays, blocks_per_cu=args.blocks_per_cu, culling=args.culling, batch=args.batch,
                                               ordering=args.ordering, direct_light=args.direct_light, grid_density=args.grid_density, **options))
    meshes = sf.meshes()
    if meshes:
        tracer.set_meshes(meshes)
    tracer.upload(geoms, mats, cam)
    tracer.bind_device_image(accum)

    def barrier():
        if dist_on:
            dist.barrier()

    # warm-up (untimed)
    tracer.render(1, args.warmup)
    tracer.sync()
    row_gather = RowGather(accum if backend == "nccl" else accum.cpu(), H, W, force=dist_on)     # buffers allocated once, outside any timed region

    def exchange(t):
        return row_gather(t) if args.exchange == "gather" else reduce_to_root(t, force=dist_on)

    if dist_on and backend == "nccl":
        exchange(accum.clone())                # RCCL communicator setup outside the timed region
    torch.cuda.synchronize()

    def max_over_ranks(dt):
        if dist_on:
            t = torch.tensor([dt], dtype=torch.float64, device=("cuda:%d" % device) if backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    def timed_pass(first, with_events):
        """exactly K steps, bracketed by barrier + synchronize; max over ranks"""
        accum.zero_()
        tracer.reset_stats()
        tracer.set_profiling(with_events)
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        tracer.render(first, args.steps)
        tracer.sync()
        torch.cuda.synchronize()
        barrier()
        return max_over_ranks(time.perf_counter() - t0)

    def timed_exchange():
        """the per-frame exchange of the accumulator just rendered, bracketed the same way; max over ranks"""
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if backend == "nccl":
            gathered[0] = exchange(accum)
        else:                                   # CPU rehearsal of the N>1 path (gloo)
            gathered[0] = exchange(accum.cpu())
        torch.cuda.synchronize()
        barrier()
        return max_over_ranks(time.perf_counter() - t0)

    gathered = [None]

    # 0) untimed passes of the SAME K-step shape until the pass time has settled (the W warm-up steps above form a
    #    differently sized launch group, and a cold chip ramps its clock over the first tens of milliseconds: seven
    #    back-to-back 4 ms passes measured 4.39 -> 3.98 ms).  At least one, at most 40 passes or 0.4 s.
    #    (Every quantity in the loop condition is the all-reduced pass time, so all ranks take the same decisions.)
    warm_passes, settled, prev, warm_total = 0, 0, None, 0.0
    while warm_passes < (args.warm_passes or 40) and (args.warm_passes or warm_total < 0.4):
        cur = timed_pass(args.warmup + 1, False)
        warm_passes += 1
        warm_total += cur
        if args.warm_passes:
            continue
        settled = settled + 1 if prev is not None and abs(cur - prev) <= 0.01 * cur else 0
        prev = cur
        if settled >= 3:
            break
    # 1) the metric: the exact K-step pass, repeated; `value` is the MEDIAN pass (min / max reported beside it)
    R = max(1, args.repeats)
    passes = [timed_pass(args.warmup + 1, False) for _ in range(R)]
    order = sorted(range(R), key=lambda k: passes[k])
    elapsed = passes[order[R // 2]] if R % 2 else 0.5 * (passes[order[R // 2 - 1]] + passes[order[R // 2]])
    raw = tracer.stats()                          # counters of the last pass: every pass renders the same iterations
    # N > 1: the frame exchange, once per frame (see the module docstring); three measurements, the median counts
    exchange_s = sorted(timed_exchange() for _ in range(3))[1] if dist_on else 0.0
    if args.dump_image and rank == 0:
        import numpy as np
        frame = gathered[0] if dist_on else accum
        np.save(args.dump_image, frame.detach().cpu().numpy().reshape(H, W, 3))
    frame_iterations = max(int(sf.iterations), args.steps)
    elapsed_frame_share = elapsed + exchange_s * args.steps / frame_iterations
    # 2) the same K steps once more with every launch bracketed by HIP events on the render stream: per-launch
    #    durations (what rocprofv3 --kernel-trace reports).  NOT used for `value` or `roofline.frac`.
    elapsed_events = None
    ev = None
    if not args.no_kernel_events:
        elapsed_events = timed_pass(args.warmup + 1, True)
        ev = tracer.stats()
    import types
    stats = types.SimpleNamespace(live=[int(raw.live[k]) for k in range(65)], emitted=int(raw.emitted), iterations=int(raw.iterations),
                                  bounce_launches=int(raw.bounce_launches))
    if dist_on:                                   # whole-job counters: the live rays / emitter hits of every rank's rows
        t = torch.tensor(stats.live + [stats.emitted, stats.bounce_launches], dtype=torch.int64, device=("cuda:%d" % device) if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        t = [int(v) for v in t.tolist()]
        stats.live, stats.emitted, stats.bounce_launches = t[:65], t[65], t[66]
    nbytes, design_bytes, live = algorithmic_bytes(stats, depth)
    result = None
    if rank == 0:
        value = W * H * args.steps * depth / elapsed_frame_share / 1e6
        launches = max(1, int(stats.bounce_launches))
        # roofline on the SAME pass as `value`: SURVEY.md 8(d)'s algorithmic bytes of the K steps / that pass's wall time
        achieved = nbytes / elapsed_frame_share / 1e9
        traffic = None
        valu = None
        provenance = None
        hbm_measured = None
        nprims = len(geoms)
        # which kernel family rendered (mirrors pt_upload_scene's choice)
        if args.ordering == 2 and not args.direct_light and nprims <= 32:
            kernel = "k_path_q (whole paths, one launch per group: generate + cull + exact tests + scatter + accumulate; rays between bounces on per-wave stacks)"
        elif args.ordering == 2 and args.direct_light and nprims <= 32 and not meshes:
            kernel = "k_path_q<NEE> (whole paths, one launch per group; the shadow ray of a diffuse hit is a record of the same typed queues, resolved before the scattered ray goes on)"
        elif args.ordering == 2 and not args.direct_light and not meshes:
            kernel = "k_path_w (whole paths, one launch per group, more than 32 primitives: grid walk, dense (ray, cell reference) and (ray, primitive) pairs, type-pure exact tests, shading stage by hit type; rays between bounces on per-wave stacks sorted by walk length" + \
                     ("; byte ids, geometry table in LDS)" if nprims <= 256 else "; wide ids, geometry gathered from global memory)")
        elif args.ordering in (1, 2) and not args.direct_light and nprims <= 32:
            kernel = "k_bounce_q (typed work queues, one launch per bounce)"
        else:
            kernel = "k_bounce_seg (cull + exact tests + scatter + accumulate + segmented compaction, one launch per bounce; bounce 0 also generates)"
        tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
        if world == 1 and os.path.exists(tpath):
            try:
                rec = json.load(open(tpath)).get(args.workload, {})
                current = kernel_source_id()
                stale = rec.get("kernel_source_id") != current or (rec.get("ordering") is not None and rec.get("ordering") != args.ordering) \
                    or bool(rec.get("direct_light", 0)) != bool(args.direct_light)
                provenance = {"file": "profiles/traffic_latest.json", "profile": rec.get("profile"), "steps_profiled": rec.get("steps_profiled"),
                              "kernel_source_id": rec.get("kernel_source_id"), "current_kernel_source_id": current, "stale": stale,
                              "note": "counter figures come from a SEPARATE rocprofv3 --pmc run of this command (tools/profile.sh), per rendered step; "
                                      "they are dropped (null) when the kernels have changed since that run or the run used another --ordering"}
                if not stale:
                    per_step = rec.get("hbm_bytes_per_step")
                    if per_step:
                        traffic = round(per_step)
                        gbs = per_step * args.steps / elapsed / 1e9
                        hbm_measured = {"bytes_per_step": round(per_step), "GB_s": round(gbs, 1), "frac_of_peak": round(gbs / HBM_PEAK_GBS, 4),
                                        "ratio_to_algorithmic_bytes": round(per_step * args.steps / max(1, nbytes), 3),
                                        "note": "HBM bytes from the PMC counters, (2*FETCH_SIZE + WRITE_SIZE)*1024 per dispatch in separate passes (MI355X_MICROARCH.md, HBM), "
                                                "per rendered step, over the wall time of the value pass: THIS is the achieved HBM bandwidth against the chip's peak"}
                    vi = rec.get("valu_wave_instructions_per_step")
                    if vi:
                        # what actually bounds these kernels: wave64 VALU issue (SURVEY.md 8d's secondary bound).  Datasheet rate: one
                        # wave64 instruction per 2 cycles per SIMD (MI355X_MICROARCH.md: SIMD-32, 2 cycles); measured ceiling on the
                        # kernels' own exact-test bodies: tools/ubench/valu_ceiling.hip -> profiles/valu_ceiling.json
                        rate = vi * args.steps / elapsed
                        ns_per = elapsed * 1024.0 / (vi * args.steps) * 1e9
                        valu = {"wave_instructions_per_step": vi, "achieved_G_wave_inst_per_s": round(rate / 1e9, 1),
                                "peak_G_wave_inst_per_s": 1228.8, "frac": round(rate / 1.2288e12, 4),
                                "ns_per_wave_instruction_per_simd": round(ns_per, 3),
                                "wave_instructions_per_64_live_ray_bounces": round(vi / max(1.0, sum(live[:depth]) / max(1, stats.iterations) / 64.0), 1),
                                "note": "instruction count from rocprofv3 SQ_INSTS_VALU (see traffic_source), duration = the value pass; peak = 1024 SIMDs x 2.4 GHz / 2 cycles per wave64 instruction"}
                        cpath = os.path.join(ROOT, "profiles", "valu_ceiling.json")
                        if os.path.exists(cpath):
                            ceil = json.load(open(cpath))
                            lo, hi = float(ceil["ns_per_wave_instruction_per_simd"]["best"]), float(ceil["ns_per_wave_instruction_per_simd"]["worst"])
                            valu["measured_ceiling_ns_per_wave_instruction_per_simd"] = [lo, hi]
                            valu["frac_of_measured_ceiling"] = [round(lo / ns_per, 4), round(hi / ns_per, 4)]
                            valu["measured_ceiling_source"] = ceil.get("source")
            except Exception:
                traffic = None
        kernel_events = None
        if ev is not None and int(ev.bounce_launches):
            # one stream: bounce_ms = the summed HIP-event durations.  Several streams: the library reports the longest
            # stream's sum (the launches overlap); every stream's sum spans about the same interval
            kernel_events = {"avg_launch_us": round(float(ev.bounce_ms) * S / int(ev.bounce_launches) * 1e3, 2),
                             "busiest_stream_bounce_ms_per_step": round(float(ev.bounce_ms) / args.steps, 4),
                             "ms_per_step_with_kernel_events": round(elapsed_events / args.steps * 1e3, 4),
                             "kernel_only_frac": round(nbytes / (float(ev.bounce_ms) * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                             "note": "from one extra pass with HIP events around every launch on the render streams (slower than the value pass); "
                                     "kernel_only_frac = algorithmic bytes / the busiest stream's summed bounce-kernel time / 8 TB/s"}
        # `bound`: what the counters say limits the kernel.  Every kernel of this path tracer is limited by vector-instruction
        # issue (IEEE-exact intersection arithmetic), not by HBM; `frac` stays SURVEY.md 8(d)'s HBM-roofline figure (a work rate in
        # survey-byte units, mostly served from L2 / LDS / registers), the measured HBM bandwidth is `hbm_measured`.
        # Without a counter record of THESE kernel sources (traffic_source.stale, or no record for the workload) the line claims no
        # limiter: "unmeasured".
        bound = "unmeasured"
        if hbm_measured and valu:
            bound = "hbm" if hbm_measured["frac_of_peak"] > valu["frac"] else "valu-issue"
        roof = {"bound": bound, "stated_roofline": "hbm (SURVEY.md 8(d) algorithmic bytes / 8 TB/s): `achieved`, `peak`, `frac` below are in those units",
                "kernel": kernel,
                "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": traffic, "traffic_per_launch": (round(traffic * args.steps / launches) if traffic else None),
                "traffic_units": "HBM bytes per rendered step (PMC); traffic_per_launch = x steps / launches",
                "hbm_measured": hbm_measured, "traffic_source": provenance,
                "algorithmic_bytes_per_step": round(nbytes / args.steps), "algorithmic_bytes_per_launch": round(nbytes / launches),
                "duration": "wall time of the median K-step pass = the pass `value` and `ms_per_step` come from (launch gaps and k_fold included): "
                            "frac = algorithmic_bytes_per_step / ms_per_step / 8e12",
                "bytes_formula": "SURVEY.md 8(d): N0*40 + sum_k[N_k*40 + N_k+1*40 + (N_k-N_k+1)*24], from the device live-ray counters",
                "design_moved": {"bytes_per_step": round(design_bytes / args.steps),
                                 "achieved": round(design_bytes / elapsed / 1e9, 1),
                                 "frac": round(design_bytes / elapsed / 1e9 / HBM_PEAK_GBS, 4),
                                 "note": "bytes a per-bounce pool implementation has to move (fused generation, no write-back at the last bounce, accumulator touched by emitter hits only)"},
                "launches": launches, "counters": "live rays, emitter hits and launches summed over all %d rank(s)" % world, "kernel_events": kernel_events, "valu_issue": valu}
        live_per_step = sum(live[:depth]) / max(1, int(stats.iterations))
        result = {
            "metric": "Mray/s (rays launched x bounces / s) at 1080p, 8 bounces" if args.workload in ("c3", "c4") else "Mray/s (rays launched x bounces / s)",
            "value": round(value, 1), "unit": "Mray/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed_frame_share / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "repeats": R, "value_is": "median of %d identical K-step passes" % R,
            "value_min": round(W * H * args.steps * depth / (max(passes) + elapsed_frame_share - elapsed) / 1e6, 1),
            "value_max": round(W * H * args.steps * depth / (min(passes) + elapsed_frame_share - elapsed) / 1e6, 1),
            "spread": round((max(passes) - min(passes)) / elapsed, 4), "passes_ms": [round(p * 1e3, 4) for p in passes],
            "live_Mray_bounces_per_s": round(live_per_step * args.steps / elapsed_frame_share / 1e6, 1),
            "config": {"workload": desc, "scene": scene_path, "resolution": [W, H], "bounces": depth,
                       "rays_per_step": W * H, "sharding": "rows interleaved over %d GPU(s), 1 RCCL %s per frame (timed separately: see exchange)" % (world, "gather of the owned rows" if args.exchange == "gather" else "reduce"),
                       "live_ray_bounces_per_step": round(live_per_step),
                       "compaction": "wave-autonomous (ballot + mbcnt ranks); " + {0: "stable order, one launch per bounce (ordering=0)", 1: "typed work queues, one launch per bounce (ordering=1)", 2: "whole paths, one launch per group (ordering=2)"}.get(args.ordering, "stable order"),
                       "primitives": nprims,
                       "direct_light": bool(args.direct_light), "streams_per_gpu": S,
                       "kernel_shape": (tracer.path_shape() if hasattr(tracer, "path_shape") and hasattr(pkg.lib(), "pt_debug_path_shape") else None),
                       "warmup_passes": "W steps + %d untimed K-step passes (same launch-group shape as the timed passes; until three in a row agree to 1 %%)" % warm_passes},
            "roofline": roof,
        }
        if dist_on:
            result["exchange"] = {
                "kind": "gather of the owned rows on rank 0" if args.exchange == "gather" else "full-frame reduce(sum) to rank 0",
                "backend": ("RCCL over xGMI" if backend == "nccl" else backend + " (CPU rehearsal)") + (" -- a world of ONE rank (PT_BENCH_DIST=1: rehearsal of the calls, nothing crosses a link)" if world == 1 else ""),
                "ms": round(exchange_s * 1e3, 4), "measured": "median of 3, barrier + synchronize on both sides, max over ranks, right after the timed passes",
                "bytes_received_by_rank0": int(W * H * 12 * (world - 1) / world) if args.exchange == "gather" else W * H * 12,
                "once_per": "frame = %d iterations (the scene's ITERATIONS)" % frame_iterations,
                "share_charged_to_value_ms": round(exchange_s * args.steps / frame_iterations * 1e3, 6),
                "k_steps_only_ms": round(elapsed * 1e3, 4),
                "value_if_exchanged_every_pass": round(W * H * args.steps * depth / (elapsed + exchange_s) / 1e6, 1)}
            # both accountings in plain view at the top level
            result["value_if_exchanged_every_pass"] = result["exchange"]["value_if_exchanged_every_pass"]
            result["exchange_accounting"] = ("`value` charges the per-frame exchange (%.3f ms) at its share of a %d-iteration frame: t = t_K + t_exchange * K / %d; "
                                             "`value_if_exchanged_every_pass` charges one whole exchange to the %d timed steps"
                                             % (exchange_s * 1e3, frame_iterations, frame_iterations, args.steps))
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(scene_file, depth, options=options) if not meshes else None    # the oracle's mesh entry point is exercised in tests/, not timed here
        print(json.dumps(result), flush=True)
    tracer.close()
    if dist_on:
        dist.destroy_process_group()
    return result


if __name__ == "__main__":
    main()
