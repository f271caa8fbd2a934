"""Where does the fixed cost of a launch group go?  Renders `steps` iterations of one rank's shard of
configs[2] (row_stride = shards) as ONE pt_render call, `reps` times, and prints wall time per call; run it
under `rocprofv3 --kernel-trace --stats` to compare with the sum of the kernel durations.
usage: tools/group_overhead.py [shards=8] [steps=20] [reps=20] [chunk_rays=0 (auto)]"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("project2-pathtracer_amd")
shards = int(sys.argv[1]) if len(sys.argv) > 1 else 8
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
chunk = int(sys.argv[4]) if len(sys.argv) > 4 else 0
sf = pkg.SceneFile(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scenes", "cornell_mirror.txt"))
g, m, cam = sf.flatten(0)
tr = pkg.PathTracer(pkg.default_config(streams=1, max_depth=8, ordering=1, row_offset=0, row_stride=shards, chunk_rays=chunk))
tr.upload(g, m, cam); tr.set_image(None)
tr.render(1, steps); tr.sync()
t0 = time.perf_counter()
for r in range(reps):
    tr.render(1 + (r + 1) * steps, steps)
    tr.sync()
dt = (time.perf_counter() - t0) / reps
print("chunk_rays %d, shards %d, %d steps per call: %.1f us per call, %.2f us per step" % (chunk, shards, steps, dt * 1e6, dt * 1e6 / steps))
