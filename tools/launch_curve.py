"""Time of one render call against the number of iterations in it, on the rows one rank of an N-way job owns: what a launch
costs beyond its rays (ramp, drain, fold, host turnaround).  usage: launch_curve.py [shards=8] [scene] [pipelined calls=4]"""
import importlib, os, sys, time
sys.path.insert(0, os.getcwd())
pkg = importlib.import_module("project2-pathtracer_amd")
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
scene = sys.argv[2] if len(sys.argv) > 2 else "scenes/cornell_mirror.txt"
pipe = int(sys.argv[3]) if len(sys.argv) > 3 else 4
sf = pkg.SceneFile(scene); g, m, cam = sf.flatten(0)
tr = pkg.PathTracer(pkg.default_config(streams=1, max_depth=8, ordering=2, row_offset=0, row_stride=world))
tr.upload(g, m, cam); tr.set_image(None)
tr.render(1, 40); tr.sync()
rows = []
for steps in (1, 2, 5, 10, 20, 40, 80, 160):
    one = 1e9
    for rep in range(5):
        t0 = time.perf_counter(); tr.render(100, steps); tr.sync(); one = min(one, time.perf_counter() - t0)
    many = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        for k in range(pipe):
            tr.render(100 + k * steps, steps)
        tr.sync(); many = min(many, (time.perf_counter() - t0) / pipe)
    rows.append((steps, one * 1e6, many * 1e6))
    print("shards %d, %3d steps per call: %8.1f us per call alone, %8.1f us per call when %d are queued back to back" % (world, steps, one * 1e6, many * 1e6, pipe), flush=True)
(s0, _, a), (s1, _, b) = rows[-2], rows[-1]
slope = (b - a) / (s1 - s0)
print("slope %.2f us per step; intercept %.1f us per call (queued)" % (slope, b - slope * s1))
for s, one, many in rows:
    print("  %3d steps: beyond slope x steps: %7.1f us alone, %7.1f us queued" % (s, one - slope * s, many - slope * s))
