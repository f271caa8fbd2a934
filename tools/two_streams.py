"""Tail filling by concurrency: N contexts on ONE device, each owning rows y % N == r (bit-identical to one
context, like the multi-GPU sharding), all enqueued before any is awaited.  Prints ms per step for N = 1, 2, 3, 4.
usage: python tools/two_streams.py [steps=192]"""
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("project2-pathtracer_amd")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 192
sf = pkg.SceneFile(os.path.join(ROOT, "scenes", "cornell_mirror.txt"))
g, m, cam = sf.flatten(0)
for n in (1, 2, 3, 4):
    trs = []
    for r in range(n):
        tr = pkg.PathTracer(pkg.default_config(streams=1, max_depth=8, ordering=1, row_offset=r, row_stride=n))
        tr.upload(g, m, cam)
        tr.set_image(None)
        trs.append(tr)
    for tr in trs:
        tr.render(1, 32)
    for tr in trs:
        tr.sync()
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        for tr in trs:
            tr.render(33 + rep * steps, steps)
        for tr in trs:
            tr.sync()
        best = min(best, (time.perf_counter() - t0) / steps * 1e3)
    print("%d context(s) on one device, %d steps: %.4f ms/step" % (n, steps, best))
    for tr in trs:
        tr.close()
