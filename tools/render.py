#!/usr/bin/env python3
"""Render a scene file on the GPU and save the gamma-corrected picture (visual sanity check).
usage: tools/render.py <scene.txt> <out.png> [iterations] [depth] [direct_light 0|1]"""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("project2-pathtracer_amd")


def main():
    scene, out = sys.argv[1], sys.argv[2]
    iters = int(sys.argv[3]) if len(sys.argv) > 3 else 500
    depth = int(sys.argv[4]) if len(sys.argv) > 4 else 8
    direct = int(sys.argv[5]) if len(sys.argv) > 5 else 0
    sf = pkg.SceneFile(scene)
    g, m, cam = sf.flatten(0)
    tr = pkg.PathTracer(pkg.default_config(streams=1, max_depth=depth, ordering=1, direct_light=direct))
    meshes = sf.meshes()
    if meshes:
        tr.set_meshes(meshes)
    tr.upload(g, m, cam)
    tr.set_image(None)
    t0 = time.time()
    tr.render(1, iters)
    img = tr.image()
    dt = time.time() - t0
    u8 = pkg.image_to_u8(img, iters, np.float32(1.0 / 2.2))
    from PIL import Image
    Image.fromarray(u8).save(out, optimize=True)
    print("%s: %dx%d, %d iterations, depth %d%s in %.2f s; mean gamma RGB %s" % (out, tr.W, tr.H, iters, depth, ", direct light sampling" if direct else "", dt, (u8.reshape(-1, 3).mean(0) / 255).round(3)))


if __name__ == "__main__":
    main()
