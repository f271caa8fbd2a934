"""Live counts and image of a mesh scene at full size: whole-path kernel (ordering 2, twice) against the per-bounce kernels (ordering 0)."""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
pkg = importlib.import_module("project2-pathtracer_amd")
scene = sys.argv[1] if len(sys.argv) > 1 else "scenes/cornell_mesh.txt"
sf = pkg.SceneFile(scene); g, m, cam = sf.flatten(0)
cam.resolution[0], cam.resolution[1] = 1920.0, 1080.0
res = []
for o in (0, 2, 2):
    tr = pkg.PathTracer(pkg.default_config(max_depth=8, ordering=o)); tr.set_meshes(sf.meshes()); tr.upload(g, m, cam); tr.set_image(None)
    tr.render(1, 2); tr.sync()
    st = tr.stats()
    res.append(([int(st.live[k]) for k in range(9)], tr.image().copy()))
    tr.close()
    print("ordering", o, res[-1][0], flush=True)
for i in (1, 2):
    same = np.array_equal(res[0][1], res[i][1])
    print("run", i, "live equal", res[0][0] == res[i][0], "image equal", same, "" if same else "pixels differing %d" % int((res[0][1] != res[i][1]).any(axis=-1).sum()))
