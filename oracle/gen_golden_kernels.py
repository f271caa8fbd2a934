#!/usr/bin/env python3
"""Regenerate tests/golden/ref_kernels.npz from oracle/_ref/ref_kernels_probe = the REFERENCE's own
intersections.h / interactions.h compiled unchanged as host code (oracle/Makefile, ref_kernels_probe.cpp).
TEST INFRASTRUCTURE ONLY.  Runs only in the build container (needs /root/reference through the prebuilt
probe).  The file written is DATA: seeded inputs and the reference functions' outputs, float32/uint32
arrays -- no reference source text.  tests/test_oracle_vs_reference_kernels.py replays the inputs through
oracle/pt_oracle.c and compares bit patterns.
"""
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(ROOT, "tests"))
GOLD = os.path.join(ROOT, "tests", "golden")
PROBE = os.path.join(HERE, "_ref", "ref_kernels_probe")


def probe(cmd, records, out_words, dtype=np.float32):
    data = np.ascontiguousarray(records).tobytes()
    out = subprocess.run([PROBE, cmd], input=data, check=True, capture_output=True).stdout
    return np.frombuffer(out, dtype).reshape(len(records), out_words).copy()


def scene_objects():
    """(type, transform16, inverse16) of every object of the sample scene and the 256-primitive scene."""
    import orc
    objs = []
    for name in ("sampleScene", "random256", "cornell_glass_4k"):
        sc = orc.load_golden_scene(name)
        for g in sc.geoms:
            objs.append((int(g.type), np.array(g.transform[:16], np.float32), np.array(g.inverseTransform[:16], np.float32)))
    return objs


def rays_for(objs, kind, n, rng):
    """n records [xf16 inv16 o3 d3] against objects of `kind`: aimed at the object (mostly hits), grazing,
    from inside, and unrelated directions (misses)."""
    pick = [o for o in objs if o[0] == kind]
    rec = np.zeros((n, 38), np.float32)
    for i in range(n):
        _, xf, inv = pick[rng.integers(len(pick))]
        centre = xf.reshape(4, 4)[:3, 3]
        ext = np.abs(xf.reshape(4, 4)[:3, :3]).sum(1) * 0.5
        mode = i % 8
        if mode == 6:                                   # origin inside the primitive
            o = centre + (rng.random(3) - 0.5) * ext * 0.5
        else:
            o = np.array([rng.uniform(-6, 6), rng.uniform(-1, 11), rng.uniform(-6, 14)])
        if mode in (0, 1, 2, 3):
            target = centre + (rng.random(3) - 0.5) * ext * 1.6
        elif mode == 4:                                 # grazing: aim at the silhouette
            target = centre + np.sign(rng.random(3) - 0.5) * ext * rng.uniform(0.95, 1.05)
        elif mode == 5:
            target = o + rng.normal(size=3)
        else:
            target = centre + rng.normal(size=3) * ext
        d = target - o
        d = (d / np.linalg.norm(d)).astype(np.float32)
        if mode == 7:                                   # axis-parallel directions: infinite inverse components
            d = np.zeros(3, np.float32); d[rng.integers(3)] = np.float32(rng.choice([-1.0, 1.0]))
        rec[i, :16], rec[i, 16:32], rec[i, 32:35], rec[i, 35:38] = xf, inv, o.astype(np.float32), d
    return rec


def main():
    rng = np.random.default_rng(565)
    objs = scene_objects()
    out = {}
    h_in = np.concatenate([np.arange(0, 16, dtype=np.uint32), rng.integers(0, 2 ** 32, 240, dtype=np.uint64).astype(np.uint32)])
    out["hash_in"] = h_in
    out["hash_out"] = probe("hash", h_in.reshape(-1, 1), 1, np.uint32).ravel()
    mv = rng.normal(size=(256, 20)).astype(np.float32)
    mv[:128, 19] = 1.0
    mv[128:, 19] = 0.0
    out["multiplymv_in"], out["multiplymv_out"] = mv, probe("multiplymv", mv, 3)
    for kind, cmd in ((0, "sphere"), (1, "box")):
        rec = rays_for(objs, kind, 4096, rng)
        out[cmd + "_in"], out[cmd + "_out"] = rec, probe(cmd, rec, 7)
    xfs = np.stack([o[1] for o in objs])
    out["radiuses_in"], out["radiuses_out"] = xfs, probe("radiuses", xfs, 3)
    for kind, cmd in ((1, "cubepoint"), (0, "spherepoint")):
        pick = [o for o in objs if o[0] == kind]
        rec = np.zeros((1024, 17), np.float32)
        for i in range(len(rec)):
            rec[i, :16] = pick[rng.integers(len(pick))][1]
            rec[i, 16] = np.float32(rng.integers(0, 1 << 24)) if i % 4 else np.float32(rng.uniform(0, 1e6))
        out[cmd + "_in"], out[cmd + "_out"] = rec, probe(cmd, rec, 3)
    hn = rng.normal(size=(4096, 3))
    hn /= np.linalg.norm(hn, axis=1, keepdims=True)
    hn = hn.astype(np.float32)
    hn[:6] = [[0, 1, 0], [1, 0, 0], [0, 0, 1], [0, -1, 0], [-1, 0, 0], [0, 0, -1]]
    hn[6:12] = [[0.01, 0, 0], [0, 10, 0], [0, 0, -0.3], [1.3e-8, -0.3, 0], [4.371139e-10, 0, 0.01], [0.57735, 0.57735, 0.57735]]
    hem = np.concatenate([hn, rng.random((4096, 2)).astype(np.float32)], axis=1)
    hem[0, 3:] = [0.25, 0.5]
    out["hemisphere_in"], out["hemisphere_out"] = hem, probe("hemisphere", hem, 3)
    por = rng.normal(size=(256, 7)).astype(np.float32)
    por[:, 6] = np.abs(por[:, 6]) * 10
    out["pointonray_in"], out["pointonray_out"] = por, probe("pointonray", por, 3)
    path = os.path.join(GOLD, "ref_kernels.npz")
    np.savez_compressed(path, **out)
    hits = {k: int((out[k + "_out"][:, 0] >= 0).sum()) for k in ("sphere", "box")}
    print("wrote %s (%d bytes); hits %s" % (path, os.path.getsize(path), hits))


if __name__ == "__main__":
    main()
