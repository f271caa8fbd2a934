#!/usr/bin/env python3
"""Write the scene files under scenes/ in the reference's scene grammar
(/root/reference/src/scene.cpp:9-263; grammar summary in SURVEY.md section 5).

These are the build's own files (typed from parameter tables below, no comments, LF endings):
  cornell.txt            the Cornell box of BASELINE config 2 -- same numbers as the reference's
                         scenes/sampleScene.txt (9 materials, 9 objects, 2 identical frames,
                         800x800, FOVY 25, 5000 iterations); tests/test_scene_loader.py checks that
                         it parses to the same PODs as the reference file does.
  cornell_c1.txt         config 1: RES 400 400, ITERATIONS 1
  cornell_mirror.txt     config 3: 1920x1080, materials 3,4,6 are perfect mirrors (REFL 1)
  random256.txt          config 4: Cornell shell + 250 random spheres/cubes, seed 565, 1920x1080
  random1024.txt         the same shell + 1 018 smaller spheres/cubes, seed 566: more than 256 primitives (not a BASELINE configuration)
  cornell_glass_4k.txt   config 5: 3840x2160, sphere 5 uses the glass material (REFR 1, IOR 2.2)
  cornell_mesh.txt       GEOMTYPE MESH (the reference only declares it): Cornell shell + light + three `*.obj` objects
                         -- a 320-triangle icosphere (diffuse), a quad-faced torus (mirror) and a glass tetrahedron,
                         rotated and non-uniformly scaled -- next to one sphere and one cube
  meshes/*.obj           the meshes, generated below (v / f lines only)
"""
import math
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# (RGB, SPECEX, SPECRGB, REFL, REFR, REFRIOR, SCATTER, ABSCOEFF, RSCTCOEFF, EMITTANCE)
CORNELL_MATERIALS = [
    ((1, 1, 1), 0, (1, 1, 1), 0, 0, 0, 0, (0, 0, 0), 0, 0),            # 0 white diffuse
    ((.63, .06, .04), 0, (1, 1, 1), 0, 0, 0, 0, (0, 0, 0), 0, 0),      # 1 red diffuse
    ((.15, .48, .09), 0, (1, 1, 1), 0, 0, 0, 0, (0, 0, 0), 0, 0),      # 2 green diffuse
    ((.63, .06, .04), 0, (1, 1, 1), 0, 0, 2, 0, (0, 0, 0), 0, 0),      # 3 red glossy
    ((1, 1, 1), 0, (1, 1, 1), 0, 0, 2, 0, (0, 0, 0), 0, 0),            # 4 white glossy
    ((0, 0, 0), 0, (1, 1, 1), 0, 1, 2.2, 0, (.02, 5.1, 5.7), 13, 0),   # 5 glass
    ((.15, .48, .09), 0, (1, 1, 1), 0, 0, 2.6, 0, (0, 0, 0), 0, 0),    # 6 green glossy
    ((1, 1, 1), 0, (0, 0, 0), 0, 0, 0, 0, (0, 0, 0), 0, 1),            # 7 dim light
    ((1, 1, 1), 0, (0, 0, 0), 0, 0, 0, 0, (0, 0, 0), 0, 15),           # 8 light
]

# (type, material, TRANS, ROTAT, SCALE)
CORNELL_OBJECTS = [
    ("cube", 0, (0, 0, 0), (0, 0, 90), (.01, 10, 10)),      # floor
    ("cube", 0, (0, 5, -5), (0, 90, 0), (.01, 10, 10)),     # back wall
    ("cube", 0, (0, 10, 0), (0, 0, 90), (.01, 10, 10)),     # ceiling
    ("cube", 1, (-5, 5, 0), (0, 0, 0), (.01, 10, 10)),      # left wall (red)
    ("cube", 2, (5, 5, 0), (0, 0, 0), (.01, 10, 10)),       # right wall (green)
    ("sphere", 4, (0, 2, 0), (0, 180, 0), (3, 3, 3)),
    ("sphere", 3, (2, 5, 2), (0, 180, 0), (2.5, 2.5, 2.5)),
    ("sphere", 6, (-2, 5, -2), (0, 180, 0), (3, 3, 3)),
    ("cube", 8, (0, 10, 0), (0, 0, 90), (.3, 3, 3)),        # light
]

CAMERA = dict(eye=(0, 4.5, 12), view=(0, 0, -1), up=(0, 1, 0), fovy=25)


def num(v):
    s = repr(float(v))
    if s.endswith(".0"):
        s = s[:-2]
    if s.startswith("0."):
        s = s[1:]
    if s.startswith("-0."):
        s = "-" + s[2:]
    return s


def triple(t):
    return " ".join(num(v) for v in t)


def write_scene(path, materials, objects, res, iterations, outfile, frames=2, camera=CAMERA):
    L = []
    for i, m in enumerate(materials):
        rgb, specex, specrgb, refl, refr, ior, scatter, absc, rsct, emit = m
        # the parser accepts the ten property lines in any order (src/scene.cpp:230-258)
        L += ["MATERIAL %d" % i, "EMITTANCE " + num(emit), "RGB " + triple(rgb), "REFL " + num(refl),
              "REFR " + num(refr), "REFRIOR " + num(ior), "SPECRGB " + triple(specrgb), "SPECEX " + num(specex),
              "SCATTER " + num(scatter), "ABSCOEFF " + triple(absc), "RSCTCOEFF " + num(rsct), ""]
    L += ["CAMERA", "FILE " + outfile, "ITERATIONS %d" % iterations, "RES %d %d" % res, "FOVY " + num(camera["fovy"])]
    for f in range(frames):
        L += ["frame %d" % f, "UP " + triple(camera["up"]), "VIEW " + triple(camera["view"]), "EYE " + triple(camera["eye"])]
    L += [""]
    for i, (typ, mat, t, r, s) in enumerate(objects):
        L += ["OBJECT %d" % i, typ, "material %d" % mat]
        for f in range(frames):
            L += ["frame %d" % f, "SCALE " + triple(s), "ROTAT " + triple(r), "TRANS " + triple(t)]
        L += [""]
    with open(path, "w", newline="\n") as fh:
        fh.write("\n".join(L))


class MinStd:
    """minstd_rand (a=48271, m=2^31-1), the engine the reference uses for its RNG."""

    def __init__(self, seed):
        self.x = seed % 2147483647 or 1

    def u(self, lo=0.0, hi=1.0):
        self.x = (self.x * 48271) % 2147483647
        return lo + (hi - lo) * ((self.x - 1) / 2147483648.0)


def random256(seed=565, extra=250, size=(0.2, 0.8)):
    mats = [
        CORNELL_MATERIALS[0], CORNELL_MATERIALS[1], CORNELL_MATERIALS[2],
        ((1, 1, 1), 0, (.9, .9, .9), 1, 0, 0, 0, (0, 0, 0), 0, 0),          # 3 mirror
        ((.25, .35, .75), 0, (1, 1, 1), 0, 0, 0, 0, (0, 0, 0), 0, 0),       # 4 blue diffuse
        ((.8, .7, .2), 0, (1, 1, 1), 0, 0, 0, 0, (0, 0, 0), 0, 0),          # 5 yellow diffuse
        ((0, 0, 0), 0, (1, 1, 1), 0, 1, 1.5, 0, (0, 0, 0), 0, 0),           # 6 glass
        CORNELL_MATERIALS[8],                                               # 7 light
    ]
    objs = [o for o in CORNELL_OBJECTS[:5]] + [("cube", 7, (0, 10, 0), (0, 0, 90), (.3, 3, 3))]
    rng = MinStd(seed)
    for i in range(extra):
        c = (round(rng.u(-4.5, 4.5), 3), round(rng.u(0.5, 9.0), 3), round(rng.u(-4.5, 4.5), 3))
        s = round(rng.u(size[0], size[1]), 3)
        if i & 1:
            rot = (round(rng.u(0, 360), 2), round(rng.u(0, 360), 2), round(rng.u(0, 360), 2))
            objs.append(("cube", i % 7, c, rot, (s, s, s)))
        else:
            objs.append(("sphere", i % 7, c, (0, 0, 0), (s, s, s)))
    return mats, objs


def icosphere(level):
    """unit-diameter icosphere, `level` subdivisions (20 * 4^level triangles)"""
    t = (1.0 + math.sqrt(5.0)) / 2.0
    v = [(-1, t, 0), (1, t, 0), (-1, -t, 0), (1, -t, 0), (0, -1, t), (0, 1, t), (0, -1, -t), (0, 1, -t),
         (t, 0, -1), (t, 0, 1), (-t, 0, -1), (-t, 0, 1)]
    f = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2), (10, 7, 6), (7, 1, 8),
         (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8), (3, 8, 9), (4, 9, 5), (2, 4, 11), (6, 2, 10), (8, 6, 7), (9, 8, 1)]
    v = [tuple(c / math.sqrt(1 + t * t) for c in p) for p in v]
    for _ in range(level):
        mid, nf = {}, []

        def m(a, b):
            key = (min(a, b), max(a, b))
            if key not in mid:
                p = [(v[a][k] + v[b][k]) / 2 for k in range(3)]
                n = math.sqrt(sum(c * c for c in p))
                v.append(tuple(c / n for c in p))
                mid[key] = len(v) - 1
            return mid[key]
        for a, b, c in f:
            ab, bc, ca = m(a, b), m(b, c), m(c, a)
            nf += [(a, ab, ca), (b, bc, ab), (c, ca, bc), (ab, bc, ca)]
        f = nf
    return [tuple(0.5 * c for c in p) for p in v], f


def torus(nu, nv, R=0.35, r=0.15):
    """quad faces (the loader fans them); axis = y"""
    v, f = [], []
    for i in range(nu):
        a = 2 * math.pi * i / nu
        for j in range(nv):
            b = 2 * math.pi * j / nv
            v.append(((R + r * math.cos(b)) * math.cos(a), r * math.sin(b), (R + r * math.cos(b)) * math.sin(a)))
    for i in range(nu):
        for j in range(nv):
            f.append((i * nv + j, ((i + 1) % nu) * nv + j, ((i + 1) % nu) * nv + (j + 1) % nv, i * nv + (j + 1) % nv))
    return v, f


def write_obj(path, v, f, relative=False):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w", newline="\n") as fh:
        fh.write("# generated by tools/make_scenes.py\n")
        for p in v:
            fh.write("v %.6f %.6f %.6f\n" % p)
        for face in f:
            if relative:                       # negative indices count back from the last vertex
                fh.write("f " + " ".join(str(i - len(v)) for i in face) + "\n")
            else:
                fh.write("f " + " ".join("%d/%d/%d" % (i + 1, i + 1, i + 1) if len(face) == 4 else str(i + 1) for i in face) + "\n")


def mesh_scene(out):
    v, f = icosphere(2)
    write_obj(os.path.join(out, "meshes", "icosphere.obj"), v, f)
    v, f = torus(20, 10)
    write_obj(os.path.join(out, "meshes", "torus.obj"), v, f)
    tet = [(-.5, -.5, -.5), (.5, -.5, .5), (-.5, .5, .5), (.5, .5, -.5)]
    write_obj(os.path.join(out, "meshes", "tetra.obj"), tet, [(0, 1, 2), (0, 3, 1), (0, 2, 3), (1, 3, 2)], relative=True)
    mats = list(CORNELL_MATERIALS)
    m = list(mats[4]); m[3] = 1; mats[4] = tuple(m)                 # 4: white mirror
    objs = [o for o in CORNELL_OBJECTS[:5]] + [CORNELL_OBJECTS[8]] + [
        ("meshes/icosphere.obj", 2, (-2.2, 2.0, 0.5), (20, 35, 0), (3.4, 3.0, 3.4)),
        ("meshes/torus.obj", 4, (1.8, 3.2, -1.0), (60, 20, 10), (5.5, 5.5, 5.5)),
        ("meshes/tetra.obj", 5, (0.4, 1.2, 2.6), (0, 30, 0), (2.0, 2.4, 2.0)),
        ("sphere", 1, (2.6, 6.8, 1.5), (0, 0, 0), (2, 2, 2)),
        ("cube", 0, (-2.8, 6.5, -2.0), (25, 40, 10), (1.6, 1.6, 1.6)),
    ]
    write_scene(os.path.join(out, "cornell_mesh.txt"), mats, objs, (800, 600), 500, "renders/cornell_mesh.bmp", frames=1)
    # one large mesh (bench.py --workload mesh5k): a 5 120-triangle icosphere, diffuse, beside a sphere and a cube
    v, f = icosphere(4)
    write_obj(os.path.join(out, "meshes", "icosphere5k.obj"), v, f)
    objs5k = [o for o in CORNELL_OBJECTS[:5]] + [CORNELL_OBJECTS[8]] + [
        ("meshes/icosphere5k.obj", 2, (-0.6, 3.2, 0.0), (20, 35, 0), (5.0, 5.0, 5.0)),
        ("sphere", 4, (2.9, 7.0, 1.5), (0, 0, 0), (2, 2, 2)),
        ("cube", 0, (-3.2, 7.0, -2.0), (25, 40, 10), (1.6, 1.6, 1.6)),
    ]
    write_scene(os.path.join(out, "cornell_mesh5k.txt"), mats, objs5k, (800, 600), 500, "renders/cornell_mesh5k.bmp", frames=1)


def main():
    out = os.path.join(ROOT, "scenes")
    os.makedirs(out, exist_ok=True)
    mesh_scene(out)
    write_scene(os.path.join(out, "cornell.txt"), CORNELL_MATERIALS, CORNELL_OBJECTS, (800, 800), 5000,
                "renders/sampleScene.bmp")
    write_scene(os.path.join(out, "cornell_c1.txt"), CORNELL_MATERIALS, CORNELL_OBJECTS, (400, 400), 1,
                "renders/sampleScene.bmp")
    mirror = list(CORNELL_MATERIALS)
    for k in (3, 4, 6):
        m = list(mirror[k]); m[3] = 1; mirror[k] = tuple(m)
    write_scene(os.path.join(out, "cornell_mirror.txt"), mirror, CORNELL_OBJECTS, (1920, 1080), 1000,
                "renders/cornell_mirror.bmp")
    mats, objs = random256()
    write_scene(os.path.join(out, "random256.txt"), mats, objs, (1920, 1080), 1000, "renders/random256.bmp", frames=1)
    # beyond 256 primitives (the reference's loop takes any count): the same room with 1 018 smaller spheres / cubes
    mats, objs = random256(seed=566, extra=1018, size=(0.12, 0.5))
    write_scene(os.path.join(out, "random1024.txt"), mats, objs, (1920, 1080), 1000, "renders/random1024.bmp", frames=1)
    glass = list(CORNELL_OBJECTS)
    glass[5] = ("sphere", 5, (0, 2, 0), (0, 180, 0), (3, 3, 3))
    write_scene(os.path.join(out, "cornell_glass_4k.txt"), CORNELL_MATERIALS, glass, (3840, 2160), 1000,
                "renders/cornell_glass_4k.bmp", frames=1)
    print("wrote", sorted(os.listdir(out)))


if __name__ == "__main__":
    main()
