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
  cornell_glass_4k.txt   config 5: 3840x2160, sphere 5 uses the glass material (REFR 1, IOR 2.2)
"""
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


def random256(seed=565, extra=250):
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
        s = round(rng.u(0.2, 0.8), 3)
        if i & 1:
            rot = (round(rng.u(0, 360), 2), round(rng.u(0, 360), 2), round(rng.u(0, 360), 2))
            objs.append(("cube", i % 7, c, rot, (s, s, s)))
        else:
            objs.append(("sphere", i % 7, c, (0, 0, 0), (s, s, s)))
    return mats, objs


def main():
    out = os.path.join(ROOT, "scenes")
    os.makedirs(out, exist_ok=True)
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
    glass = list(CORNELL_OBJECTS)
    glass[5] = ("sphere", 5, (0, 2, 0), (0, 180, 0), (3, 3, 3))
    write_scene(os.path.join(out, "cornell_glass_4k.txt"), CORNELL_MATERIALS, glass, (3840, 2160), 1000,
                "renders/cornell_glass_4k.bmp", frames=1)
    print("wrote", sorted(os.listdir(out)))


if __name__ == "__main__":
    main()
