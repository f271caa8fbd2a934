#!/usr/bin/env python3
"""Regenerate tests/golden/ref_*.{json,bmp,png,f32} from oracle/_ref (the reference's own
host code compiled where it lies by oracle/Makefile).  TEST INFRASTRUCTURE ONLY.

Runs only in the build container (needs /root/reference for the sample scene and the
prebuilt oracle/_ref/ref_probe + thrust_probe).  The outputs are DATA: parsed-scene dumps,
matrices, RNG known answers, image bytes -- no reference source text.
"""
import json
import os
import struct
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLD = os.path.join(ROOT, "tests", "golden")
REF = os.environ.get("PT_REFERENCE", "/root/reference")
PROBE = os.path.join(HERE, "_ref", "ref_probe")
THRUST = os.path.join(HERE, "_ref", "thrust_probe")


def run(*args):
    return subprocess.run(list(args), check=True, capture_output=True, text=True).stdout


def lcg(state):
    return (state * 1664525 + 1013904223) & 0xFFFFFFFF


def main():
    os.makedirs(GOLD, exist_ok=True)
    # 1. the reference parser on the reference's own sample scene
    dump = json.loads(run(PROBE, "scene", os.path.join(REF, "scenes", "sampleScene.txt")))
    dump["source"] = "ref_probe scene /root/reference/scenes/sampleScene.txt (reference scene.cpp + utilities.cpp)"
    json.dump(dump, open(os.path.join(GOLD, "ref_scene_sampleScene.json"), "w"), separators=(",", ":"))
    # ... and on the build's own re-typed scenes (parser parity for files the GPU box will read)
    for name in sorted(os.listdir(os.path.join(ROOT, "scenes"))):
        if name.endswith(".txt"):
            d = json.loads(run(PROBE, "scene", os.path.join(ROOT, "scenes", name)))
            d["source"] = "ref_probe scene scenes/%s" % name
            json.dump(d, open(os.path.join(GOLD, "ref_scene_%s.json" % name[:-4]), "w"), separators=(",", ":"))
    # 2. buildTransformationMatrix + glm::inverse on a spread of TRS triples
    st = 565
    cases = [([0, 0, 0], [0, 0, 90], [.01, 10, 10]), ([0, 10, 0], [0, 0, 90], [.3, 3, 3]),
             ([2, 5, 2], [0, 180, 0], [2.5, 2.5, 2.5]), ([0, 0, 0], [0, 0, 0], [1, 1, 1])]
    for _ in range(60):
        vals = []
        for k in range(9):
            st = lcg(st)
            u = (st >> 8) / 16777216.0
            vals.append(round((u * 20 - 10) if k < 3 else (u * 360 if k < 6 else u * 4 + 0.05), 4))
        cases.append((vals[0:3], vals[3:6], vals[6:9]))
    out = []
    for t, r, s in cases:
        args = [repr(float(v)) for v in (t + r + s)]
        res = json.loads(run(PROBE, "transform", *args))
        out.append({"t": t, "r": r, "s": s, **res})
    json.dump({"source": "ref_probe transform (utilities.cpp buildTransformationMatrix + glm::inverse)",
               "floats_are": "binary32 bit patterns", "cases": out},
              open(os.path.join(GOLD, "ref_transforms.json"), "w"), separators=(",", ":"))
    # 3. GLM vector ops
    g = json.loads(run(PROBE, "glm"))
    g["source"] = "ref_probe glm (vendored GLM 0.9.3.4 normalize/cross/dot/length/distance)"
    json.dump(g, open(os.path.join(GOLD, "ref_glm.json"), "w"), separators=(",", ":"))
    # 4. Thrust RNG known answers (rocThrust 7.2 in this image)
    t = json.loads(run(THRUST))
    t["source"] = "thrust_probe (rocThrust 7.2 default_random_engine + uniform_real_distribution<float>)"
    json.dump(t, open(os.path.join(GOLD, "ref_thrust_rng.json"), "w"), separators=(",", ":"))
    # 5. image class: gamma/clamp/u8 + BMP and PNG encoders on a small synthetic accumulator
    W, H, div = 24, 16, 7
    st = 99
    vals = []
    for _ in range(W * H * 3):
        st = lcg(st)
        vals.append(((st >> 8) / 16777216.0) * 9.0 - 0.5)     # some <0, some >divisor
    raw = struct.pack("<%df" % len(vals), *vals)
    open(os.path.join(GOLD, "ref_image_in.f32"), "wb").write(raw)
    for ext in ("bmp", "png"):
        outp = os.path.join(GOLD, "ref_image_out." + ext)
        run(PROBE, "image", str(W), str(H), str(div), repr(1.0 / 2.2), os.path.join(GOLD, "ref_image_in.f32"), outp)
    json.dump({"W": W, "H": H, "divisor": div, "gamma": 1.0 / 2.2,
               "source": "ref_probe image (reference image.cpp + stb_image_write.c)"},
              open(os.path.join(GOLD, "ref_image_meta.json"), "w"))
    # 6. tokenizer
    lines = ["RGB         .63 .06 .04       ", "MATERIAL 0\t\t\t\t//white diffuse", "frame 0", "SCALE       .01 10 10 ", ""]
    json.dump({"source": "ref_probe tokens (utilityCore::tokenizeString)",
               "cases": [{"line": l, "tokens": json.loads(run(PROBE, "tokens", l))} for l in lines]},
              open(os.path.join(GOLD, "ref_tokens.json"), "w"))
    print("golden vectors written to", GOLD)


if __name__ == "__main__":
    sys.exit(main())
