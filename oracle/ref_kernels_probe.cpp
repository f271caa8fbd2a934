// ref_kernels_probe.cpp -- harness around the REFERENCE's own hot-path bodies, compiled where they lie.
//
// TEST INFRASTRUCTURE ONLY (see oracle/pt_oracle.h).  This file contains no reference code: it
// #includes /root/reference/src/intersections.h and interactions.h (hash, multiplyMV, the sphere and
// box tests, getRadiuses, getRandomPointOnCube/Sphere, calculateRandomDirectionInHemisphere -- rows
// (a)1, (a)6-(a)9 of SURVEY.md section 8) UNCHANGED and calls them as ordinary host functions (they
// are `__host__ __device__`).  Built by oracle/Makefile into oracle/_ref/ (git-ignored).
//
// How it compiles here without stand-in files:
//   * <cuda_runtime.h> (sceneStructs.h:11, cudaMat4.h:10) is the real CUDA header set inside the
//     image's triton package (the same one oracle/_ref/ref_probe already uses);
//   * <thrust/random.h> (intersections.h:13) is the image's rocThrust 7.2, which needs a HIP compiler:
//     the TU is compiled by `hipcc -x hip --cuda-host-only`;
//   * HIP and CUDA both define uchar4/float4/dim3/make_float4...: the CUDA copies are left out by
//     pre-defining CUDA's own include guards (-D__VECTOR_TYPES_H__ -D__VECTOR_FUNCTIONS_H__) on the
//     command line, so HIP's definitions serve both.  No reference text and no substitute header exists
//     in this repo for either dependency.
// Overload resolution differs from the reference's 2012 nvcc in two documented places:
//   * min/max(float,float): CUDA declares float overloads in the global namespace; HIP's host headers
//     declare only int ones (the cause of the truncated sphere rows in SURVEY.md 8c).  The two
//     using-declarations below put std::min/std::max (exact match for float) in scope instead.
//   * pow(radius, 2) (intersections.h:180): libstdc++'s C++11 overload returns double where CUDA 4.0's
//     pow(float,int) returned float, so `radicand` is rounded once from double here.  The oracle follows
//     CUDA; tests/test_oracle_vs_reference_kernels.py checks the oracle's pow-in-double variant bit for
//     bit and the float variant to within that single rounding.
// abs/sqrt/cos/sin(float) resolve to libstdc++'s float overloads via <math.h> (what CUDA provides too).
//
// Protocol: `ref_kernels_probe <command>` reads little-endian binary32/uint32 records from stdin and
// writes binary records to stdout (oracle/gen_golden_kernels.py drives it):
//   hash        in: u32                       out: u32
//   multiplymv  in: m[16] v[4]                out: r[3]
//   sphere      in: xf[16] inv[16] o[3] d[3]  out: t P[3] N[3]
//   box         (same)
//   radiuses    in: xf[16]                    out: r[3]
//   cubepoint   in: xf[16] seed               out: p[3]
//   spherepoint in: xf[16] seed               out: p[3]
//   hemisphere  in: n[3] xi1 xi2              out: d[3]
//   pointonray  in: o[3] d[3] t               out: p[3]
#include <math.h>
#include <stdio.h>
#include <string.h>
#include <algorithm>
#include <vector>

using std::max;
using std::min;

#include "intersections.h"
#include "interactions.h"

static cudaMat4 mat_from(const float *m) {
    cudaMat4 r;
    r.x = glm::vec4(m[0], m[1], m[2], m[3]);
    r.y = glm::vec4(m[4], m[5], m[6], m[7]);
    r.z = glm::vec4(m[8], m[9], m[10], m[11]);
    r.w = glm::vec4(m[12], m[13], m[14], m[15]);
    return r;
}

static bool take(float *dst, size_t n) { return fread(dst, 4, n, stdin) == n; }
static void put(const float *src, size_t n) { fwrite(src, 4, n, stdout); }
static void put3(glm::vec3 v) { float o[3] = {v.x, v.y, v.z}; put(o, 3); }

int main(int argc, char **argv) {
    if (argc < 2) { fprintf(stderr, "usage: ref_kernels_probe <command> < in > out\n"); return 2; }
    const char *cmd = argv[1];
    float in[40];
    if (!strcmp(cmd, "hash")) {
        unsigned int a;
        while (fread(&a, 4, 1, stdin) == 1) { unsigned int h = hash(a); fwrite(&h, 4, 1, stdout); }
    } else if (!strcmp(cmd, "multiplymv")) {
        while (take(in, 20)) put3(multiplyMV(mat_from(in), glm::vec4(in[16], in[17], in[18], in[19])));
    } else if (!strcmp(cmd, "sphere") || !strcmp(cmd, "box")) {
        const bool sph = cmd[0] == 's';
        while (take(in, 38)) {
            staticGeom g;
            g.type = sph ? SPHERE : CUBE;
            g.materialid = 0;
            g.transform = mat_from(in);
            g.inverseTransform = mat_from(in + 16);
            ray r;
            r.origin = glm::vec3(in[32], in[33], in[34]);
            r.direction = glm::vec3(in[35], in[36], in[37]);
            glm::vec3 P(0, 0, 0), N(0, 0, 0);
            float t = sph ? sphereIntersectionTest(g, r, P, N) : boxIntersectionTest(g, r, P, N);
            put(&t, 1); put3(P); put3(N);
        }
    } else if (!strcmp(cmd, "radiuses") || !strcmp(cmd, "cubepoint") || !strcmp(cmd, "spherepoint")) {
        const size_t n = cmd[0] == 'r' ? 16 : 17;
        while (take(in, n)) {
            staticGeom g;
            g.type = cmd[0] == 's' ? SPHERE : CUBE;
            g.materialid = 0;
            g.transform = mat_from(in);
            g.inverseTransform = mat_from(in);      // unused by these three
            if (cmd[0] == 'r') put3(getRadiuses(g));
            else if (cmd[0] == 'c') put3(getRandomPointOnCube(g, in[16]));
            else put3(getRandomPointOnSphere(g, in[16]));
        }
    } else if (!strcmp(cmd, "hemisphere")) {
        while (take(in, 5)) put3(calculateRandomDirectionInHemisphere(glm::vec3(in[0], in[1], in[2]), in[3], in[4]));
    } else if (!strcmp(cmd, "pointonray")) {
        while (take(in, 7)) {
            ray r;
            r.origin = glm::vec3(in[0], in[1], in[2]);
            r.direction = glm::vec3(in[3], in[4], in[5]);
            put3(getPointOnRay(r, in[6]));
        }
    } else {
        fprintf(stderr, "unknown command %s\n", cmd);
        return 2;
    }
    return 0;
}
