// pt_device.hpp -- device-side primitives of the gfx950 path tracer.
//
// Each function states the reference function whose result it reproduces (paths relative to
// /root/reference).  Arithmetic contract: binary32, round-to-nearest-even,
// no FMA contraction (this TU is compiled with -ffp-contract=off), correctly rounded
// division and sqrt (hipcc default), sums associated left to right exactly as the
// reference's C++ expressions are.  The expression trees are deliberately spelled out:
// the CPU oracle must agree with these kernels bit for bit (up to the sign of zero).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pt_camrec.hpp"

// Every function here is host-callable too: the host evaluates per-primitive constants (the box-face
// shading frames, pt_api.hip) with the very same expression trees, compiled by the same hipcc run with
// the same -ffp-contract=off, so a tabulated value has the bits the kernel would have computed.
#define PTD_FN __host__ __device__ __forceinline__

namespace ptd {

// src/utilities.h:20-26
#define PT_PI 3.1415926535897932384626422832795028841971f
#define PT_TWO_PI 6.2831853071795864769252867665590057683943f
#define PT_SQRT_OF_ONE_THIRD 0.5773502691896257645091487805019574556476f
#define PT_EPSILON .000000001f
#define PT_RAY_BIAS 0.0002f
#define PT_TRANSMIT_BIAS 0.001f   // build-defined

struct f3 { float x, y, z; };

PTD_FN f3 mk(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
PTD_FN f3 operator+(f3 a, f3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
PTD_FN f3 operator-(f3 a, f3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
PTD_FN f3 operator*(f3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
PTD_FN f3 operator*(f3 a, f3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
PTD_FN f3 operator/(f3 a, float s) { return mk(a.x / s, a.y / s, a.z / s); }
PTD_FN f3 neg(f3 a) { return mk(-a.x, -a.y, -a.z); }
// glm::dot / cross / length / normalize (src/glm/core/func_geometric.inl:158-167,199-211,59-68,239-248)
PTD_FN float dot(f3 a, f3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
PTD_FN f3 cross(f3 x, f3 y) {
    return mk(x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y);
}
PTD_FN float length(f3 a) { return __builtin_sqrtf(dot(a, a)); }
PTD_FN f3 normalize(f3 a) { return a * (1.0f / __builtin_sqrtf(dot(a, a))); }

// ---------------------------------------------------------------- RNG ------------------
// hash: src/intersections.h:26-34
PTD_FN uint32_t hash(uint32_t a) {
    a = (a + 0x7ed55d16u) + (a << 12);
    a = (a ^ 0xc761c23cu) ^ (a >> 19);
    a = (a + 0x165667b1u) + (a << 5);
    a = (a + 0xd3a2646cu) ^ (a << 9);
    a = (a + 0xfd7046c5u) + (a << 3);
    a = (a ^ 0xb55a4f09u) ^ (a >> 16);
    return a;
}
// thrust::default_random_engine = minstd_rand (a = 48271, m = 2^31-1); m is a Mersenne prime so
// x mod m folds as (x & m) + (x >> 31) with one conditional subtract (no 64-bit division).
PTD_FN uint32_t lcg_seed(uint32_t s) {
    uint32_t r = (s & 0x7FFFFFFFu) + (s >> 31);
    if (r >= 0x7FFFFFFFu) r -= 0x7FFFFFFFu;
    return r == 0u ? 1u : r;
}
PTD_FN uint32_t lcg_next(uint32_t x) {
    uint64_t p = (uint64_t)x * 48271ull;
    uint32_t r = (uint32_t)(p & 0x7FFFFFFFull) + (uint32_t)(p >> 31);
    if (r >= 0x7FFFFFFFu) r -= 0x7FFFFFFFu;
    return r;
}
// uniform_real_distribution<float>(0,1): float(x - 1) / 2^31
PTD_FN float u01(uint32_t x) { return (float)(x - 1u) / 2147483648.0f; }
// build-defined stream seed: ONE hash of an injective mix of (pixel, iteration, stream)
// -- `hash` is a bijection on 32 bits, so two streams coincide only where the mixes do
PTD_FN uint32_t stream_seed(uint32_t pixel, uint32_t iteration, uint32_t stream) {
    return hash(pixel + 0x9E3779B9u * iteration + 0x85EBCA6Bu * stream);
}

// sin/cos of a in [0, 2pi]: Cody-Waite by pi/2 + Cephes-coefficient polynomials, +,-,* only.
PTD_FN void sincos_poly(float a, float &s, float &c) {
    int k = (int)((a * 0.636619772367581343f) + 0.5f);
    float kf = (float)k;
    float r = ((a - (kf * 1.5703125f)) - (kf * 4.837512969970703125e-4f)) - (kf * 7.54978995489188216e-8f);
    float z = r * r;
    float sp = (((((-1.9515295891e-4f * z) + 8.3321608736e-3f) * z) - 1.6666654611e-1f) * z) * r + r;
    float cp = ((((((2.443315711809948e-5f * z) - 1.388731625493765e-3f) * z) + 4.166664568298827e-2f) * z) * z
                - (0.5f * z)) + 1.0f;
    int q = k & 3;
    float s0 = (q & 1) ? cp : sp;
    float c0 = (q & 1) ? sp : cp;
    s = (q & 2) ? -s0 : s0;
    c = (q == 1 || q == 2) ? -c0 : c0;
}

// ---------------------------------------------------------------- tables ---------------
// One primitive as the kernels read it: rows x,y,z of inverseTransform and transform
// (multiplyMV never reads row w, src/intersections.h:53-59), 28 dwords = 112 B so that every
// row is a 16-byte LDS read; + the culling box: 36 dwords = 144 B (a stride that keeps the
// per-lane gathers of up to 9 different primitives bank-conflict free).
struct __attribute__((aligned(16))) GeomRec {
    float inv[12];
    float xf[12];
    int type;
    int mat;
    int inside_hits;   // box only: material is refractive (build extension)
    float slack;       // world-space slack subtracted from the AABB entry distance before it is
                       // compared with the best hit (sphere: the 1e-4 object-space pull-back, scaled)
    float bmin[4];     // conservative world-space AABB (inflated), xyz + pad
    float bmax[4];
};
// material fields the scatter reads (src/sceneStructs.h:62-73), 12 dwords
struct MatRec {
    float color[3];
    float emittance;
    float spec[3];
    float refl;
    float refr;
    float ior;
    float pad[2];
};

// multiplyMV with w = 1 (src/intersections.h:53-59): ((m0*x + m1*y) + m2*z) + m3*1
PTD_FN f3 mul_point(const float *m, f3 v) {
    return mk((((m[0] * v.x) + (m[1] * v.y)) + (m[2] * v.z)) + m[3],
              (((m[4] * v.x) + (m[5] * v.y)) + (m[6] * v.z)) + m[7],
              (((m[8] * v.x) + (m[9] * v.y)) + (m[10] * v.z)) + m[11]);
}
// multiplyMV with w = 0: the "+ m3*0" term only affects the sign of a zero result
PTD_FN f3 mul_vector(const float *m, f3 v) {
    return mk(((m[0] * v.x) + (m[1] * v.y)) + (m[2] * v.z),
              ((m[4] * v.x) + (m[5] * v.y)) + (m[6] * v.z),
              ((m[8] * v.x) + (m[9] * v.y)) + (m[10] * v.z));
}

// sphereIntersectionTest (src/intersections.h:168-204) incl. getPointOnRay's second
// normalize and its double-precision `t-.0001` (:46-48).  Returns world distance or -1.
PTD_FN float sphere_test(const float *inv, const float *xf, f3 o, f3 d, f3 &P, f3 &N) {
    f3 ro = mul_point(inv, o);
    f3 rd = normalize(mul_vector(inv, d));
    float b = dot(ro, rd);
    float radicand = b * b - (dot(ro, ro) - (0.5f * 0.5f));
    if (radicand < 0.0f) return -1.0f;
    float sq = __builtin_sqrtf(radicand);
    float first = -b;
    float t1 = first + sq;
    float t2 = first - sq;
    float t;
    if (t1 < 0.0f && t2 < 0.0f) return -1.0f;
    else if (t1 > 0.0f && t2 > 0.0f) t = fminf(t1, t2);
    else t = fmaxf(t1, t2);
    float tt = (float)((double)t - .0001);
    f3 pobj = ro + normalize(rd) * tt;
    f3 wp = mul_point(xf, pobj);
    f3 centre = mk(xf[3], xf[7], xf[11]);          // multiplyMV(transform, (0,0,0,1))
    P = wp;
    N = normalize(wp - centre);
    return length(o - wp);
}

// boxIntersectionTest on the unit cube (src/intersections.h:73-164); inside_hits is the
// build's extension for refractive boxes (a ray starting inside leaves through tmax).
// box_test_face returns the world distance (or -1), the hit point and the FACE instead of the normal:
// face = col + 3*(negative side), col = 0,1,2 for x,y,z in the reference's cascade order +x,+y,+z,-x,-y,-z
// (:143-155), -1 when the cascade finds no face.  N = multiplyMV(transform,(n,0)) = +-column `col` of xf.
PTD_FN float box_test_face(const float *inv, const float *xf, int inside_hits, f3 o, f3 d, f3 &P, int &face) {
    f3 ro = mul_point(inv, o);
    f3 p1 = mul_point(inv, o + d);
    f3 rd = normalize(p1 - ro);
    float ix = 1.0f / rd.x, iy = 1.0f / rd.y, iz = 1.0f / rd.z;
    float tmin, tmax, tymin, tymax, tzmin, tzmax;
    if (!(ix < 0.0f)) { tmin = (-.5f - ro.x) * ix; tmax = (.5f - ro.x) * ix; }
    else              { tmin = (.5f - ro.x) * ix;  tmax = (-.5f - ro.x) * ix; }
    if (!(iy < 0.0f)) { tymin = (-.5f - ro.y) * iy; tymax = (.5f - ro.y) * iy; }
    else              { tymin = (.5f - ro.y) * iy;  tymax = (-.5f - ro.y) * iy; }
    if ((tmin > tymax) || (tymin > tmax)) return -1.0f;
    if (tymin > tmin) tmin = tymin;
    if (tymax < tmax) tmax = tymax;
    if (!(iz < 0.0f)) { tzmin = (-.5f - ro.z) * iz; tzmax = (.5f - ro.z) * iz; }
    else              { tzmin = (.5f - ro.z) * iz;  tzmax = (-.5f - ro.z) * iz; }
    if ((tmin > tzmax) || (tzmin > tmax)) return -1.0f;
    if (tzmin > tmin) tmin = tzmin;
    if (tzmax < tmax) tmax = tzmax;
    float th = tmin;
    if (tmin < 0.0f) {
        if (!inside_hits || !(tmax > 0.0f)) return -1.0f;
        th = tmax;
    }
    f3 os = ro + rd * th;
    int fc = -1;
    if (fabsf(os.x - .5f) < .001f) { fc = 0; }
    else if (fabsf(os.y - .5f) < .001f) { fc = 1; }
    else if (fabsf(os.z - .5f) < .001f) { fc = 2; }
    else if (fabsf(os.x + .5f) < .001f) { fc = 3; }
    else if (fabsf(os.y + .5f) < .001f) { fc = 4; }
    else if (fabsf(os.z + .5f) < .001f) { fc = 5; }
    f3 wp = mul_point(xf, os);
    P = wp;
    face = fc;
    return length(wp - o);
}
// the reference's normal of a box face: +-column of transform, unnormalised (:161); (0,0,0) for face -1
PTD_FN f3 box_face_normal(const float *xf, int face) {
    if (face < 0) return mk(0.0f, 0.0f, 0.0f);
    const int col = face >= 3 ? face - 3 : face;
    const float sgn = face >= 3 ? -1.0f : 1.0f;
    return mk(xf[col] * sgn, xf[4 + col] * sgn, xf[8 + col] * sgn);
}
PTD_FN float box_test(const float *inv, const float *xf, int inside_hits, f3 o, f3 d, f3 &P, f3 &N) {
    int face = -1;
    f3 p = mk(0.0f, 0.0f, 0.0f);
    const float depth = box_test_face(inv, xf, inside_hits, o, d, p, face);
    if (depth < 0.0f) return depth;                      // a miss leaves P, N untouched like the reference
    P = p;
    N = box_face_normal(xf, face);
    return depth;
}

// ---------------------------------------------------------------- MESH (build-defined) --
// The reference declares GEOMTYPE MESH and leaves its kernel branch empty (src/raytraceKernel.cu:144-145); the
// build's definition is DESIGN.md section 3.8 (oracle: orc_triangle_test / orc_mesh_test).  One triangle as the
// kernels read it: v0, the two edges, the object-space geometric normal e1 x e2 (all evaluated by the host with
// the functions of this header) and the triangle's index in the file (ties go to the earlier triangle): 64 bytes.
struct __attribute__((aligned(16))) MeshTri {
    float v0[3]; int index;
    float e1[3]; float pad1;
    float e2[3]; float pad2;
    float ng[3]; float pad3;
};
// threaded BVH node (depth-first order, the child on the lower side of the cut first).  A ray visits the children of an inner
// node NEAR SIDE FIRST -- the lower one first if it travels up the cut's axis, else the higher one -- so that a hit found early
// prunes the far side; the order therefore depends on the octant of the ray's direction (bit k: d[k] < 0), and so does the node
// at which traversal continues once a node's subtree is finished or the node is missed: skip[octant] (-1 = done).
// leaf >= 0: first triangle | count << 27; leaf < 0: inner, far = axis of the cut | index of the higher child << 2 (the lower
// child is the next node).  64 bytes.
struct __attribute__((aligned(16))) MeshNode {
    float bmin[3]; int leaf;
    float bmax[3]; int far;
    int skip[8];
};

// two-sided Moeller-Trumbore in this exact expression order; returns t > 0 or -1
PTD_FN float triangle_test(f3 v0, f3 e1, f3 e2, f3 ro, f3 rd) {
    const f3 pvec = cross(rd, e2);
    const float det = dot(e1, pvec);
    if (det == 0.0f) return -1.0f;
    const float inv_det = 1.0f / det;
    const f3 tvec = ro - v0;
    const float u = dot(tvec, pvec) * inv_det;
    if (u < 0.0f || u > 1.0f) return -1.0f;
    const f3 qvec = cross(tvec, e1);
    const float v = dot(rd, qvec) * inv_det;
    if (v < 0.0f || u + v > 1.0f) return -1.0f;
    const float t = dot(e2, qvec) * inv_det;
    if (!(t > 0.0f)) return -1.0f;
    return t;
}
// hit point and normal of the winning triangle (object-space t, geometric normal ng)
PTD_FN float mesh_finish(const float *inv, const float *xf, f3 o, f3 ro, f3 rd, float t, f3 ng, f3 &P, f3 &N) {
    const float tt = (float)((double)t - .0001);
    const f3 wp = mul_point(xf, ro + rd * tt);
    const f3 n = mk((inv[0] * ng.x + inv[4] * ng.y) + inv[8] * ng.z,
                    (inv[1] * ng.x + inv[5] * ng.y) + inv[9] * ng.z,
                    (inv[2] * ng.x + inv[6] * ng.y) + inv[10] * ng.z);
    P = wp;
    N = normalize(n);
    return length(o - wp);
}

// ---------------------------------------------------------------- light sampling -------
// getRadiuses / getRandomPointOnCube / getRandomPointOnSphere (src/intersections.h:207-286): no
// call sites in the reference; provided (and parity-tested against the oracle) for next-event
// estimation.  Quirks kept: float seed truncated to unsigned, sphere sampler non-uniform / NaN-prone.
PTD_FN f3 get_radiuses(const float *xf) {
    const f3 origin = mul_point(xf, mk(0.0f, 0.0f, 0.0f));
    const f3 xmax = mul_point(xf, mk(.5f, 0.0f, 0.0f));
    const f3 ymax = mul_point(xf, mk(0.0f, .5f, 0.0f));
    const f3 zmax = mul_point(xf, mk(0.0f, 0.0f, .5f));
    return mk(length(xmax - origin), length(ymax - origin), length(zmax - origin));
}
PTD_FN float dist_ab(uint32_t x, float a, float b) { return (u01(x) * (b - a)) + a; }

PTD_FN f3 random_point_on_cube(const float *xf, float randomSeed) {
    uint32_t st = lcg_seed(hash((uint32_t)randomSeed));
    const f3 radii = get_radiuses(xf);
    const float side1 = radii.x * radii.y * 4.0f;
    const float side2 = radii.z * radii.y * 4.0f;
    const float side3 = radii.x * radii.z * 4.0f;
    const float totalarea = 2.0f * (side1 + side2 + side3);
    st = lcg_next(st); const float rr = u01(st);
    st = lcg_next(st); const float a = dist_ab(st, -0.5f, 0.5f);
    st = lcg_next(st); const float b = dist_ab(st, -0.5f, 0.5f);
    f3 point;
    if (rr < (side1 / totalarea)) point = mk(a, b, .5f);
    else if (rr < ((side1 * 2) / totalarea)) point = mk(a, b, -.5f);
    else if (rr < (((side1 * 2) + (side2)) / totalarea)) point = mk(.5f, a, b);
    else if (rr < (((side1 * 2) + (side2 * 2)) / totalarea)) point = mk(-.5f, a, b);
    else if (rr < (((side1 * 2) + (side2 * 2) + (side3)) / totalarea)) point = mk(a, .5f, b);
    else point = mk(a, -.5f, b);
    return mul_point(xf, point);
}

PTD_FN f3 random_point_on_sphere(const float *xf, float randomSeed) {
    uint32_t st = lcg_seed(hash((uint32_t)randomSeed));
    st = lcg_next(st); const float x = dist_ab(st, -0.5f, 0.5f);
    st = lcg_next(st); const float y = dist_ab(st, -0.5f, 0.5f);
    st = lcg_next(st); const float rr = u01(st);
    float z = __builtin_sqrtf(0.5f * 0.5f - x * x - y * y);
    if (!(rr < 0.5f)) z = -z;
    return mul_point(xf, mk(x, y, z));
}

// Light sample for next-event estimation: the reference's samplers above plus
// the reciprocal of the density they induce per unit world area (cube: 1/totalarea; sphere: |z|
// on the unit-diameter sphere over the area scale (2*radii.x)^2).  false = unusable sample.
PTD_FN bool sample_light(const float *xf, int type, float randomSeed, f3 &Q, float &inv_pdf_area) {
    const f3 radii = get_radiuses(xf);
    bool ok = false;
    Q = mk(0.0f, 0.0f, 0.0f);
    inv_pdf_area = 0.0f;
    if (type == 1) {
        const float side1 = radii.x * radii.y * 4.0f;
        const float side2 = radii.z * radii.y * 4.0f;
        const float side3 = radii.x * radii.z * 4.0f;
        const float totalarea = 2.0f * (side1 + side2 + side3);
        Q = random_point_on_cube(xf, randomSeed);
        inv_pdf_area = totalarea;
        ok = totalarea > 0.0f;
    } else if (type == 0) {
        uint32_t st = lcg_seed(hash((uint32_t)randomSeed));
        st = lcg_next(st); const float x = dist_ab(st, -0.5f, 0.5f);
        st = lcg_next(st); const float y = dist_ab(st, -0.5f, 0.5f);
        const float z = __builtin_sqrtf(0.5f * 0.5f - x * x - y * y);
        Q = random_point_on_sphere(xf, randomSeed);
        const float s = 2.0f * radii.x;
        inv_pdf_area = (s * s) / z;
        ok = z > 0.0f;
    }
    return ok;
}

// ---------------------------------------------------------------- scatter --------------
// calculateRandomDirectionInHemisphere (src/interactions.h:62-87), in two halves: the tangent frame depends only
// on the normal (tabulated per box face by the host, pt_api.hip), the combination on the two random numbers.
PTD_FN void hemisphere_frame(f3 normal, f3 &p1, f3 &p2) {
    f3 dnn;
    if (fabsf(normal.x) < PT_SQRT_OF_ONE_THIRD) dnn = mk(1.0f, 0.0f, 0.0f);
    else if (fabsf(normal.y) < PT_SQRT_OF_ONE_THIRD) dnn = mk(0.0f, 1.0f, 0.0f);
    else dnn = mk(0.0f, 0.0f, 1.0f);
    p1 = normalize(cross(normal, dnn));
    p2 = normalize(cross(normal, p1));
}
PTD_FN f3 hemisphere_combine(f3 normal, f3 p1, f3 p2, float xi1, float xi2) {
    float up = __builtin_sqrtf(xi1);
    float over = __builtin_sqrtf(1.0f - up * up);
    float around = xi2 * PT_TWO_PI;
    float sn, cs;
    sincos_poly(around, sn, cs);
    return ((normal * up) + (p1 * (cs * over))) + (p2 * (sn * over));
}
PTD_FN f3 hemisphere(f3 normal, float xi1, float xi2) {
    f3 p1, p2;
    hemisphere_frame(normal, p1, p2);
    return hemisphere_combine(normal, p1, p2, xi1, xi2);
}
// calculateReflectionDirection (stub src/interactions.h:47-50)
PTD_FN f3 reflect_dir(f3 n, f3 i) {
    float k = 2.0f * dot(n, i);
    return i - n * k;
}
// calculateFresnel + calculateTransmissionDirection (stubs src/interactions.h:42-44,53-59)
// evaluated together: returns the reflection coefficient (1 = total internal reflection) and
// the transmitted direction.
PTD_FN float fresnel_transmit(f3 n, f3 i, float ior_i, float ior_t, f3 &tdir) {
    float eta = ior_i / ior_t;
    float c = -dot(n, i);
    float k = 1.0f - ((eta * eta) * (1.0f - (c * c)));
    if (k < 0.0f) { tdir = mk(0.0f, 0.0f, 0.0f); return 1.0f; }
    float ct = __builtin_sqrtf(k);
    float rs = ((ior_i * c) - (ior_t * ct)) / ((ior_i * c) + (ior_t * ct));
    float rp = ((ior_i * ct) - (ior_t * c)) / ((ior_i * ct) + (ior_t * c));
    float R = 0.5f * ((rs * rs) + (rp * rp));
    float a = (eta * c) - ct;
    tdir = (i * eta) + (n * a);
    return R;
}

// The calculateBSDF contract (stub src/interactions.h:96-103; spec DESIGN.md section 3.5).
// Returns 0 diffuse, 1 reflected, 2 transmitted, 3 ended on an emitter (L set), 4 degenerate.
// (Single exit with value selects: keeps o/d/thr in registers instead of scratch.)
PTD_FN int scatter(const MatRec &m, f3 P, f3 N, float u_sel, float xi1, float xi2,
                                       f3 &o, f3 &d, f3 &thr, f3 &L) {
    int code;
    f3 no = o, nd = d, nthr = thr, nL = mk(0.0f, 0.0f, 0.0f);
    const float nn = dot(N, N);
    if (m.emittance > 0.0f) {
        const f3 e = mk(m.color[0], m.color[1], m.color[2]) * m.emittance;
        nL = thr * e;
        code = 3;
    } else if (!(nn > 0.0f)) {
        code = 4;
    } else {
        const f3 n = N * (1.0f / __builtin_sqrtf(nn));
        const float cosi = dot(n, d);
        const f3 nf = (cosi > 0.0f) ? neg(n) : n;
        const f3 spec = mk(m.spec[0], m.spec[1], m.spec[2]);
        if (m.refr > 0.0f) {
            const float ior = (m.ior > 0.0f) ? m.ior : 1.0f;
            const bool entering = !(cosi > 0.0f);
            const float ior_i = entering ? 1.0f : ior, ior_t = entering ? ior : 1.0f;
            f3 tdir;
            const float R = fresnel_transmit(nf, d, ior_i, ior_t, tdir);
            nthr = thr * spec;
            const bool refl = u_sel < R;
            const f3 rdir = reflect_dir(nf, d);
            const f3 o_r = P + nf * PT_RAY_BIAS, o_t = P - nf * PT_TRANSMIT_BIAS;
            nd = refl ? rdir : tdir;
            no = refl ? o_r : o_t;
            code = refl ? 1 : 2;
        } else if (m.refl > 0.0f) {
            nd = reflect_dir(nf, d);
            nthr = thr * spec;
            no = P + nf * PT_RAY_BIAS;
            code = 1;
        } else {
            nd = hemisphere(nf, xi1, xi2);
            nthr = thr * mk(m.color[0], m.color[1], m.color[2]);
            no = P + nf * PT_RAY_BIAS;
            code = 0;
        }
    }
    o = no; d = nd; thr = nthr; L = nL;
    return code;
}

// Shading frames of a box primitive, one record per axis (column of transform): everything scatter() derives
// from the face normal alone.  A hit on the +-`col` face has N = +-column (box_face_normal), so
// n = N/|N| = +-n and nf, the normal facing the incoming ray, is n or its exact negation; the tangent frame
// hemisphere_frame(nf) has two possible values per axis.  The host fills these with the functions above
// (same expression trees, same compiler run): the bits are the ones the generic path computes per ray.
struct __attribute__((aligned(16))) FaceFrame {
    float n[4];               // normalize(column); n[3] = 1 if |column|^2 > 0 else 0 (degenerate: scatter code 4)
    float pp[6];              // hemisphere_frame(+n): p1, p2
    float pm[6];              // hemisphere_frame(-n): p1, p2
};                            // 64 bytes

PTD_FN void make_face_frame(const float *xf, int col, FaceFrame *out) {
    const f3 N = mk(xf[col], xf[4 + col], xf[8 + col]);
    const float nn = dot(N, N);
    const f3 n = N * (1.0f / __builtin_sqrtf(nn));
    f3 a, b, c, d;
    hemisphere_frame(n, a, b);
    hemisphere_frame(neg(n), c, d);
    out->n[0] = n.x; out->n[1] = n.y; out->n[2] = n.z; out->n[3] = nn > 0.0f ? 1.0f : 0.0f;
    out->pp[0] = a.x; out->pp[1] = a.y; out->pp[2] = a.z; out->pp[3] = b.x; out->pp[4] = b.y; out->pp[5] = b.z;
    out->pm[0] = c.x; out->pm[1] = c.y; out->pm[2] = c.z; out->pm[3] = d.x; out->pm[4] = d.y; out->pm[5] = d.z;
}

// scatter() for a box hit given as (face, the primitive's three FaceFrames): same results bit for bit, without the
// per-ray normalisation and tangent frame (2 of the 3 normalize + both cross products of the generic path).
PTD_FN int scatter_box(const MatRec &m, f3 P, int face, const FaceFrame *frames, float u_sel, float xi1, float xi2,
                       f3 &o, f3 &d, f3 &thr, f3 &L) {
    int code;
    f3 no = o, nd = d, nthr = thr, nL = mk(0.0f, 0.0f, 0.0f);
    if (m.emittance > 0.0f) {
        const f3 e = mk(m.color[0], m.color[1], m.color[2]) * m.emittance;
        nL = thr * e;
        code = 3;
    } else {
        const int col = face >= 3 ? face - 3 : (face < 0 ? 0 : face);
        const FaceFrame *F = frames + col;
        const float nvx = F->n[0], nvy = F->n[1], nvz = F->n[2], nvw = F->n[3];
        if (face < 0 || !(nvw > 0.0f)) {
            code = 4;
        } else {
            const bool minus = face >= 3;
            const f3 np = mk(nvx, nvy, nvz);
            const f3 n = minus ? neg(np) : np;
            const float cosi = dot(n, d);
            const bool back = cosi > 0.0f;
            const f3 nf = back ? neg(n) : n;
            const f3 spec = mk(m.spec[0], m.spec[1], m.spec[2]);
            if (m.refr > 0.0f) {
                const float ior = (m.ior > 0.0f) ? m.ior : 1.0f;
                const bool entering = !back;
                const float ior_i = entering ? 1.0f : ior, ior_t = entering ? ior : 1.0f;
                f3 tdir;
                const float R = fresnel_transmit(nf, d, ior_i, ior_t, tdir);
                nthr = thr * spec;
                const bool refl = u_sel < R;
                const f3 rdir = reflect_dir(nf, d);
                const f3 o_r = P + nf * PT_RAY_BIAS, o_t = P - nf * PT_TRANSMIT_BIAS;
                nd = refl ? rdir : tdir;
                no = refl ? o_r : o_t;
                code = refl ? 1 : 2;
            } else if (m.refl > 0.0f) {
                nd = reflect_dir(nf, d);
                nthr = thr * spec;
                no = P + nf * PT_RAY_BIAS;
                code = 1;
            } else {
                const bool flip = minus != back;                   // nf == -normalize(column)
                const float *fr = flip ? F->pm : F->pp;
                nd = hemisphere_combine(nf, mk(fr[0], fr[1], fr[2]), mk(fr[3], fr[4], fr[5]), xi1, xi2);
                nthr = thr * mk(m.color[0], m.color[1], m.color[2]);
                no = P + nf * PT_RAY_BIAS;
                code = 0;
            }
        }
    }
    o = no; d = nd; thr = nthr; L = nL;
    return code;
}

// global pixel index (x + y*W, y owned by this context) -> index among the context's owned pixels (local row * W + x):
// the per-iteration accumulator planes of a launch group hold only the owned rows
PTD_FN uint32_t owned_index(const CamRec &c, uint32_t pixel) {
    const uint32_t y = (uint32_t)(((unsigned long long)pixel * c.mW) >> c.shW);
    const uint32_t x = pixel - y * (uint32_t)c.W;
    const uint32_t ly = (uint32_t)(((unsigned long long)(y - (uint32_t)c.row_offset) * c.mS) >> c.shS);
    return ly * (uint32_t)c.W + x;
}

// ---------------------------------------------------------------- camera ---------------
// per-pixel half of raycastFromCameraKernel (src/raytraceKernel.cu:62-74)
PTD_FN void camera_ray(const CamRec &c, uint32_t pixel, uint32_t iteration, f3 &o, f3 &d) {
    int x = (int)(pixel % (uint32_t)c.W), y = (int)(pixel / (uint32_t)c.W);
    f3 E = mk(c.E[0], c.E[1], c.E[2]);
    float fx = (float)x, fy = (float)y;
    float lu = 0.0f, lv = 0.0f;
    bool lens = (c.camera_mode == 1) && (c.aperture > 0.0f);
    if (c.antialias || lens) {
        uint32_t st = lcg_seed(stream_seed(pixel, iteration, 0u));
        st = lcg_next(st); float jx = u01(st) - 0.5f;
        st = lcg_next(st); float jy = u01(st) - 0.5f;
        st = lcg_next(st); lu = u01(st);
        st = lcg_next(st); lv = u01(st);
        if (c.antialias) { fx = fx + jx; fy = fy + jy; }
    }
    float sx = fx / c.wm1;
    float sy = fy / c.hm1;
    f3 P = (mk(c.M[0], c.M[1], c.M[2]) + mk(c.H[0], c.H[1], c.H[2]) * ((2.0f * sx) - 1.0f))
           + mk(c.V[0], c.V[1], c.V[2]) * ((2.0f * sy) - 1.0f);
    f3 PmE = P - E;
    f3 oo = E, dd;
    if (c.camera_mode == 0) {
        f3 R = E + (PmE * 200.0f) / length(PmE);
        dd = normalize(R);                 // the reference normalises the POINT R (:67-69)
    } else {
        f3 dn = normalize(PmE);
        dd = dn;
        if (lens) {
            f3 Cn = mk(c.Cn[0], c.Cn[1], c.Cn[2]), Ah = mk(c.Ah[0], c.Ah[1], c.Ah[2]), Bh = mk(c.Bh[0], c.Bh[1], c.Bh[2]);
            float tf = c.focal / dot(dn, Cn);
            f3 F = E + dn * tf;
            float rr = c.aperture * __builtin_sqrtf(lu);
            float sn, cs;
            sincos_poly(lv * PT_TWO_PI, sn, cs);
            f3 Eo = (E + Ah * (rr * cs)) + Bh * (rr * sn);
            oo = Eo;
            dd = normalize(F - Eo);
        }
    }
    o = oo;
    d = dd;
}

}  // namespace ptd
