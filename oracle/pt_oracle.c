/*
 * pt_oracle.c -- CPU restatement (plain C) of the CIS565 Project2-Pathtracer hot path.
 * TEST INFRASTRUCTURE ONLY -- see pt_oracle.h for scope, citations and pinning status.
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math -fopenmp -fPIC -shared (oracle/Makefile).
 * Every float expression below is written with explicit parentheses in the order the
 * reference's C++ evaluates it (left-to-right), so that the HIP kernels can mirror it
 * operation for operation.
 */
#include "pt_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* src/utilities.h:20-26 -- float constants exactly as the reference declares them */
static const float ORC_PI = 3.1415926535897932384626422832795028841971f;
static const float ORC_TWO_PI = 6.2831853071795864769252867665590057683943f;
static const float ORC_SQRT_OF_ONE_THIRD = 0.5773502691896257645091487805019574556476f;
static const float ORC_EPSILON = .000000001f;
static const float ORC_RAY_BIAS = 0.0002f;
static const float ORC_TRANSMIT_BIAS = 0.001f; /* build-defined, DESIGN.md section 3.6 */

typedef struct { float x, y, z; } v3;

static inline v3 V(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 vadd(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 vsub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 vscale(v3 a, float s) { return V(a.x * s, a.y * s, a.z * s); }
static inline v3 vdivs(v3 a, float s) { return V(a.x / s, a.y / s, a.z / s); }
static inline v3 vneg(v3 a) { return V(-a.x, -a.y, -a.z); }
static inline v3 vmul(v3 a, v3 b) { return V(a.x * b.x, a.y * b.y, a.z * b.z); }
/* glm::dot  src/glm/core/func_geometric.inl:158-167 */
static inline float vdot(v3 a, v3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
/* glm::cross  src/glm/core/func_geometric.inl:199-211 */
static inline v3 vcross(v3 x, v3 y) {
    return V(x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y);
}
/* glm::length / glm::normalize  src/glm/core/func_geometric.inl:59-68, 239-248;
 * inversesqrt = 1/sqrt  src/glm/core/func_exponential.inl:145-153 */
static inline float vlength(v3 a) { return sqrtf(vdot(a, a)); }
static inline v3 vnormalize(v3 a) { return vscale(a, 1.0f / sqrtf(vdot(a, a))); }
static inline v3 vload(const float *p) { return V(p[0], p[1], p[2]); }
static inline void vstore(float *p, v3 a) { p[0] = a.x; p[1] = a.y; p[2] = a.z; }

/* ------------------------------------------------------------------ RNG ---------- */

/* src/intersections.h:26-34 */
uint32_t orc_hash(uint32_t a) {
    a = (a + 0x7ed55d16u) + (a << 12);
    a = (a ^ 0xc761c23cu) ^ (a >> 19);
    a = (a + 0x165667b1u) + (a << 5);
    a = (a + 0xd3a2646cu) ^ (a << 9);
    a = (a + 0xfd7046c5u) + (a << 3);
    a = (a ^ 0xb55a4f09u) ^ (a >> 16);
    return a;
}

/* minstd_rand seeding: x = s mod m, and 0 -> 1 because c == 0
 * (linear_congruential_engine::seed, rocThrust random/detail/linear_congruential_engine.inl:43-50) */
uint32_t orc_lcg_seed(uint32_t s) {
    uint32_t x = s % 2147483647u;
    return x == 0u ? 1u : x;
}

uint32_t orc_lcg_next(uint32_t x) {
    return (uint32_t)(((uint64_t)x * 48271ull) % 2147483647ull);
}

/* uniform_real_distribution<float>(0,1): float(x - min) / (1 + float(max - min)), min = 1,
 * max = m - 1; the denominator rounds to 2^31 in binary32. */
float orc_u01(uint32_t x) {
    return (float)(x - 1u) / 2147483648.0f;
}

/* Build-defined integer stream seed: one independent minstd stream per
 * (global pixel index, 1-based iteration, stream id); stream 0 = camera sample,
 * stream 1+b = scatter at bounce b.  One hash (a bijection on 32 bits) of an injective mix. */
uint32_t orc_stream_seed(uint32_t pixel, uint32_t iteration, uint32_t stream) {
    return orc_hash(pixel + 0x9E3779B9u * iteration + 0x85EBCA6Bu * stream);
}

/* src/raytraceKernel.cu:30-37: seed = hash(index*time) with index int, time float, i.e. a
 * float product truncated to unsigned.  Defined only while the product is < 2^32. */
void orc_rng_from_thread(float resx, float resy, float time, int x, int y, float out[3]) {
    (void)resy;
    int index = (int)((float)x + ((float)y * resx));
    uint32_t s = (uint32_t)((float)index * time);
    uint32_t st = orc_lcg_seed(orc_hash(s));
    st = orc_lcg_next(st); out[0] = orc_u01(st);
    st = orc_lcg_next(st); out[1] = orc_u01(st);
    st = orc_lcg_next(st); out[2] = orc_u01(st);
}

/* sin and cos of a in [0, 2*pi] (a = xi2*TWO_PI, src/interactions.h:68,85) in binary32
 * +,-,* only: three-constant Cody-Waite reduction by pi/2, then degree-7 / degree-8
 * polynomials on [-pi/4, pi/4] (coefficients: Cephes single precision sinf/cosf). */
void orc_sincos(float a, float *s, float *c) {
    int k = (int)((a * 0.636619772367581343f) + 0.5f);
    float kf = (float)k;
    float r = ((a - (kf * 1.5703125f)) - (kf * 4.837512969970703125e-4f)) - (kf * 7.54978995489188216e-8f);
    float z = r * r;
    float sp = (((((-1.9515295891e-4f * z) + 8.3321608736e-3f) * z) - 1.6666654611e-1f) * z) * r + r;
    float cp = ((((((2.443315711809948e-5f * z) - 1.388731625493765e-3f) * z) + 4.166664568298827e-2f) * z) * z
                - (0.5f * z)) + 1.0f;
    switch (k & 3) {
        case 0: *s = sp;  *c = cp;  break;
        case 1: *s = cp;  *c = -sp; break;
        case 2: *s = -sp; *c = -cp; break;
        default: *s = -cp; *c = sp; break;
    }
}

/* ------------------------------------------------------------------ camera ------- */

/* host half of raycastFromCameraKernel (src/raytraceKernel.cu:47-60) */
void orc_camera_setup(const orc_camera *cam, orc_camera_basis *b) {
    v3 E = vload(cam->position), C = vload(cam->view), U = vload(cam->up);
    float fovx = cam->fov[0], fovy = cam->fov[1];
    float CD = vlength(C);
    v3 A = vcross(C, U);
    v3 B = vcross(A, C);
    v3 M = vadd(E, C);
    v3 H = vdivs(vscale(A, CD * tanf(fovx * (ORC_PI / 180.0f))), vlength(A));
    v3 Vv = vdivs(vscale(B, CD * tanf(-fovy * (ORC_PI / 180.0f))), vlength(B));
    vstore(b->E, E); vstore(b->M, M); vstore(b->H, H); vstore(b->V, Vv);
    vstore(b->Cn, vdivs(C, CD));
    vstore(b->Ah, vdivs(A, vlength(A)));
    vstore(b->Bh, vdivs(B, vlength(B)));
    b->W = cam->resolution[0];
    b->Hres = cam->resolution[1];
    b->inv_wm1 = 0.0f; b->inv_hm1 = 0.0f;
}

/* per-pixel half of raycastFromCameraKernel (src/raytraceKernel.cu:62-74).
 * jx,jy: anti-alias offsets (0 when off); lu,lv: lens sample in [0,1) (thin lens only). */
void orc_camera_ray(const orc_camera_basis *b, const orc_config *cfg, int x, int y,
                    float jx, float jy, float lu, float lv, float origin[3], float dir[3]) {
    v3 E = vload(b->E), M = vload(b->M), H = vload(b->H), Vv = vload(b->V);
    float fx = (float)x, fy = (float)y;
    if (cfg && cfg->antialias) { fx = fx + jx; fy = fy + jy; }
    float sx = fx / (b->W - 1.0f);
    float sy = fy / (b->Hres - 1.0f);
    v3 P = vadd(vadd(M, vscale(H, (2.0f * sx) - 1.0f)), vscale(Vv, (2.0f * sy) - 1.0f));
    v3 PmE = vsub(P, E);
    if (!cfg || cfg->camera_mode == 0) {
        v3 R = vadd(E, vdivs(vscale(PmE, 200.0f), vlength(PmE)));
        vstore(dir, vnormalize(R));            /* the reference normalises the POINT R (:67-69) */
        vstore(origin, E);
        return;
    }
    v3 d = vnormalize(PmE);
    if (cfg->aperture > 0.0f) {
        /* thin lens, DESIGN.md section 3.3: focal point on the plane at focal_distance along view */
        v3 Cn = vload(b->Cn), Ah = vload(b->Ah), Bh = vload(b->Bh);
        float tf = cfg->focal_distance / vdot(d, Cn);
        v3 F = vadd(E, vscale(d, tf));
        float rr = cfg->aperture * sqrtf(lu);
        float sn, cs;
        orc_sincos(lv * ORC_TWO_PI, &sn, &cs);
        v3 Eo = vadd(vadd(E, vscale(Ah, rr * cs)), vscale(Bh, rr * sn));
        vstore(origin, Eo);
        vstore(dir, vnormalize(vsub(F, Eo)));
        return;
    }
    vstore(origin, E);
    vstore(dir, d);
}

/* ------------------------------------------------------------------ intersections - */

/* src/intersections.h:53-59 */
void orc_multiply_mv(const float m[16], const float v[4], float out[3]) {
    out[0] = (m[0] * v[0]) + (m[1] * v[1]) + (m[2] * v[2]) + (m[3] * v[3]);
    out[1] = (m[4] * v[0]) + (m[5] * v[1]) + (m[6] * v[2]) + (m[7] * v[3]);
    out[2] = (m[8] * v[0]) + (m[9] * v[1]) + (m[10] * v[2]) + (m[11] * v[3]);
}

static inline v3 mulmv(const float m[16], v3 v, float w) {
    float in[4] = {v.x, v.y, v.z, w}, out[3];
    orc_multiply_mv(m, in, out);
    return V(out[0], out[1], out[2]);
}

/* src/intersections.h:168-204.  `pow(radius,2)` is 0.25f (C++03 float overload, as the
 * reference's 2012 toolchain resolves it); `float(t-.0001)` (getPointOnRay :46-48) is a
 * double subtraction rounded once to float, kept as such.
 * int_minmax = 0: min/max are the float overloads CUDA puts in the global namespace (what
 * the reference's own toolchain does).  int_minmax = 1 reproduces the SURVEY's host-shim
 * build, where `min(t1,t2)`/`max(t1,t2)` resolved to HIP's host-side int overloads and
 * truncated the roots -- used ONLY by tests/test_oracle_kats.py to explain the three sphere
 * rows of SURVEY.md section 8c (t_obj there is exactly floor(near root)). */
static float sphere_test_impl(const orc_geom *g, int variant, const float o[3], const float d[3],
                              float P[3], float N[3]) {
    const int int_minmax = variant & 1, pow_double = variant & 2;
    const float radius = .5f;
    v3 ro = mulmv(g->inverseTransform, vload(o), 1.0f);
    v3 rd = vnormalize(mulmv(g->inverseTransform, vload(d), 0.0f));
    float vDotDirection = vdot(ro, rd);
    float radicand = vDotDirection * vDotDirection - (vdot(ro, ro) - (radius * radius));
    /* pow_double (tests only): C++11's pow(float,int) returns double, so a present-day host build of the
     * reference evaluates the two subtractions in double and rounds once (oracle/ref_kernels_probe.cpp) */
    if (pow_double) radicand = (float)((double)(vDotDirection * vDotDirection) - ((double)vdot(ro, ro) - 0.25));
    if (radicand < 0.0f) return -1.0f;
    float squareRoot = sqrtf(radicand);
    float firstTerm = -vDotDirection;
    float t1 = firstTerm + squareRoot;
    float t2 = firstTerm - squareRoot;
    float t;
    if (t1 < 0.0f && t2 < 0.0f) return -1.0f;
    else if (t1 > 0.0f && t2 > 0.0f) t = int_minmax ? (float)(((int)t1 < (int)t2) ? (int)t1 : (int)t2) : fminf(t1, t2);
    else t = int_minmax ? (float)(((int)t1 > (int)t2) ? (int)t1 : (int)t2) : fmaxf(t1, t2);
    /* getPointOnRay(rt, t): origin + float(t-.0001)*normalize(direction) */
    float tt = (float)((double)t - .0001);
    v3 pobj = vadd(ro, vscale(vnormalize(rd), tt));
    v3 realIntersectionPoint = mulmv(g->transform, pobj, 1.0f);
    v3 realOrigin = mulmv(g->transform, V(0.0f, 0.0f, 0.0f), 1.0f);
    vstore(P, realIntersectionPoint);
    vstore(N, vnormalize(vsub(realIntersectionPoint, realOrigin)));
    return vlength(vsub(vload(o), realIntersectionPoint));
}

float orc_sphere_test(const orc_geom *g, const float o[3], const float d[3], float P[3], float N[3]) {
    return sphere_test_impl(g, 0, o, d, P, N);
}

float orc_sphere_test_intminmax(const orc_geom *g, const float o[3], const float d[3], float P[3], float N[3]) {
    return sphere_test_impl(g, 1, o, d, P, N);
}

float orc_sphere_test_powdouble(const orc_geom *g, const float o[3], const float d[3], float P[3], float N[3]) {
    return sphere_test_impl(g, 2, o, d, P, N);
}

/* getPointOnRay (src/intersections.h:46-48): origin + float(t - .0001) * normalize(direction) */
void orc_point_on_ray(const float o[3], const float d[3], float t, float out[3]) {
    float tt = (float)((double)t - .0001);
    vstore(out, vadd(vload(o), vscale(vnormalize(vload(d)), tt)));
}

/* src/intersections.h:73-164 (unit cube [-.5,.5]^3).  inside_hits = 0 is the reference
 * (tmin<0 -> miss, :138-140); inside_hits = 1 (only passed for refractive materials) is
 * the build's extension: a ray starting inside leaves through the tmax face. */
float orc_box_test(const orc_geom *g, int inside_hits, const float o[3], const float d[3],
                   float P[3], float N[3]) {
    v3 O = vload(o), D = vload(d);
    v3 iP0 = mulmv(g->inverseTransform, O, 1.0f);
    v3 iP1 = mulmv(g->inverseTransform, vadd(O, D), 1.0f);
    v3 ro = iP0;
    v3 rd = vnormalize(vsub(iP1, iP0));
    float ix = 1.0f / rd.x, iy = 1.0f / rd.y, iz = 1.0f / rd.z;   /* 1.0/x in double == float division */
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
    v3 os = vadd(ro, vscale(rd, th));
    v3 n = V(0.0f, 0.0f, 0.0f);
    if (fabsf(os.x - .5f) < .001f) n = V(1.0f, 0.0f, 0.0f);
    else if (fabsf(os.y - .5f) < .001f) n = V(0.0f, 1.0f, 0.0f);
    else if (fabsf(os.z - .5f) < .001f) n = V(0.0f, 0.0f, 1.0f);
    else if (fabsf(os.x + .5f) < .001f) n = V(-1.0f, 0.0f, 0.0f);
    else if (fabsf(os.y + .5f) < .001f) n = V(0.0f, -1.0f, 0.0f);
    else if (fabsf(os.z + .5f) < .001f) n = V(0.0f, 0.0f, -1.0f);
    v3 ip = mulmv(g->transform, os, 1.0f);
    vstore(P, ip);
    vstore(N, mulmv(g->transform, n, 0.0f));   /* not inverse-transpose, not normalised (:161) */
    return vlength(vsub(ip, O));
}

/* ------------------------------------------------------------------ MESH (build-defined) - */
/* The reference declares GEOMTYPE MESH (src/sceneStructs.h:14), tags `*.obj` objects in the parser
 * (src/scene.cpp:55-64) and leaves the kernel branch empty (src/raytraceKernel.cu:144-145).  The build's
 * definition, restated here as a brute-force loop over the triangles in file order:
 *   object-space ray as in the sphere test (:171-172): ro = M^-1 (o,1), rd = normalize(M^-1 (d,0));
 *   per triangle (v0, e1 = v1-v0, e2 = v2-v0): two-sided Moeller-Trumbore with this exact expression order;
 *   nearest t > 0 wins, a tie goes to the earlier triangle;
 *   hit point like getPointOnRay (:46-48) without its second normalize: P = M (ro + float(t-.0001) rd, 1);
 *   normal = normalize((M^-1)^T (e1 x e2)); returned depth = |o - P| in world space like the other tests. */
typedef struct {
    int geom_index;
    const float *vertices;      /* nvertices x 3, object space */
    int nvertices;
    const int *indices;         /* ntriangles x 3 */
    int ntriangles;
} orc_mesh_rec;

static orc_mesh_rec g_meshes[64];
static int g_nmeshes = 0;

/* Registers the meshes of the scene about to be rendered (pointers are kept, not copied); n = 0 clears.
 * A MESH primitive without a registered mesh is skipped like in the reference. */
int orc_set_meshes(const int *geom_index, const float *const *vertices, const int *nvertices,
                   const int *const *indices, const int *ntriangles, int n) {
    if (n < 0 || n > 64) return -1;
    for (int i = 0; i < n; i++) {
        g_meshes[i].geom_index = geom_index[i];
        g_meshes[i].vertices = vertices[i]; g_meshes[i].nvertices = nvertices[i];
        g_meshes[i].indices = indices[i]; g_meshes[i].ntriangles = ntriangles[i];
    }
    g_nmeshes = n;
    return 0;
}

/* one triangle; returns t (object space) or -1 */
float orc_triangle_test(const float v0a[3], const float e1a[3], const float e2a[3], const float roa[3], const float rda[3]) {
    v3 v0 = vload(v0a), e1 = vload(e1a), e2 = vload(e2a), ro = vload(roa), rd = vload(rda);
    v3 pvec = vcross(rd, e2);
    float det = vdot(e1, pvec);
    if (det == 0.0f) return -1.0f;
    float inv_det = 1.0f / det;
    v3 tvec = vsub(ro, v0);
    float u = vdot(tvec, pvec) * inv_det;
    if (u < 0.0f || u > 1.0f) return -1.0f;
    v3 qvec = vcross(tvec, e1);
    float v = vdot(rd, qvec) * inv_det;
    if (v < 0.0f || u + v > 1.0f) return -1.0f;
    float t = vdot(e2, qvec) * inv_det;
    if (!(t > 0.0f)) return -1.0f;
    return t;
}

float orc_mesh_test(const orc_geom *g, const float *vertices, const int *indices, int ntriangles,
                    const float o[3], const float d[3], float P[3], float N[3], int *triangle) {
    v3 ro = mulmv(g->inverseTransform, vload(o), 1.0f);
    v3 rd = vnormalize(mulmv(g->inverseTransform, vload(d), 0.0f));
    float roa[3], rda[3];
    vstore(roa, ro); vstore(rda, rd);
    float best = -1.0f;
    int win = -1;
    v3 wng = V(0.0f, 0.0f, 0.0f);
    for (int k = 0; k < ntriangles; k++) {
        v3 v0 = vload(vertices + 3 * indices[3 * k]);
        v3 e1 = vsub(vload(vertices + 3 * indices[3 * k + 1]), v0);
        v3 e2 = vsub(vload(vertices + 3 * indices[3 * k + 2]), v0);
        float v0a[3], e1a[3], e2a[3];
        vstore(v0a, v0); vstore(e1a, e1); vstore(e2a, e2);
        float t = orc_triangle_test(v0a, e1a, e2a, roa, rda);
        if (t > 0.0f && (win < 0 || t < best)) { best = t; win = k; wng = vcross(e1, e2); }
    }
    if (triangle) *triangle = win;
    if (win < 0) return -1.0f;
    float tt = (float)((double)best - .0001);
    v3 pobj = vadd(ro, vscale(rd, tt));
    v3 wp = mulmv(g->transform, pobj, 1.0f);
    const float *m = g->inverseTransform;
    v3 n = V((m[0] * wng.x + m[4] * wng.y) + m[8] * wng.z,
             (m[1] * wng.x + m[5] * wng.y) + m[9] * wng.z,
             (m[2] * wng.x + m[6] * wng.y) + m[10] * wng.z);
    vstore(P, wp);
    vstore(N, vnormalize(n));
    return vlength(vsub(vload(o), wp));
}

/* geometry loop of raytraceRay (src/raytraceKernel.cu:134-153): first strictly nearer wins */
int orc_nearest_hit(const orc_geom *geoms, int ngeoms, const orc_material *mats,
                    const float o[3], const float d[3], float *t, float P[3], float N[3]) {
    float MAX_DEPTH = 100000000000000000.0f;
    int hit = -1;
    for (int i = 0; i < ngeoms; i++) {
        float p[3], n[3], depth = -1.0f;
        if (geoms[i].type == 0) depth = orc_sphere_test(&geoms[i], o, d, p, n);
        else if (geoms[i].type == 1) {
            int inside = mats ? (mats[geoms[i].materialid].hasRefractive > 0.0f) : 0;
            depth = orc_box_test(&geoms[i], inside, o, d, p, n);
        } else {
            /* MESH: empty branch in the reference (:144-145); the build's definition when a mesh is registered */
            const orc_mesh_rec *mr = NULL;
            for (int k = 0; k < g_nmeshes; k++)
                if (g_meshes[k].geom_index == i) mr = &g_meshes[k];
            if (!mr || geoms[i].type != 2) continue;
            depth = orc_mesh_test(&geoms[i], mr->vertices, mr->indices, mr->ntriangles, o, d, p, n, NULL);
        }
        if (depth < MAX_DEPTH && depth > -ORC_EPSILON) {
            MAX_DEPTH = depth;
            hit = i;
            memcpy(P, p, sizeof p);
            memcpy(N, n, sizeof n);
        }
    }
    if (t) *t = MAX_DEPTH;
    return hit;
}

/* ------------------------------------------------------------------ light sampling - */

/* getRadiuses (src/intersections.h:207-216): half-dimensions of the transformed unit cube */
void orc_get_radiuses(const orc_geom *g, float out[3]) {
    v3 origin = mulmv(g->transform, V(0.0f, 0.0f, 0.0f), 1.0f);
    v3 xmax = mulmv(g->transform, V(.5f, 0.0f, 0.0f), 1.0f);
    v3 ymax = mulmv(g->transform, V(0.0f, .5f, 0.0f), 1.0f);
    v3 zmax = mulmv(g->transform, V(0.0f, 0.0f, .5f), 1.0f);
    out[0] = vlength(vsub(xmax, origin));         /* glm::distance(p0,p1) = length(p1-p0) */
    out[1] = vlength(vsub(ymax, origin));
    out[2] = vlength(vsub(zmax, origin));
}

/* uniform_real_distribution<float>(a,b): u01*(b-a) + a */
static inline float dist_ab(uint32_t x, float a, float b) { return (orc_u01(x) * (b - a)) + a; }

/* getRandomPointOnCube (src/intersections.h:220-262).  No call sites in the reference; restated for
 * completeness ("parity unpinned": intersections.h cannot be built here).  `hash(randomSeed)` converts
 * the float seed to unsigned by truncation; the two u02 draws inside one glm::vec3(...) constructor are
 * unsequenced in C++ -- taken left to right. */
void orc_random_point_on_cube(const orc_geom *cube, float randomSeed, float out[3]) {
    uint32_t st = orc_lcg_seed(orc_hash((uint32_t)randomSeed));
    float radii[3];
    orc_get_radiuses(cube, radii);
    float side1 = radii[0] * radii[1] * 4.0f;
    float side2 = radii[2] * radii[1] * 4.0f;
    float side3 = radii[0] * radii[2] * 4.0f;
    float totalarea = 2.0f * (side1 + side2 + side3);
    st = orc_lcg_next(st);
    float russianRoulette = orc_u01(st);
    st = orc_lcg_next(st); float a = dist_ab(st, -0.5f, 0.5f);
    st = orc_lcg_next(st); float b = dist_ab(st, -0.5f, 0.5f);
    v3 point;
    if (russianRoulette < (side1 / totalarea)) point = V(a, b, .5f);
    else if (russianRoulette < ((side1 * 2) / totalarea)) point = V(a, b, -.5f);
    else if (russianRoulette < (((side1 * 2) + (side2)) / totalarea)) point = V(.5f, a, b);
    else if (russianRoulette < (((side1 * 2) + (side2 * 2)) / totalarea)) point = V(-.5f, a, b);
    else if (russianRoulette < (((side1 * 2) + (side2 * 2) + (side3)) / totalarea)) point = V(a, .5f, b);
    else point = V(a, -.5f, b);
    vstore(out, mulmv(cube->transform, point, 1.0f));
}

/* getRandomPointOnSphere (src/intersections.h:265-286): x,y ~ U(-.5,.5), z = +-sqrt(r^2-x^2-y^2) -- as
 * shipped this is neither uniform nor always real (x^2+y^2 can exceed r^2 = .25 -> NaN); restated as is. */
void orc_random_point_on_sphere(const orc_geom *sphere, float randomSeed, float out[3]) {
    const float radius = .5f;
    uint32_t st = orc_lcg_seed(orc_hash((uint32_t)randomSeed));
    st = orc_lcg_next(st); float x = dist_ab(st, -0.5f, 0.5f);
    st = orc_lcg_next(st); float y = dist_ab(st, -0.5f, 0.5f);
    st = orc_lcg_next(st); float russianRoulette = orc_u01(st);
    float z = sqrtf(radius * radius - x * x - y * y);
    if (!(russianRoulette < 0.5f)) z = -z;
    vstore(out, mulmv(sphere->transform, V(x, y, z), 1.0f));
}

/* Light sample for next-event estimation (build-defined): the reference's own
 * samplers above (src/intersections.h:220-286, written for exactly this use, no call sites there) and
 * the reciprocal of the density they induce per unit WORLD area.
 *   cube:   faces are picked by area and points are uniform on a face -> density 1/totalarea;
 *   sphere: (x,y) uniform on the unit square, z = +-sqrt(r^2-x^2-y^2) with probability 1/2 each ->
 *           density |z| per unit area of the unit-diameter sphere (dA = dx dy / |z/r|, r = 1/2), divided
 *           by the area scale (2*radii.x)^2 of the (uniformly scaled) transform; samples outside the
 *           disk (NaN) are unusable and contribute nothing, which the density accounts for. */
int orc_sample_light(const orc_geom *g, float randomSeed, float Q[3], float *inv_pdf_area) {
    float radii[3];
    orc_get_radiuses(g, radii);
    if (g->type == 1) {
        float side1 = radii[0] * radii[1] * 4.0f;
        float side2 = radii[2] * radii[1] * 4.0f;
        float side3 = radii[0] * radii[2] * 4.0f;
        float totalarea = 2.0f * (side1 + side2 + side3);
        orc_random_point_on_cube(g, randomSeed, Q);
        *inv_pdf_area = totalarea;
        return totalarea > 0.0f;
    }
    if (g->type == 0) {
        const float radius = .5f;
        uint32_t st = orc_lcg_seed(orc_hash((uint32_t)randomSeed));
        st = orc_lcg_next(st); float x = dist_ab(st, -0.5f, 0.5f);
        st = orc_lcg_next(st); float y = dist_ab(st, -0.5f, 0.5f);
        float z = sqrtf(radius * radius - x * x - y * y);      /* |z| of the point the sampler returns */
        orc_random_point_on_sphere(g, randomSeed, Q);
        float s = 2.0f * radii[0];
        *inv_pdf_area = (s * s) / z;
        return z > 0.0f;                                       /* false for NaN and for the rim */
    }
    return 0;
}

/* ------------------------------------------------------------------ scatter ------ */

/* src/interactions.h:62-87 */
void orc_hemisphere(const float nrm[3], float xi1, float xi2, float out[3]) {
    v3 normal = vload(nrm);
    float up = sqrtf(xi1);
    float over = sqrtf(1.0f - up * up);
    float around = xi2 * ORC_TWO_PI;
    v3 directionNotNormal;
    if (fabsf(normal.x) < ORC_SQRT_OF_ONE_THIRD) directionNotNormal = V(1.0f, 0.0f, 0.0f);
    else if (fabsf(normal.y) < ORC_SQRT_OF_ONE_THIRD) directionNotNormal = V(0.0f, 1.0f, 0.0f);
    else directionNotNormal = V(0.0f, 0.0f, 1.0f);
    v3 p1 = vnormalize(vcross(normal, directionNotNormal));
    v3 p2 = vnormalize(vcross(normal, p1));
    float sn, cs;
    orc_sincos(around, &sn, &cs);
    v3 r = vadd(vadd(vscale(normal, up), vscale(p1, cs * over)), vscale(p2, sn * over));
    vstore(out, r);
}

/* calculateReflectionDirection (stub src/interactions.h:47-50): r = i - 2(n.i)n */
void orc_reflection_direction(const float n[3], const float i[3], float out[3]) {
    v3 nn = vload(n), ii = vload(i);
    float k = 2.0f * vdot(nn, ii);
    vstore(out, vsub(ii, vscale(nn, k)));
}

/* calculateTransmissionDirection (stub src/interactions.h:42-44): Snell, n faces against i.
 * Returns 0 on total internal reflection. */
int orc_transmission_direction(const float n[3], const float i[3], float ior_i, float ior_t, float out[3]) {
    v3 nn = vload(n), ii = vload(i);
    float eta = ior_i / ior_t;
    float c = -vdot(nn, ii);
    float k = 1.0f - ((eta * eta) * (1.0f - (c * c)));
    if (k < 0.0f) { out[0] = out[1] = out[2] = 0.0f; return 0; }
    float a = (eta * c) - sqrtf(k);
    vstore(out, vadd(vscale(ii, eta), vscale(nn, a)));
    return 1;
}

/* calculateFresnel (stub src/interactions.h:53-59): unpolarised dielectric Fresnel */
void orc_fresnel(const float n[3], const float i[3], float ior_i, float ior_t,
                 float *reflection, float *transmission) {
    v3 nn = vload(n), ii = vload(i);
    float eta = ior_i / ior_t;
    float c = -vdot(nn, ii);
    float k = 1.0f - ((eta * eta) * (1.0f - (c * c)));
    if (k < 0.0f) { *reflection = 1.0f; *transmission = 0.0f; return; }
    float ct = sqrtf(k);
    float rs = ((ior_i * c) - (ior_t * ct)) / ((ior_i * c) + (ior_t * ct));
    float rp = ((ior_i * ct) - (ior_t * c)) / ((ior_i * ct) + (ior_t * c));
    float R = 0.5f * ((rs * rs) + (rp * rp));
    *reflection = R;
    *transmission = 1.0f - R;
}

/* The calculateBSDF contract (stub src/interactions.h:96-103), DESIGN.md section 3.5. `dir` holds the
 * incoming direction on entry and the scattered direction on return. */
int orc_scatter(const orc_material *m, const float Pp[3], const float Nn[3],
                float u_sel, float xi1, float xi2,
                float origin[3], float dir[3], float thr[3], float L[3]) {
    v3 P = vload(Pp), N = vload(Nn), d = vload(dir), T = vload(thr);
    if (m->emittance > 0.0f) {
        v3 e = vscale(vload(m->color), m->emittance);
        vstore(L, vmul(T, e));
        return 3;
    }
    float nn = vdot(N, N);
    if (!(nn > 0.0f)) return 4;
    v3 n = vscale(N, 1.0f / sqrtf(nn));
    float cosi = vdot(n, d);
    v3 nf = (cosi > 0.0f) ? vneg(n) : n;
    float nfa[3], da[3], out[3];
    vstore(nfa, nf); vstore(da, d);
    if (m->hasRefractive > 0.0f) {
        float ior = (m->indexOfRefraction > 0.0f) ? m->indexOfRefraction : 1.0f;
        int entering = !(cosi > 0.0f);
        float ior_i = entering ? 1.0f : ior, ior_t = entering ? ior : 1.0f;
        float R, Tr;
        orc_fresnel(nfa, da, ior_i, ior_t, &R, &Tr);
        T = vmul(T, vload(m->specularColor));
        vstore(thr, T);
        if (u_sel < R) {
            orc_reflection_direction(nfa, da, out);
            vstore(origin, vadd(P, vscale(nf, ORC_RAY_BIAS)));
            memcpy(dir, out, sizeof out);
            return 1;
        }
        orc_transmission_direction(nfa, da, ior_i, ior_t, out);
        vstore(origin, vsub(P, vscale(nf, ORC_TRANSMIT_BIAS)));
        memcpy(dir, out, sizeof out);
        return 2;
    }
    if (m->hasReflective > 0.0f) {
        orc_reflection_direction(nfa, da, out);
        vstore(thr, vmul(T, vload(m->specularColor)));
        vstore(origin, vadd(P, vscale(nf, ORC_RAY_BIAS)));
        memcpy(dir, out, sizeof out);
        return 1;
    }
    orc_hemisphere(nfa, xi1, xi2, out);
    vstore(thr, vmul(T, vload(m->color)));
    vstore(origin, vadd(P, vscale(nf, ORC_RAY_BIAS)));
    memcpy(dir, out, sizeof out);
    return 0;
}

/* ------------------------------------------------------------------ transforms --- */

typedef struct { float c[4][4]; } m4;   /* column-major like glm::mat4: c[col][row] */

static m4 m4_identity(void) {
    m4 r; memset(&r, 0, sizeof r);
    r.c[0][0] = r.c[1][1] = r.c[2][2] = r.c[3][3] = 1.0f;
    return r;
}

/* tmat4x4 operator* (src/glm/core/type_mat4x4.inl:757-778): column k of the product =
 * ((A0*B[k][0] + A1*B[k][1]) + A2*B[k][2]) + A3*B[k][3] */
static m4 m4_mul(const m4 *a, const m4 *b) {
    m4 r;
    for (int k = 0; k < 4; k++)
        for (int row = 0; row < 4; row++)
            r.c[k][row] = ((a->c[0][row] * b->c[k][0] + a->c[1][row] * b->c[k][1])
                           + a->c[2][row] * b->c[k][2]) + a->c[3][row] * b->c[k][3];
    return r;
}

/* glm::translate (src/glm/gtc/matrix_transform.inl:32-42) */
static m4 m4_translate(const m4 *m, const float v[3]) {
    m4 r = *m;
    for (int row = 0; row < 4; row++)
        r.c[3][row] = ((m->c[0][row] * v[0] + m->c[1][row] * v[1]) + m->c[2][row] * v[2]) + m->c[3][row];
    return r;
}

/* glm::rotate (src/glm/gtc/matrix_transform.inl:44-80); angle in degrees
 * (radians(): src/glm/core/func_trigonometric.inl:35-44) */
static m4 m4_rotate(const m4 *m, float angle, const float v[3]) {
    const float pi = 3.1415926535897932384626433832795f;
    float a = angle * (pi / 180.0f);
    float c = cosf(a);
    float s = sinf(a);
    v3 axis = vnormalize(vload(v));
    float ax[3] = {axis.x, axis.y, axis.z};
    float temp[3] = {(1.0f - c) * ax[0], (1.0f - c) * ax[1], (1.0f - c) * ax[2]};
    float R[3][3];
    R[0][0] = c + temp[0] * ax[0];
    R[0][1] = 0 + temp[0] * ax[1] + s * ax[2];
    R[0][2] = 0 + temp[0] * ax[2] - s * ax[1];
    R[1][0] = 0 + temp[1] * ax[0] - s * ax[2];
    R[1][1] = c + temp[1] * ax[1];
    R[1][2] = 0 + temp[1] * ax[2] + s * ax[0];
    R[2][0] = 0 + temp[2] * ax[0] + s * ax[1];
    R[2][1] = 0 + temp[2] * ax[1] - s * ax[0];
    R[2][2] = c + temp[2] * ax[2];
    m4 r;
    for (int k = 0; k < 3; k++)
        for (int row = 0; row < 4; row++)
            r.c[k][row] = (m->c[0][row] * R[k][0] + m->c[1][row] * R[k][1]) + m->c[2][row] * R[k][2];
    for (int row = 0; row < 4; row++) r.c[3][row] = m->c[3][row];
    return r;
}

/* glm::scale (src/glm/gtc/matrix_transform.inl:82-95) */
static m4 m4_scale(const m4 *m, const float v[3]) {
    m4 r;
    for (int row = 0; row < 4; row++) {
        r.c[0][row] = m->c[0][row] * v[0];
        r.c[1][row] = m->c[1][row] * v[1];
        r.c[2][row] = m->c[2][row] * v[2];
        r.c[3][row] = m->c[3][row];
    }
    return r;
}

/* glm::inverse for mat4, cofactor form (src/glm/core/func_matrix.inl:523-583) */
static m4 m4_inverse(const m4 *mm) {
#define m(i, j) (mm->c[i][j])
    float Coef00 = m(2,2) * m(3,3) - m(3,2) * m(2,3);
    float Coef02 = m(1,2) * m(3,3) - m(3,2) * m(1,3);
    float Coef03 = m(1,2) * m(2,3) - m(2,2) * m(1,3);
    float Coef04 = m(2,1) * m(3,3) - m(3,1) * m(2,3);
    float Coef06 = m(1,1) * m(3,3) - m(3,1) * m(1,3);
    float Coef07 = m(1,1) * m(2,3) - m(2,1) * m(1,3);
    float Coef08 = m(2,1) * m(3,2) - m(3,1) * m(2,2);
    float Coef10 = m(1,1) * m(3,2) - m(3,1) * m(1,2);
    float Coef11 = m(1,1) * m(2,2) - m(2,1) * m(1,2);
    float Coef12 = m(2,0) * m(3,3) - m(3,0) * m(2,3);
    float Coef14 = m(1,0) * m(3,3) - m(3,0) * m(1,3);
    float Coef15 = m(1,0) * m(2,3) - m(2,0) * m(1,3);
    float Coef16 = m(2,0) * m(3,2) - m(3,0) * m(2,2);
    float Coef18 = m(1,0) * m(3,2) - m(3,0) * m(1,2);
    float Coef19 = m(1,0) * m(2,2) - m(2,0) * m(1,2);
    float Coef20 = m(2,0) * m(3,1) - m(3,0) * m(2,1);
    float Coef22 = m(1,0) * m(3,1) - m(3,0) * m(1,1);
    float Coef23 = m(1,0) * m(2,1) - m(2,0) * m(1,1);
    const float SignA[4] = {+1, -1, +1, -1}, SignB[4] = {-1, +1, -1, +1};
    float Fac0[4] = {Coef00, Coef00, Coef02, Coef03};
    float Fac1[4] = {Coef04, Coef04, Coef06, Coef07};
    float Fac2[4] = {Coef08, Coef08, Coef10, Coef11};
    float Fac3[4] = {Coef12, Coef12, Coef14, Coef15};
    float Fac4[4] = {Coef16, Coef16, Coef18, Coef19};
    float Fac5[4] = {Coef20, Coef20, Coef22, Coef23};
    float Vec0[4] = {m(1,0), m(0,0), m(0,0), m(0,0)};
    float Vec1[4] = {m(1,1), m(0,1), m(0,1), m(0,1)};
    float Vec2[4] = {m(1,2), m(0,2), m(0,2), m(0,2)};
    float Vec3[4] = {m(1,3), m(0,3), m(0,3), m(0,3)};
    m4 inv;
    for (int k = 0; k < 4; k++) {
        inv.c[0][k] = SignA[k] * ((Vec1[k] * Fac0[k] - Vec2[k] * Fac1[k]) + Vec3[k] * Fac2[k]);
        inv.c[1][k] = SignB[k] * ((Vec0[k] * Fac0[k] - Vec2[k] * Fac3[k]) + Vec3[k] * Fac4[k]);
        inv.c[2][k] = SignA[k] * ((Vec0[k] * Fac1[k] - Vec1[k] * Fac3[k]) + Vec3[k] * Fac5[k]);
        inv.c[3][k] = SignB[k] * ((Vec0[k] * Fac2[k] - Vec1[k] * Fac4[k]) + Vec2[k] * Fac5[k]);
    }
    /* Row0 = (Inverse[0][0], Inverse[1][0], Inverse[2][0], Inverse[3][0]); det = dot(m[0], Row0) */
    float det = ((m(0,0) * inv.c[0][0] + m(0,1) * inv.c[1][0]) + m(0,2) * inv.c[2][0]) + m(0,3) * inv.c[3][0];
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) inv.c[i][j] = inv.c[i][j] / det;
#undef m
    return inv;
}

/* glmMat4ToCudaMat4 (src/utilities.cpp:79-86): rows of the matrix become x,y,z,w */
static void m4_to_rows(const m4 *a, float out[16]) {
    for (int row = 0; row < 4; row++)
        for (int col = 0; col < 4; col++) out[row * 4 + col] = a->c[col][row];
}

/* buildTransformationMatrix (src/utilities.cpp:70-77) = T * (Rx*Ry*Rz) * S, then the pair
 * (transform, inverse) as scene.cpp:123-125 stores it */
int orc_build_transform(const float t[3], const float r[3], const float s[3],
                        float transform[16], float inverse[16]) {
    const float X[3] = {1, 0, 0}, Y[3] = {0, 1, 0}, Z[3] = {0, 0, 1};
    m4 I = m4_identity();
    m4 translationMat = m4_translate(&I, t);
    m4 rotationMat = m4_rotate(&I, r[0], X);
    m4 ry = m4_rotate(&I, r[1], Y);
    rotationMat = m4_mul(&rotationMat, &ry);
    m4 rz = m4_rotate(&I, r[2], Z);
    rotationMat = m4_mul(&rotationMat, &rz);
    m4 scaleMat = m4_scale(&I, s);
    m4 tr = m4_mul(&translationMat, &rotationMat);
    m4 full = m4_mul(&tr, &scaleMat);
    m4 inv = m4_inverse(&full);
    m4_to_rows(&full, transform);
    m4_to_rows(&inv, inverse);
    return 0;
}

/* ------------------------------------------------------------------ image out ---- */

/* sendImageToPBO (src/raytraceKernel.cu:88-119): x255, clamp only above, truncate */
void orc_display_pixel(const float rgb[3], uint8_t out_xyzw[4]) {
    float c[3];
    for (int k = 0; k < 3; k++) {
        c[k] = rgb[k] * 255.0f;
        if (c[k] > 255.0f) c[k] = 255.0f;
    }
    out_xyzw[0] = (uint8_t)c[0];
    out_xyzw[1] = (uint8_t)c[1];
    out_xyzw[2] = (uint8_t)c[2];
    out_xyzw[3] = 0;
}

/* main.cpp:143-147 + image.cpp: value = pow(sum/divisor, gamma), x255, clamp 0..255, u8 */
void orc_image_to_u8(const float *rgb, int n, float divisor, float gamma, uint8_t *out_rgb) {
    for (int i = 0; i < 3 * n; i++) {
        float v = powf(rgb[i] / divisor, gamma) * 255.0f;
        if (v < 0.0f) v = 0.0f; else if (v > 255.0f) v = 255.0f;
        out_rgb[i] = (uint8_t)v;
    }
}

/* ------------------------------------------------------------------ whole path --- */

int orc_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

static int clamp_threads(int n) {
    int mx = orc_max_threads();
    if (n <= 0 || n > mx) n = mx;
    return n;
}

int orc_raycast_flat(const orc_geom *geoms, int ngeoms, const orc_material *mats, int nmats,
                     const orc_camera *cam, float *image_rgb, int *hit_id, int nthreads) {
    (void)nmats;
    orc_camera_basis cb;
    orc_camera_setup(cam, &cb);
    int W = (int)cam->resolution[0], H = (int)cam->resolution[1];
    nthreads = clamp_threads(nthreads);
#pragma omp parallel for schedule(dynamic, 4) num_threads(nthreads)
    for (int y = 0; y < H; y++) {
        for (int x = 0; x < W; x++) {
            float o[3], d[3], t, P[3], N[3];
            orc_camera_ray(&cb, NULL, x, y, 0, 0, 0, 0, o, d);
            int hit = orc_nearest_hit(geoms, ngeoms, NULL, o, d, &t, P, N);
            int idx = x + y * W;
            if (hit_id) hit_id[idx] = hit;
            if (hit >= 0) memcpy(&image_rgb[3 * idx], mats[geoms[hit].materialid].color, 3 * sizeof(float));
        }
    }
    return 0;
}

typedef struct {
    float o[3], d[3], thr[3];
    int alive;
    int count_emission;     /* direct_light: the next emitter hit adds its radiance (camera ray / after a
                               specular event); after a diffuse event the shadow ray already did */
} path_state;

typedef struct {
    int n;
    int *ids;               /* primitives whose material emits, in index order */
} light_list;

static light_list collect_lights(const orc_geom *geoms, int ngeoms, const orc_material *mats) {
    light_list l;
    l.n = 0;
    l.ids = (int *)malloc(sizeof(int) * (size_t)(ngeoms > 0 ? ngeoms : 1));
    for (int i = 0; i < ngeoms; i++)
        if (mats[geoms[i].materialid].emittance > 0.0f) l.ids[l.n++] = i;
    return l;
}

/* Next-event estimation at a diffuse hit.  `st` is the bounce's RNG stream after
 * the three scatter draws; ps holds the scattered ray (origin = biased hit point, thr already multiplied
 * by the albedo); N is the geometric normal of the hit, d_in the incoming direction. */
static void direct_light(const orc_geom *geoms, int ngeoms, const orc_material *mats, const light_list *lights,
                         uint32_t st, const float Nn[3], const float d_in[3], const path_state *ps, float L[3]) {
    if (lights->n <= 0) return;
    st = orc_lcg_next(st); float u_l = orc_u01(st);
    st = orc_lcg_next(st); float seedf = (float)(st & 0xFFFFFFu);
    int li = (int)(u_l * (float)lights->n);
    if (li > lights->n - 1) li = lights->n - 1;
    int lid = lights->ids[li];
    float Q[3], invpdf;
    if (!orc_sample_light(&geoms[lid], seedf, Q, &invpdf)) return;
    v3 o = vload(ps->o);
    v3 wv = vsub(vload(Q), o);
    float dist2 = vdot(wv, wv);
    if (!(dist2 > 0.0f)) return;
    float dist = sqrtf(dist2);
    v3 w = vscale(wv, 1.0f / dist);
    v3 N = vload(Nn);
    v3 n = vscale(N, 1.0f / sqrtf(vdot(N, N)));
    v3 nf = (vdot(n, vload(d_in)) > 0.0f) ? vneg(n) : n;
    float cos_s = vdot(nf, w);
    if (!(cos_s > 0.0f)) return;
    float wa[3], t, Ph[3], Nh[3];
    vstore(wa, w);
    int hit = orc_nearest_hit(geoms, ngeoms, mats, ps->o, wa, &t, Ph, Nh);
    if (hit != lid) return;
    float tol = 1e-3f * (dist > 1.0f ? dist : 1.0f);
    if (!(t + tol >= dist)) return;                      /* a nearer face of the same emitter hides Q */
    v3 NH = vload(Nh);
    float nl2 = vdot(NH, NH);
    if (!(nl2 > 0.0f)) return;
    float cos_l = fabsf(vdot(NH, w)) / sqrtf(nl2);
    float geomf = (((cos_s * cos_l) * invpdf) / (ORC_PI * dist2)) * (float)lights->n;
    const orc_material *lm = &mats[geoms[lid].materialid];
    v3 Le = vscale(vload(lm->color), lm->emittance);
    v3 C = vscale(vmul(vload(ps->thr), Le), geomf);
    L[0] = L[0] + C.x; L[1] = L[1] + C.y; L[2] = L[2] + C.z;
}

static void path_generate(const orc_camera_basis *cb, const orc_config *cfg, int W, uint32_t pixel,
                          uint32_t iteration, path_state *ps) {
    int x = (int)(pixel % (uint32_t)W), y = (int)(pixel / (uint32_t)W);
    float jx = 0, jy = 0, lu = 0, lv = 0;
    if (cfg->antialias || (cfg->camera_mode == 1 && cfg->aperture > 0.0f)) {
        uint32_t st = orc_lcg_seed(orc_stream_seed(pixel, iteration, 0u));
        st = orc_lcg_next(st); jx = orc_u01(st) - 0.5f;
        st = orc_lcg_next(st); jy = orc_u01(st) - 0.5f;
        st = orc_lcg_next(st); lu = orc_u01(st);
        st = orc_lcg_next(st); lv = orc_u01(st);
    }
    orc_camera_ray(cb, cfg, x, y, jx, jy, lu, lv, ps->o, ps->d);
    ps->thr[0] = ps->thr[1] = ps->thr[2] = 1.0f;
    ps->alive = 1;
    ps->count_emission = 1;
}

/* one bounce of one path; returns 1 while the path stays alive; adds radiance into L */
static int path_bounce(const orc_geom *geoms, int ngeoms, const orc_material *mats, const light_list *lights,
                       uint32_t pixel, uint32_t iteration, int bounce, int last, path_state *ps, float L[3]) {
    float t, P[3], N[3];
    int hit = orc_nearest_hit(geoms, ngeoms, mats, ps->o, ps->d, &t, P, N);
    if (hit < 0) return 0;
    const orc_material *m = &mats[geoms[hit].materialid];
    if (last && !(m->emittance > 0.0f)) return 1;   /* depth exhausted: alive, contributes 0 */
    uint32_t st = orc_lcg_seed(orc_stream_seed(pixel, iteration, 1u + (uint32_t)bounce));
    st = orc_lcg_next(st); float u_sel = orc_u01(st);
    st = orc_lcg_next(st); float xi1 = orc_u01(st);
    st = orc_lcg_next(st); float xi2 = orc_u01(st);
    float d_in[3] = {ps->d[0], ps->d[1], ps->d[2]};
    float Le[3] = {0, 0, 0};
    int code = orc_scatter(m, P, N, u_sel, xi1, xi2, ps->o, ps->d, ps->thr, Le);
    if (code == 3 && (!lights || ps->count_emission)) {
        /* contributions of one path are summed in bounce order, starting from zero */
        L[0] = L[0] + Le[0]; L[1] = L[1] + Le[1]; L[2] = L[2] + Le[2];
    }
    if (lights) {
        if (code == 0) direct_light(geoms, ngeoms, mats, lights, st, N, d_in, ps, L);
        ps->count_emission = (code == 1 || code == 2);
    }
    return code <= 2;
}

int orc_render(const orc_geom *geoms, int ngeoms, const orc_material *mats, int nmats,
               const orc_camera *cam, const orc_config *cfg, int first_iteration, int count,
               float *image_rgb, uint64_t *live, int nthreads) {
    (void)nmats;
    orc_camera_basis cb;
    orc_camera_setup(cam, &cb);
    int W = (int)cam->resolution[0], H = (int)cam->resolution[1];
    int D = cfg->max_depth;
    int stride = cfg->row_stride > 0 ? cfg->row_stride : 1;
    int offset = cfg->row_offset;
    if (D < 0 || D > 64) return -1;
    nthreads = clamp_threads(nthreads);
    uint64_t *tl = (uint64_t *)calloc((size_t)nthreads * 65, sizeof(uint64_t));
    if (!tl) return -2;
    light_list ll = collect_lights(geoms, ngeoms, mats);
    const light_list *lights = cfg->direct_light ? &ll : NULL;
#pragma omp parallel for schedule(dynamic, 2) num_threads(nthreads)
    for (int y = 0; y < H; y++) {
        if (y % stride != offset) continue;
#ifdef _OPENMP
        uint64_t *mine = tl + (size_t)omp_get_thread_num() * 65;
#else
        uint64_t *mine = tl;
#endif
        for (int x = 0; x < W; x++) {
            uint32_t pixel = (uint32_t)(x + y * W);
            for (int it = first_iteration; it < first_iteration + count; it++) {
                path_state ps;
                float L[3] = {0, 0, 0};
                path_generate(&cb, cfg, W, pixel, (uint32_t)it, &ps);
                mine[0]++;
                int b;
                for (b = 0; b < D; b++) {
                    if (!path_bounce(geoms, ngeoms, mats, lights, pixel, (uint32_t)it, b, b == D - 1, &ps, L)) break;
                    mine[b + 1]++;
                }
                /* one add per pixel per iteration, in iteration order (main.cpp:146 divides later) */
                image_rgb[3 * pixel + 0] = image_rgb[3 * pixel + 0] + L[0];
                image_rgb[3 * pixel + 1] = image_rgb[3 * pixel + 1] + L[1];
                image_rgb[3 * pixel + 2] = image_rgb[3 * pixel + 2] + L[2];
            }
        }
    }
    if (live) {
        for (int k = 0; k <= D; k++) {
            uint64_t s = 0;
            for (int t = 0; t < nthreads; t++) s += tl[(size_t)t * 65 + k];
            live[k] = s;
        }
    }
    free(tl);
    free(ll.ids);
    return 0;
}

int orc_trace_pool(const orc_geom *geoms, int ngeoms, const orc_material *mats, int nmats,
                   const orc_camera *cam, const orc_config *cfg, int iteration, int bounces,
                   float *ox, float *oy, float *oz, float *dx, float *dy, float *dz,
                   float *tr, float *tg, float *tb, uint32_t *pixel) {
    (void)nmats;
    orc_camera_basis cb;
    orc_camera_setup(cam, &cb);
    int W = (int)cam->resolution[0], H = (int)cam->resolution[1];
    int stride = cfg->row_stride > 0 ? cfg->row_stride : 1;
    int n = 0;
    light_list ll = collect_lights(geoms, ngeoms, mats);
    const light_list *lights = cfg->direct_light ? &ll : NULL;
    for (int y = 0; y < H; y++) {
        if (y % stride != cfg->row_offset) continue;
        for (int x = 0; x < W; x++) {
            uint32_t p = (uint32_t)(x + y * W);
            path_state ps;
            float L[3] = {0, 0, 0};
            path_generate(&cb, cfg, W, p, (uint32_t)iteration, &ps);
            int alive = 1;
            for (int b = 0; b < bounces && alive; b++)
                alive = path_bounce(geoms, ngeoms, mats, lights, p, (uint32_t)iteration, b,
                                    b == cfg->max_depth - 1, &ps, L);
            if (!alive) continue;
            if (ox) ox[n] = ps.o[0];
            if (oy) oy[n] = ps.o[1];
            if (oz) oz[n] = ps.o[2];
            if (dx) dx[n] = ps.d[0];
            if (dy) dy[n] = ps.d[1];
            if (dz) dz[n] = ps.d[2];
            if (tr) tr[n] = ps.thr[0];
            if (tg) tg[n] = ps.thr[1];
            if (tb) tb[n] = ps.thr[2];
            /* direct_light: bit 31 carries the path's count_emission flag, as in the device pool */
            if (pixel) pixel[n] = p | ((lights && ps.count_emission) ? 0x80000000u : 0u);
            n++;
        }
    }
    free(ll.ids);
    return n;
}
