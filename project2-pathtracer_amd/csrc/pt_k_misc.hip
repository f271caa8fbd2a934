// pt_k_misc.hip -- the small kernels: k_fold (adds the per-iteration accumulator planes of a launch group to the image
// in iteration order), k_flat (the reference kernel as shipped + primary-hit parity hook), k_display (sendImageToPBO)
// and the known-answer kernels of the parity hooks.
#include "pt_kernels.hpp"

namespace ptk {

// ------------------------------------------------------------------ fold (batched iterations) ---
// image[p] = (((image[p] + plane_0[p]) + plane_1[p]) + ...) in iteration order -- the same sum, in the
// same order, as rendering the iterations one after the other -- and clears the planes for the next
// batch.  One thread per owned pixel.

__global__ __launch_bounds__(kBlock) void k_fold(FoldArgs a) {
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= a.n_own) return;
    const uint32_t W = (uint32_t)a.W;
    const uint32_t lr = gid / W, x = gid - lr * W;
    const size_t p = ((size_t)(lr * (uint32_t)a.row_stride + (uint32_t)a.row_offset) * W + x) * 3;
    float r = a.image[p], g = a.image[p + 1], b = a.image[p + 2];
    for (uint32_t s = 0; s < a.batch; ++s) {
        float *q = a.planes + (size_t)s * a.plane_stride + (size_t)gid * 3;       // planes hold the owned pixels only
        const float qr = q[0], qg = q[1], qb = q[2];
        r = r + qr; g = g + qg; b = b + qb;
        // most entries are still zero (only paths that met an emitter wrote): clearing only the others saves most of the fold's writes
        if (qr != 0.0f || qg != 0.0f || qb != 0.0f) { q[0] = 0.0f; q[1] = 0.0f; q[2] = 0.0f; }
    }
    a.image[p] = r; a.image[p + 1] = g; a.image[p + 2] = b;
}

// ------------------------------------------------------------------ flat (reference) ---

// raytraceRay as shipped (src/raytraceKernel.cu:123-159): nearest hit, flat material colour
// OVERWRITES the pixel, misses leave it untouched.  Also the primary-hit parity hook.
__global__ __launch_bounds__(kBlock) void k_flat(FlatArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    GeomRec *lg;
    MatRec *lm;
    stage_tables(smem, a.geoms, a.G, a.mats, a.M, true, lg, lm);
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= a.n_own) return;
    const uint32_t W = (uint32_t)a.cam.W;
    const uint32_t lr = gid / W, x = gid - lr * W;
    const uint32_t pixel = (lr * (uint32_t)a.cam.row_stride + (uint32_t)a.cam.row_offset) * W + x;
    f3 o, d, P = mk(0, 0, 0), N = mk(0, 0, 0);
    CamRec c = a.cam;
    c.camera_mode = 0; c.antialias = 0;
    camera_ray(c, pixel, 1u, o, d);
    float t;
    const int hit = nearest_hit(lg, a.G, o, d, t, P, N);
    if (a.write_image && hit >= 0) {
        const MatRec m = lm[lg[hit].mat];
        float *px = a.image + (size_t)pixel * 3;
        px[0] = m.color[0]; px[1] = m.color[1]; px[2] = m.color[2];
    }
    if (a.hit) a.hit[pixel] = hit;
    if (a.t) a.t[pixel] = t;
    if (a.dir) { a.dir[3 * pixel] = d.x; a.dir[3 * pixel + 1] = d.y; a.dir[3 * pixel + 2] = d.z; }
    if (a.P) { a.P[3 * pixel] = P.x; a.P[3 * pixel + 1] = P.y; a.P[3 * pixel + 2] = P.z; }
    if (a.N) { a.N[3 * pixel] = N.x; a.N[3 * pixel + 1] = N.y; a.N[3 * pixel + 2] = N.z; }
}

// sendImageToPBO (src/raytraceKernel.cu:88-119); scale = 1 is the reference
__global__ __launch_bounds__(kBlock) void k_display(const float *image, uchar4 *out, uint32_t n, float scale) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float r = (image[3 * i] * scale) * 255.0f, g = (image[3 * i + 1] * scale) * 255.0f,
          b = (image[3 * i + 2] * scale) * 255.0f;
    if (r > 255.0f) r = 255.0f;
    if (g > 255.0f) g = 255.0f;
    if (b > 255.0f) b = 255.0f;
    uchar4 v;
    v.w = 0; v.x = (unsigned char)r; v.y = (unsigned char)g; v.z = (unsigned char)b;
    out[i] = v;
}

// ------------------------------------------------------------------ KAT kernels --------
// generateRandomNumberFromThread (src/raytraceKernel.cu:30-37)
__global__ void k_rng_from_thread(float resx, float time, int n, const int *xy, float *out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int x = xy[2 * i], y = xy[2 * i + 1];
    const int index = (int)((float)x + ((float)y * resx));
    const uint32_t s = (uint32_t)((float)index * time);
    uint32_t st = lcg_seed(hash(s));
    st = lcg_next(st); out[3 * i] = u01(st);
    st = lcg_next(st); out[3 * i + 1] = u01(st);
    st = lcg_next(st); out[3 * i + 2] = u01(st);
}

__global__ void k_hemisphere(int n, const float *nrm, const float *xi, float *out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    f3 r = hemisphere(mk(nrm[3 * i], nrm[3 * i + 1], nrm[3 * i + 2]), xi[2 * i], xi[2 * i + 1]);
    out[3 * i] = r.x; out[3 * i + 1] = r.y; out[3 * i + 2] = r.z;
}

__global__ void k_light_points(const GeomRec *g, int n, const float *seeds, float *out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const f3 p = g->type == 0 ? random_point_on_sphere(g->xf, seeds[i]) : random_point_on_cube(g->xf, seeds[i]);
    out[3 * i] = p.x; out[3 * i + 1] = p.y; out[3 * i + 2] = p.z;
}

__global__ void k_sincos(int n, const float *a, float *s, float *c) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float sn, cs;
    sincos_poly(a[i], sn, cs);
    s[i] = sn; c[i] = cs;
}

// ------------------------------------------------------------------ host side ---------
void fold_launch(hipStream_t stream, const FoldArgs &f) {
    hipLaunchKernelGGL(k_fold, dim3((f.n_own + kBlock - 1) / kBlock), dim3(kBlock), 0, stream, f);
}
void flat_launch(hipStream_t stream, const FlatArgs &f, uint32_t lds_bytes) {
    hipLaunchKernelGGL(k_flat, dim3((f.n_own + kBlock - 1) / kBlock), dim3(kBlock), lds_bytes, stream, f);
}
void display_launch(hipStream_t stream, const float *image, uchar4 *out, uint32_t n, float scale) {
    hipLaunchKernelGGL(k_display, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, stream, image, out, n, scale);
}
void kat_rng_from_thread(hipStream_t stream, float resx, float time, int n, const int *xy, float *out) {
    hipLaunchKernelGGL(k_rng_from_thread, dim3((n + 255) / 256), dim3(256), 0, stream, resx, time, n, xy, out);
}
void kat_hemisphere(hipStream_t stream, int n, const float *nrm, const float *xi, float *out) {
    hipLaunchKernelGGL(k_hemisphere, dim3((n + 255) / 256), dim3(256), 0, stream, n, nrm, xi, out);
}
void kat_light_points(hipStream_t stream, const GeomRec *g, int n, const float *seeds, float *out) {
    hipLaunchKernelGGL(k_light_points, dim3((n + 255) / 256), dim3(256), 0, stream, g, n, seeds, out);
}
void kat_sincos(hipStream_t stream, int n, const float *a, float *s, float *c) {
    hipLaunchKernelGGL(k_sincos, dim3((n + 255) / 256), dim3(256), 0, stream, n, a, s, c);
}

}  // namespace ptk
