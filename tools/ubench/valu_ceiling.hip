// valu_ceiling.hip -- what does the chip sustain on the exact intersection tests when nothing but
// VALU work is in the way?  Rays live in registers, the primitive's matrices are wave-uniform
// (scalar loads), each lane runs `iters` dependent tests.  Prints tests/s and the implied
// instructions/cycle/SIMD for the instruction counts measured from the ISA (box 209, sphere 223).
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I../../project2-pathtracer_amd/csrc valu_ceiling.hip -o valu_ceiling
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#include "pt_device.hpp"
using namespace ptd;

template <int KIND>
__global__ __launch_bounds__(256) void k(const GeomRec *__restrict__ g, int iters, float *out) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t st = lcg_seed(hash(tid));
    st = lcg_next(st); float a = u01(st);
    st = lcg_next(st); float b = u01(st);
    st = lcg_next(st); float c = u01(st);
    f3 o = mk(a * 8 - 4, b * 8 + 1, c * 8 - 4 + 12), d = normalize(mk(0.3f - a, 0.4f - b, -1.0f));
    float acc = 0.0f;
    for (int i = 0; i < iters; ++i) {
        f3 P, N;
        float t;
        if (KIND == 0) t = box_test(g[0].inv, g[0].xf, 0, o, d, P, N);
        else if (KIND == 1) t = sphere_test(g[1].inv, g[1].xf, o, d, P, N);
        else if (KIND == 2) { t = o.x / d.z; P = o; N = d; }                       // IEEE division
        else if (KIND == 3) { t = __builtin_sqrtf(fabsf(o.x)); P = o; N = d; }      // IEEE sqrt
        else { t = o.x * d.z + o.y; P = o; N = d; }                                // mul + add
        acc += t;
        o.x = o.x + t * 1e-7f;                                                      // dependency, keeps the loop honest
        d.z = d.z - 1e-9f * acc;
    }
    out[tid] = acc + o.x;
}

// the same cube test, but structured like the render kernel's exact pass: 9 primitives staged in LDS,
// every lane fetches ITS primitive's matrices with a per-lane index (GATHER), optionally only a
// fraction of the lanes takes part (HALF: every other lane; TENTH: 6 of 64)
template <int MODE>
__global__ __launch_bounds__(256) void kg(const GeomRec *__restrict__ g, int iters, float *out) {
    __shared__ GeomRec lg[9];
    for (int i = threadIdx.x; i < 9 * (int)(sizeof(GeomRec) / 4); i += 256)
        reinterpret_cast<uint32_t *>(lg)[i] = reinterpret_cast<const uint32_t *>(g)[i % (sizeof(GeomRec) / 4)];
    __syncthreads();
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t st = lcg_seed(hash(tid));
    st = lcg_next(st); float a = u01(st);
    st = lcg_next(st); float b = u01(st);
    st = lcg_next(st); float c = u01(st);
    f3 o = mk(a * 8 - 4, b * 8 + 1, c * 8 - 4 + 12), d = normalize(mk(0.3f - a, 0.4f - b, -1.0f));
    float acc = 0.0f;
    const int lane = threadIdx.x & 63;
    const bool active = MODE == 0 ? true : MODE == 1 ? (lane & 1) == 0 : lane < 6;
    for (int i = 0; i < iters; ++i) {
        if (active) {
            const GeomRec *p = &lg[(lane + i) % 9];
            f3 P, N;
            float t = box_test(p->inv, p->xf, 0, o, d, P, N);
            acc += t;
            o.x = o.x + t * 1e-7f;
            d.z = d.z - 1e-9f * acc;
        }
    }
    out[tid] = acc + o.x;
}

template <int MODE>
double rung(const GeomRec *dg, float *dout, int blocks, int iters) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(kg<MODE>, dim3(blocks), dim3(256), 0, 0, dg, iters, dout);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(kg<MODE>, dim3(blocks), dim3(256), 0, 0, dg, iters, dout);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms;
}

template <int KIND>
double run(const GeomRec *dg, float *dout, int blocks, int iters) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, dg, iters, dout);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, dg, iters, dout);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms;
}

int main() {
    GeomRec h[2] = {};
    // wall: scale (.01,10,10) rot z 90 at origin (object 0 of the Cornell box); sphere: scale 3 at (0,2,0)
    float inv0[12] = {-4.37113886e-06f, 100.000008f, 0, 0, -0.100000001f, -4.37113856e-09f, 0, 0, 0, 0, 0.100000009f, 0};
    float xf0[12] = {-4.37113873e-10f, -10, 0, 0, 0.00999999978f, -4.37113897e-07f, 0, 0, 0, 0, 9.99999905f, 0};
    float inv1[12] = {0.333333343f, 0, 0, 0, 0, 0.333333343f, 0, -0.666666687f, 0, 0, 0.333333343f, 0};
    float xf1[12] = {3, 0, 0, 0, 0, 3, 0, 2, 0, 0, 3, 0};
    memcpy(h[0].inv, inv0, 48); memcpy(h[0].xf, xf0, 48); h[0].type = 1;
    memcpy(h[1].inv, inv1, 48); memcpy(h[1].xf, xf1, 48); h[1].type = 0;
    GeomRec *dg; float *dout;
    hipMalloc(&dg, sizeof h); hipMemcpy(dg, h, sizeof h, hipMemcpyHostToDevice);
    const int names_n = 5;
    const char *names[names_n] = {"box_test", "sphere_test", "ieee_div(+3)", "ieee_sqrt(+4)", "mul_add(+3)"};
    const double instr[names_n] = {209 + 3, 223 + 3, 10 + 3, 15 + 4, 2 + 3};
    for (int per_cu : {4, 8}) {                  // 256-thread blocks per CU -> 4 or 8 waves per SIMD
        const int blocks = 256 * per_cu;
        hipMalloc(&dout, (size_t)blocks * 256 * 4);
        for (int kind = 0; kind < names_n; ++kind) {
            const int iters = kind < 2 ? 2000 : 40000;
            double ms = kind == 0 ? run<0>(dg, dout, blocks, iters) : kind == 1 ? run<1>(dg, dout, blocks, iters)
                      : kind == 2 ? run<2>(dg, dout, blocks, iters) : kind == 3 ? run<3>(dg, dout, blocks, iters) : run<4>(dg, dout, blocks, iters);
            const double tests = (double)blocks * 256 * iters;
            const double wave_instr = tests / 64 * instr[kind];
            printf("{\"kernel\":\"%s\",\"waves_per_simd\":%d,\"ms\":%.3f,\"Gtests_per_s\":%.2f,\"approx_valu_per_test\":%.0f,\"ns_per_wave_instr_per_simd\":%.3f}\n",
                   names[kind], per_cu, ms, tests / ms / 1e6, instr[kind], ms * 1e6 / (wave_instr / 1024));
        }
        const char *gn[3] = {"box_test gather all lanes", "box_test gather 32/64 lanes", "box_test gather 6/64 lanes"};
        for (int mode = 0; mode < 3; ++mode) {
            const int iters = 2000;
            const double ms = mode == 0 ? rung<0>(dg, dout, blocks, iters) : mode == 1 ? rung<1>(dg, dout, blocks, iters) : rung<2>(dg, dout, blocks, iters);
            const double wave_iters = (double)blocks * 4 * iters;          // wave-level loop iterations
            printf("{\"kernel\":\"%s\",\"waves_per_simd\":%d,\"ms\":%.3f,\"ns_per_wave_iteration_per_simd\":%.1f,\"ns_per_wave_instr_per_simd_at_212\":%.3f}\n",
                   gn[mode], per_cu, ms, ms * 1e6 / (wave_iters / 1024), ms * 1e6 / (wave_iters * 212 / 1024));
        }
        hipFree(dout);
    }
    return 0;
}
