// pk_rate.hip -- issue cost of plain vs packed binary32 VALU instructions on gfx950, per wave64
// instruction per SIMD, with 1..8 waves per SIMD, independent streams and one dependent chain.
// Question it answers: is hand-packing the intersection math into
// v_pk_{mul,add,fma}_f32 a lever for the VALU-issue-bound bounce kernel?
// Build: hipcc --offload-arch=gfx950 -O3 pk_rate.hip -o pk_rate
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float v2f __attribute__((ext_vector_type(2)));

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

// KIND 0 v_mul_f32, 1 v_pk_mul_f32, 2 v_add_f32, 3 v_pk_add_f32, 4 v_fma_f32, 5 v_pk_fma_f32
// DEP 0: 16 independent accumulators; DEP 1: one accumulator, every instruction reads the previous result
template <int KIND, int DEP>
__global__ __launch_bounds__(256) void k(int iters, float m0, float *out) {
    float a[16];
    v2f p[16];
    for (int i = 0; i < 16; ++i) { a[i] = 1.0f + threadIdx.x * 1e-6f + i; p[i] = v2f{a[i], a[i] + 0.5f}; }
    const float m = m0;
    const v2f pm = v2f{m0, m0};
    for (int it = 0; it < iters; ++it) {
#define S_MUL(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[DEP ? 0 : i]) : "v"(m));
#define P_MUL(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[DEP ? 0 : i]) : "v"(pm));
#define S_ADD(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[DEP ? 0 : i]) : "v"(m));
#define P_ADD(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[DEP ? 0 : i]) : "v"(pm));
#define S_FMA(i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[DEP ? 0 : i]) : "v"(m));
#define P_FMA(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[DEP ? 0 : i]) : "v"(pm));
        if (KIND == 0) { REP16(S_MUL) }
        if (KIND == 1) { REP16(P_MUL) }
        if (KIND == 2) { REP16(S_ADD) }
        if (KIND == 3) { REP16(P_ADD) }
        if (KIND == 4) { REP16(S_FMA) }
        if (KIND == 5) { REP16(P_FMA) }
    }
    float s = 0.0f;
    for (int i = 0; i < 16; ++i) s += a[i] + p[i].x + p[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND, int DEP>
double run(int blocks, int iters, float *dout) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((k<KIND, DEP>), dim3(blocks), dim3(256), 0, 0, iters, 1.0000001f, dout);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL((k<KIND, DEP>), dim3(blocks), dim3(256), 0, 0, iters, 1.0000001f, dout);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    hipEventDestroy(a); hipEventDestroy(b);
    return ms;
}

int main() {
    const char *names[6] = {"v_mul_f32", "v_pk_mul_f32", "v_add_f32", "v_pk_add_f32", "v_fma_f32", "v_pk_fma_f32"};
    float *dout;
    hipMalloc(&dout, (size_t)256 * 8 * 256 * 4);
    int clk_khz = 0;
    hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0);
    for (int dep = 0; dep < 2; ++dep)
        for (int per_cu : {1, 2, 4, 8}) {          // 256-thread blocks per CU = waves per SIMD
            const int blocks = 256 * per_cu, iters = 20000;
            for (int kind = 0; kind < 6; ++kind) {
                double ms;
#define RUN(K) (dep ? run<K, 1>(blocks, iters, dout) : run<K, 0>(blocks, iters, dout))
                ms = kind == 0 ? RUN(0) : kind == 1 ? RUN(1) : kind == 2 ? RUN(2) : kind == 3 ? RUN(3) : kind == 4 ? RUN(4) : RUN(5);
                const double instr_per_simd = (double)per_cu * iters * 16;      // wave instructions issued by one SIMD
                const double ns = ms * 1e6 / instr_per_simd;
                printf("{\"instr\":\"%s\",\"dependent\":%d,\"waves_per_simd\":%d,\"ms\":%.3f,\"ns_per_wave_instr_per_simd\":%.3f,\"cycles_at_2.4GHz\":%.2f,\"clock_attr_khz\":%d}\n",
                       names[kind], dep, per_cu, ms, ns, ns * 2.4, clk_khz);
            }
        }
    return 0;
}
