// atomic_denorm.hip -- do the memory-side float atomics (global_atomic_add_f32, what unsafeAtomicAdd emits) keep
// denormals?  The queue kernel adds emitter radiance to the accumulator with them and claims the bits of a plain
// read-modify-write; that holds only if a denormal operand or result is not flushed.
// build: hipcc --offload-arch=gfx950 -O2 tools/ubench/atomic_denorm.hip -o tools/ubench/atomic_denorm
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>

__global__ void k(float *p, const float *add, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) (void)unsafeAtomicAdd(p + i, add[i]);
}

int main() {
    const int n = 6;
    // initial value, addend: denormal + denormal, 0 + denormal, normal + negative normal -> denormal, normal + denormal, ...
    float init[n] = {1e-40f, 0.0f, 2.0e-38f, 1.0f, 3e-39f, -0.0f};
    float add[n] = {1e-40f, 7e-41f, -1.9e-38f, 1e-40f, -3e-39f, 1e-45f};
    float *d_p, *d_a, out[n];
    hipMalloc(&d_p, sizeof init); hipMalloc(&d_a, sizeof add);
    hipMemcpy(d_p, init, sizeof init, hipMemcpyHostToDevice); hipMemcpy(d_a, add, sizeof add, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d_p, d_a, n);
    hipMemcpy(out, d_p, sizeof out, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < n; ++i) {
        const float want = init[i] + add[i];                 // host IEEE addition keeps denormals
        unsigned a, b;
        memcpy(&a, &out[i], 4); memcpy(&b, &want, 4);
        printf("%d: %.9g + %.9g -> %.9g (0x%08x), IEEE %.9g (0x%08x) %s\n", i, init[i], add[i], out[i], a, want, b, a == b ? "same" : "DIFFERENT");
        bad += a != b;
    }
    printf("%s\n", bad ? "float atomics FLUSH denormals" : "float atomics keep denormals");
    return 0;
}
