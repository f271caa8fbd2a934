// exact_ops.hip -- exhaustive check of shorter correctly-rounded reciprocal / sqrt sequences against
// hipcc's IEEE expansions (`1.0f / x`, `__builtin_sqrtf(x)`) over ALL 2^32 binary32 bit patterns.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off exact_ops.hip -o exact_ops
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>

__device__ __forceinline__ float rcp_lean(float x) {
    float y = __builtin_amdgcn_rcpf(x);
    float e = __builtin_fmaf(-x, y, 1.0f);
    y = __builtin_fmaf(e, y, y);
    e = __builtin_fmaf(-x, y, 1.0f);
    y = __builtin_fmaf(e, y, y);
    return __builtin_amdgcn_div_fixupf(y, x, 1.0f);
}
__device__ __forceinline__ float rcp_lean3(float x) {   // one more correction round
    float y = __builtin_amdgcn_rcpf(x);
    float e = __builtin_fmaf(-x, y, 1.0f);
    y = __builtin_fmaf(e, y, y);
    e = __builtin_fmaf(-x, y, 1.0f);
    y = __builtin_fmaf(e, y, y);
    e = __builtin_fmaf(-x, y, 1.0f);
    y = __builtin_fmaf(e, y, y);
    return __builtin_amdgcn_div_fixupf(y, x, 1.0f);
}
__device__ __forceinline__ float sqrt_lean(float x) {
    float s = __builtin_amdgcn_sqrtf(x);
    const int si = __float_as_int(s);
    const float dn = __int_as_float(si - 1), up = __int_as_float(si + 1);
    const float rdn = __builtin_fmaf(-dn, s, x), rup = __builtin_fmaf(-up, s, x);
    s = rdn <= 0.0f ? dn : s;
    s = rup > 0.0f ? up : s;
    return s;
}

__device__ __forceinline__ bool same(float a, float b) {
    return __float_as_uint(a) == __float_as_uint(b) || (a != a && b != b);
}

// counters: [0] rcp_lean mismatches, [1] rcp_lean3, [2] sqrt_lean; lo/hi bit patterns of the mismatching |x|
__global__ void k(unsigned long long *cnt, unsigned *lo, unsigned *hi) {
    const unsigned long long tid = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x;
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    for (unsigned long long b = tid; b < (1ull << 32); b += stride) {
        const float x = __uint_as_float((unsigned)b);
        const unsigned mag = (unsigned)b & 0x7FFFFFFFu;
        const float want = 1.0f / x;
        if (!same(rcp_lean(x), want)) { atomicAdd(&cnt[0], 1ull); atomicMin(&lo[0], mag); atomicMax(&hi[0], mag); }
        if (!same(rcp_lean3(x), want)) { atomicAdd(&cnt[1], 1ull); atomicMin(&lo[1], mag); atomicMax(&hi[1], mag); }
        if (!same(sqrt_lean(x), __builtin_sqrtf(x))) {
            atomicAdd(&cnt[2], 1ull); atomicMin(&lo[2], mag); atomicMax(&hi[2], mag);
            if ((b >> 31) == 0) atomicAdd(&cnt[3], 1ull);      // non-negative inputs only
        }
        // restricted domains: normal range used by the kernels
        if (mag >= 0x00800000u && mag <= 0x7E800000u) {       // 2^-126 <= |x| <= 2^126
            if (!same(rcp_lean(x), want)) atomicAdd(&cnt[4], 1ull);
        }
        if ((b >> 31) == 0 && mag >= 0x00800000u && mag <= 0x7F7FFFFFu) {
            if (!same(sqrt_lean(x), __builtin_sqrtf(x))) atomicAdd(&cnt[5], 1ull);
        }
    }
}

int main() {
    unsigned long long *cnt; unsigned *lo, *hi;
    hipMalloc(&cnt, 64); hipMalloc(&lo, 32); hipMalloc(&hi, 32);
    hipMemset(cnt, 0, 64); hipMemset(lo, 0xFF, 32); hipMemset(hi, 0, 32);
    hipLaunchKernelGGL(k, dim3(256 * 16), dim3(256), 0, 0, cnt, lo, hi);
    hipDeviceSynchronize();
    unsigned long long c[8]; unsigned l[8], h[8];
    hipMemcpy(c, cnt, 64, hipMemcpyDeviceToHost); hipMemcpy(l, lo, 32, hipMemcpyDeviceToHost); hipMemcpy(h, hi, 32, hipMemcpyDeviceToHost);
    const char *names[3] = {"rcp_lean (2 rounds)", "rcp_lean3 (3 rounds)", "sqrt_lean"};
    for (int i = 0; i < 3; ++i) {
        float fl, fh; memcpy(&fl, &l[i], 4); memcpy(&fh, &h[i], 4);
        printf("%-22s mismatches over all 2^32 patterns: %llu  |x| range of mismatches: [%g (0x%08x), %g (0x%08x)]\n", names[i], c[i], fl, l[i], fh, h[i]);
    }
    printf("sqrt_lean mismatches on non-negative inputs: %llu\n", c[3]);
    printf("rcp_lean mismatches for 2^-126 <= |x| <= 2^126: %llu\n", c[4]);
    printf("sqrt_lean mismatches for normal positive x: %llu\n", c[5]);
    return 0;
}
