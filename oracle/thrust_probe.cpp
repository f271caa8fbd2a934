// thrust_probe.cpp -- known answers from the image's Thrust (rocThrust 7.2, /opt/rocm/include/thrust)
// for the RNG classes the reference uses (thrust::default_random_engine +
// uniform_real_distribution<float>; call sites /root/reference/src/raytraceKernel.cu:33-36,
// src/intersections.h:222-224).  Thrust is NOT vendored by the reference (it shipped with
// CUDA 4.0); rocThrust is the only Thrust in this image and needs a HIP compiler, so this
// probe is built with `hipcc -x hip --cuda-host-only` and runs on the host without a GPU.
// TEST INFRASTRUCTURE ONLY (see oracle/pt_oracle.h).
#include <cstdio>
#include <cstring>
#include <thrust/random.h>

static unsigned bits(float f) { unsigned u; memcpy(&u, &f, 4); return u; }

int main() {
    printf("{\"seeds\":[");
    const unsigned seeds[] = {0u, 1u, 7u, 12345u, 2147483646u, 2147483647u, 2147483648u, 4294967295u,
                              1800329511u, 3028713910u, 48271u, 399268537u};
    for (unsigned i = 0; i < sizeof(seeds) / sizeof(seeds[0]); i++) {
        thrust::default_random_engine rng(seeds[i]);
        thrust::default_random_engine raw(seeds[i]);
        thrust::uniform_real_distribution<float> u01(0, 1);
        thrust::uniform_real_distribution<float> u02(-0.5, 0.5);
        printf("%s{\"seed\":%u,\"raw\":[", i ? "," : "", seeds[i]);
        for (int k = 0; k < 6; k++) printf("%s%u", k ? "," : "", (unsigned)raw());
        printf("],\"u01_bits\":[");
        for (int k = 0; k < 6; k++) printf("%s%u", k ? "," : "", bits((float)u01(rng)));
        thrust::default_random_engine rng2(seeds[i]);
        printf("],\"u02_bits\":[");
        for (int k = 0; k < 6; k++) printf("%s%u", k ? "," : "", bits((float)u02(rng2)));
        printf("]}");
    }
    thrust::minstd_rand d;   // default seed 1; ISO C++: 10000th value is 399268537
    unsigned v = 0;
    for (int k = 0; k < 10000; k++) v = d();
    printf("],\"minstd_10000th\":%u}\n", v);
    return 0;
}
