/* thrust/random.h -- headless shim for the HOST translation units only (/root/reference/src/raytraceKernel.h:12 pulls
 * it into main.cpp, which uses nothing from it).  The kernels' random numbers are the library's own minstd_rand
 * restatement (csrc/pt_device.hpp), pinned against the image's rocThrust (tests/golden/ref_thrust_rng.json). */
#ifndef PTMI355_SHIM_THRUST_RANDOM_H
#define PTMI355_SHIM_THRUST_RANDOM_H
#endif
