/* cuda.h -- headless shim (see cuda_runtime.h): /root/reference/src/raytraceKernel.h:13 includes it, host code uses nothing from it */
#ifndef PTMI355_SHIM_CUDA_H
#define PTMI355_SHIM_CUDA_H
#include "cuda_runtime.h"
#endif
