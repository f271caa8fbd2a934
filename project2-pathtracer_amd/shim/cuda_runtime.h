/* cuda_runtime.h -- headless shim for building the reference's HOST code (main.cpp, scene.cpp, utilities.cpp,
 * image.cpp, glslUtility.cpp) unchanged on a machine without the CUDA toolkit, against libptmi355.so.
 *
 * Part of the drop-in boundary, not of any parity claim: it declares only what those five
 * translation units use from <cuda_runtime.h> (/root/reference/src/sceneStructs.h:11, cudaMat4.h:10, main.h:19):
 * the __host__/__device__ decorations, `uchar4` (its NAME is part of cudaRaytraceCore's mangled symbol, so the adaptor
 * TU must be compiled against this same header), cudaDeviceReset (main.cpp:159,171) -- and, transitively, the C
 * headers scene.cpp relies on without including them (strcmp/atoi/atof). */
#ifndef PTMI355_SHIM_CUDA_RUNTIME_H
#define PTMI355_SHIM_CUDA_RUNTIME_H

#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifndef __host__
#define __host__
#endif
#ifndef __device__
#define __device__
#endif
#ifndef __global__
#define __global__
#endif

struct uchar4 { unsigned char x, y, z, w; };
struct float3 { float x, y, z; };
struct float4 { float x, y, z, w; };

typedef int cudaError_t;
#define cudaSuccess 0

#ifdef __cplusplus
extern "C" {
#endif
/* drops every device-side state of the adaptor (ptmi355_adaptor_reset); defined in headless_gl.cpp */
cudaError_t cudaDeviceReset(void);
cudaError_t cudaThreadSynchronize(void);
cudaError_t cudaGetLastError(void);
const char *cudaGetErrorString(cudaError_t);
#ifdef __cplusplus
}
#endif

#endif
