/* cuda_gl_interop.h -- headless shim (/root/reference/src/main.h:22).  The "pixel buffer object" main.cpp registers
 * and maps (main.cpp:109-111,129,284-286,369) is plain host memory owned by headless_gl.cpp: cudaRaytraceCore's
 * adaptor writes sendImageToPBO's bytes into it (pt_display with a host pointer). */
#ifndef PTMI355_SHIM_CUDA_GL_INTEROP_H
#define PTMI355_SHIM_CUDA_GL_INTEROP_H
#include "cuda_runtime.h"
#ifdef __cplusplus
extern "C" {
#endif
cudaError_t cudaGLSetGLDevice(int device);
cudaError_t cudaGLRegisterBufferObject(unsigned int buffer);
cudaError_t cudaGLUnregisterBufferObject(unsigned int buffer);
cudaError_t cudaGLMapBufferObject(void **devPtr, unsigned int buffer);
cudaError_t cudaGLUnmapBufferObject(unsigned int buffer);
#ifdef __cplusplus
}
#endif
#endif
