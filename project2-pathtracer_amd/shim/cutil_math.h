/* cutil_math.h -- headless shim: the CUDA SDK's float3/float4 operator helpers (/root/reference/src/raytraceKernel.h:16);
 * the reference's host code uses none of them (its vectors are GLM's). */
#ifndef PTMI355_SHIM_CUTIL_MATH_H
#define PTMI355_SHIM_CUTIL_MATH_H
#endif
