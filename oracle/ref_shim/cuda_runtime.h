/* TEST INFRASTRUCTURE (oracle/_ref): lets hipcc compile the reference's own CUDA kernel files where they lie under
 * /root/reference, so that the CPU oracle can be checked against the reference's real kernels running on the GPU.
 * Nothing in the product includes this file. */
#pragma once
#include <hip/hip_runtime.h>
#define cudaStream_t hipStream_t
#define cudaTextureObject_t hipTextureObject_t
#define cudaError hipError_t
#define cudaError_t hipError_t
#define cudaSuccess hipSuccess
#define cudaGetLastError hipGetLastError
#define cudaGetErrorString hipGetErrorString
#define cudaDeviceSynchronize hipDeviceSynchronize
#define cudaStreamSynchronize hipStreamSynchronize
#define cudaMemsetAsync hipMemsetAsync
#define cudaMemcpyAsync hipMemcpyAsync
#define cudaMemcpyDeviceToHost hipMemcpyDeviceToHost
#define cudaMemcpyHostToDevice hipMemcpyHostToDevice
#define cudaMemcpyDeviceToDevice hipMemcpyDeviceToDevice
#define cudaMalloc hipMalloc
#define cudaFree hipFree
/* OpenGL interop of the viewer: never called by the harness */
typedef struct smx_ref_graphics_resource* cudaGraphicsResource_t;
static inline hipError_t cudaGraphicsMapResources(int, cudaGraphicsResource_t*, hipStream_t) { return hipErrorNotSupported; }
static inline hipError_t cudaGraphicsUnmapResources(int, cudaGraphicsResource_t*, hipStream_t) { return hipErrorNotSupported; }
static inline hipError_t cudaGraphicsResourceGetMappedPointer(void**, size_t*, cudaGraphicsResource_t) { return hipErrorNotSupported; }
