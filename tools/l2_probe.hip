// l2_probe.hip -- does a kernel boundary on ONE stream cost the kernels of OTHER streams their L2 contents?  (The eight XCDs'
// L2s are not coherent with each other; visibility across XCDs at a kernel boundary means write-back + invalidate.)  Not part
// of the product:  hipcc --offload-arch=gfx950 -O3 tools/l2_probe.hip -o build/l2_probe ; run on the GPU box.
//
// k_reread: 256 workgroups, each re-reads its own slice (96 KB: 24 MB in all, 3 MB per XCD) `iters` times -- L2 hits after
// the first pass.  Timed (a) alone, (b) beside a second stream that launches tiny kernels back to back, (c) beside a second
// stream that runs ONE long kernel without boundaries (same occupancy as the tiny ones: one wavefront).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__global__ void __launch_bounds__(256) k_reread(const uint4* __restrict__ buf, int per_wg, int iters, uint32_t* out) {
  const uint4* p = buf + (size_t)blockIdx.x * per_wg;
  uint32_t acc = 0;
  for (int it = 0; it < iters; ++it)
    for (int i = threadIdx.x; i < per_wg; i += 256) { const uint4 v = p[i]; acc += v.x ^ v.w; asm volatile("" : "+v"(acc)); }
  if (acc == 0x12345u) out[0] = acc;
}
__global__ void k_tiny(uint32_t* p) { if (p == nullptr) asm volatile("s_nop 0"); }
__global__ void k_long(uint32_t* p, int n) {
  uint32_t a = threadIdx.x;
  for (int i = 0; i < n; ++i) { a = a * 1664525u + 1013904223u; __builtin_amdgcn_s_sleep(16); }
  if (a == 0x12345u) p[0] = a;
}

int main() {
  hipStream_t s0, s1;
  CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  const int per_wg = 96 * 1024 / 16;   // uint4 per workgroup
  uint4* buf; uint32_t* out;
  CK(hipMalloc(&buf, (size_t)256 * per_wg * 16)); CK(hipMemset(buf, 1, (size_t)256 * per_wg * 16)); CK(hipMalloc(&out, 64));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int iters = 400;
  auto run = [&](int mode) {
    CK(hipDeviceSynchronize());
    if (mode == 1) for (int i = 0; i < 4000; ++i) hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, s1, out);
    if (mode == 2) hipLaunchKernelGGL(k_long, dim3(1), dim3(64), 0, s1, out, 200000);
    CK(hipEventRecord(e0, s0));
    hipLaunchKernelGGL(k_reread, dim3(256), dim3(256), 0, s0, buf, per_wg, iters, out);
    CK(hipEventRecord(e1, s0));
    CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipDeviceSynchronize());
    return ms;
  };
  run(0);
  for (int rep = 0; rep < 2; ++rep) {
    const float a = run(0), b = run(1), c = run(2);
    const double gb = 256.0 * per_wg * 16 * iters / 1e9;
    printf("re-read 24 MB x %d: alone %.3f ms (%.1f TB/s) | beside 4000 tiny kernels on another stream %.3f ms (%.1f TB/s) | beside one long "
           "one-wavefront kernel %.3f ms (%.1f TB/s)\n", iters, a, gb / a, b, gb / b, c, gb / c);
  }
  return 0;
}
