// ubench.hip -- micro-measurements that the design decisions in DESIGN.md cite (gfx950).  Not part of the product:
// built by tools/ubench_build.sh into build/ubench, run on the GPU box, output kept under profiles/.
//   1. per-pixel atomics (the association images): min / add32 / add64, device scope vs workgroup scope,
//      random vs sorted pixel order;
//   2. random 16-byte / 32-byte record gathers and 16-byte scattered stores over a 5 M-slot array (the regulariser);
//   3. launch boundary, cross-stream event hand-off, and the same dependency pattern replayed from a hipGraph.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

constexpr int kBlock = 256;

__host__ __device__ inline uint32_t hash32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

// mode: 0 = min agent, 1 = add32 agent, 2 = add64 agent, 3 = min workgroup, 4 = add32 workgroup, 5 = add64 workgroup,
// 6 = all three agent (the associate kernel's mix), 7 = all three workgroup, 8 = plain stores (no atomic), 9 = packed add64 + min
template <int kMode>
__global__ void __launch_bounds__(kBlock)
k_atomics(uint32_t* __restrict__ img32, unsigned long long* __restrict__ img64, uint32_t* __restrict__ img32b,
          const uint32_t* __restrict__ pix, uint32_t n) {
  for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
    const uint32_t p = pix[i];
    const uint32_t v = i;
    if (kMode == 0 || kMode == 6 || kMode == 9) atomicMin(&img32[p], v);
    if (kMode == 1 || kMode == 6) atomicAdd(&img32b[p], 1u);
    if (kMode == 2 || kMode == 6 || kMode == 9) atomicAdd(&img64[p], (unsigned long long)v * 977u);
    if (kMode == 3 || kMode == 7) __hip_atomic_fetch_min(&img32[p], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (kMode == 4 || kMode == 7) __hip_atomic_fetch_add(&img32b[p], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (kMode == 5 || kMode == 7) __hip_atomic_fetch_add(&img64[p], (unsigned long long)v * 977u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (kMode == 8) { img32[p] = v; img32b[p] = 1u; img64[p] = v; }
  }
}

__global__ void __launch_bounds__(kBlock)
k_gather16(const float4* __restrict__ rec, const uint32_t* __restrict__ idx, uint32_t n, float* __restrict__ out) {
  float acc = 0;
  for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
    const float4 a = rec[idx[i]];
    acc += a.x + a.w;
  }
  if (acc == 12345.678f) out[0] = acc;
}
// four gathers in flight per lane (as the regulariser has: 4 neighbours)
__global__ void __launch_bounds__(kBlock)
k_gather16x4(const float4* __restrict__ rec, const uint32_t* __restrict__ idx, uint32_t n, float* __restrict__ out) {
  float acc = 0;
  for (uint32_t i = (blockIdx.x * kBlock + threadIdx.x) * 4; i + 3 < n; i += gridDim.x * kBlock * 4) {
    const uint4 j = *reinterpret_cast<const uint4*>(&idx[i]);
    const float4 a = rec[j.x], b = rec[j.y], c = rec[j.z], d = rec[j.w];
    acc += a.x + b.y + c.z + d.w;
  }
  if (acc == 12345.678f) out[0] = acc;
}
// two 16-byte records from two arrays (S and T today) vs one 32-byte record
__global__ void __launch_bounds__(kBlock)
k_gather2x16x4(const float4* __restrict__ recA, const float4* __restrict__ recB, const uint32_t* __restrict__ idx, uint32_t n, float* __restrict__ out) {
  float acc = 0;
  for (uint32_t i = (blockIdx.x * kBlock + threadIdx.x) * 4; i + 3 < n; i += gridDim.x * kBlock * 4) {
    const uint4 j = *reinterpret_cast<const uint4*>(&idx[i]);
    const float4 a = recA[j.x], b = recA[j.y], c = recA[j.z], d = recA[j.w];
    const float4 a2 = recB[j.x], b2 = recB[j.y], c2 = recB[j.z], d2 = recB[j.w];
    acc += a.x + b.y + c.z + d.w + a2.x + b2.y + c2.z + d2.w;
  }
  if (acc == 12345.678f) out[0] = acc;
}
__global__ void __launch_bounds__(kBlock)
k_gather32x4(const float4* __restrict__ rec, const uint32_t* __restrict__ idx, uint32_t n, float* __restrict__ out) {
  float acc = 0;
  for (uint32_t i = (blockIdx.x * kBlock + threadIdx.x) * 4; i + 3 < n; i += gridDim.x * kBlock * 4) {
    const uint4 j = *reinterpret_cast<const uint4*>(&idx[i]);
    const float4 a = rec[2 * (size_t)j.x], b = rec[2 * (size_t)j.y], c = rec[2 * (size_t)j.z], d = rec[2 * (size_t)j.w];
    const float4 a2 = rec[2 * (size_t)j.x + 1], b2 = rec[2 * (size_t)j.y + 1], c2 = rec[2 * (size_t)j.z + 1], d2 = rec[2 * (size_t)j.w + 1];
    acc += a.x + b.y + c.z + d.w + a2.x + b2.y + c2.z + d2.w;
  }
  if (acc == 12345.678f) out[0] = acc;
}
__global__ void __launch_bounds__(kBlock)
k_scatter16(float4* __restrict__ rec, const uint32_t* __restrict__ idx, uint32_t n) {
  for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock)
    rec[idx[i]] = make_float4((float)i, 1.f, 2.f, 3.f);
}
__global__ void __launch_bounds__(kBlock)
k_stream16(const float4* __restrict__ rec, uint32_t n, float* __restrict__ out) {
  float acc = 0;
  for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) acc += rec[i].x;
  if (acc == 12345.678f) out[0] = acc;
}
__global__ void k_tiny(uint32_t* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1; }
__global__ void __launch_bounds__(kBlock) k_work(float* p, int iters) {
  float a = p[blockIdx.x * kBlock + threadIdx.x];
  for (int i = 0; i < iters; ++i) a = a * 1.0001f + 0.5f;
  p[blockIdx.x * kBlock + threadIdx.x] = a;
}

template <typename F>
float time_ms(F f, int reps, hipStream_t st = nullptr) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  f();
  CK(hipStreamSynchronize(st));
  CK(hipEventRecord(a, st));
  for (int r = 0; r < reps; ++r) f();
  CK(hipEventRecord(b, st));
  CK(hipEventSynchronize(b));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, a, b));
  CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
  return ms / reps;
}

int main() {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  printf("# device %s, %d CUs\n", prop.name, prop.multiProcessorCount);
  const int grid = prop.multiProcessorCount * 8;

  // ---- 1. per-pixel atomics -------------------------------------------------------------------------------------
  {
    const uint32_t P = 640 * 480, n = 720000;   // ~0.36 M visible surfels x 2 pixels
    uint32_t *img32, *img32b, *pix;
    unsigned long long* img64;
    CK(hipMalloc(&img32, P * 4)); CK(hipMalloc(&img32b, P * 4)); CK(hipMalloc(&img64, P * 8)); CK(hipMalloc(&pix, n * 4));
    std::vector<uint32_t> h(n);
    for (int pattern = 0; pattern < 3; ++pattern) {
      // 0: random pixels; 1: image-coherent (consecutive surfels -> neighbouring pixels in 20-px column strips, as
      // surfels created by a panning camera are); 2: fully sorted by pixel
      for (uint32_t i = 0; i < n; ++i) {
        if (pattern == 0) h[i] = hash32(i) % P;
        else if (pattern == 1) { const uint32_t s = i / 2, strip = s / (20 * 480), in = s % (20 * 480); const uint32_t y = in / 20, x = (strip * 20 + in % 20) % 640; h[i] = y * 640 + x + (i & 1 ? (hash32(i) & 1 ? 1 : 640) : 0); if (h[i] >= P) h[i] = P - 1; }
        else h[i] = (uint32_t)((uint64_t)i * P / n);
      }
      CK(hipMemcpy(pix, h.data(), n * 4, hipMemcpyHostToDevice));
      const char* pn[3] = {"random", "strips", "sorted"};
      const char* mn[10] = {"min.agent", "add32.agent", "add64.agent", "min.wg", "add32.wg", "add64.wg", "all3.agent", "all3.wg", "plain-stores", "min+add64.agent"};
      float ms[10];
      ms[0] = time_ms([&] { hipLaunchKernelGGL(k_atomics<0>, dim3(grid), dim3(kBlock), 0, 0, img32, img64, img32b, pix, n); }, 20);
      ms[1] = time_ms([&] { hipLaunchKernelGGL(k_atomics<1>, dim3(grid), dim3(kBlock), 0, 0, img32, img64, img32b, pix, n); }, 20);
      ms[2] = time_ms([&] { hipLaunchKernelGGL(k_atomics<2>, dim3(grid), dim3(kBlock), 0, 0, img32, img64, img32b, pix, n); }, 20);
      ms[3] = time_ms([&] { hipLaunchKernelGGL(k_atomics<3>, dim3(grid), dim3(kBlock), 0, 0, img32, img64, img32b, pix, n); }, 20);
      ms[4] = time_ms([&] { hipLaunchKernelGGL(k_atomics<4>, dim3(grid), dim3(kBlock), 0, 0, img32, img64, img32b, pix, n); }, 20);
      ms[5] = time_ms([&] { hipLaunchKernelGGL(k_atomics<5>, dim3(grid), dim3(kBlock), 0, 0, img32, img64, img32b, pix, n); }, 20);
      ms[6] = time_ms([&] { hipLaunchKernelGGL(k_atomics<6>, dim3(grid), dim3(kBlock), 0, 0, img32, img64, img32b, pix, n); }, 20);
      ms[7] = time_ms([&] { hipLaunchKernelGGL(k_atomics<7>, dim3(grid), dim3(kBlock), 0, 0, img32, img64, img32b, pix, n); }, 20);
      ms[8] = time_ms([&] { hipLaunchKernelGGL(k_atomics<8>, dim3(grid), dim3(kBlock), 0, 0, img32, img64, img32b, pix, n); }, 20);
      ms[9] = time_ms([&] { hipLaunchKernelGGL(k_atomics<9>, dim3(grid), dim3(kBlock), 0, 0, img32, img64, img32b, pix, n); }, 20);
      for (int m = 0; m < 10; ++m)
        printf("atomics %-7s %-16s n=%u  %8.2f us  %7.2f Gop/s\n", pn[pattern], mn[m], n, ms[m] * 1e3, n / (ms[m] * 1e-3) / 1e9);
    }
    CK(hipFree(img32)); CK(hipFree(img32b)); CK(hipFree(img64)); CK(hipFree(pix));
  }

  // ---- 2. record gathers / scattered stores ---------------------------------------------------------------------
  {
    const uint32_t N = 5200000, n = 1600000;   // slots; window edges per frame
    float4 *rec, *rec2; uint32_t* idx; float* out;
    CK(hipMalloc(&rec, (size_t)N * 32)); CK(hipMalloc(&rec2, (size_t)N * 16)); CK(hipMalloc(&idx, n * 4)); CK(hipMalloc(&out, 64));
    CK(hipMemset(rec, 0, (size_t)N * 32)); CK(hipMemset(rec2, 0, (size_t)N * 16));
    std::vector<uint32_t> h(n);
    for (int pattern = 0; pattern < 2; ++pattern) {
      // 0: uniformly random slots; 1: clustered like the recent set (400 k slots at 28 % density in ~1400 segments,
      // neighbours within +-20 / +-600 slots)
      for (uint32_t i = 0; i < n; ++i) {
        if (pattern == 0) h[i] = hash32(i) % N;
        else { const uint32_t src = (hash32(i / 4) % 1400) * 3600 + (hash32(i / 4 + 77) % 1024); const int d[4] = {-1, 1, -600, 600}; h[i] = (uint32_t)((int)src + d[i & 3] + (int)(hash32(i) % 5)) % N; }
      }
      CK(hipMemcpy(idx, h.data(), n * 4, hipMemcpyHostToDevice));
      const char* pn[2] = {"random", "clustered"};
      const float g1 = time_ms([&] { hipLaunchKernelGGL(k_gather16, dim3(grid), dim3(kBlock), 0, 0, rec, idx, n, out); }, 20);
      const float g4 = time_ms([&] { hipLaunchKernelGGL(k_gather16x4, dim3(grid), dim3(kBlock), 0, 0, rec, idx, n, out); }, 20);
      const float g216 = time_ms([&] { hipLaunchKernelGGL(k_gather2x16x4, dim3(grid), dim3(kBlock), 0, 0, rec, rec2, idx, n, out); }, 20);
      const float g32 = time_ms([&] { hipLaunchKernelGGL(k_gather32x4, dim3(grid), dim3(kBlock), 0, 0, rec, idx, n, out); }, 20);
      const float s16 = time_ms([&] { hipLaunchKernelGGL(k_scatter16, dim3(grid), dim3(kBlock), 0, 0, rec2, idx, n); }, 20);
      printf("gather  %-9s 16B x1/lane   n=%u %8.2f us  %6.2f G/s\n", pn[pattern], n, g1 * 1e3, n / (g1 * 1e-3) / 1e9);
      printf("gather  %-9s 16B x4/lane   n=%u %8.2f us  %6.2f G/s\n", pn[pattern], n, g4 * 1e3, n / (g4 * 1e-3) / 1e9);
      printf("gather  %-9s 2x16B x4/lane n=%u %8.2f us  %6.2f G/s\n", pn[pattern], n, g216 * 1e3, n / (g216 * 1e-3) / 1e9);
      printf("gather  %-9s 32B x4/lane   n=%u %8.2f us  %6.2f G/s\n", pn[pattern], n, g32 * 1e3, n / (g32 * 1e-3) / 1e9);
      printf("scatter %-9s 16B stores    n=%u %8.2f us  %6.2f G/s\n", pn[pattern], n, s16 * 1e3, n / (s16 * 1e-3) / 1e9);
    }
    const float st = time_ms([&] { hipLaunchKernelGGL(k_stream16, dim3(grid * 4), dim3(kBlock), 0, 0, rec, N, out); }, 20);
    printf("stream  16B records N=%u %8.2f us  %6.2f TB/s\n", N, st * 1e3, (double)N * 16 / (st * 1e-3) / 1e12);
    CK(hipFree(rec)); CK(hipFree(rec2)); CK(hipFree(idx)); CK(hipFree(out));
  }

  // ---- 3. launch boundaries and cross-stream hand-offs ------------------------------------------------------------
  {
    uint32_t* c; float* w;
    CK(hipMalloc(&c, 64)); CK(hipMemset(c, 0, 64));
    CK(hipMalloc(&w, 2048 * kBlock * 4)); CK(hipMemset(w, 0, 2048 * kBlock * 4));
    hipStream_t s0, s1;
    int lo = 0, hi = 0;
    CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
    CK(hipStreamCreateWithPriority(&s1, hipStreamNonBlocking, hi));
    hipEvent_t e01, e10;
    CK(hipEventCreateWithFlags(&e01, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&e10, hipEventDisableTiming));
    const int chain = 200;
    // (a) dependent tiny kernels on one stream
    const float same = time_ms([&] { for (int i = 0; i < chain; ++i) hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, s0, c); }, 5, s0);
    printf("boundary same-stream tiny kernels: %6.2f us per kernel\n", same * 1e3 / chain);
    // (b) dependent 20-us kernels on one stream: per-kernel time minus the body
    const int iters = 4000;
    const float body = time_ms([&] { hipLaunchKernelGGL(k_work, dim3(2048), dim3(kBlock), 0, s0, w, iters); }, 20, s0);
    const float same_w = time_ms([&] { for (int i = 0; i < chain; ++i) hipLaunchKernelGGL(k_work, dim3(2048), dim3(kBlock), 0, s0, w, iters); }, 3, s0);
    printf("boundary same-stream %5.1f-us kernels: %6.2f us per kernel (body alone %6.2f)\n", body * 1e3, same_w * 1e3 / chain, body * 1e3);
    // (c) ping-pong between two streams through events
    auto pingpong = [&] {
      for (int i = 0; i < chain / 2; ++i) {
        hipLaunchKernelGGL(k_work, dim3(2048), dim3(kBlock), 0, s0, w, iters);
        CK(hipEventRecord(e01, s0)); CK(hipStreamWaitEvent(s1, e01, 0));
        hipLaunchKernelGGL(k_work, dim3(2048), dim3(kBlock), 0, s1, w, iters);
        CK(hipEventRecord(e10, s1)); CK(hipStreamWaitEvent(s0, e10, 0));
      }
    };
    const float pp = time_ms(pingpong, 3, s0);
    printf("hand-off two-stream ping-pong:         %6.2f us per kernel -> %6.2f us per hand-off\n", pp * 1e3 / chain, pp * 1e3 / chain - same_w * 1e3 / chain);
    // (d) the same ping-pong captured into a hipGraph
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s0, hipStreamCaptureModeGlobal));
    pingpong();
    CK(hipStreamEndCapture(s0, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    const float gp = time_ms([&] { CK(hipGraphLaunch(ge, s0)); }, 3, s0);
    printf("hand-off ping-pong as hipGraph replay:  %6.2f us per kernel\n", gp * 1e3 / chain);
    // (e) single-stream chain captured into a graph
    hipGraph_t g2; hipGraphExec_t ge2;
    CK(hipStreamBeginCapture(s0, hipStreamCaptureModeGlobal));
    for (int i = 0; i < chain; ++i) hipLaunchKernelGGL(k_work, dim3(2048), dim3(kBlock), 0, s0, w, iters);
    CK(hipStreamEndCapture(s0, &g2));
    CK(hipGraphInstantiate(&ge2, g2, nullptr, nullptr, 0));
    const float gs = time_ms([&] { CK(hipGraphLaunch(ge2, s0)); }, 3, s0);
    printf("boundary same-stream chain as hipGraph:  %6.2f us per kernel\n", gs * 1e3 / chain);
  }
  return 0;
}
