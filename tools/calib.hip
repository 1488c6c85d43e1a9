// calib.hip -- calibration of the two peaks bench.py's fractions divide by.  Not part of the
// product: built by tools/calib.sh into build/calib, run on the GPU box, summary kept as
// profiles/r17_counter_calibration.md.
//
//   calib pmc     one launch of each memory-pattern kernel over buffers far larger than L2 + Infinity Cache, each
//                 with a KNOWN byte count (printed as "expect <kernel> <read bytes> <written bytes>").  Run under
//                 `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes); tools/calib_summary.py divides.
//                 The patterns are this design's own: wide coalesced streams (pass A / pass B), 16-byte record gathers
//                 (one per 64-byte line, one per 128-byte line, and the "mostly consecutive" shape of the list kernels),
//                 16-byte record stores into otherwise untouched 64-byte lines (G, in-segment sums, far-term bins), byte
//                 stores (flag table).
//   calib valu    VALU issue rate: independent v_fma_f32 / v_pk_fma_f32 / v_add_f32 chains at 1, 2, 4, 8 wavefronts per
//                 SIMD -> wave-instructions per cycle per CU (shader clock from s_memtime-free wall clock + clock64).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

constexpr int kBlock = 256;
typedef float v2f __attribute__((ext_vector_type(2)));

// ---- memory patterns ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock) cal_stream_read16(const uint4* __restrict__ a, size_t n, uint32_t* __restrict__ out) {
  uint32_t acc = 0;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) { const uint4 v = a[i]; acc ^= v.x ^ v.w; }
  if (acc == 0x12345u) out[0] = acc;
}
// 64 contiguous bytes per lane (pass A / pass B: four 16-byte records of consecutive slots)
__global__ void __launch_bounds__(kBlock) cal_stream_read64_per_lane(const uint4* __restrict__ a, size_t n, uint32_t* __restrict__ out) {
  uint32_t acc = 0;
  for (size_t i = ((size_t)blockIdx.x * kBlock + threadIdx.x) * 4; i + 3 < n; i += (size_t)gridDim.x * kBlock * 4) {
    const uint4 v0 = a[i], v1 = a[i + 1], v2 = a[i + 2], v3 = a[i + 3];
    acc ^= v0.x ^ v1.y ^ v2.z ^ v3.w;
  }
  if (acc == 0x12345u) out[0] = acc;
}
__global__ void __launch_bounds__(kBlock) cal_stream_read4(const uint32_t* __restrict__ a, size_t n, uint32_t* __restrict__ out) {
  uint32_t acc = 0;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) acc ^= a[i];
  if (acc == 0x12345u) out[0] = acc;
}
__global__ void __launch_bounds__(kBlock) cal_stream_read1(const uint8_t* __restrict__ a, size_t n, uint32_t* __restrict__ out) {
  uint32_t acc = 0;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) acc ^= a[i];
  if (acc == 0x12345u) out[0] = acc;
}
// one 16-byte record per `stride` records, lane k of the grid reads record perm-free index k * stride (+ a hash offset
// inside the stride window that keeps it in its own line): every touched line is touched once
template <int kTag>
__global__ void __launch_bounds__(kBlock) cal_gather16(const uint4* __restrict__ a, const uint32_t* __restrict__ idx, uint32_t n, uint32_t* __restrict__ out) {
  uint32_t acc = 0;
  for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) { const uint4 v = a[idx[i]]; acc ^= v.x ^ v.w; }
  if (acc == 0x12345u) out[0] = acc;
}
template <int kTag>
__global__ void __launch_bounds__(kBlock) cal_scatter16(uint4* __restrict__ a, const uint32_t* __restrict__ idx, uint32_t n) {
  for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) a[idx[i]] = make_uint4(i, i, i, i);
}
__global__ void __launch_bounds__(kBlock) cal_scatter1(uint8_t* __restrict__ a, const uint32_t* __restrict__ idx, uint32_t n) {
  for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) a[idx[i]] = (uint8_t)i;
}
__global__ void __launch_bounds__(kBlock) cal_stream_write16(uint4* __restrict__ a, size_t n) {
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) a[i] = make_uint4((uint32_t)i, 1u, 2u, 3u);
}
__global__ void __launch_bounds__(kBlock) cal_stream_write4(uint32_t* __restrict__ a, size_t n) {
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) a[i] = (uint32_t)i;
}
__global__ void __launch_bounds__(kBlock) cal_stream_write1(uint8_t* __restrict__ a, size_t n) {
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) a[i] = (uint8_t)i;
}
// read-modify-write of a 16-byte record in its own line (integrate: P / N / C of a visible slot)
__global__ void __launch_bounds__(kBlock) cal_rmw16(uint4* __restrict__ a, const uint32_t* __restrict__ idx, uint32_t n) {
  for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) { uint4 v = a[idx[i]]; v.x += 1u; a[idx[i]] = v; }
}

static std::vector<uint32_t> make_idx(uint32_t n, uint32_t stride, size_t records, int shape) {
  // shape 0: one record per `stride` records in a shuffled order (sparse gather); 1: runs of 48 consecutive records
  // followed by a gap of 208 (the list kernels' "four fifths consecutive" shape, every line touched by one run only)
  std::vector<uint32_t> v(n);
  if (shape == 0) {
    for (uint32_t i = 0; i < n; ++i) v[i] = i * stride;
    uint64_t s = 0x9E3779B97F4A7C15ull;
    for (uint32_t i = n - 1; i > 0; --i) { s = s * 6364136223846793005ull + 1442695040888963407ull; const uint32_t j = (uint32_t)((s >> 33) % (i + 1)); std::swap(v[i], v[j]); }
  } else {
    for (uint32_t i = 0; i < n; ++i) v[i] = (i / 48) * 256 + (i % 48);
  }
  for (uint32_t i = 0; i < n; ++i) if ((size_t)v[i] >= records) { printf("index out of range\n"); exit(1); }
  return v;
}

static void run_pmc() {
  const size_t bytes = (size_t)3 << 30;   // 3 GiB: twelve times the Infinity Cache
  const size_t rec = bytes / 16;
  uint4* a; uint32_t* out; uint32_t* idx;
  CK(hipMalloc(&a, bytes)); CK(hipMalloc(&out, 64)); CK(hipMemset(a, 1, bytes));
  const uint32_t n = 8u << 20;            // 8 Mi records per gather / scatter pattern
  CK(hipMalloc(&idx, (size_t)n * 4));
  const int grid = 256 * 16;
  auto flush = [&]() { CK(hipDeviceSynchronize()); };
  auto load_idx = [&](uint32_t stride, int shape, uint32_t count) {
    const std::vector<uint32_t> h = make_idx(count, stride, rec, shape);
    CK(hipMemcpy(idx, h.data(), (size_t)count * 4, hipMemcpyHostToDevice));
  };
  // streams over 1 GiB each (distinct GiB per kernel so that nothing is warm)
  const size_t g16 = ((size_t)1 << 30) / 16;
  hipLaunchKernelGGL(cal_stream_read16, dim3(grid), dim3(kBlock), 0, 0, a, g16, out); flush();
  printf("expect cal_stream_read16 %zu 0\n", g16 * 16);
  hipLaunchKernelGGL(cal_stream_read64_per_lane, dim3(grid), dim3(kBlock), 0, 0, a + g16, g16, out); flush();
  printf("expect cal_stream_read64_per_lane %zu 0\n", g16 * 16);
  hipLaunchKernelGGL(cal_stream_read4, dim3(grid), dim3(kBlock), 0, 0, reinterpret_cast<const uint32_t*>(a + 2 * g16), g16 * 4, out); flush();
  printf("expect cal_stream_read4 %zu 0\n", g16 * 16);
  hipLaunchKernelGGL(cal_stream_read1, dim3(grid), dim3(kBlock), 0, 0, reinterpret_cast<const uint8_t*>(a), (size_t)256 << 20, out); flush();
  printf("expect cal_stream_read1 %zu 0\n", (size_t)256 << 20);
  // gathers: one record per 64-byte line (stride 4), per 128-byte line (stride 8), and in consecutive runs
  struct G { const char* name; uint32_t stride; int shape; } gs[3] = {{"gather16_per_64B_line", 4, 0}, {"gather16_per_128B_line", 8, 0}, {"gather16_runs_of_48", 0, 1}};
  for (int t = 0; t < 3; ++t) {
    const G& g = gs[t];
    load_idx(g.stride, g.shape, n);
    if (t == 0) hipLaunchKernelGGL(cal_gather16<0>, dim3(grid), dim3(kBlock), 0, 0, a, idx, n, out);
    if (t == 1) hipLaunchKernelGGL(cal_gather16<1>, dim3(grid), dim3(kBlock), 0, 0, a, idx, n, out);
    if (t == 2) hipLaunchKernelGGL(cal_gather16<2>, dim3(grid), dim3(kBlock), 0, 0, a, idx, n, out);
    flush();
    printf("expect cal_gather16<%d> %zu 0 # %s (+ %zu index bytes read)\n", t, (size_t)n * 16 + (size_t)n * 4, g.name, (size_t)n * 4);
  }
  for (int t = 0; t < 3; ++t) {
    const G& g = gs[t];
    load_idx(g.stride, g.shape, n);
    if (t == 0) hipLaunchKernelGGL(cal_scatter16<0>, dim3(grid), dim3(kBlock), 0, 0, a, idx, n);
    if (t == 1) hipLaunchKernelGGL(cal_scatter16<1>, dim3(grid), dim3(kBlock), 0, 0, a, idx, n);
    if (t == 2) hipLaunchKernelGGL(cal_scatter16<2>, dim3(grid), dim3(kBlock), 0, 0, a, idx, n);
    flush();
    printf("expect cal_scatter16<%d> %zu %zu # %s (index bytes read)\n", t, (size_t)n * 4, (size_t)n * 16, g.name);
  }
  load_idx(4, 0, n);
  hipLaunchKernelGGL(cal_rmw16, dim3(grid), dim3(kBlock), 0, 0, a, idx, n); flush();
  printf("expect cal_rmw16 %zu %zu # one record per 64-byte line (incl. %zu index bytes read)\n", (size_t)n * 16 + (size_t)n * 4, (size_t)n * 16, (size_t)n * 4);
  load_idx(16, 0, n);   // one byte per 16 bytes of a byte array (idx = byte offsets)
  hipLaunchKernelGGL(cal_scatter1, dim3(grid), dim3(kBlock), 0, 0, reinterpret_cast<uint8_t*>(a), idx, n); flush();
  printf("expect cal_scatter1 %zu %zu # one byte per 16 bytes\n", (size_t)n * 4, (size_t)n);
  hipLaunchKernelGGL(cal_stream_write16, dim3(grid), dim3(kBlock), 0, 0, a, g16); flush();
  printf("expect cal_stream_write16 0 %zu\n", g16 * 16);
  hipLaunchKernelGGL(cal_stream_write4, dim3(grid), dim3(kBlock), 0, 0, reinterpret_cast<uint32_t*>(a + g16), g16 * 4); flush();
  printf("expect cal_stream_write4 0 %zu\n", g16 * 16);
  hipLaunchKernelGGL(cal_stream_write1, dim3(grid), dim3(kBlock), 0, 0, reinterpret_cast<uint8_t*>(a + 2 * g16), (size_t)256 << 20); flush();
  printf("expect cal_stream_write1 0 %zu\n", (size_t)256 << 20);
  CK(hipFree(a)); CK(hipFree(out)); CK(hipFree(idx));
}

// ---- VALU issue -----------------------------------------------------------------------------------------------------
// kOp: 0 = v_fma_f32, 1 = v_pk_fma_f32, 2 = v_add_f32, 3 = v_pk_add_f32, 4 = v_add_u32 (VALU integer), 5 = v_exp_f32
// 16 independent accumulators (no dependent-issue stall), 64 instructions per loop body
template <int kOp>
__global__ void __launch_bounds__(256) cal_valu(int iters, float* __restrict__ out, long long* __restrict__ cycles) {
  float a[16];
  v2f p[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) { a[k] = (float)(threadIdx.x + k); p[k] = v2f{a[k], a[k] + 1.0f}; }
  const float m = 0.999f, c = 0.001f;
  const v2f m2 = {m, m}, c2 = {c, c};
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int rep = 0; rep < 4; ++rep) {
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        if (kOp == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(m), "v"(c));
        if (kOp == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[k]) : "v"(m2), "v"(c2));
        if (kOp == 2) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[k]) : "v"(c));
        if (kOp == 3) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[k]) : "v"(c2));
        if (kOp == 4) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[k]) : "v"(c));
        if (kOp == 5) asm volatile("v_exp_f32 %0, %0" : "+v"(a[k]));
      }
    }
  }
  const long long t1 = clock64();
  float s = 0;
#pragma unroll
  for (int k = 0; k < 16; ++k) s += a[k] + p[k].x + p[k].y;
  if (s == 1234.5678f) out[0] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cycles[0] = t1 - t0;
}

template <int kOp>
static void valu_case(const char* name, int cus) {
  float* out; long long* cyc;
  CK(hipMalloc(&out, 64)); CK(hipMalloc(&cyc, 8));
  const int iters = 20000;   // x 64 instructions
  for (int waves_per_simd : {1, 2, 4, 8}) {
    // 256-lane workgroups = one wavefront per SIMD of a CU; `waves_per_simd` workgroups per CU
    const int grid = cus * waves_per_simd;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(cal_valu<kOp>, dim3(grid), dim3(256), 0, 0, 1000, out, cyc);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(cal_valu<kOp>, dim3(grid), dim3(256), 0, 0, iters, out, cyc);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    long long hc = 0; CK(hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost));
    const double wave_insts = (double)grid * 4.0 * (double)iters * 64.0;
    // clock64() = s_memtime: a constant-rate counter (100 MHz on this family), not the shader clock; the rate per
    // shader cycle is therefore given against the 2.4 GHz boost clock and against what the run itself implies is irrelevant
    const double per_s = wave_insts / (ms * 1e-3);
    printf("valu %-12s waves/SIMD %d  %8.3f ms  %7.1f G wave-inst/s  = %.3f per CU per cycle at 2.4 GHz  (memtime ticks %lld)\n",
           name, waves_per_simd, ms, per_s / 1e9, per_s / (cus * 2.4e9), hc);
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
  }
  CK(hipFree(out)); CK(hipFree(cyc));
}

// ---- achievable bandwidth per pattern (calib bw): the same kernels, timed; bytes = what the request counters showed
// they move (128 bytes per gathered record, 32 per sparse 16-byte store)
template <class F>
static double time_ms(F f) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  f(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0, 0));
  for (int r = 0; r < 3; ++r) f();
  CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
  float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / 3.0;
}
static void run_bw() {
  const size_t bytes = (size_t)3 << 30;
  const size_t rec = bytes / 16;
  uint4* a; uint32_t* out; uint32_t* idx;
  CK(hipMalloc(&a, bytes)); CK(hipMalloc(&out, 64)); CK(hipMemset(a, 1, bytes));
  const uint32_t n = 8u << 20;
  CK(hipMalloc(&idx, (size_t)n * 4));
  const size_t g16 = ((size_t)1 << 30) / 16;
  auto load_idx = [&](uint32_t stride, int shape) {
    const std::vector<uint32_t> h = make_idx(n, stride, rec, shape);
    CK(hipMemcpy(idx, h.data(), (size_t)n * 4, hipMemcpyHostToDevice));
  };
  for (int grid : {256 * 8, 256 * 32}) {
    double ms = time_ms([&] { hipLaunchKernelGGL(cal_stream_read16, dim3(grid), dim3(kBlock), 0, 0, a, g16 * 3, out); });
    printf("bw grid %5d  stream read 3 GiB            %7.3f ms  %6.2f TB/s\n", grid, ms, 3.0 * 1.0737 / ms);
    ms = time_ms([&] { hipLaunchKernelGGL(cal_stream_write16, dim3(grid), dim3(kBlock), 0, 0, a, g16 * 3); });
    printf("bw grid %5d  stream write 3 GiB           %7.3f ms  %6.2f TB/s\n", grid, ms, 3.0 * 1.0737 / ms);
    load_idx(8, 0);
    ms = time_ms([&] { hipLaunchKernelGGL(cal_gather16<1>, dim3(grid), dim3(kBlock), 0, 0, a, idx, n, out); });
    printf("bw grid %5d  8 Mi shuffled 16 B gathers   %7.3f ms  %6.2f G gathers/s  %6.2f TB/s of 128-byte lines\n", grid, ms, n / ms / 1e6, n * 128.0 / ms / 1e9);
    ms = time_ms([&] { hipLaunchKernelGGL(cal_scatter16<1>, dim3(grid), dim3(kBlock), 0, 0, a, idx, n); });
    printf("bw grid %5d  8 Mi shuffled 16 B stores    %7.3f ms  %6.2f G stores/s\n", grid, ms, n / ms / 1e6);
    ms = time_ms([&] { hipLaunchKernelGGL(cal_rmw16, dim3(grid), dim3(kBlock), 0, 0, a, idx, n); });
    printf("bw grid %5d  8 Mi shuffled 16 B rmw       %7.3f ms  %6.2f G rmw/s\n", grid, ms, n / ms / 1e6);
    load_idx(0, 1);
    ms = time_ms([&] { hipLaunchKernelGGL(cal_gather16<2>, dim3(grid), dim3(kBlock), 0, 0, a, idx, n, out); });
    printf("bw grid %5d  8 Mi gathers in runs of 48   %7.3f ms  %6.2f G gathers/s  %6.2f TB/s of record bytes\n", grid, ms, n / ms / 1e6, n * 16.0 / ms / 1e9);
  }
  CK(hipFree(a)); CK(hipFree(out)); CK(hipFree(idx));
}

int main(int argc, char** argv) {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  printf("# device %s, %d CUs, clockRate %d kHz\n", prop.name, prop.multiProcessorCount, prop.clockRate);
  if (argc > 1 && !strcmp(argv[1], "pmc")) { run_pmc(); return 0; }
  if (argc > 1 && !strcmp(argv[1], "bw")) { run_bw(); return 0; }
  const int cus = prop.multiProcessorCount;
  valu_case<0>("v_fma_f32", cus);
  valu_case<1>("v_pk_fma_f32", cus);
  valu_case<2>("v_add_f32", cus);
  valu_case<3>("v_pk_add_f32", cus);
  valu_case<4>("v_add_u32", cus);
  valu_case<5>("v_exp_f32", cus);
  return 0;
}
