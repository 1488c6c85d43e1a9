// boundary.hip -- what a launch boundary costs behind a kernel that leaves dirty lines in the L2s, and which store flavour
// makes it cheaper (MI355X_MICROARCH.md price table, row "boundary": 1.45-1.9 us + B / 6 TB/s for the B
// bytes the predecessor leaves dirty).  Not part of the product:
//   hipcc --offload-arch=gfx950 -O3 tools/boundary.hip -o build/boundary ; run on the GPU box; output kept under profiles/.
//
// A writer kernel stores B bytes as 16-byte records (dense: consecutive lanes, consecutive records; sparse: every second
// 16-byte slot, i.e. half-filled 32-byte sectors -- what the far-bin records of k_reg_accumulate look like), with
//   plain   global_store_dwordx4
//   nt      __builtin_nontemporal_store (global_store_dwordx4 ... nt)
//   sc1     global_store_dwordx4 ... sc1          (write-through, the line is dropped from the L2)
//   sc0sc1  global_store_dwordx4 ... sc0 sc1
// followed IN THE SAME STREAM by a trivial dependent kernel of 256 workgroups.  Every workgroup of both kernels stamps the
// device's wall clock (s_memrealtime, 100 MHz) at its begin and end; boundary = min(begin of the follower's workgroups) -
// max(end of the writer's workgroups), median of 31 repetitions.  Also reported: the writer's own duration (first begin ->
// last end) and the pair's period when 40 pairs run back to back (events around the chain), which is what a frame pays.
// A third column repeats the measurement with the follower on ANOTHER stream behind an event (the frame's hand-overs).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

constexpr int kBlock = 256;

__device__ inline uint64_t now() { return __builtin_amdgcn_s_memrealtime(); }

enum { kPlain = 0, kNt = 1, kSc1 = 2, kSc0Sc1 = 3 };

typedef uint32_t v4u __attribute__((ext_vector_type(4)));

template <int kMode>
__device__ inline void store16(uint4* p, uint4 v) {
  const v4u w = {v.x, v.y, v.z, v.w};
  if (kMode == kPlain) *p = v;
  else if (kMode == kNt) __builtin_nontemporal_store(w, reinterpret_cast<v4u*>(p));
  else if (kMode == kSc1) asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(w) : "memory");
  else asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(p), "v"(w) : "memory");
}

// n records of 16 bytes; stride 1 (dense) or 2 (every second slot).  Grid-stride, 4 records per lane per step like the
// product's streaming kernels.
template <int kMode>
__global__ void __launch_bounds__(kBlock) k_writer(uint4* __restrict__ out, uint32_t n, uint32_t stride, uint64_t* __restrict__ stamps) {
  const uint64_t t0 = now();
  for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock)
    store16<kMode>(out + (size_t)i * stride, make_uint4(i, i ^ 0x5555u, blockIdx.x, 7u));
  __syncthreads();
  if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = t0; stamps[2 * blockIdx.x + 1] = now(); }
}

__global__ void __launch_bounds__(kBlock) k_follower(const uint4* __restrict__ in, uint32_t n, uint32_t stride, uint64_t* __restrict__ stamps, uint32_t* __restrict__ sink) {
  const uint64_t t0 = now();
  // (reads one record the writer stored: a real dependency; the value is checked on the host through `sink`)
  const uint32_t i = (blockIdx.x * 9973u) % (n ? n : 1u);
  const uint4 v = n ? in[(size_t)i * stride] : make_uint4(0, 0, 0, 0);
  if (threadIdx.x == 0) { sink[blockIdx.x] = v.x ^ i; stamps[2 * blockIdx.x] = t0; stamps[2 * blockIdx.x + 1] = now(); }
}

// The same follower with a private-segment (scratch) frame of 20 bytes per lane that is hardly used -- what k_integrate's code
// object asks for (SGPR spill slots): does a dispatch that needs scratch start later?
__global__ void __launch_bounds__(kBlock) k_follower_scratch(const uint4* __restrict__ in, uint32_t n, uint32_t stride, uint64_t* __restrict__ stamps, uint32_t* __restrict__ sink) {
  const uint64_t t0 = now();
  volatile uint32_t tmp[5];
  tmp[threadIdx.x % 5u] = threadIdx.x;
  const uint32_t i = (blockIdx.x * 9973u) % (n ? n : 1u);
  const uint4 v = n ? in[(size_t)i * stride] : make_uint4(0, 0, 0, 0);
  if (threadIdx.x == 0) { sink[blockIdx.x] = (v.x ^ i) + (tmp[0] & 0u); stamps[2 * blockIdx.x] = t0; stamps[2 * blockIdx.x + 1] = now(); }
}

struct Result { double boundary_us, writer_us, pair_period_us; int bad; };

template <int kMode>
static Result measure(uint4* buf, size_t bytes, uint32_t stride, bool cross_stream, hipStream_t s0, hipStream_t s1, uint64_t* d_st, uint32_t* d_sink) {
  const uint32_t n = (uint32_t)(bytes / 16);
  const int wg_w = 256 * 8, wg_f = 256;
  std::vector<uint64_t> st(2 * (wg_w + wg_f));
  std::vector<uint32_t> sink(wg_f);
  std::vector<double> bnd, wr;
  hipEvent_t ev;
  CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  Result r{};
  for (int rep = 0; rep < 31; ++rep) {
    hipLaunchKernelGGL(k_writer<kMode>, dim3(wg_w), dim3(kBlock), 0, s0, buf, n, stride, d_st);
    if (cross_stream) { CK(hipEventRecord(ev, s0)); CK(hipStreamWaitEvent(s1, ev, 0)); }
    hipLaunchKernelGGL(k_follower, dim3(wg_f), dim3(kBlock), 0, cross_stream ? s1 : s0, buf, n, stride, d_st + 2 * wg_w, d_sink);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(st.data(), d_st, st.size() * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(sink.data(), d_sink, sink.size() * 4, hipMemcpyDeviceToHost));
    uint64_t w_b = ~0ull, w_e = 0, f_b = ~0ull;
    for (int b = 0; b < wg_w; ++b) { w_b = std::min(w_b, st[2 * b]); w_e = std::max(w_e, st[2 * b + 1]); }
    for (int b = 0; b < wg_f; ++b) f_b = std::min(f_b, st[2 * (wg_w + b)]);
    bnd.push_back(((double)f_b - (double)w_e) * 0.01);   // 100 MHz -> us
    wr.push_back(((double)w_e - (double)w_b) * 0.01);
    if (n) for (int b = 0; b < wg_f; ++b) if (sink[b] != 0u) ++r.bad;   // v.x == i: the follower saw the writer's record
  }
  std::sort(bnd.begin(), bnd.end());
  std::sort(wr.begin(), wr.end());
  r.boundary_us = bnd[bnd.size() / 2];
  r.writer_us = wr[wr.size() / 2];
  // 40 pairs back to back
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0, s0));
  for (int i = 0; i < 40; ++i) {
    hipLaunchKernelGGL(k_writer<kMode>, dim3(wg_w), dim3(kBlock), 0, s0, buf, n, stride, d_st);
    if (cross_stream) { CK(hipEventRecord(ev, s0)); CK(hipStreamWaitEvent(s1, ev, 0)); }
    hipLaunchKernelGGL(k_follower, dim3(wg_f), dim3(kBlock), 0, cross_stream ? s1 : s0, buf, n, stride, d_st + 2 * wg_w, d_sink);
    if (cross_stream) { CK(hipEventRecord(ev, s1)); CK(hipStreamWaitEvent(s0, ev, 0)); }
  }
  CK(hipEventRecord(e1, s0));
  CK(hipDeviceSynchronize());
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  r.pair_period_us = ms * 1e3 / 40;
  CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1)); CK(hipEventDestroy(ev));
  return r;
}

int main() {
  const size_t max_bytes = 128u << 20;
  uint4* buf; uint64_t* d_st; uint32_t* d_sink;
  CK(hipMalloc(&buf, 2 * max_bytes));
  CK(hipMalloc(&d_st, 2 * (256 * 8 + 256) * 8));
  CK(hipMalloc(&d_sink, 256 * 4));
  CK(hipMemset(buf, 0, 2 * max_bytes));
  hipStream_t s0, s1;
  CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  const char* names[4] = {"plain", "nt", "sc1", "sc0sc1"};
  const size_t sizes[] = {0, 1u << 20, 8u << 20, 16u << 20, 32u << 20, 64u << 20};
  printf("# boundary behind a writer of B bytes of 16-byte records (us; median of 31; s_memrealtime stamps)\n");
  printf("# %-7s %-6s %6s | same stream: %8s %8s %8s | other stream (event): %8s %8s | stale\n", "stores", "layout", "MB",
         "boundary", "writer", "pair", "boundary", "pair");
  for (int cross_first = 0; cross_first < 1; ++cross_first)
    for (uint32_t stride = 1; stride <= 2; ++stride)
      for (size_t bytes : sizes)
        for (int m = 0; m < 4; ++m) {
          Result a, b;
          switch (m) {
            case 0: a = measure<kPlain>(buf, bytes, stride, false, s0, s1, d_st, d_sink); b = measure<kPlain>(buf, bytes, stride, true, s0, s1, d_st, d_sink); break;
            case 1: a = measure<kNt>(buf, bytes, stride, false, s0, s1, d_st, d_sink); b = measure<kNt>(buf, bytes, stride, true, s0, s1, d_st, d_sink); break;
            case 2: a = measure<kSc1>(buf, bytes, stride, false, s0, s1, d_st, d_sink); b = measure<kSc1>(buf, bytes, stride, true, s0, s1, d_st, d_sink); break;
            default: a = measure<kSc0Sc1>(buf, bytes, stride, false, s0, s1, d_st, d_sink); b = measure<kSc0Sc1>(buf, bytes, stride, true, s0, s1, d_st, d_sink); break;
          }
          printf("  %-7s %-6s %6.0f |              %8.2f %8.2f %8.2f |                       %8.2f %8.2f | %d\n", names[m],
                 stride == 1 ? "dense" : "sparse", bytes / 1048576.0, a.boundary_us, a.writer_us, a.pair_period_us, b.boundary_us,
                 b.pair_period_us, a.bad + b.bad);
          fflush(stdout);
        }
  // ---- does a follower that needs scratch start later?  (same stream, plain stores, 0 and 16 MB)
  printf("# follower with a 20-byte scratch frame vs without (same stream; boundary us, pair period us)\n");
  for (size_t bytes : {(size_t)0, (size_t)(16u << 20)}) {
    for (int scratch = 0; scratch < 2; ++scratch) {
      const uint32_t n = (uint32_t)(bytes / 16);
      const int wg_w = 256 * 8, wg_f = 256;
      std::vector<uint64_t> st(2 * (wg_w + wg_f));
      std::vector<double> bnd;
      for (int rep = 0; rep < 31; ++rep) {
        hipLaunchKernelGGL(k_writer<kPlain>, dim3(wg_w), dim3(kBlock), 0, s0, buf, n, 1u, d_st);
        if (scratch) hipLaunchKernelGGL(k_follower_scratch, dim3(wg_f), dim3(kBlock), 0, s0, buf, n, 1u, d_st + 2 * wg_w, d_sink);
        else hipLaunchKernelGGL(k_follower, dim3(wg_f), dim3(kBlock), 0, s0, buf, n, 1u, d_st + 2 * wg_w, d_sink);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(st.data(), d_st, st.size() * 8, hipMemcpyDeviceToHost));
        uint64_t w_e = 0, f_b = ~0ull;
        for (int b = 0; b < wg_w; ++b) w_e = std::max(w_e, st[2 * b + 1]);
        for (int b = 0; b < wg_f; ++b) f_b = std::min(f_b, st[2 * (wg_w + b)]);
        bnd.push_back(((double)f_b - (double)w_e) * 0.01);
      }
      std::sort(bnd.begin(), bnd.end());
      hipEvent_t e0, e1;
      CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      CK(hipEventRecord(e0, s0));
      for (int i = 0; i < 40; ++i) {
        hipLaunchKernelGGL(k_writer<kPlain>, dim3(wg_w), dim3(kBlock), 0, s0, buf, n, 1u, d_st);
        if (scratch) hipLaunchKernelGGL(k_follower_scratch, dim3(wg_f), dim3(kBlock), 0, s0, buf, n, 1u, d_st + 2 * wg_w, d_sink);
        else hipLaunchKernelGGL(k_follower, dim3(wg_f), dim3(kBlock), 0, s0, buf, n, 1u, d_st + 2 * wg_w, d_sink);
      }
      CK(hipEventRecord(e1, s0));
      CK(hipDeviceSynchronize());
      float ms = 0;
      CK(hipEventElapsedTime(&ms, e0, e1));
      printf("  %2.0f MB  follower %-12s boundary %6.2f  pair %6.2f\n", bytes / 1048576.0, scratch ? "with scratch" : "plain", bnd[bnd.size() / 2], ms * 1e3 / 40);
      CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    }
  }
  return 0;
}
