// handoff.hip -- what does it cost to hand a dependency from one HIP stream to another on this chip, and is there a
// cheaper way than an event?  (two hand-offs of ~11 us sit on the frame's cycle A.)  Not part of the product:
//   hipcc --offload-arch=gfx950 -O3 tools/handoff.hip -o build/handoff ; run on the GPU box.
// A ping-pong of short kernels between two streams, 200 hand-offs per measurement:
//   (a) hipEventRecord + hipStreamWaitEvent, default event;
//   (b) the same with hipEventDisableTiming | hipEventReleaseToDevice (what libsmx uses);
//   (c) hipStreamWriteValue32 + hipStreamWaitValue32 on signal memory;
//   (d) device-side flag: a one-thread "set" kernel behind the producer, a one-wavefront "gate" kernel (polls the flag)
//       in front of the consumer;
//   (e) the producer kernel sets the flag itself at its end (all its workgroups count down on a counter first).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <chrono>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

// body: `iters` dependent FMAs per lane (iters = 0: nothing)
__global__ void __launch_bounds__(256) k_body(float* out, int iters) {
  float a = (float)threadIdx.x;
  for (int i = 0; i < iters; ++i) a = a * 0.999f + 0.001f;
  if (a == 1234.5678f) out[0] = a;
}
__global__ void k_set(uint32_t* flag, uint32_t v) { __hip_atomic_store(flag, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); }
__global__ void __launch_bounds__(64) k_gate(const uint32_t* flag, uint32_t v) {
  if (threadIdx.x == 0)
    while ((int32_t)(__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) - v) < 0) __builtin_amdgcn_s_sleep(8);
}
// body whose last workgroup sets the flag (every workgroup counts down on `pending` when it is done)
__global__ void __launch_bounds__(256) k_body_signal(float* out, int iters, uint32_t* pending, uint32_t* flag, uint32_t v) {
  float a = (float)threadIdx.x;
  for (int i = 0; i < iters; ++i) a = a * 0.999f + 0.001f;
  if (a == 1234.5678f) out[0] = a;
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    if (atomicAdd(pending, 1u) + 1u == gridDim.x) { *pending = 0; __hip_atomic_store(flag, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); }
  }
}

template <class F>
static double time_us(F f, hipStream_t s0, hipStream_t s1) {
  f();
  CK(hipStreamSynchronize(s0)); CK(hipStreamSynchronize(s1));
  const auto t0 = std::chrono::steady_clock::now();
  f();
  CK(hipStreamSynchronize(s0)); CK(hipStreamSynchronize(s1));
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
}

int main() {
  hipStream_t s0, s1;
  int lo = 0, hi = 0;
  CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
  CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
  CK(hipStreamCreateWithPriority(&s1, hipStreamNonBlocking, hi));
  float* out; CK(hipMalloc(&out, 64));
  uint32_t* flags; CK(hipMalloc(&flags, 4096)); CK(hipMemset(flags, 0, 4096));
  uint32_t* sig = nullptr;
  const bool have_sig = hipExtMallocWithFlags(reinterpret_cast<void**>(&sig), 64, hipMallocSignalMemory) == hipSuccess;
  if (have_sig) CK(hipMemset(sig, 0, 64));
  const int n = 200;   // hand-offs per measurement (n / 2 round trips)
  hipEvent_t ev_def[2], ev_dev[2];
  for (int k = 0; k < 2; ++k) { CK(hipEventCreate(&ev_def[k])); CK(hipEventCreateWithFlags(&ev_dev[k], hipEventDisableTiming | hipEventReleaseToDevice)); }
  uint32_t epoch = 0;
  for (int grid : {1, 2048}) {
    for (int iters : {0, 4000}) {
      // the kernel alone, back to back in one stream
      const double same = time_us([&] { for (int i = 0; i < n; ++i) hipLaunchKernelGGL(k_body, dim3(grid), dim3(256), 0, s0, out, iters); }, s0, s1) / n;
      auto pp_event = [&](hipEvent_t* ev) {
        return time_us([&] {
          for (int i = 0; i < n / 2; ++i) {
            hipLaunchKernelGGL(k_body, dim3(grid), dim3(256), 0, s0, out, iters);
            CK(hipEventRecord(ev[0], s0)); CK(hipStreamWaitEvent(s1, ev[0], 0));
            hipLaunchKernelGGL(k_body, dim3(grid), dim3(256), 0, s1, out, iters);
            CK(hipEventRecord(ev[1], s1)); CK(hipStreamWaitEvent(s0, ev[1], 0));
          }
        }, s0, s1) / n;
      };
      const double a = pp_event(ev_def), b = pp_event(ev_dev);
      double c = -1;
      if (have_sig) {
        c = time_us([&] {
          for (int i = 0; i < n / 2; ++i) {
            hipLaunchKernelGGL(k_body, dim3(grid), dim3(256), 0, s0, out, iters);
            ++epoch; CK(hipStreamWriteValue32(s0, sig, epoch, 0)); CK(hipStreamWaitValue32(s1, sig, epoch, hipStreamWaitValueGte, 0xFFFFFFFFu));
            hipLaunchKernelGGL(k_body, dim3(grid), dim3(256), 0, s1, out, iters);
            ++epoch; CK(hipStreamWriteValue32(s1, sig + 8, epoch, 0)); CK(hipStreamWaitValue32(s0, sig + 8, epoch, hipStreamWaitValueGte, 0xFFFFFFFFu));
          }
        }, s0, s1) / n;
      }
      const double d = time_us([&] {
        for (int i = 0; i < n / 2; ++i) {
          hipLaunchKernelGGL(k_body, dim3(grid), dim3(256), 0, s0, out, iters);
          ++epoch; hipLaunchKernelGGL(k_set, dim3(1), dim3(1), 0, s0, flags, epoch); hipLaunchKernelGGL(k_gate, dim3(1), dim3(64), 0, s1, flags, epoch);
          hipLaunchKernelGGL(k_body, dim3(grid), dim3(256), 0, s1, out, iters);
          ++epoch; hipLaunchKernelGGL(k_set, dim3(1), dim3(1), 0, s1, flags + 64, epoch); hipLaunchKernelGGL(k_gate, dim3(1), dim3(64), 0, s0, flags + 64, epoch);
        }
      }, s0, s1) / n;
      const double e = time_us([&] {
        for (int i = 0; i < n / 2; ++i) {
          ++epoch; hipLaunchKernelGGL(k_body_signal, dim3(grid), dim3(256), 0, s0, out, iters, flags + 128, flags, epoch);
          hipLaunchKernelGGL(k_gate, dim3(1), dim3(64), 0, s1, flags, epoch);
          ++epoch; hipLaunchKernelGGL(k_body_signal, dim3(grid), dim3(256), 0, s1, out, iters, flags + 192, flags + 64, epoch);
          hipLaunchKernelGGL(k_gate, dim3(1), dim3(64), 0, s0, flags + 64, epoch);
        }
      }, s0, s1) / n;
      printf("grid %4d iters %4d | same stream %6.2f us/kernel | per kernel incl. hand-off: event default %6.2f  event device-release %6.2f  "
             "stream write/wait value %6.2f  set + gate kernels %6.2f  kernel signals + gate %6.2f\n", grid, iters, same, a, b, c, d, e);
    }
  }
  return 0;
}
