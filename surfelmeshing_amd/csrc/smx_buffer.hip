// smx_buffer.hip -- pitched device buffers and their copies/fills.
// Replaces the libvis helpers VIS/cuda/cuda_buffer.{h,cuh,cu}, cuda_buffer_inl.h
// (CUDABuffer<T> / CUDABuffer_<T>) behind the C-ABI of include/smx.h.
#include <stdarg.h>

#include <stdlib.h>
#include "smx_common.hpp"
#include <hip/hip_ext.h>

namespace smx {

static thread_local char g_error[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_error, sizeof(g_error), fmt, ap);
  va_end(ap);
}

}  // namespace smx

// The frame loop keeps three to five HIP streams busy at once (caller's stream, internal stream, one or two preprocessing
// queues, the stamp copies), at three priorities.  The HIP runtime maps streams onto at most GPU_MAX_HW_QUEUES hardware
// queues (default 4); when two BUSY streams land on one queue their packets serialise, and the frame falls into a slower
// mode for the rest of the process -- 3 of 14 runs at C2, 5 670 instead of 6 380 frames/s (period 165 instead of 150 us: the
// internal stream waits 20 - 35 us for the front), 4 of 128 with 8 queues (profiles/r5_ab_notes.md: the rate also rises with
// every stream the process creates, busy or not -- which is why the library creates none it can do without).  The variable is
// read when the runtime initialises, i.e. at the first HIP call of the PROCESS, so it is the application's to set (rounds 4-5
// raised it from a constructor of this library: process-global state changed behind the host's back, a setenv racing other
// threads' getenv, and no effect at all when HIP was already up -- advisor r5).  The library only REPORTS:
// smx_runtime_advice(), and one line on stderr from the first smx_recon_create of a process that runs without it
// (SMX_QUIET=1 silences it).  bench.py, the Python binding (_lib.py, before it imports torch) and the shim's
// vis::SmxSetRecommendedRuntimeDefaults() (called by the application at the top of main) set it.
static int runtime_advice(char* text, size_t capacity) {
  const char* q = getenv("GPU_MAX_HW_QUEUES");
  const int have = q ? atoi(q) : 0;
  if (have >= 8) { if (text && capacity) text[0] = 0; return 0; }
  if (text && capacity)
    snprintf(text, capacity, "GPU_MAX_HW_QUEUES is %s: the HIP runtime multiplexes this library's 4-5 busy streams onto %s hardware "
             "queues, and about one run in five is a tenth slower; export GPU_MAX_HW_QUEUES=8 before the process's first HIP call",
             q ? q : "unset", q ? q : "4");
  return 1;
}

struct smx_buffer_s {
  smx_buffer_desc desc;
  int32_t elem_bytes;
};

using namespace smx;

// Fill kernel: one thread per 4-byte word where possible.  The reference's
// CUDABufferClearKernel (VIS/cuda/cuda_buffer.cu:40-59) writes one element per
// thread of a 32x8 block; here rows are written as 16-byte words (rows are
// 256-byte aligned, so the tail of the pitch may be overwritten safely).
template <typename T>
__global__ void k_fill(T* __restrict__ base, size_t pitch, int width, int height, T value) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y;
  if (x < width && y < height)
    *reinterpret_cast<T*>(reinterpret_cast<char*>(base) + (size_t)y * pitch + (size_t)x * sizeof(T)) = value;
}

// Empty kernel whose only purpose is to show up in kernel traces (rocprofv3 --kernel-trace) so that
// tools/prof_summary.py can cut out the timed region of a benchmark run.
__global__ void k_smx_marker(int id) { (void)id; }
// (the hand-over probe's empty kernel: NOT the marker, which the profile summaries look for)
__global__ void k_smx_probe(int id) { (void)id; }

extern "C" {

int smx_debug_marker(smx_stream s, int32_t id) {
  hipLaunchKernelGGL(k_smx_marker, dim3(1), dim3(64), 0, (hipStream_t)s, (int)id);
  SMX_LAUNCH_CHECK();
  return SMX_OK;
}

const char* smx_last_error(void) { return g_error; }

int smx_runtime_advice(char* text, size_t capacity) { return runtime_advice(text, capacity); }

int smx_device_count(int* count) {
  SMX_CHECK_ARG(count != nullptr);
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) { n = 0; (void)hipGetLastError(); }
  *count = n;
  return SMX_OK;
}

int smx_set_device(int device) {
  SMX_HIP(hipSetDevice(device));
  return SMX_OK;
}

int smx_device_name(int device, char* name, size_t capacity) {
  SMX_CHECK_ARG(name != nullptr && capacity > 0);
  hipDeviceProp_t prop;
  SMX_HIP(hipGetDeviceProperties(&prop, device));
  snprintf(name, capacity, "%s (%s)", prop.name, prop.gcnArchName);
  return SMX_OK;
}

int smx_stream_create(smx_stream* out) {
  SMX_CHECK_ARG(out != nullptr);
  hipStream_t s;
  SMX_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  *out = (smx_stream)s;
  return SMX_OK;
}

int smx_stream_create_with_priority(smx_stream* out, int32_t priority_class) {
  SMX_CHECK_ARG(out != nullptr && priority_class >= -1 && priority_class <= 1);
  int least = 0, greatest = 0;
  SMX_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));  // numerically: greatest <= 0 <= least
  const int prio = priority_class > 0 ? greatest : priority_class < 0 ? least : 0;
  hipStream_t s;
  SMX_HIP(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, prio));
  *out = (smx_stream)s;
  return SMX_OK;
}

int smx_stream_create_with_cu_mask(smx_stream* out, const uint32_t* mask_words, uint32_t n_words) {
  SMX_CHECK_ARG(out != nullptr && mask_words != nullptr && n_words >= 1 && n_words <= 32);
  uint32_t any = 0;
  for (uint32_t k = 0; k < n_words; ++k) any |= mask_words[k];
  SMX_CHECK_ARG(any != 0);
  hipStream_t s;
  SMX_HIP(hipExtStreamCreateWithCUMask(&s, n_words, mask_words));
  *out = (smx_stream)s;
  return SMX_OK;
}

// cudaHostAlloc / cudaFreeHost of the caller's upload staging (APP/main.cc:917, 825-829)
// (measurement) n ping-pongs of an empty kernel between two streams, each leg handed over by an event: mean time per leg.
int smx_debug_handover_probe(smx_stream sa, smx_stream sb, int32_t n, float* us_per_handover) {
  SMX_CHECK_ARG(n > 0 && n <= 100000 && us_per_handover != nullptr);
  hipStream_t a = (hipStream_t)sa, b = (hipStream_t)sb;
  hipEvent_t t0 = nullptr, t1 = nullptr, ea = nullptr, eb = nullptr;
  // (every exit destroys what was created: the SMX_HIP early returns used to leak the four events)
  struct Cleanup { hipEvent_t *a, *b, *c, *d; ~Cleanup() { for (hipEvent_t* e : {a, b, c, d}) if (*e) (void)hipEventDestroy(*e); } } cleanup{&t0, &t1, &ea, &eb};
  SMX_HIP(hipEventCreate(&t0)); SMX_HIP(hipEventCreate(&t1));
  SMX_HIP(hipEventCreateWithFlags(&ea, hipEventDisableTiming | hipEventReleaseToDevice));
  SMX_HIP(hipEventCreateWithFlags(&eb, hipEventDisableTiming | hipEventReleaseToDevice));
  SMX_HIP(hipStreamSynchronize(a)); SMX_HIP(hipStreamSynchronize(b));
  auto leg = [&](hipStream_t from, hipStream_t to, hipEvent_t e) {
    hipLaunchKernelGGL(k_smx_probe, dim3(1), dim3(64), 0, from, 0);
    (void)hipEventRecord(e, from);
    (void)hipStreamWaitEvent(to, e, 0);
  };
  for (int i = 0; i < 8; ++i) { leg(a, b, ea); leg(b, a, eb); }   // (warm-up)
  SMX_HIP(hipEventRecord(t0, a));
  for (int i = 0; i < n; ++i) { leg(a, b, ea); leg(b, a, eb); }
  SMX_HIP(hipEventRecord(t1, a));
  SMX_HIP(hipEventSynchronize(t1));
  float ms = 0;
  SMX_HIP(hipEventElapsedTime(&ms, t0, t1));
  *us_per_handover = ms * 1e3f / (2.0f * (float)n);
  SMX_LAUNCH_CHECK();
  return SMX_OK;
}

int smx_host_alloc(void** out, size_t bytes, int32_t write_combined) {
  SMX_CHECK_ARG(out != nullptr && bytes > 0);
  void* p = nullptr;
  SMX_HIP(hipHostMalloc(&p, bytes, write_combined ? hipHostMallocWriteCombined : hipHostMallocDefault));
  *out = p;
  return SMX_OK;
}

// (device view of a page-locked host range, or null; *room = bytes of the allocation behind the pointer, SIZE_MAX if unknown)
static void* pagelocked_device_view(const void* p, size_t* room) {
  void* dsrc = nullptr;
  if (p == nullptr || hipHostGetDevicePointer(&dsrc, const_cast<void*>(p), 0) != hipSuccess || dsrc == nullptr) {
    (void)hipGetLastError();
    return nullptr;
  }
  hipDeviceptr_t base = nullptr;
  size_t size = 0;
  *room = SIZE_MAX;
  if (hipMemGetAddressRange(&base, &size, dsrc) == hipSuccess && base != nullptr && size != 0) {
    const size_t off = (size_t)(static_cast<char*>(dsrc) - static_cast<char*>(base));
    *room = off <= size ? size - off : 0;
  } else {
    (void)hipGetLastError();   // (no range known -- memory registered by the caller: its size is the caller's word)
  }
  return dsrc;
}

int smx_host_is_page_locked(const void* p, size_t bytes, int32_t* yes) {
  SMX_CHECK_ARG(yes != nullptr);
  size_t room = 0;
  *yes = (pagelocked_device_view(p, &room) != nullptr && bytes <= room) ? 1 : 0;
  return SMX_OK;
}

int smx_host_free(void* p) {
  if (p) SMX_HIP(hipHostFree(p));
  return SMX_OK;
}

int smx_stream_destroy(smx_stream s) {
  if (s) SMX_HIP(hipStreamDestroy((hipStream_t)s));
  return SMX_OK;
}

int smx_stream_synchronize(smx_stream s) {
  SMX_HIP(hipStreamSynchronize((hipStream_t)s));
  return SMX_OK;
}

int smx_event_create(smx_event* out) {
  SMX_CHECK_ARG(out != nullptr);
  hipEvent_t e;
  // (ordering of GPU streams only: device-scope release at the record, no flush towards the host)
  SMX_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming | hipEventReleaseToDevice));
  *out = (smx_event)e;
  return SMX_OK;
}

int smx_event_create_timed(smx_event* out) {
  SMX_CHECK_ARG(out != nullptr);
  hipEvent_t e;
  SMX_HIP(hipEventCreate(&e));
  *out = (smx_event)e;
  return SMX_OK;
}

int smx_event_elapsed_ms(smx_event start, smx_event stop, float* ms) {
  SMX_CHECK_ARG(start != nullptr && stop != nullptr && ms != nullptr);
  SMX_HIP(hipEventSynchronize((hipEvent_t)stop));
  SMX_HIP(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
  return SMX_OK;
}

int smx_event_destroy(smx_event e) {
  if (e) SMX_HIP(hipEventDestroy((hipEvent_t)e));
  return SMX_OK;
}

int smx_event_record(smx_event e, smx_stream s) {
  SMX_HIP(hipEventRecord((hipEvent_t)e, (hipStream_t)s));
  return SMX_OK;
}

int smx_stream_wait_event(smx_stream s, smx_event e) {
  SMX_HIP(hipStreamWaitEvent((hipStream_t)s, (hipEvent_t)e, 0));
  return SMX_OK;
}

int smx_buffer_create(int32_t height, int32_t width, int32_t elem_bytes, smx_buffer* out) {
  SMX_CHECK_ARG(out != nullptr && height > 0 && width > 0 && elem_bytes > 0);
  smx_buffer_s* b = new smx_buffer_s;
  b->elem_bytes = elem_bytes;
  b->desc.height = height;
  b->desc.width = width;
  // 256-byte row alignment: whole rows can be moved with 16-byte lane accesses.
  const size_t row_bytes = (size_t)width * elem_bytes;
  b->desc.pitch = (row_bytes + 255) / 256 * 256;
  b->desc.address = nullptr;
  hipError_t e = hipMalloc(&b->desc.address, b->desc.pitch * (size_t)height);
  if (e != hipSuccess) {
    set_error("hipMalloc(%zu bytes) failed: %s", b->desc.pitch * (size_t)height, hipGetErrorString(e));
    delete b;
    return SMX_ERR_HIP;
  }
  *out = b;
  return SMX_OK;
}

int smx_buffer_destroy(smx_buffer b) {
  if (!b) return SMX_OK;
  hipError_t e = hipFree(b->desc.address);
  delete b;
  if (e != hipSuccess) { set_error("hipFree failed: %s", hipGetErrorString(e)); return SMX_ERR_HIP; }
  return SMX_OK;
}

int smx_buffer_get_desc(smx_buffer b, smx_buffer_desc* out) {
  SMX_CHECK_ARG(b != nullptr && out != nullptr);
  *out = b->desc;
  return SMX_OK;
}

int smx_buffer_upload(smx_buffer b, smx_stream s, const void* src, size_t src_pitch) {
  SMX_CHECK_ARG(b != nullptr && src != nullptr);
  const size_t row_bytes = (size_t)b->desc.width * b->elem_bytes;
  if (src_pitch == 0) src_pitch = row_bytes;
  SMX_HIP(hipMemcpy2DAsync(b->desc.address, b->desc.pitch, src, src_pitch, row_bytes, b->desc.height,
                           hipMemcpyHostToDevice, (hipStream_t)s));
  return SMX_OK;
}

// (rows of row_bytes bytes: 16 bytes per lane where source, destination and both pitches allow, single bytes otherwise)
__global__ void __launch_bounds__(256) k_copy_rows(char* __restrict__ dst, size_t dst_pitch, const char* __restrict__ src, size_t src_pitch,
                                                   size_t row_bytes, int height, int wide) {
  if (wide) {
    const size_t per_row = row_bytes / 16, total = per_row * (size_t)height;
    for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < total; k += (size_t)gridDim.x * blockDim.x) {
      const size_t y = k / per_row, x = k - y * per_row;
      *reinterpret_cast<uint4*>(dst + y * dst_pitch + 16 * x) = *reinterpret_cast<const uint4*>(src + y * src_pitch + 16 * x);
    }
  } else {
    const size_t total = row_bytes * (size_t)height;
    for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < total; k += (size_t)gridDim.x * blockDim.x) {
      const size_t y = k / row_bytes, x = k - y * row_bytes;
      dst[y * dst_pitch + x] = src[y * src_pitch + x];
    }
  }
}

int smx_buffer_upload_by_kernel(smx_buffer b, smx_stream s, const void* src, size_t src_pitch, smx_event done) {
  SMX_CHECK_ARG(b != nullptr && src != nullptr);
  const size_t row_bytes = (size_t)b->desc.width * b->elem_bytes;
  if (src_pitch == 0) src_pitch = row_bytes;
  // The kernel reads height rows of row_bytes at src_pitch: the page-locked allocation has to cover them (the copy engine
  // would refuse a range that runs past it; a kernel would fault the GPU instead).
  size_t room = 0;
  void* dsrc = pagelocked_device_view(src, &room);
  if (dsrc == nullptr) {
    set_error("smx_buffer_upload_by_kernel: the source is not page-locked (smx_host_alloc) memory");
    return SMX_ERR_INVALID_ARGUMENT;
  }
  const size_t need = b->desc.height > 0 ? (size_t)(b->desc.height - 1) * src_pitch + row_bytes : 0;
  if (need > room) {
    set_error("smx_buffer_upload_by_kernel: the source needs %zu bytes, its page-locked allocation holds %zu behind the pointer", need, room);
    return SMX_ERR_INVALID_ARGUMENT;
  }
  const int wide = (row_bytes % 16 == 0 && src_pitch % 16 == 0 && b->desc.pitch % 16 == 0 &&
                    reinterpret_cast<uintptr_t>(dsrc) % 16 == 0 && reinterpret_cast<uintptr_t>(b->desc.address) % 16 == 0) ? 1 : 0;
  // 32 workgroups: 128 KB of 16-byte loads in flight cover the bus latency; the copy must not take the chip from the frame
  hipExtLaunchKernelGGL(k_copy_rows, dim3(32), dim3(256), 0, (hipStream_t)s, nullptr, (hipEvent_t)done, 0,
                        static_cast<char*>(b->desc.address), b->desc.pitch, static_cast<const char*>(dsrc), src_pitch, row_bytes,
                        b->desc.height, wide);
  SMX_LAUNCH_CHECK();
  return SMX_OK;
}

int smx_buffer_download(smx_buffer b, smx_stream s, void* dst, size_t dst_pitch) {
  SMX_CHECK_ARG(b != nullptr && dst != nullptr);
  const size_t row_bytes = (size_t)b->desc.width * b->elem_bytes;
  if (dst_pitch == 0) dst_pitch = row_bytes;
  SMX_HIP(hipMemcpy2DAsync(dst, dst_pitch, b->desc.address, b->desc.pitch, row_bytes, b->desc.height,
                           hipMemcpyDeviceToHost, (hipStream_t)s));
  return SMX_OK;
}

int smx_buffer_upload_part(smx_buffer b, smx_stream s, size_t start, size_t length, const void* src) {
  SMX_CHECK_ARG(b != nullptr && src != nullptr);
  SMX_CHECK_ARG(start + length <= b->desc.pitch * (size_t)b->desc.height);
  SMX_HIP(hipMemcpyAsync(reinterpret_cast<char*>(b->desc.address) + start, src, length, hipMemcpyHostToDevice,
                         (hipStream_t)s));
  return SMX_OK;
}

int smx_buffer_download_part(smx_buffer b, smx_stream s, size_t start, size_t length, void* dst) {
  SMX_CHECK_ARG(b != nullptr && dst != nullptr);
  SMX_CHECK_ARG(start + length <= b->desc.pitch * (size_t)b->desc.height);
  SMX_HIP(hipMemcpyAsync(dst, reinterpret_cast<char*>(b->desc.address) + start, length, hipMemcpyDeviceToHost,
                         (hipStream_t)s));
  return SMX_OK;
}

int smx_buffer_clear(smx_buffer b, smx_stream s, const void* pattern) {
  SMX_CHECK_ARG(b != nullptr && pattern != nullptr);
  hipStream_t st = (hipStream_t)s;
  const int w = b->desc.width, h = b->desc.height;
  dim3 block(256, 1, 1), grid(div_up(w, 256), h, 1);
  switch (b->elem_bytes) {
    case 1: { uint8_t v; memcpy(&v, pattern, 1);
      hipLaunchKernelGGL(k_fill<uint8_t>, grid, block, 0, st, (uint8_t*)b->desc.address, b->desc.pitch, w, h, v); break; }
    case 2: { uint16_t v; memcpy(&v, pattern, 2);
      hipLaunchKernelGGL(k_fill<uint16_t>, grid, block, 0, st, (uint16_t*)b->desc.address, b->desc.pitch, w, h, v); break; }
    case 4: { uint32_t v; memcpy(&v, pattern, 4);
      hipLaunchKernelGGL(k_fill<uint32_t>, grid, block, 0, st, (uint32_t*)b->desc.address, b->desc.pitch, w, h, v); break; }
    case 8: { uint64_t v; memcpy(&v, pattern, 8);
      hipLaunchKernelGGL(k_fill<uint64_t>, grid, block, 0, st, (uint64_t*)b->desc.address, b->desc.pitch, w, h, v); break; }
    default: {
      // odd element sizes (e.g. uchar3): expand the pattern over one row and replicate rows.
      const size_t row_bytes = (size_t)w * b->elem_bytes;
      char* host_row = new char[row_bytes];
      for (size_t i = 0; i < row_bytes; ++i) host_row[i] = ((const char*)pattern)[i % b->elem_bytes];
      hipError_t e = hipSuccess;
      for (int y = 0; y < h && e == hipSuccess; ++y)
        e = hipMemcpyAsync((char*)b->desc.address + (size_t)y * b->desc.pitch, host_row, row_bytes,
                           hipMemcpyHostToDevice, st);
      if (e == hipSuccess) e = hipStreamSynchronize(st);
      delete[] host_row;
      if (e != hipSuccess) { set_error("clear failed: %s", hipGetErrorString(e)); return SMX_ERR_HIP; }
      return SMX_OK;
    }
  }
  SMX_LAUNCH_CHECK();
  return SMX_OK;
}

int smx_buffer_set_to(smx_buffer dst, smx_buffer src, smx_stream s) {
  SMX_CHECK_ARG(dst != nullptr && src != nullptr);
  SMX_CHECK_ARG(dst->desc.width == src->desc.width && dst->desc.height == src->desc.height &&
                dst->elem_bytes == src->elem_bytes);
  SMX_HIP(hipMemcpy2DAsync(dst->desc.address, dst->desc.pitch, src->desc.address, src->desc.pitch,
                           (size_t)src->desc.width * src->elem_bytes, src->desc.height, hipMemcpyDeviceToDevice,
                           (hipStream_t)s));
  return SMX_OK;
}

}  // extern "C"
