/*
 * smx.h -- C-ABI of the MI355X-native surfel-integration library (libsmx.so).
 *
 * This is the drop-in boundary for the hot path of puzzlepaint/surfelmeshing:
 * every entry point replaces one piece of the reference's CUDA-side interface
 * (cited per declaration; paths relative to the reference checkout,
 * APP = applications/surfel_meshing/src/surfel_meshing, VIS = libvis/src/libvis).
 * Plain C types only: opaque handles, POD descriptors, pointers and sizes.
 * The C++ shim include/smx_shim.hpp re-creates the reference's class names
 * (CUDABuffer<T>, CUDASurfelReconstruction, CUDASurfelsCPU) on top of it; the
 * Python mirror is surfelmeshing_amd/api.py.  See INTEGRATION.md.
 *
 * Conventions:
 *  - every function returns 0 on success or a negative smx_status; the last
 *    error text is available from smx_last_error().  The reference aborts via
 *    LOG(FATAL) (VIS/cuda/cuda_util.h:35-49); the shims convert non-zero into
 *    an abort / exception to keep that behaviour.
 *  - all work is enqueued on the HIP stream passed in (a hipStream_t cast to
 *    void*; NULL = the default stream).  Calls return asynchronously; unlike
 *    the reference, Integrate does not block the host (the surfel count lives
 *    in device memory), so counts are read with smx_recon_counts(), which
 *    synchronises the stream.
 *  - single caller thread per object; no process-global state -- the library neither keeps any nor touches the process
 *    environment (smx_runtime_advice reports what the application should set) -- (the reference's
 *    function-local static buffer at APP/cuda_surfel_reconstruction_kernels.cc:479
 *    is per-object here), so one object per GPU / per stream works.
 *  - camera cx, cy are in the pixel-CORNER convention, as
 *    PinholeCamera4f::parameters()[2..3] in the reference.
 */
#ifndef SMX_H_
#define SMX_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  SMX_OK = 0,
  SMX_ERR_INVALID_ARGUMENT = -1,
  SMX_ERR_HIP = -2,          /* a HIP runtime call failed */
  SMX_ERR_NO_DEVICE = -3,    /* no usable gfx950 device */
  SMX_ERR_UNSUPPORTED = -4
} smx_status;

typedef void* smx_stream;                 /* hipStream_t */
typedef struct smx_buffer_s* smx_buffer;  /* owns pitched device memory */
typedef struct smx_recon_s* smx_recon;    /* CUDASurfelReconstruction */
typedef struct smx_nn_s* smx_nn;          /* radius-neighbor search index */

/* POD passed to kernels by value, identical in layout to the reference's
 * CUDABuffer_<T> {T* address_; int height_; int width_; size_t pitch_;}
 * (VIS/cuda/cuda_buffer.cuh:44-119). */
typedef struct {
  void* address;
  int32_t height;
  int32_t width;
  size_t pitch;   /* bytes */
} smx_buffer_desc;

const char* smx_last_error(void);
/* Runtime settings the frame loop wants and the library does not make itself (process-global: the application's to set):
 * returns the number of recommendations (0 = none) and their text.  Today: GPU_MAX_HW_QUEUES >= 8, read by the HIP runtime at
 * the process's first HIP call (INTEGRATION.md "Streams, queues, priorities").  The first smx_recon_create of a process prints
 * the text once on stderr unless SMX_QUIET=1. */
int smx_runtime_advice(char* text, size_t capacity);
/* Number of visible HIP devices / select one for the calling thread. */
int smx_device_count(int* count);
int smx_set_device(int device);
int smx_device_name(int device, char* name, size_t capacity);
int smx_stream_create(smx_stream* out);
/* priority_class: -1 = lowest, 0 = default, +1 = highest priority the device offers (cudaStreamCreateWithPriority) */
int smx_stream_create_with_priority(smx_stream* out, int32_t priority_class);
/* A stream whose kernels only run on a subset of the compute units (hipExtStreamCreateWithCUMask; no CUDA counterpart):
 * bit k of mask_words[k / 32] set = compute unit k may be used.  On this part consecutive bits fall on consecutive XCDs
 * (bit k -> XCD k % 8), so the low n bits are n / 8 compute units of every XCD.  Default priority.  Measured use:
 * a partition between the preprocessing queue and the surfel queues (profiles/r6_ab_notes.md section 10). */
int smx_stream_create_with_cu_mask(smx_stream* out, const uint32_t* mask_words, uint32_t n_words);
/* Page-locked host memory for upload staging (cudaHostAlloc(..., cudaHostAllocWriteCombined) / cudaFreeHost,
 * APP/main.cc:825-829, 917): copies from it are asynchronous to the host.  write_combined memory is fast to
 * upload from and slow for the CPU to read. */
int smx_host_alloc(void** out, size_t bytes, int32_t write_combined);
int smx_host_free(void* p);
/* *yes = 1 if [p, p + bytes) lies inside page-locked memory the device can read (smx_host_alloc / hipHostMalloc /
 * hipHostRegister), else 0.  What smx_buffer_upload_by_kernel requires of its source; a caller with a choice of routes asks
 * before it enqueues anything (smx_driver does). */
int smx_host_is_page_locked(const void* p, size_t bytes, int32_t* yes);
int smx_stream_destroy(smx_stream s);
int smx_stream_synchronize(smx_stream s);
/* Events for cross-stream ordering (hipEvent_t, timing disabled): the frame driver overlaps depth
 * preprocessing of the next frame with the integration of the current one, as APP/main.cc overlaps uploads
 * (main.cc:902, 995).  An smx_event orders the streams of ONE device: it is created with a device-scope release
 * (hipEventReleaseToDevice), so work behind it is visible to kernels and copies of that GPU; it is not meant to be
 * waited for by the host or by another GPU (use smx_stream_synchronize for the host). */
typedef void* smx_event;
int smx_event_create(smx_event* out);
int smx_event_destroy(smx_event e);
int smx_event_record(smx_event e, smx_stream s);
int smx_stream_wait_event(smx_stream s, smx_event e);
/* Events that carry a time stamp (measurement only: the driver's per-stage profile of the preprocessing stream);
 * smx_event_elapsed_ms blocks until `stop` has completed. */
int smx_event_create_timed(smx_event* out);
int smx_event_elapsed_ms(smx_event start, smx_event stop, float* ms);
/* Launches an empty kernel (k_smx_marker) that delimits regions in kernel traces. */
int smx_debug_marker(smx_stream s, int32_t id);
/* (measurement) n ping-pongs of an empty kernel between two streams, each leg handed over by an event record + a stream wait:
 * mean time per leg in microseconds.  Synchronises both streams. */
int smx_debug_handover_probe(smx_stream a, smx_stream b, int32_t n, float* us_per_handover);

/* ---- CUDABuffer<T>  (VIS/cuda/cuda_buffer.h:45-129, cuda_buffer_inl.h:36-172) ---- */
/* CUDABuffer(int height, int width): cudaMallocPitch */
int smx_buffer_create(int32_t height, int32_t width, int32_t elem_bytes, smx_buffer* out);
int smx_buffer_destroy(smx_buffer b);
/* ToCUDA() */
int smx_buffer_get_desc(smx_buffer b, smx_buffer_desc* out);
/* UploadAsync / UploadPitchedAsync (src_pitch = 0: dense rows of width*elem_bytes) */
int smx_buffer_upload(smx_buffer b, smx_stream s, const void* src, size_t src_pitch);
/* The same copy done by a KERNEL that reads the page-locked source over the bus (src must come from smx_host_alloc /
 * hipHostMalloc; otherwise SMX_ERR_INVALID_ARGUMENT) -- an addition for callers that stage uploads on a stream of their own:
 * a kernel's stores reach later kernels on other streams through the ordinary event ordering, whereas a copy-engine write
 * followed by a cross-stream wait that the runtime finds already satisfied is dropped together with the cache invalidation
 * the consumer needs (measured: stale reads now and then).  Few workgroups (the copy is bound by the bus, not the chip).
 * done: optional completion event of the launch (no packet of its own on the stream). */
int smx_buffer_upload_by_kernel(smx_buffer b, smx_stream s, const void* src_pagelocked, size_t src_pitch, smx_event done);
/* DownloadAsync / DownloadPitchedAsync */
int smx_buffer_download(smx_buffer b, smx_stream s, void* dst, size_t dst_pitch);
/* UploadPartAsync / DownloadPartAsync: byte range [start, start+length) of the allocation */
int smx_buffer_upload_part(smx_buffer b, smx_stream s, size_t start, size_t length, const void* src);
int smx_buffer_download_part(smx_buffer b, smx_stream s, size_t start, size_t length, void* dst);
/* Clear(T value, stream): pattern points at one element (elem_bytes bytes) */
int smx_buffer_clear(smx_buffer b, smx_stream s, const void* pattern);
/* SetTo(const CUDABuffer<T>& other, stream) */
int smx_buffer_set_to(smx_buffer dst, smx_buffer src, smx_stream s);

/* ---- depth preprocessing free functions (APP/cuda_depth_processing.cuh:43-122) ---- */
/* BilateralFilteringAndDepthCutoffCUDA, APP/cuda_depth_processing.cu:120-158 */
int smx_bilateral_filtering_and_depth_cutoff(
    smx_stream s, float sigma_xy, float sigma_value_factor, uint16_t value_to_ignore,
    float radius_factor, uint16_t max_depth, float depth_valid_region_radius,
    const smx_buffer_desc* input_depth /*u16*/, const smx_buffer_desc* output_depth /*u16*/);
/* OutlierDepthMapFusionCUDA<count,u16>, both overloads (cu:229-285 and :399-455):
 * other_count = count-1 in {2,4,6,8}; required_count < 0 selects the
 * all-must-agree overload.  others_TR_reference: other_count row-major 3x4. */
int smx_outlier_depth_map_fusion(
    smx_stream s, int32_t other_count, int32_t required_count, float tolerance,
    const smx_buffer_desc* input_depth, float fx, float fy, float cx, float cy,
    const smx_buffer_desc* other_depths /*[other_count]*/, const float* others_TR_reference,
    const smx_buffer_desc* output_depth);
/* BilateralFilteringAndDepthCutoffCUDA (value_to_ignore = 0) followed by OutlierDepthMapFusionCUDA, as the reference's
 * caller chains them (APP/main.cc:1015-1115), in ONE launch where the shape allows it (eight other frames, filter radius 1..8):
 * the outlier test of a pixel only needs that pixel's own filtered depth.  Same output image as the two calls; scratch_depth
 * (the intermediate image of the two-call form) is only written when the shape needs the two launches. */
int smx_bilateral_outlier_fusion(
    smx_stream s, float sigma_xy, float sigma_value_factor, float radius_factor, uint16_t max_depth,
    float depth_valid_region_radius, const smx_buffer_desc* input_depth,
    int32_t other_count, int32_t required_count, float tolerance, float fx, float fy, float cx, float cy,
    const smx_buffer_desc* other_depths /*[other_count]*/, const float* others_TR_reference,
    const smx_buffer_desc* scratch_depth, const smx_buffer_desc* output_depth);
/* ErodeDepthMapCUDA (radius 1..3), cu:540-579; CopyWithoutBorderCUDA, cu:609-633 */
int smx_erode_depth_map(smx_stream s, int32_t radius, const smx_buffer_desc* input_depth,
                        const smx_buffer_desc* output_depth);
int smx_copy_without_border(smx_stream s, const smx_buffer_desc* input_depth,
                            const smx_buffer_desc* output_depth);
/* MedianFilterAndDensifyDepthMap, APP/main.cc:206-252: in the reference a CPU loop before the upload (its TODO at
 * main.cc:928 asks for the GPU), one call per `median_filter_and_densify_iterations`.  Input and output must differ. */
int smx_median_filter_and_densify_depth_map(smx_stream s, const smx_buffer_desc* input_depth,
                                            const smx_buffer_desc* output_depth);
/* Image<u16>::DownscaleUsingMedianWhileExcluding, VIS/image.h:1003-1053 -- the depth half of --pyramid_level
 * (APP/main.cc:941-962), a CPU loop in the reference.  The output size selects the source blocks. */
int smx_downscale_using_median_while_excluding(smx_stream s, uint16_t value_to_ignore, const smx_buffer_desc* input,
                                               const smx_buffer_desc* output);
/* The colour half of --pyramid_level (APP/main.cc:973-981): ImagePyramid(frame, pyramid_level) = pyramid_level times
 * Image<Vec3u8>::DownscaleToHalfSize (VIS/image_cache.h:203-243, VIS/image.h:929-948: per channel
 * a/4 + b/4 + c/4 + d/4, each term truncated), a CPU loop in the reference.  3-byte pixels; 1 <= pyramid_level <= 4;
 * the input size must be divisible by 2^pyramid_level and the output buffer must have the resulting size. */
int smx_color_image_pyramid(smx_stream s, int32_t pyramid_level, const smx_buffer_desc* input,
                            const smx_buffer_desc* output);
/* ComputeNormalsAndDropBadPixelsCUDA, cu:720-762 */
int smx_compute_normals_and_drop_bad_pixels(
    smx_stream s, float observation_angle_threshold_deg, float depth_scaling,
    float fx, float fy, float cx, float cy,
    const smx_buffer_desc* in_depth, const smx_buffer_desc* out_depth,
    const smx_buffer_desc* out_normals /*float2*/);
/* ComputePointRadiiAndRemoveIsolatedPixelsCUDA, cu:839-883 */
int smx_compute_point_radii_and_remove_isolated_pixels(
    smx_stream s, float point_radius_extension_factor, float point_radius_clamp_factor,
    float depth_scaling, float fx, float fy, float cx, float cy,
    const smx_buffer_desc* depth_buffer, const smx_buffer_desc* radius_buffer /*float*/,
    const smx_buffer_desc* out_depth);

/* The three calls above that always follow each other in the caller (APP/main.cc:1128-1191: ErodeDepthMapCUDA or, for
 * erosion_radius 0, CopyWithoutBorderCUDA; ComputeNormalsAndDropBadPixelsCUDA; ComputePointRadiiAndRemoveIsolatedPixelsCUDA)
 * as ONE launch: the two intermediate depth images stay in LDS tiles.  out_depth, out_normals and radius_buffer receive
 * exactly what the three separate calls leave in their last outputs.  in_depth and out_depth must differ. */
int smx_erode_normals_radii(smx_stream s, int32_t erosion_radius, float observation_angle_threshold_deg,
                            float point_radius_extension_factor, float point_radius_clamp_factor, float depth_scaling,
                            float fx, float fy, float cx, float cy, const smx_buffer_desc* in_depth,
                            const smx_buffer_desc* out_depth, const smx_buffer_desc* out_normals /*float2*/,
                            const smx_buffer_desc* radius_buffer /*float*/);
/* The same launch with `done` (may be null) as its own completion event: equivalent to smx_event_record(done, s) behind the
 * call, without a packet of its own on the stream (the frame loop's "preprocessed" mark: smx_driver.cpp). */
int smx_erode_normals_radii_signal(smx_stream s, int32_t erosion_radius, float observation_angle_threshold_deg,
                                   float point_radius_extension_factor, float point_radius_clamp_factor, float depth_scaling,
                                   float fx, float fy, float cx, float cy, const smx_buffer_desc* in_depth,
                                   const smx_buffer_desc* out_depth, const smx_buffer_desc* out_normals /*float2*/,
                                   const smx_buffer_desc* radius_buffer /*float*/, smx_event done);

/* ---- CUDASurfelReconstruction (APP/cuda_surfel_reconstruction.h:44-176) ---- */
/* trailing arguments of Integrate(), .h:59-77; defaults APP/main.cc:323-368 */
typedef struct {
  float sensor_noise_factor;
  float max_surfel_confidence;
  float regularizer_weight;
  int32_t regularization_frame_window_size;
  int32_t do_blending;
  int32_t measurement_blending_radius;
  int32_t regularization_iterations_per_integration_iteration;
  float radius_factor_for_regularization_neighbors;
  float normal_compatibility_threshold_deg;
  int32_t surfel_integration_active_window_size;
} smx_integrate_params;

/* CUDASurfelBuffersCPU, APP/cuda_surfels_cpu.h:40-74 */
typedef struct {
  uint32_t frame_index;
  size_t surfel_count;
  float* surfel_x_buffer;
  float* surfel_y_buffer;
  float* surfel_z_buffer;
  float* surfel_radius_squared_buffer;
  float* surfel_normal_x_buffer;
  float* surfel_normal_y_buffer;
  float* surfel_normal_z_buffer;
  uint32_t* surfel_last_update_stamp_buffer;
} smx_surfel_buffers_cpu;

/* ctor, .h:47-53 / .cc:44-91 (the three GL resources and the render window are dropped).
 * device_id: the HIP device the object lives on, -1 = the calling thread's current device.  The object remembers
 * it: every smx_recon_* call makes it current for its duration (and restores the caller's device), so one process
 * can hold one object per GPU and call them from one thread each -- or from one thread -- without smx_set_device
 * in between.  Streams and buffers passed to a call must belong to the object's device. */
int smx_recon_create(uint32_t max_surfel_count, int32_t width, int32_t height,
                     float fx, float fy, float cx, float cy, int32_t device_id, smx_recon* out);
int smx_recon_destroy(smx_recon r);
/* Integrate, .h:59-77 / .cc:112-320.  depth is MUTATED by blending as in the
 * reference; global_T_local is row-major 3x4 (SE3f::matrix3x4()).
 * frame_index normally counts up from call to call (the reference's caller does, APP/main.cc:1015; stamps are compared
 * with it in windows, kernels.cu:77-87, 2132): pass A keeps, per 1024-slot segment, the newest stamp it has seen and
 * skips segments whose stamps have left the regulariser window, which presumes that time moves forward.  A call with a
 * smaller frame_index than the previous one is accepted like in the reference: it drops that cache (the call reads
 * every segment again) and costs one stream join.
 * measurement_blending_radius is only read (and range-checked, 2..255) when do_blending != 0. */
int smx_recon_integrate(smx_recon r, smx_stream s, uint32_t frame_index, float depth_scaling,
                        const smx_buffer_desc* depth /*u16*/, const smx_buffer_desc* normals /*float2*/,
                        const smx_buffer_desc* radius /*float*/, const smx_buffer_desc* color /*uchar3*/,
                        const float global_T_local[12], const smx_integrate_params* params);
/* Regularize, .h:82-87 / .cc:322-337 */
int smx_recon_regularize(smx_recon r, smx_stream s, uint32_t frame_index, float regularizer_weight,
                         float radius_factor_for_regularization_neighbors,
                         int32_t regularization_frame_window_size);
/* TransferAllToCPU, .h:91-94 / .cc:339-359.  Fills frame_index and surfel_count,
 * enqueues the 8 row downloads; the caller synchronises the stream
 * (APP/main.cc:1266-1267).  Reads the device-side count first (one small
 * blocking copy). */
int smx_recon_transfer_all_to_cpu(smx_recon r, smx_stream s, uint32_t frame_index,
                                  smx_surfel_buffers_cpu* buffers);
/* Changed-surfel delta for the mesher (not in the reference: SURVEY.md 8f-1, the step right after this path).
 * TransferAllToCPU moves 32 B x N over PCIe every time and leaves it to the CPU to find out what changed
 * (SurfelMeshing::IntegrateCUDABuffers, APP/surfel_meshing.cc:190-300 walks all N).  With tracking on, every kernel
 * that changes one of the eight transferred attributes of a slot marks the slot; the transfer compacts the marked
 * slots on the GPU, downloads (slot index ascending, the eight attributes) for those only and clears the marks.
 * Contract: applying every delta since a full transfer to that transfer's arrays reproduces the current full arrays
 * bit for bit (a delta may contain slots whose values did not change).  Enabling marks every existing slot.
 * smx_recon_transfer_changed_to_cpu is synchronous (it returns with the arrays filled); if count > capacity it fails,
 * reports the needed count and keeps the marks. */
typedef struct {
  uint32_t capacity;      /* in: entries each array can hold */
  uint32_t count;         /* out: entries written */
  uint32_t frame_index;   /* out */
  uint32_t surfel_count;  /* out: slots in use (as smx_surfel_buffers_cpu.surfel_count) */
  uint32_t* surfel_index;
  float* x;
  float* y;
  float* z;
  float* radius_squared;
  float* normal_x;
  float* normal_y;
  float* normal_z;
  uint32_t* last_update_stamp;
} smx_surfel_delta_cpu;
int smx_recon_set_delta_tracking(smx_recon r, smx_stream s, int32_t enabled);
int smx_recon_transfer_changed_to_cpu(smx_recon r, smx_stream s, uint32_t frame_index, smx_surfel_delta_cpu* delta);
/* ExportVertices, .h:109-112 / .cc:405-410: position 1 x 3N float, colour 1 x 3N u8 */
int smx_recon_export_vertices(smx_recon r, smx_stream s, const smx_buffer_desc* position_buffer,
                              const smx_buffer_desc* color_buffer);
/* GetTimings, .h:115-122 / .cc:412-429: data association, merging, blending, integration, neighbor update, new surfel
 * creation, regularization (ms) of the LAST smx_recon_integrate call; like the reference it waits until that call is
 * through (cudaEventSynchronize(regularization_end_event_), cc:420).  On from the first call; cost 1.4 - 2.2 % of the frame rate at
 * 640 x 480 / 5 M surfels, nothing measurable at 1280 x 960 (bench.py: stage_timing_cost; smx_recon_set_timing_enabled(r, 0)
 * switches the stamps off): the stages are not bracketed by event records (each a packet between two kernels of a stream that is never idle: fourteen
 * of them cost a third of the frame rate here) but stamped by the kernels themselves -- device wall clock, first
 * workgroup in of the launch that begins a stage / of the launch that follows it on the same stream, last workgroups out
 * where nothing follows -- into a per-call record.  Stages the
 * design fuses into another stage's launch report 0 -- out_ms[1] (surfel_merging: decided in the association kernel,
 * applied by the integration kernel) and out_ms[5] (new_surfel_creation: the first workgroups of the neighbour-update
 * launch): their time is INSIDE out_ms[0] / out_ms[3] and out_ms[4], so the seven values still add up to the call; a caller
 * that accumulates the reference's seven columns (APP/main.cc:1511-1530) gets two empty ones. */
int smx_recon_get_timings(smx_recon r, float out_ms[7]);
/* The same for a frame loop that must not wait: the stage times of the NEWEST call whose record has been handed over --
 * every call copies the record of the call before the previous one (complete by stream order at that point) into
 * page-locked host memory, so the read lags the queue by two calls and touches neither the device nor any stream.
 * *call_number: that call's 1-based number, 0 (and zeros) if there is none yet. */
int smx_recon_get_timings_nowait(smx_recon r, float out_ms[7], uint64_t* call_number);
/* How the front of a pipelined smx_recon_integrate call (pass A .. blend, on the caller's stream) hands over to the
 * internal stream: 1 = the blend's workgroups count themselves in a device word and a one-wavefront gate kernel in front of the
 * integration polls it (default: no event packet on the internal stream, and no release of the XCDs' L2s behind the blend --
 * its output leaves write-through -- + 3.6 % at 640 x 480, profiles/r6_ab_notes.md section 13), 0 = an event (rounds 3 - 6).
 * Results identical.  The two measurement modes of smx_recon_set_timing_enabled that bracket kernels with event records of
 * their own (bits 0 and 1) keep the event whatever the mode. */
int smx_recon_set_handover_mode(smx_recon r, int32_t mode);
/* The mode in use.  smx_recon_create starts an object in mode 0 when the process runs under a profiler that collects hardware
 * counters (ROCPROF_COUNTER_COLLECTION set, i.e. rocprofv3 --pmc): such a tool serialises the kernel dispatches of ALL queues, the
 * gate can then reach the chip in front of the launch it waits for, and nothing else is let on.  The gate's poll is bounded
 * (0.25 s); one that gives up invalidates the map, the next smx_recon_counts / smx_recon_get_stats returns SMX_ERR_UNSUPPORTED, and
 * the object goes back to mode 0 with its next smx_recon_integrate call (the gate leaves its mark in page-locked memory). */
int smx_recon_get_handover_mode(smx_recon r, int32_t* mode);
/* Experiment: the object's internal stream re-created on a subset of the compute units (mask as for
 * smx_stream_create_with_cu_mask; n_words = 0: all of them again, at the highest priority).  Waits for the object's work. */
int smx_recon_set_internal_cu_mask(smx_recon r, const uint32_t* mask_words, uint32_t n_words);
/* Measurement: the object's internal stream (for smx_debug_handover_probe; never enqueue work on it). */
int smx_recon_debug_internal_stream(smx_recon r, smx_stream* out);
/* Measurement: the raw stage-stamp records of the last 8 smx_recon_integrate calls (8 x 16 words of device wall clock,
 * rate in *wall_clock_khz; word 0 = the call's number, then: cull begin, tiles end*, blend begin, blend end*, integrate
 * begin, integrate end*, update begin, update end*, pass B begin, step end*, pass A begin, tiles begin, edge kernel begin,
 * step begin; * = maximum over the last workgroups dispatched).  bench.py turns them into the in-frame timeline of the
 * pipelined run -- no profiler, no event packets.  Call after synchronising. */
int smx_recon_debug_stamp_ring(smx_recon r, uint64_t* out, int32_t capacity_words, int32_t* wall_clock_khz);
/* enabled: bit 2 = stage stamps (default ON: what smx_recon_get_timings reads), bit 0 = the reference's own 14 stage
 * events instead (measurement: smx_recon_get_timings then reads those), bit 1 = events around every kernel */
int smx_recon_set_timing_enabled(smx_recon r, int32_t enabled);
/* Per-kernel device times of the last Integrate call (needs timing bit 1); slot names from
 * smx_recon_kernel_slot_name(0 .. smx_recon_kernel_slot_count()-1). */
int smx_recon_kernel_slot_count(void);
const char* smx_recon_kernel_slot_name(int32_t slot);
int smx_recon_get_kernel_timings(smx_recon r, float* out_ms, int32_t capacity);
/* HIP-event timing of ONE kernel slot over many Integrate calls (2 event records per frame on the
 * launch stream): begin, run up to max_frames frames, end -> average launch duration. */
int smx_recon_profile_begin(smx_recon r, int32_t slot, int32_t max_frames);
int smx_recon_profile_end(smx_recon r, float* avg_ms, int32_t* frames);
/* surfel_count() = slots - merged, surfels_size() = slots, .h:125-128.  Synchronises s. */
int smx_recon_counts(smx_recon r, smx_stream s, uint32_t* surfel_count, uint32_t* surfels_size);

/* Value distributions of the last Integrate call: synchronises s. */
typedef struct {
  uint32_t surfels_size, merge_count;
  uint32_t n_visible;      /* slots projecting into the image with z > 0 */
  uint32_t n_new, n_merged, n_recent, n_edges;
  uint32_t n_integrated, n_replaced, n_conflict_hits;
  uint32_t capacity_clamped;  /* 1 if new-surfel creation hit max_surfel_count */
  uint32_t n_window_edges;    /* neighbour links whose target lies inside the regulariser window */
  uint32_t n_contributors;    /* slots with at least one such link */
  uint32_t n_segments_skipped;  /* 1024-slot segments pass A did not have to read (out of view, unchanged) */
  uint32_t regularizer_saturated;  /* sticky since creation / state upload: a regulariser gradient term reached the
                                    * +-16 m range of the exact fixed-point sums, or one slot collected >= 100 senders
                                    * of one neighbour-count class; smooth positions may then deviate from the
                                    * reference.  Terms are 2 * regularizer_weight / count * (n . d) * n: with
                                    * neighbour distances of centimetres any weight below ~100 is far inside. */
  uint32_t n_pairs;            /* (slot, pixel) pairs pass A appended to the association tiles' bins */
  uint32_t n_overflow_pairs;   /* ... of which went through the overflow list (a tile's bin was full) */
  uint32_t max_tile_pairs;     /* pairs of the fullest tile */
} smx_recon_stats;
int smx_recon_get_stats(smx_recon r, smx_stream s, smx_recon_stats* out);
/* The n_* counters above are single-address atomics; they are collected only while enabled
 * (default on; benchmarks switch them off for the timed region).  surfels_size / merge_count are
 * always exact. */
int smx_recon_set_stats_enabled(smx_recon r, int32_t enabled);

/* Test / benchmark hooks (not part of the reference interface): raw access to
 * the surfel SoA rows (25 rows as in APP/cuda_surfel_reconstruction_kernels.cuh:49-78,
 * dense [25][count] on the host side) and to the per-pixel association images. */
int smx_recon_debug_download_surfels(smx_recon r, smx_stream s, float* rows, uint32_t count);
int smx_recon_debug_upload_surfels(smx_recon r, smx_stream s, const float* rows, uint32_t count,
                                   uint32_t merge_count);
enum {
  SMX_SCRATCH_SUPPORTING = 0,      /* u32 [H][W] */
  SMX_SCRATCH_SUPPORT_COUNTS = 1,  /* u32 */
  SMX_SCRATCH_DEPTH_SUMS = 2,      /* i64, 2^-32 fixed point */
  SMX_SCRATCH_CONFLICTING = 3,     /* u32, decoded index or 0xFFFFFFFF */
  SMX_SCRATCH_FIRST_DEPTH = 4,     /* f32 */
  SMX_SCRATCH_NEW_FLAGS = 5,       /* u8 [W*H] */
  SMX_SCRATCH_NEW_INDICES = 6      /* u32 [W*H], exclusive ranks */
};
int smx_recon_debug_download_scratch(smx_recon r, smx_stream s, int32_t which, void* dst);
/* Number of 1024-slot segments the last regulariser link scan did not have to read (every link of theirs stays among
 * slots nothing happened to; only with the statistics counters off -- the edge counters visit every link).  Tests. */
int smx_recon_debug_count_skipped_segments(smx_recon r, smx_stream s, uint32_t* out);
/* A/B switches; results are identical in every mode.  bit 0: every surfel kernel scans all slots like
 * the reference does instead of the compacted lists; bit 1: measurement blending as the reference's
 * start + iteration launches instead of the fused LDS kernel; bit 2: the regulariser's link scan gathers the flag byte
 * of every far link (no hot-group filter); bit 3: association bins of 16 pairs per tile, so that most pairs travel
 * through the overflow list; bit 4: pass A reserves bin space pair by pair instead of per (workgroup, tile) through
 * an LDS table (the path a pair takes that finds no room in that table); bit 5: the regulariser's far-term bins hold 4 records
 * per destination segment, bit 6: a sender workgroup addresses 2 destination segments through the bins -- the other far
 * terms take the atomic accumulators (the overflow paths of those bins); bit 7: the blend's other tile size (the
 * library picks 32 x 32 or 40 x 40 pixels by the number of tiles per compute unit; this bit swaps the choice);
 * bit 8: the list kernels run on a grid of four workgroups, so that every workgroup walks many steps;
 * bit 9: the regulariser's pass B and its edge kernel as ONE launch (the workgroup of a segment does the segment's edge work
 * itself, the work list stays in LDS) instead of two (pass B writes the work lists to memory, k_reg_accumulate walks
 * them: the default -- the fused launch is shorter alone and longer in the frame). */
int smx_recon_set_scan_mode(smx_recon r, int32_t mode);
/* TIMING ONLY -- the map is WRONG afterwards: leaves launches of smx_recon_integrate out, for the upper-bound runs of
 * bench.py --ub (what would the frame rate be without this chain?).  bit 0: no regulariser (pass B, edges, step);
 * bit 1: the front of the frame only (pass A, association tiles, blend): no integration, neighbour update, creation
 * or regulariser either.  bit 2: the internal stream does not wait for the front of the frame (blend -> integrate
 * hand-over left out), bit 3: the caller's stream does not wait for update + create (-> next pass A): what the two
 * cross-stream hand-overs cost the frame -- results undefined; bit 4: another arrangement of the streams (integrate + update
 * stay on the caller's stream behind the blend and wait for the previous call's edge kernel only, the internal stream keeps
 * pass B / edges / step and waits for update + create: the step kernel off every cycle, two hand-overs on the critical one) --
 * measured 5 % slower even as an upper bound, profiles/r6_ab_notes.md; bit 5 (test only): the front gate waits for one workgroup
 * more than the blend has and gives up after its bound; bit 6: the caller's stream is released behind the integration launch
 * instead of behind update + create; bit 7: the edge kernel works on the first 256 entries of every segment only; bit 8: the blend
 * stops behind its start ring (bits 6 - 8: upper bounds, profiles/r6_ab_notes.md sections 16, 21, 24).  0 = off. */
int smx_recon_debug_set_skip(smx_recon r, int32_t mask);
/* Frame pipelining (default on): the regulariser of a frame runs on an internal stream beside the first
 * kernels of the next smx_recon_integrate call (which only read what the regulariser does not write).
 * Every entry point that takes a stream first orders that stream after the pending regulariser, so the
 * one-stream semantics of CUDASurfelReconstruction are kept; results are identical on and off. */
int smx_recon_set_overlap(smx_recon r, int32_t enabled);
/* Dependency routing for a caller that runs its own pipeline around Integrate (smx_driver does: preprocessing of
 * later frames on a second stream).  An event record or wait costs a stream 6 - 8 us on this hardware, and the
 * caller's stream carries the frame-to-frame critical chain; these two hooks move one record and one wait per frame
 * from it to the internal stream.  Both are one-shot: they apply to the NEXT smx_recon_integrate call (also when
 * that call fails); either may be null.
 *   inputs_consumed  is recorded at the point from which that call no longer reads its four input images.
 *   chain_after      must have been recorded already; the call's internal completion mark waits for it.  The next
 *                    call orders its integration kernels -- and so every call after that all of its kernels -- after
 *                    that mark: work covered by chain_after is complete before the call AFTER the next one starts to
 *                    read its inputs, without any wait on the caller's stream.
 * With pipelining off both act on the caller's stream at the same points.
 * CONTRACT CHANGE for a caller that passes inputs_consumed (pipelining on): the caller's stream no longer waits for the
 * call's second half (integration, neighbour update, creation) when the call returns -- that wait is deferred into
 * the NEXT smx_recon_integrate call.  Until then the stream is NOT ordered behind the kernels that write the blended
 * depths back into the depth image and read the colour image: a caller that touches those four images (or reuses them)
 * outside another smx_recon_* entry point must first make its stream wait for inputs_consumed itself.  smx_driver does:
 * every run ends with that wait, and its frame upload / render / work-image download entry points wait for the steps
 * in flight.  Every other smx_recon_* entry point still orders its stream behind all internal work. */
int smx_recon_integrate_hooks(smx_recon r, smx_event inputs_consumed, smx_event chain_after);
/* A third hook of the same kind (one-shot, may be null): `inputs_ready` must have been recorded already (e.g. at the end of
 * the preprocessing of the frame, on the caller's preprocessing stream); the NEXT smx_recon_integrate call waits for it
 * on the caller's stream itself, AFTER its first kernel -- the all-slot scan, which does not read the four input images --
 * instead of the caller waiting in front of the call. */
int smx_recon_integrate_inputs_ready(smx_recon r, smx_event inputs_ready);

/* ---- radius-neighbor search (replaces CompressedOctree::FindNearestSurfelsWithinRadius,
 * APP/octree.h:470-477, APP/octree.cc:313-470, for batched queries) ---- */
/* Build a uniform-grid index over n points given as three device or host rows.
 * cell_size > 0; queries with radius <= cell_size touch at most 27 cells.  Points with a non-finite coordinate
 * are not indexed (no finite ball contains them).  The grid is sparse (sorted cell keys + a hash table of the occupied
 * 4x4x4-cell bricks), so cell_size is kept at any scene extent; results do not depend on it.  Synchronises s (two
 * small read-backs); the rows may be released / overwritten when the call returns.  Workspace is kept in the handle. */
/* device_id as in smx_recon_create; the index owns its workspace and reuses it from call to call. */
int smx_nn_create(int32_t device_id, smx_nn* out);
int smx_nn_destroy(smx_nn nn);
int smx_nn_build(smx_nn nn, smx_stream s, const float* x, const float* y, const float* z,
                 uint32_t n, float cell_size, int32_t rows_on_device);
/* For each query: up to k nearest points with dist^2 <= r2[q], ascending by
 * (dist^2, index).  state (may be NULL): points whose state byte has a bit of
 * skip_mask set are skipped (octree.cc:330-335).  Outputs are device or host
 * pointers according to outputs_on_device; out_idx/out_d2 are [nq][k].
 * With device-resident queries and outputs the call only enqueues kernels on s (no allocation once the workspace
 * has grown to the batch size, no synchronisation); with host pointers it returns after the results have arrived. */
int smx_nn_query_batch(smx_nn nn, smx_stream s, uint32_t nq, const float* qx, const float* qy,
                       const float* qz, const float* r2, int32_t k, const uint8_t* state,
                       uint8_t skip_mask, int32_t queries_on_device,
                       uint32_t* out_idx, float* out_d2, int32_t* out_count,
                       int32_t outputs_on_device);

/* Index geometry and (while enabled) counters of the queries since smx_nn_set_stats_enabled: what the C5 roofline of
 * SURVEY.md 8(d) is computed from.  The counters are single-address atomics (one per tile): off by default. */
typedef struct {
  uint32_t n_points, n_indexed, n_bricks;   /* given to the last build / with finite coordinates / occupied 4x4x4-cell bricks */
  float cell_size;                          /* the caller's, unless the 2^21-cells-per-axis key range forced it up */
  int32_t dim[3];                           /* cells per axis of the (sparse) grid */
  int32_t key_bits;                         /* width of the sort keys = 8 bits per radix pass */
  uint64_t tiles;                           /* query tiles (<= 64 queries of one brick) */
  uint64_t staged_candidates;               /* points staged in LDS, summed over the tiles */
  uint64_t distance_tests;                  /* exact tests, summed over the queries */
  uint64_t results;                         /* entries returned */
} smx_nn_stats;
/* Every indexed point queries its own neighbourhood (the full-retriangulation pattern, config C5 of SURVEY.md 8d;
 * APP/surfel_meshing.cc:549, 819-823): the same results as smx_nn_query_batch with the points' own positions, with
 * r^2 = factor * radius_squared[i] (device array indexed like the build rows) or, if radius_squared is NULL, r^2 =
 * factor for all.  Rows are indexed by point; points without finite coordinates get count 0.  Nothing is keyed, sorted
 * or gathered: a tile is an occupied brick, its queries are the brick's own records.  Device pointers only; enqueues
 * kernels on s, no allocation, no synchronisation. */
int smx_nn_query_self(smx_nn nn, smx_stream s, const float* radius_squared, float factor, int32_t k,
                      const uint8_t* state, uint8_t skip_mask, uint32_t* out_idx, float* out_d2, int32_t* out_count);
/* A/B switch of the query kernel (results are identical): 2 = one LANE per query over the brick tiles staged in LDS,
 * queries it cannot hold (more than 32 matches, very large regions) answered by kernel 0 afterwards (default);
 * 0 = one wavefront per query over the same staged tiles; 1 = one wavefront per query reading the brick ranges
 * through L1 / L2. */
int smx_nn_set_query_mode(smx_nn nn, int32_t mode);
int smx_nn_set_stats_enabled(smx_nn nn, smx_stream s, int32_t enabled);
int smx_nn_get_stats(smx_nn nn, smx_stream s, smx_nn_stats* out);

/* ---- the loop-closure hook the reference describes but does not ship (README.md:152-176; its call site is the
 * "### Loop closures ###" block of main.cc:1194-1200, between preprocessing and Integrate) ----
 * Every live surfel created at frame c < n_frames moves by the rigid correction frame_T[c] (row-major 3x4,
 * new_global_T_old_global; identity rows for frames that stay): offset = T * (X,Y,Z) - (X,Y,Z) is added to the raw
 * and to the smooth position (README.md:160-165), the normal becomes R * normal (:166-168); where reactivate[c] != 0
 * (array may be NULL) LastUpdateStamp is set to frame_index, which makes the surfel active for integration again
 * (:172-174).  Merged slots and slots created at c >= n_frames are untouched.  frame_T / reactivate are device
 * pointers if inputs_on_device, host pointers (synchronous call) otherwise. */
int smx_recon_deform_by_creation_frame(smx_recon r, smx_stream s, const float* frame_T, uint32_t n_frames,
                                       const uint8_t* reactivate, uint32_t frame_index, int32_t inputs_on_device);

/* ---- candidate lists for the mesher, straight from the device-resident map ----
 * Replaces, for the surfels of one batch (e.g. one changed-surfel delta), the per-surfel octree query at the top of
 * SurfelMeshing::TriangulateSurfel (APP/surfel_meshing.cc:417-425) with the widest radius that function can ask for,
 * radius_factor_squared * radius_squared (surfel_meshing.cc:359-360, --max_neighbor_search_range_increase_factor):
 * a query with a smaller radius and the same K is the prefix of this list with dist^2 <= that radius.
 *
 * build_neighbor_index: (re)builds `nn` over the smooth positions of all surfels_size() slots without a host
 * round trip; merged slots are left out (cuda_surfel_reconstruction.cc:348-358 hands the mesher the same rows and it
 * removes merged surfels from its octree).  The index is a snapshot: rebuild it after Integrate / Regularize.
 * neighbor_candidates: for q < n_indices, up to k (<= 64) nearest indexed surfels within the ball of slot
 * surfel_indices[q], ascending by (dist^2, index); an out-of-range or merged slot gets count 0.  state / skip_mask
 * as in smx_nn_query_batch, one byte per slot (surfels_size() bytes).  surfel_indices and state are device or host
 * pointers according to inputs_on_device, the three outputs according to outputs_on_device. */
int smx_recon_build_neighbor_index(smx_recon r, smx_stream s, smx_nn nn, float cell_size);
int smx_recon_neighbor_candidates(smx_recon r, smx_stream s, smx_nn nn, const uint32_t* surfel_indices,
                                  uint32_t n_indices, float radius_factor_squared, int32_t k,
                                  const uint8_t* state, uint8_t skip_mask, int32_t inputs_on_device,
                                  uint32_t* out_idx, float* out_d2, int32_t* out_count, int32_t outputs_on_device);

/* The per-triangle tests of SurfelMeshing::CheckRemeshing (APP/surfel_meshing.cc:590-650) for a batch of triangles
 * (three slot indices each, [n_triangles][3]) against the device-resident map; the mesher keeps the sequential part
 * (RemeshTrianglesAt and the visiting order).  long_edge_total_factor_squared as in surfel_meshing.cc:171-173.
 * flags[t]: bit 0 = long-edge condition (:605-617; independent of the pivot vertex); bits 1..3 = the triangle normal
 * formed from pivot index(0) / index(1) / index(2) (right = next, left = previous vertex, :576-589) is inconsistent
 * with all three surfel normals (:632-635); bit 4 = a vertex is merged (:559-570) or out of range (then no other bit
 * is set).  triangles and flags are device pointers if on_device, host pointers (synchronous call) otherwise. */
int smx_recon_check_triangles(smx_recon r, smx_stream s, const uint32_t* triangles, uint32_t n_triangles,
                              float long_edge_total_factor_squared, uint8_t* flags, int32_t on_device);

/* ---- benchmark input generator (not part of the reference's interface) ----
 * Renders one frame of the synthetic room stream into device buffers:
 * depth u16 = round(depth_scaling * z) with sigma = noise_sigma * z^2 noise and coherent 8x8
 * drop-outs, colour uchar3 = hash of the 10 cm world cell.  Pure function of its arguments. */
int smx_synth_render_room(smx_stream s, const smx_buffer_desc* depth_out, const smx_buffer_desc* color_out,
                          float fx, float fy, float cx, float cy, const float global_T_frame[12],
                          uint32_t seed, uint32_t frame_index, float depth_scaling, float noise_sigma,
                          float dropout);

#ifdef __cplusplus
}
#endif
#endif
