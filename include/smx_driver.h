/*
 * smx_driver.h -- native per-frame driver: the reference caller's frame loop (APP/main.cc:1015-1223) in C++,
 * written against the shim classes of smx_shim.hpp (CUDABuffer<T>, CUDASurfelReconstruction, the depth
 * preprocessing free functions).  It exists so that a stream of frames can be enqueued without a scripting
 * language between the launches; every frame goes through exactly the C-ABI entry points of smx.h.
 */
#ifndef SMX_DRIVER_H_
#define SMX_DRIVER_H_

#include "smx.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct smx_driver_s* smx_driver;

/* Defaults: APP/main.cc:279, 415-475 and 323-368. */
typedef struct {
  int32_t width, height;
  float fx, fy, cx, cy;                  /* pixel-corner convention */
  uint32_t max_surfel_count;
  float depth_scaling;
  float max_depth;
  float depth_valid_region_radius;
  float observation_angle_threshold_deg;
  int32_t depth_erosion_radius;
  int32_t outlier_filtering_required_inliers;   /* -1: all */
  float bilateral_filter_sigma_xy;
  float bilateral_filter_radius_factor;
  float bilateral_filter_sigma_depth_factor;
  float outlier_filtering_depth_tolerance_factor;
  float point_radius_extension_factor;
  float point_radius_clamp_factor;
  smx_integrate_params integrate;
} smx_driver_config;

/* One frame of work: which resident frames serve as the outlier-cull neighbours, their relative poses
 * (APP/main.cc:1039-1059) and the camera pose. */
typedef struct {
  uint32_t frame_index;
  int32_t other_count;                   /* 0, 2, 4, 6 or 8 */
  uint32_t other_frames[8];
  float others_TR_reference[8][12];
  float global_T_frame[12];
} smx_driver_step;

int smx_driver_create(const smx_driver_config* config, smx_driver* out);
int smx_driver_destroy(smx_driver d);
/* The reconstruction object the driver owns (borrowed handle). */
int smx_driver_recon(smx_driver d, smx_recon* out);
/* Frame store (raw u16 depth + uchar3 colour, APP/main.cc:905-984): upload from host, or render the synthetic
 * room (smx_synth_render_room), release when no later frame needs it. */
int smx_driver_upload_frame(smx_driver d, smx_stream s, uint32_t frame_index, const uint16_t* depth, const uint8_t* color);
int smx_driver_render_frame(smx_driver d, smx_stream s, uint32_t frame_index, const float global_T_frame[12],
                            uint32_t seed, float noise_sigma, float dropout);
int smx_driver_release_frame(smx_driver d, uint32_t frame_index);
int smx_driver_frame_descs(smx_driver d, uint32_t frame_index, smx_buffer_desc* depth, smx_buffer_desc* color);
/* Enqueue n frames (preprocessing + Integrate each) on the stream; returns without synchronising. */
int smx_driver_run(smx_driver d, smx_stream s, const smx_driver_step* steps, int32_t n);
/* The same with the frames arriving from host memory (the reference caller's staging, APP/main.cc:905-984):
 * uploads[i] = a frame (index + dense host images) that has to be in the frame store before step i reads it, or
 * depth == NULL for none.  The copy is enqueued on the stream that runs step i's preprocessing, directly in front of
 * it -- with the overlap on that is the driver's preprocessing stream, so the copy runs beside Integrate(i-1) -- and
 * is asynchronous to the host for page-locked sources (smx_host_alloc); pageable sources work and block the caller.
 * A slot that earlier steps read is overwritten only after those steps have finished. */
typedef struct {
  uint32_t frame_index;
  const uint16_t* depth;   /* [height][width] */
  const uint8_t* color;    /* [height][width][3] */
} smx_driver_host_frame;
int smx_driver_run_streamed(smx_driver d, smx_stream s, const smx_driver_step* steps,
                            const smx_driver_host_frame* uploads, int32_t n);
/* Overlap of the depth preprocessing of later frames (own stream, further sets of work images) with Integrate(f);
 * default on.  Results are identical either way. */
int smx_driver_set_overlap(smx_driver d, int32_t enabled);
/* A/B switch: bilateral filter + outlier cull as one fused launch where the library has one, or (default) as the reference's
 * two calls; same images.  The fused launch is the slower one on the loaded chip (profiles/r17_ab_notes.md, r17q: the cull's
 * gathers have nothing to hide behind in a kernel that runs one 310-register wavefront per SIMD). */
int smx_driver_set_fused_head(smx_driver d, int32_t enabled);
/* Two preprocessing queues (smx_driver_run with overlap on): the bilateral filter on one, the outlier cull and the
 * erosion / normals / radii launch of the same frame behind it on a second one, so that the filter of frame f + 1 -- VALU-bound,
 * one 310-register wavefront per SIMD -- runs beside the gathers of frame f's cull instead of behind them.  Same images.
 * Frames that arrive with their step (smx_driver_run_streamed) keep the single queue: the copy and its readers share it.
 * Default: on for images of 1024 x 768 pixels and more (where the preprocessing queue paces the frame: + 2 % at 1280 x 960), off
 * below (- 2 to - 3 % at 640 x 480). */
int smx_driver_set_split_preprocessing(smx_driver d, int32_t enabled);
/* A/B switch: erosion + normals + radii as one fused launch (default) or as the reference's three calls; same images. */
int smx_driver_set_fused_tail(smx_driver d, int32_t enabled);
/* smx_driver_run with overlap on (default OFF, results identical; measured 1 % slower than the plain loop, see
 * profiles/r04c_runahead_ab.txt: the extra wait lands on the internal stream, which is the longer chain): the preprocessing runs two steps ahead of Integrate
 * (three sets of work images) and its dependencies are routed through smx_recon_integrate_hooks, so that from the third
 * step of a call on the caller's stream carries one event record and one wait per frame instead of two and two. */
int smx_driver_set_run_ahead(smx_driver d, int32_t enabled);
/* Experiment (results identical): the preprocessing queues re-created on a subset of the compute units (mask as for
 * smx_stream_create_with_cu_mask; n_words = 0: plain queues again).  Waits for the driver's work. */
int smx_driver_set_pre_cu_mask(smx_driver d, const uint32_t* mask_words, uint32_t n_words);
/* Measurement hooks (bench.py).  smx_driver_debug_prepare: preprocess the n steps now, each into work images of its
 * own, and wait; the following smx_driver_run calls integrate those images in order instead of preprocessing (same
 * results; the preprocessing queue stays empty while the frames run).  smx_driver_profile_begin / _end: time stamps
 * around one preprocessing stage (0 = bilateral filter, 1 = outlier cull, 2 = erosion + normals + radii) of the next
 * max_frames frames, on the stream the stage runs on; _end returns the average duration. */
int smx_driver_debug_prepare(smx_driver d, smx_stream s, const smx_driver_step* steps, int32_t n);
int smx_driver_profile_begin(smx_driver d, int32_t stage, int32_t max_frames);
int smx_driver_profile_end(smx_driver d, float* avg_ms, int32_t* frames);
/* Measurement: the driver's two preprocessing queues (for smx_debug_handover_probe; never enqueue work on them). */
int smx_driver_debug_streams(smx_driver d, smx_stream out[2]);
/* smx_driver_run_streamed with overlap on (default ON; results identical): the frames that arrive with their steps are copied
 * by kernels on a staging queue of its own (page-locked sources: smx_host_alloc), so that the copy for step i + 1 runs beside
 * the preprocessing of step i; 0 = the copy engine in front of the step's preprocessing, in the same queue (rounds 1-4). */
int smx_driver_set_staged_uploads(smx_driver d, int32_t enabled);
/* How many frames of smx_driver_run_streamed took the staged route and how many the copy engine (pageable source, staging
 * switched off, no overlap, or two preprocessing queues): a frame loop that believes it stages and does not would otherwise
 * only show up as a lower frame rate.  reset = 1 zeroes the counters afterwards. */
int smx_driver_upload_counts(smx_driver d, uint64_t* staged, uint64_t* copy_engine, int32_t reset);
/* The reference's frame loop reads the seven stage times after EVERY Integrate (APP/main.cc:1511-1524) and adds them to
 * running sums.  mode 1: every step of smx_driver_run / _run_streamed does the same with GetTimingsNoWait (no stall: the
 * newest call known to be through; each call is added once); mode 2: with the blocking GetTimings (the reference's
 * semantics: the host waits for the frame before it enqueues the next); 0 (default): no reads.
 * smx_driver_timing_sums: the sums in ms and the number of calls in them; reset = 1 zeroes them afterwards. */
int smx_driver_set_read_timings(smx_driver d, int32_t mode);
int smx_driver_timing_sums(smx_driver d, double sums_ms[7], uint64_t* calls, int32_t reset);
/* Working buffers after the last frame: final (blended) depth, normals, radius. */
int smx_driver_work_descs(smx_driver d, smx_buffer_desc* depth, smx_buffer_desc* normals, smx_buffer_desc* radius);

/* Blocking downloads (dense host arrays) of a stored frame / of the working buffers. */
int smx_driver_download_frame(smx_driver d, smx_stream s, uint32_t frame_index, uint16_t* depth, uint8_t* color);
int smx_driver_download_work(smx_driver d, smx_stream s, uint16_t* depth, float* normals, float* radius);

#ifdef __cplusplus
}
#endif
#endif
