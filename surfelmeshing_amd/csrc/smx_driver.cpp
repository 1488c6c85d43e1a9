// smx_driver.cpp -- the reference caller's per-frame sequence (APP/main.cc:1015-1223) as native host code,
// written against the shim API (include/smx_shim.hpp).  No kernels here.
#include <cstdlib>
#include <map>
#include <memory>
#include <vector>

#include "smx_driver.h"
#include "smx_shim.hpp"

using namespace vis;

struct Frame {
  CUDABuffer<u16> depth;
  CUDABuffer<Vec3u8> color;
  unsigned long long last_reader = 0;   // 1 + index of the last enqueued step that reads this frame (0 = none)
  smx_event uploaded = nullptr;         // (staged uploads) completion of the copy kernels that filled the frame
  Frame(int h, int w) : depth(h, w), color(h, w) {}
  ~Frame() { if (uploaded) smx_event_destroy(uploaded); }
};

// The per-frame work images (main.cc:801-812).  Three sets: the preprocessing of frames f+1 and f+2 runs on its own
// stream while Integrate(f) still reads its set.
struct WorkSet {
  CUDABuffer<u16> filtered_depth_buffer_A, filtered_depth_buffer_B;
  CUDABuffer<float2_> normals_buffer;
  CUDABuffer<float> radius_buffer;
  CUDABuffer<u16>* final_depth;
  smx_event preprocessed = nullptr, integrated = nullptr;
  smx_event filtered = nullptr;   // (two preprocessing queues) the bilateral filter's image is written
  bool used = false;
  WorkSet(int h, int w) : filtered_depth_buffer_A(h, w), filtered_depth_buffer_B(h, w), normals_buffer(h, w),
                          radius_buffer(h, w), final_depth(&filtered_depth_buffer_A) {
    radius_buffer.Clear(0.0f, nullptr);
    SMX_SHIM_CHECK(smx_event_create(&preprocessed));
    SMX_SHIM_CHECK(smx_event_create(&integrated));
    SMX_SHIM_CHECK(smx_event_create(&filtered));
  }
  ~WorkSet() { smx_event_destroy(preprocessed); smx_event_destroy(integrated); smx_event_destroy(filtered); }
};

// Priority class of the preprocessing queues: 0 = the device's default.  Rounds 2-4 ran them at the LOWEST priority (-1: the
// preprocessing only has to keep up) -- which is what the frame loop's slow mode hangs on: one run in ten to one in five a
// tenth slower for its whole length (period 165 instead of 150 us at C2: both stream hand-overs twice as long), 0 of 40 runs
// with the preprocessing at the default priority, at 1.7 % less than the fast mode (6330 against 6440 frames/s; C3 1810
// against 1804): the same in expectation, without the tail (profiles/r5_ab_notes.md).
#ifndef SMX_PRE_PRIORITY
#define SMX_PRE_PRIORITY 0
#endif
#ifndef SMX_PRE2_PRIORITY
#define SMX_PRE2_PRIORITY SMX_PRE_PRIORITY
#endif
// Two preprocessing queues by default from this many pixels on (smx_driver_set_split_preprocessing overrides): at 1280 x 960
// the single queue is the frame's pace-maker -- the caller's stream waits 150 - 230 us a frame for its images -- and two
// queues are worth + 2.0 % (1 866 - 1 871 against 1 825 - 1 836 frames/s, profiles/r8o_C3_timelines.jsonl); at 640 x 480 the
// surfel chains pace the frame and the second queue costs 2 - 3 % (profiles/r6_ab_notes.md section 2, r8p).
#ifndef SMX_SPLIT_PRE_MIN_PIXELS
#define SMX_SPLIT_PRE_MIN_PIXELS (1024 * 768)
#endif
struct smx_driver_s {
  smx_driver_config cfg;
  PinholeCamera4f camera;
  CUDASurfelReconstruction reconstruction;
  WorkSet work0, work1, work2;
  WorkSet* last;
  WorkSet* prev;            // the set of the step before `last`
  WorkSet* set(unsigned long long k) { return (k % 3 == 0) ? &work0 : (k % 3 == 1) ? &work1 : &work2; }
  std::map<u32, std::unique_ptr<Frame>> frames;
  cudaStream_t pre_stream = nullptr;   // depth preprocessing of the next frame
  cudaStream_t pre_stream2 = nullptr;  // (split_pre) outlier cull + erosion / normals / radii, behind the filter of the same frame
  // Frames that arrive with their step (smx_driver_run_streamed) are copied by KERNELS on a staging queue of its own (round 5):
  // the copy of step i + 1's frame then runs beside the preprocessing of step i instead of in front of it in the same queue
  // (55 us of bus time in front of 125 us of preprocessing made that queue the pace-maker of the streamed loop: 5 200
  // against 6 100 frames/s), and a kernel's stores reach the preprocessing queue through an ordinary event -- the copy
  // ENGINE's do not when the runtime finds the event already complete (upload_on).
  // The staging queue is the second preprocessing queue, which is idle unless smx_driver_set_split_preprocessing is on (then
  // the uploads take the old route): a stream of its own for it made the slow mode of smx_runtime_defaults come back for
  // every run of the process, the resident ones included (4 of 14: profiles/r5_ab_notes.md).
  bool staged_uploads = true;
  uint64_t uploads_staged = 0, uploads_copy_engine = 0;   // frames of smx_driver_run_streamed by the route they took
  // Two preprocessing queues: the bilateral filter of frame f + 1 (VALU-bound, one 310-register wavefront per SIMD) runs
  // beside the outlier cull and the tail of frame f (gathers) instead of behind them.  At 1280 x 960 the single queue is
  // busy all of the time and the frame waits for it (profiles/r21_timeline_c3.md).
  bool split_pre = false;   // (set by the constructor from the image size)
  smx_event run_start = nullptr;
  bool overlap = true;
  bool run_ahead = false;  // smx_driver_run: preprocessing two steps ahead, dependencies routed off the caller's stream (A/B: -1 %)
  bool fuse_tail = true;   // erosion + normals + radii as one launch (A/B: smx_driver_set_fused_tail)
  bool fuse_head = false;  // bilateral filter + outlier cull as one launch (A/B: smx_driver_set_fused_head; measured slower, see smx_driver.h)
  unsigned long long frame_counter = 0;
  // smx_driver_debug_prepare: work sets preprocessed ahead of time, consumed in order by the next runs (measurement)
  std::vector<std::unique_ptr<WorkSet>> prepared;
  size_t prepared_next = 0;
  // smx_driver_profile_begin: timed events around one preprocessing stage, one pair per frame
  int read_timings = 0;            // smx_driver_set_read_timings
  double timing_sums[7] = {0, 0, 0, 0, 0, 0, 0};
  uint64_t timing_calls = 0, timing_last_call = 0;
  int prof_stage = -1;
  std::vector<smx_event> prof_ev;
  size_t prof_n = 0;

  explicit smx_driver_s(const smx_driver_config& c, const float* intr)
      : cfg(c), camera(c.width, c.height, intr), reconstruction(c.max_surfel_count, camera),
        work0(c.height, c.width), work1(c.height, c.width), work2(c.height, c.width), last(&work0), prev(&work0) {
    // (preprocessing runs ahead of the frame loop; its priority: see SMX_PRE_PRIORITY)
    SMX_SHIM_CHECK(smx_stream_create_with_priority(&pre_stream, SMX_PRE_PRIORITY));
    SMX_SHIM_CHECK(smx_stream_create_with_priority(&pre_stream2, SMX_PRE2_PRIORITY));
    SMX_SHIM_CHECK(smx_event_create(&run_start));
    SMX_SHIM_CHECK(smx_stream_synchronize(nullptr));
    split_pre = (long long)c.width * c.height >= (long long)SMX_SPLIT_PRE_MIN_PIXELS;
  }
  ~smx_driver_s() {
    smx_stream_synchronize(pre_stream); smx_stream_synchronize(pre_stream2); smx_stream_synchronize(nullptr);
    smx_stream_destroy(pre_stream); smx_stream_destroy(pre_stream2); smx_event_destroy(run_start);
    for (smx_event e : prof_ev) smx_event_destroy(e);
  }
};

// (measurement) brackets one preprocessing stage with timed events while a profile is running
struct StageTimer {
  smx_driver_s* d; cudaStream_t st; bool on;
  StageTimer(smx_driver_s* d_, cudaStream_t st_, int stage) : d(d_), st(st_) {
    on = d->prof_stage == stage && 2 * d->prof_n + 1 < d->prof_ev.size();
    if (on) (void)smx_event_record(d->prof_ev[2 * d->prof_n], st);
  }
  ~StageTimer() { if (on) { (void)smx_event_record(d->prof_ev[2 * d->prof_n + 1], st); ++d->prof_n; } }
};

namespace smx { void set_error(const char* fmt, ...); }  // libsmx's thread-local error text (smx_last_error)
static int fail(const char* msg) { smx::set_error("smx_driver: %s", msg); return SMX_ERR_INVALID_ARGUMENT; }

// Depth preprocessing of one frame, APP/main.cc:1015-1191.
// (tail_stream: the queue of everything behind the bilateral filter -- `stream` itself, or the second preprocessing queue)
// *done_on: the queue the last launch went to; null if that launch carried ws->preprocessed as its completion event itself
// (mark_done: the caller wants that event).
static int preprocess_frame(smx_driver d, cudaStream_t stream, const smx_driver_step& st, WorkSet* ws, cudaStream_t tail_stream,
                            cudaStream_t* done_on, bool mark_done = false) {
  const smx_driver_config& c = d->cfg;
  // validated before any array below is indexed and before any frame is stamped (main.cc:1084 rejects the same values)
  if (st.other_count != 0 && st.other_count != 2 && st.other_count != 4 && st.other_count != 6 && st.other_count != 8)
    return fail("Unsupported value for outlier_filtering_frame_count");
  auto it = d->frames.find(st.frame_index);
  if (it == d->frames.end()) return fail("frame not resident");
  for (int i = 0; i < st.other_count; ++i)
    if (d->frames.find(st.other_frames[i]) == d->frames.end()) return fail("outlier-cull neighbour frame not resident");
  CUDABuffer<u16>& depth_buffer = it->second->depth;
  it->second->last_reader = d->frame_counter;   // (run_one has already counted this step)
  const float* cam = d->camera.parameters();

  const u16 max_depth_u16 = (u16)(c.depth_scaling * c.max_depth > 65535.f ? 65535.f : c.depth_scaling * c.max_depth);
  const bool fused_head = d->fuse_head && st.other_count > 0;
  // Bilateral filtering and depth cutoff (:1015-1024)
  if (!fused_head) { StageTimer t_(d, stream, 0);
  BilateralFilteringAndDepthCutoffCUDA(stream, c.bilateral_filter_sigma_xy, c.bilateral_filter_sigma_depth_factor,
                                       /*value_to_ignore*/ 0, c.bilateral_filter_radius_factor,
                                       (u16)(c.depth_scaling * c.max_depth > 65535.f ? 65535.f : c.depth_scaling * c.max_depth),
                                       c.depth_valid_region_radius, depth_buffer.ToCUDA(),
                                       &ws->filtered_depth_buffer_A.ToCUDA()); }
  if (fused_head) tail_stream = stream;
  if (tail_stream != stream) {
    SMX_SHIM_CHECK(smx_event_record(ws->filtered, stream));
    SMX_SHIM_CHECK(smx_stream_wait_event(tail_stream, ws->filtered));
    stream = tail_stream;
  }
  CUDABuffer<u16>* src = &ws->filtered_depth_buffer_A;
  CUDABuffer<u16>* dst = &ws->filtered_depth_buffer_B;

  // Depth outlier filtering (:1037-1115)
  if (st.other_count > 0) {
    const CUDABuffer_<u16>* other_depths[8];
    CUDAMatrix3x4 others_TR_reference[8];
    for (int i = 0; i < st.other_count; ++i) {
      auto o = d->frames.find(st.other_frames[i]);
      if (o == d->frames.end()) return fail("outlier-cull neighbour frame not resident");
      other_depths[i] = &o->second->depth.ToCUDA();
      o->second->last_reader = d->frame_counter;
      others_TR_reference[i] = CUDAMatrix3x4(st.others_TR_reference[i]);
    }
    const int req = c.outlier_filtering_required_inliers;
    const bool all = (req == -1 || req == st.other_count);
#define SMX_CALL_OUTLIER_FUSION(n)                                                                                     \
  do {                                                                                                                 \
    if (all) OutlierDepthMapFusionCUDA<n + 1, u16>(stream, c.outlier_filtering_depth_tolerance_factor, src->ToCUDA(), \
                                                   cam[0], cam[1], cam[2], cam[3], other_depths, others_TR_reference,  \
                                                   &dst->ToCUDA());                                                    \
    else OutlierDepthMapFusionCUDA<n + 1, u16>(stream, req, c.outlier_filtering_depth_tolerance_factor, src->ToCUDA(), \
                                               cam[0], cam[1], cam[2], cam[3], other_depths, others_TR_reference,      \
                                               &dst->ToCUDA());                                                        \
  } while (0)
    // (fused head: the filter and the cull of its own output pixel in one launch -- timed as stage 0)
#define SMX_CALL_FUSED_HEAD(n)                                                                                          \
  BilateralFilteringAndOutlierFusionCUDA<n + 1>(stream, c.bilateral_filter_sigma_xy, c.bilateral_filter_sigma_depth_factor,  \
                                                c.bilateral_filter_radius_factor, max_depth_u16, c.depth_valid_region_radius,   \
                                                depth_buffer.ToCUDA(), all ? -1 : req, c.outlier_filtering_depth_tolerance_factor, \
                                                cam[0], cam[1], cam[2], cam[3], other_depths, others_TR_reference,           \
                                                &src->ToCUDA(), &dst->ToCUDA())
    StageTimer t_(d, stream, fused_head ? 0 : 1);
    switch (st.other_count) {
      case 2: if (fused_head) SMX_CALL_FUSED_HEAD(2); else SMX_CALL_OUTLIER_FUSION(2); break;
      case 4: if (fused_head) SMX_CALL_FUSED_HEAD(4); else SMX_CALL_OUTLIER_FUSION(4); break;
      case 6: if (fused_head) SMX_CALL_FUSED_HEAD(6); else SMX_CALL_OUTLIER_FUSION(6); break;
      case 8: if (fused_head) SMX_CALL_FUSED_HEAD(8); else SMX_CALL_OUTLIER_FUSION(8); break;
      default: return fail("Unsupported value for outlier_filtering_frame_count");  // main.cc:1084
    }
#undef SMX_CALL_FUSED_HEAD
#undef SMX_CALL_OUTLIER_FUSION
    std::swap(src, dst);
  }
  // Depth map erosion (:1128-1140), normals (:1154-1164), radii (:1180-1191): one fused launch (same images)
  if (d->fuse_tail) {
    StageTimer t_(d, stream, 2);
    ErodeNormalsRadiiCUDA(stream, c.depth_erosion_radius, c.observation_angle_threshold_deg, c.point_radius_extension_factor,
                          c.point_radius_clamp_factor, c.depth_scaling, cam[0], cam[1], cam[2], cam[3], src->ToCUDA(),
                          &dst->ToCUDA(), &ws->normals_buffer.ToCUDA(), &ws->radius_buffer.ToCUDA(),
                          (mark_done && d->prof_stage != 2) ? ws->preprocessed : nullptr);
    if (mark_done && d->prof_stage != 2) stream = nullptr;
    std::swap(src, dst);
  } else {
    if (c.depth_erosion_radius > 0) ErodeDepthMapCUDA(stream, c.depth_erosion_radius, src->ToCUDA(), &dst->ToCUDA());
    else CopyWithoutBorderCUDA(stream, src->ToCUDA(), &dst->ToCUDA());
    std::swap(src, dst);
    ComputeNormalsAndDropBadPixelsCUDA(stream, c.observation_angle_threshold_deg, c.depth_scaling, cam[0], cam[1], cam[2],
                                       cam[3], src->ToCUDA(), &dst->ToCUDA(), &ws->normals_buffer.ToCUDA());
    std::swap(src, dst);
    ComputePointRadiiAndRemoveIsolatedPixelsCUDA(stream, c.point_radius_extension_factor, c.point_radius_clamp_factor,
                                                 c.depth_scaling, cam[0], cam[1], cam[2], cam[3], src->ToCUDA(),
                                                 &ws->radius_buffer.ToCUDA(), &dst->ToCUDA());
    std::swap(src, dst);
  }
  ws->final_depth = src;
  *done_on = stream;
  return SMX_OK;
}

// Surfel reconstruction of one preprocessed frame, APP/main.cc:1205-1223.
static int integrate_frame(smx_driver d, cudaStream_t stream, const smx_driver_step& st, WorkSet* ws) {
  const smx_driver_config& c = d->cfg;
  auto it = d->frames.find(st.frame_index);
  if (it == d->frames.end()) return fail("frame not resident");

  // Surfel reconstruction (:1205-1223)
  const smx_integrate_params& p = c.integrate;
  d->reconstruction.Integrate(stream, st.frame_index, c.depth_scaling, ws->final_depth, ws->normals_buffer, ws->radius_buffer,
                              it->second->color, SE3f(st.global_T_frame), p.sensor_noise_factor,
                              p.max_surfel_confidence, p.regularizer_weight, p.regularization_frame_window_size,
                              p.do_blending != 0, p.measurement_blending_radius,
                              p.regularization_iterations_per_integration_iteration,
                              p.radius_factor_for_regularization_neighbors, p.normal_compatibility_threshold_deg,
                              p.surfel_integration_active_window_size);
  return SMX_OK;
}

extern "C" {

int smx_driver_create(const smx_driver_config* config, smx_driver* out) {
  if (!config || !out) return fail("null argument");
  const float intr[4] = {config->fx, config->fy, config->cx, config->cy};
  *out = new smx_driver_s(*config, intr);
  return SMX_OK;
}

int smx_driver_destroy(smx_driver d) { delete d; return SMX_OK; }

int smx_driver_recon(smx_driver d, smx_recon* out) {
  if (!d || !out) return fail("null argument");
  *out = d->reconstruction.handle();
  return SMX_OK;
}

// Integrate marks "inputs consumed / map complete" on its INTERNAL stream and defers the caller's stream's wait for it
// into the next call (smx_recon_integrate_hooks): after a run the caller's stream is therefore not ordered behind the
// last steps' second halves -- k_integrate writes the blended depths back into the work set, update + create read the
// frame's colour.  Whoever touches a work set or a frame image outside the frame loop orders its stream here first.
static int wait_for_steps_in_flight(smx_driver d, cudaStream_t stream) {
  if (d->prev != d->last && d->prev->used) SMX_SHIM_CHECK(smx_stream_wait_event(stream, d->prev->integrated));
  if (d->last->used) SMX_SHIM_CHECK(smx_stream_wait_event(stream, d->last->integrated));
  return SMX_OK;
}

static Frame* get_or_make(smx_driver d, uint32_t f) {
  auto& slot = d->frames[f];
  if (!slot) slot.reset(new Frame(d->cfg.height, d->cfg.width));
  return slot.get();
}

int smx_driver_upload_frame(smx_driver d, smx_stream s, uint32_t frame_index, const uint16_t* depth, const uint8_t* color) {
  if (!d || !depth || !color) return fail("null argument");
  Frame* f = get_or_make(d, frame_index);
  if (f->last_reader != 0) { const int rc = wait_for_steps_in_flight(d, s); if (rc != SMX_OK) return rc; }
  f->depth.UploadAsync(s, depth);
  f->color.UploadAsync(s, reinterpret_cast<const Vec3u8*>(color));
  return SMX_OK;
}

int smx_driver_render_frame(smx_driver d, smx_stream s, uint32_t frame_index, const float global_T_frame[12],
                            uint32_t seed, float noise_sigma, float dropout) {
  if (!d || !global_T_frame) return fail("null argument");
  Frame* f = get_or_make(d, frame_index);
  if (f->last_reader != 0) { const int rc = wait_for_steps_in_flight(d, s); if (rc != SMX_OK) return rc; }
  return smx_synth_render_room(s, f->depth.ToCUDA().desc(), f->color.ToCUDA().desc(), d->cfg.fx, d->cfg.fy, d->cfg.cx,
                               d->cfg.cy, global_T_frame, seed, frame_index, d->cfg.depth_scaling, noise_sigma, dropout);
}

int smx_driver_release_frame(smx_driver d, uint32_t frame_index) {
  if (!d) return fail("null argument");
  d->frames.erase(frame_index);
  return SMX_OK;
}

int smx_driver_frame_descs(smx_driver d, uint32_t frame_index, smx_buffer_desc* depth, smx_buffer_desc* color) {
  if (!d) return fail("null argument");
  auto it = d->frames.find(frame_index);
  if (it == d->frames.end()) return fail("frame not resident");
  if (depth) *depth = *it->second->depth.ToCUDA().desc();
  if (color) *color = *it->second->color.ToCUDA().desc();
  return SMX_OK;
}

// Copy of one host frame into the frame store, on the stream whose next kernels read it.  (Not on a stream of its
// own: a copy-engine write followed by a kernel in the SAME stream gets the cache invalidation it needs, while a
// cross-stream event wait that finds the event already complete is dropped by the runtime together with it --
// measured: frames copied on a separate upload stream one step ahead were read stale from L2 now and then.)
// A slot that steps still in flight read is overwritten only after the last of them has finished.
static int upload_on(smx_driver d, cudaStream_t stream, const smx_driver_host_frame& u) {
  if (u.depth == nullptr) return SMX_OK;
  if (u.color == nullptr) return fail("upload without a colour image");
  Frame* f = get_or_make(d, u.frame_index);
  if (f->last_reader != 0) {
    // readers: the last enqueued step (its event), or older ones (the other work set's event covers every step
    // before the last)
    WorkSet* w = (f->last_reader >= d->frame_counter) ? d->last : d->prev;
    if (w->used) SMX_SHIM_CHECK(smx_stream_wait_event(stream, w->integrated));
  }
  f->depth.UploadAsync(stream, u.depth);
  f->color.UploadAsync(stream, reinterpret_cast<const Vec3u8*>(u.color));
  return SMX_OK;
}

// The staged form: both images copied by kernels on the staging queue (page-locked sources only); `consumer` -- the queue
// whose next kernels read the frame -- waits for the copy's completion event.  ONE route per frame: both sources are probed
// before anything is enqueued, and a frame with a pageable image takes the copy-engine route as a whole (*staged = false,
// nothing enqueued here; round 5 found out half-way, after the reuse wait and the depth copy were on the staging queue, and
// finished a mixed frame with a host-side synchronisation inside the loop -- advisor r5).
static int upload_staged(smx_driver d, cudaStream_t consumer, const smx_driver_host_frame& u, bool* staged) {
  *staged = false;
  if (u.depth == nullptr) { *staged = true; return SMX_OK; }
  if (u.color == nullptr) return fail("upload without a colour image");
  const size_t px = (size_t)d->cfg.width * d->cfg.height;
  int32_t depth_ok = 0, color_ok = 0;
  SMX_SHIM_CHECK(smx_host_is_page_locked(u.depth, px * sizeof(u16), &depth_ok));
  SMX_SHIM_CHECK(smx_host_is_page_locked(u.color, px * 3, &color_ok));
  if (!depth_ok || !color_ok) return SMX_OK;   // (the caller falls back to upload_on)
  Frame* f = get_or_make(d, u.frame_index);
  if (!f->uploaded) SMX_SHIM_CHECK(smx_event_create(&f->uploaded));
  if (f->last_reader != 0) {
    WorkSet* w = (f->last_reader >= d->frame_counter) ? d->last : d->prev;
    if (w->used) SMX_SHIM_CHECK(smx_stream_wait_event(d->pre_stream2, w->integrated));
  }
  if (!f->depth.UploadByKernelAsync(d->pre_stream2, u.depth) ||
      !f->color.UploadByKernelAsync(d->pre_stream2, reinterpret_cast<const Vec3u8*>(u.color), f->uploaded))
    return fail("staged upload refused a source that was probed page-locked");
  SMX_SHIM_CHECK(smx_stream_wait_event(consumer, f->uploaded));
  *staged = true;
  return SMX_OK;
}

// One frame of the loop: [copy of a frame that arrives with this step] + preprocessing (own stream when
// overlapping) + Integrate.
static int run_one(smx_driver d, smx_stream s, const smx_driver_step& step, const smx_driver_host_frame* arriving) {
  int rc;
  if (arriving) {
    bool staged = false;
    // (with two preprocessing queues the staging queue is the second of them: the copy-engine route, counted below)
    if (d->overlap && d->staged_uploads && !d->split_pre) { rc = upload_staged(d, d->pre_stream, *arriving, &staged); if (rc != SMX_OK) return rc; }
    if (arriving->depth != nullptr) ++(staged ? d->uploads_staged : d->uploads_copy_engine);
    if (!staged) {
      rc = upload_on(d, d->overlap ? d->pre_stream : (cudaStream_t)s, *arriving);
      if (rc != SMX_OK) return rc;
    }
  }
  WorkSet* ws = d->set(d->frame_counter++);
  if (d->overlap) {
    // preprocessing(f) on its own stream: it may start as soon as Integrate(f-3) has released this work set,
    // i.e. it overlaps Integrate(f-1); Integrate(f) then waits for it.
    if (ws->used) SMX_SHIM_CHECK(smx_stream_wait_event(d->pre_stream, ws->integrated));
    // (frames that arrive with their step are copied on pre_stream and read behind it in the same queue: see upload_on)
    cudaStream_t done_on = nullptr;
    rc = preprocess_frame(d, d->pre_stream, step, ws, (d->split_pre && !arriving) ? d->pre_stream2 : d->pre_stream, &done_on, true);
    if (rc != SMX_OK) return rc;
    if (done_on) SMX_SHIM_CHECK(smx_event_record(ws->preprocessed, done_on));
    // Both dependencies are routed through Integrate itself: it waits for the preprocessed images behind its all-slot
    // scan (which does not read them), and marks "images consumed" on its internal stream -- the caller's stream, which
    // carries the front chain of the frame, gets neither a wait in front of the call nor a record behind it.
    SMX_SHIM_CHECK(smx_recon_integrate_inputs_ready(d->reconstruction.handle(), ws->preprocessed));
    SMX_SHIM_CHECK(smx_recon_integrate_hooks(d->reconstruction.handle(), ws->integrated, nullptr));
    rc = integrate_frame(d, s, step, ws);
    if (rc != SMX_OK) {
      (void)smx_recon_integrate_hooks(d->reconstruction.handle(), nullptr, nullptr);
      (void)smx_recon_integrate_inputs_ready(d->reconstruction.handle(), nullptr);
      return rc;
    }
  } else {
    cudaStream_t done_on = nullptr;
    rc = preprocess_frame(d, s, step, ws, s, &done_on);
    if (rc != SMX_OK) return rc;
    rc = integrate_frame(d, s, step, ws);
    if (rc != SMX_OK) return rc;
    SMX_SHIM_CHECK(smx_event_record(ws->integrated, s));
  }
  ws->used = true;
  d->prev = d->last;
  d->last = ws;
  if (d->read_timings) {   // APP/main.cc:1511-1524
    float t[7];
    if (d->read_timings == 2) {
      d->reconstruction.GetTimings(&t[0], &t[1], &t[2], &t[3], &t[4], &t[5], &t[6]);
      for (int k = 0; k < 7; ++k) d->timing_sums[k] += t[k];
      ++d->timing_calls;
    } else {
      const uint64_t call = d->reconstruction.GetTimingsNoWait(&t[0], &t[1], &t[2], &t[3], &t[4], &t[5], &t[6]);
      if (call > d->timing_last_call) {
        for (int k = 0; k < 7; ++k) d->timing_sums[k] += t[k];
        ++d->timing_calls;
        d->timing_last_call = call;
      }
    }
  }
  return SMX_OK;
}

// Preprocessing of one step on the preprocessing stream, into the next work set in rotation.
static int preprocess_ahead(smx_driver d, const smx_driver_step& step, WorkSet** out) {
  WorkSet* ws = d->set(d->frame_counter++);
  if (ws->used) SMX_SHIM_CHECK(smx_stream_wait_event(d->pre_stream, ws->integrated));
  cudaStream_t done_on = nullptr;
  const int rc = preprocess_frame(d, d->pre_stream, step, ws, d->split_pre ? d->pre_stream2 : d->pre_stream, &done_on, true);
  if (rc != SMX_OK) return rc;
  if (done_on) SMX_SHIM_CHECK(smx_event_record(ws->preprocessed, done_on));
  *out = ws;
  return SMX_OK;
}

// The frame loop over frames that are resident, with the preprocessing two steps ahead of Integrate.  The caller's
// stream carries the frame-to-frame critical chain (Integrate(f + 1) needs the map Integrate(f) leaves), and every
// event record or wait in it costs 6 - 8 us of a 300 us frame.  The dependencies on the preprocessing stream are
// therefore routed through the reconstruction's internal stream (smx_recon_integrate_hooks): "work set free again"
// is recorded there, and its completion mark for step i -- which Integrate(i + 1) waits for anyway -- also waits for
// the preprocessing of step i + 2.  Steps 0 and 1 of a call wait on the caller's stream as before.
// A step failed with work sets preprocessed but not integrated: bring the driver back to a state from which the next
// run or upload is correct whatever it touches -- everything enqueued so far completes, no work set or frame counts as
// in flight any more.
static int abandon_run(smx_driver d, smx_stream s, int rc) {
  (void)smx_recon_integrate_hooks(d->reconstruction.handle(), nullptr, nullptr);
  (void)smx_recon_integrate_inputs_ready(d->reconstruction.handle(), nullptr);
  (void)smx_stream_synchronize(d->pre_stream);
  (void)smx_stream_synchronize(d->pre_stream2);
  (void)smx_stream_synchronize(s);
  d->work0.used = d->work1.used = d->work2.used = false;
  for (auto& f : d->frames) f.second->last_reader = 0;
  return rc;
}

static int run_ahead(smx_driver d, smx_stream s, const smx_driver_step* steps, int32_t n) {
  WorkSet* ring[3] = {nullptr, nullptr, nullptr};
  int rc;
  for (int i = 0; i < n && i < 2; ++i)
    if ((rc = preprocess_ahead(d, steps[i], &ring[i % 3])) != SMX_OK) return abandon_run(d, s, rc);
  for (int i = 0; i < n; ++i) {
    smx_event chain = nullptr;
    if (i + 2 < n) {
      if ((rc = preprocess_ahead(d, steps[i + 2], &ring[(i + 2) % 3])) != SMX_OK) return abandon_run(d, s, rc);
      chain = ring[(i + 2) % 3]->preprocessed;
    }
    WorkSet* ws = ring[i % 3];
    if (i < 2) SMX_SHIM_CHECK(smx_stream_wait_event(s, ws->preprocessed));
    SMX_SHIM_CHECK(smx_recon_integrate_hooks(d->reconstruction.handle(), ws->integrated, chain));
    if ((rc = integrate_frame(d, s, steps[i], ws)) != SMX_OK) return abandon_run(d, s, rc);
    ws->used = true;
    d->prev = d->last;
    d->last = ws;
  }
  return SMX_OK;
}

// (measurement: bench.py --ub hoist-pre) Preprocesses the n steps NOW, each into a work set of its own, and waits for
// them; the next smx_driver_run calls integrate those sets in order instead of preprocessing -- the same images, the same
// results, with the preprocessing queue empty while the frames are timed.
int smx_driver_debug_prepare(smx_driver d, smx_stream s, const smx_driver_step* steps, int32_t n) {
  if (!d || (!steps && n > 0)) return fail("null argument");
  SMX_SHIM_CHECK(smx_stream_synchronize(s));
  if (d->prepared_next >= d->prepared.size()) {
    // (the sets are freed: nothing may refer to them any more, and whatever still reads them has to be through)
    { uint32_t live = 0, slots = 0;   // (joins s with the reconstruction's internal stream and waits for both)
      SMX_SHIM_CHECK(smx_recon_counts(d->reconstruction.handle(), s, &live, &slots)); }
    d->last = d->prev = &d->work0;
    d->prepared.clear(); d->prepared_next = 0;
  }
  for (int i = 0; i < n; ++i) {
    std::unique_ptr<WorkSet> ws(new WorkSet(d->cfg.height, d->cfg.width));
    // (the set's constructor clears its radius image on the null stream, which the preprocessing queue is not ordered
    // behind: without this the clear now and then landed on top of the radii -- an intermittent mismatch of the prepared run)
    SMX_SHIM_CHECK(smx_stream_synchronize(nullptr));
    ++d->frame_counter;
    cudaStream_t done_on = nullptr;
    const int rc = preprocess_frame(d, d->pre_stream, steps[i], ws.get(), d->pre_stream, &done_on);
    if (rc != SMX_OK) return rc;
    d->prepared.push_back(std::move(ws));
  }
  SMX_SHIM_CHECK(smx_stream_synchronize(d->pre_stream));
  return SMX_OK;
}

int smx_driver_profile_begin(smx_driver d, int32_t stage, int32_t max_frames) {
  if (!d || stage < 0 || stage > 2 || max_frames <= 0) return fail("invalid argument");
  while (d->prof_ev.size() < 2 * (size_t)max_frames) {
    smx_event e = nullptr;
    SMX_SHIM_CHECK(smx_event_create_timed(&e));
    d->prof_ev.push_back(e);
  }
  d->prof_stage = stage; d->prof_n = 0;
  return SMX_OK;
}

int smx_driver_profile_end(smx_driver d, float* avg_ms, int32_t* frames) {
  if (!d || !avg_ms || !frames) return fail("null argument");
  double sum = 0;
  for (size_t i = 0; i < d->prof_n; ++i) {
    float ms = 0;
    SMX_SHIM_CHECK(smx_event_elapsed_ms(d->prof_ev[2 * i], d->prof_ev[2 * i + 1], &ms));
    sum += ms;
  }
  *frames = (int32_t)d->prof_n;
  *avg_ms = d->prof_n ? (float)(sum / (double)d->prof_n) : 0.0f;
  d->prof_stage = -1; d->prof_n = 0;
  return SMX_OK;
}

int smx_driver_run(smx_driver d, smx_stream s, const smx_driver_step* steps, int32_t n) {
  if (!d || (!steps && n > 0)) return fail("null argument");
  if (d->prepared_next < d->prepared.size()) {
    for (int i = 0; i < n; ++i) {
      if (d->prepared_next >= d->prepared.size()) return fail("fewer prepared work sets than steps");
      WorkSet* ws = d->prepared[d->prepared_next++].get();   // (smx_driver_debug_prepare counted the step when it preprocessed it)
      // (the same bookkeeping as run_one: "inputs consumed" marked by Integrate itself, the set counts as in flight)
      SMX_SHIM_CHECK(smx_recon_integrate_hooks(d->reconstruction.handle(), ws->integrated, nullptr));
      const int rc = integrate_frame(d, s, steps[i], ws);
      if (rc != SMX_OK) { (void)smx_recon_integrate_hooks(d->reconstruction.handle(), nullptr, nullptr); return rc; }
      ws->used = true;
      d->prev = d->last;
      d->last = ws;
    }
    return wait_for_steps_in_flight(d, s);
  }
  // Everything already enqueued on s (frame uploads / renders, earlier runs) precedes the first preprocessing.
  if (d->overlap) {
    SMX_SHIM_CHECK(smx_event_record(d->run_start, s));
    SMX_SHIM_CHECK(smx_stream_wait_event(d->pre_stream, d->run_start));
    if (d->run_ahead) { const int rc = run_ahead(d, s, steps, n); return rc != SMX_OK ? rc : wait_for_steps_in_flight(d, s); }
  }
  for (int i = 0; i < n; ++i) {
    const int rc = run_one(d, s, steps[i], nullptr);
    if (rc != SMX_OK) return rc;
  }
  // (one wait per RUN, not per frame: what follows the run on s -- downloads, uploads, other entry points' work on the
  // work sets -- finds the last steps' second halves complete)
  return d->overlap ? wait_for_steps_in_flight(d, s) : SMX_OK;
}

int smx_driver_run_streamed(smx_driver d, smx_stream s, const smx_driver_step* steps,
                            const smx_driver_host_frame* uploads, int32_t n) {
  if (!d || ((!steps || !uploads) && n > 0)) return fail("null argument");
  if (d->overlap) {
    SMX_SHIM_CHECK(smx_event_record(d->run_start, s));
    SMX_SHIM_CHECK(smx_stream_wait_event(d->pre_stream, d->run_start));
    SMX_SHIM_CHECK(smx_stream_wait_event(d->pre_stream2, d->run_start));
  }
  for (int i = 0; i < n; ++i) {
    const int rc = run_one(d, s, steps[i], &uploads[i]);
    if (rc != SMX_OK) return rc;
  }
  return d->overlap ? wait_for_steps_in_flight(d, s) : SMX_OK;
}

int smx_driver_set_overlap(smx_driver d, int32_t enabled) {
  if (!d) return fail("null argument");
  d->overlap = enabled != 0;
  return SMX_OK;
}

int smx_driver_set_read_timings(smx_driver d, int32_t mode) {
  if (!d || mode < 0 || mode > 2) return fail("invalid argument");
  d->read_timings = mode;
  return SMX_OK;
}

int smx_driver_timing_sums(smx_driver d, double sums_ms[7], uint64_t* calls, int32_t reset) {
  if (!d || !sums_ms || !calls) return fail("null argument");
  for (int k = 0; k < 7; ++k) sums_ms[k] = d->timing_sums[k];
  *calls = d->timing_calls;
  if (reset) { for (int k = 0; k < 7; ++k) d->timing_sums[k] = 0; d->timing_calls = 0; }
  return SMX_OK;
}

int smx_driver_debug_streams(smx_driver d, smx_stream out[2]) {   // (measurement: the two preprocessing queues)
  if (!d || !out) return fail("null argument");
  out[0] = d->pre_stream; out[1] = d->pre_stream2;
  return SMX_OK;
}

int smx_driver_set_staged_uploads(smx_driver d, int32_t enabled) {
  if (!d) return fail("null argument");
  d->staged_uploads = enabled != 0;
  return SMX_OK;
}

int smx_driver_upload_counts(smx_driver d, uint64_t* staged, uint64_t* copy_engine, int32_t reset) {
  if (!d) return fail("null argument");
  if (staged) *staged = d->uploads_staged;
  if (copy_engine) *copy_engine = d->uploads_copy_engine;
  if (reset) { d->uploads_staged = 0; d->uploads_copy_engine = 0; }
  return SMX_OK;
}

int smx_driver_set_run_ahead(smx_driver d, int32_t enabled) {
  if (!d) return fail("null argument");
  d->run_ahead = enabled != 0;
  return SMX_OK;
}

int smx_driver_set_fused_head(smx_driver d, int32_t enabled) {
  if (!d) return fail("null argument");
  d->fuse_head = enabled != 0;
  return SMX_OK;
}

int smx_driver_set_split_preprocessing(smx_driver d, int32_t enabled) {
  if (!d) return fail("null argument");
  d->split_pre = enabled != 0;
  return SMX_OK;
}

int smx_driver_set_pre_cu_mask(smx_driver d, const uint32_t* mask_words, uint32_t n_words) {
  if (!d || (n_words && !mask_words)) return fail("null argument");
  SMX_SHIM_CHECK(smx_stream_synchronize(d->pre_stream));
  SMX_SHIM_CHECK(smx_stream_synchronize(d->pre_stream2));
  cudaStream_t a = nullptr, b = nullptr;
  if (n_words) {
    SMX_SHIM_CHECK(smx_stream_create_with_cu_mask(&a, mask_words, n_words));
    SMX_SHIM_CHECK(smx_stream_create_with_cu_mask(&b, mask_words, n_words));
  } else {
    SMX_SHIM_CHECK(smx_stream_create_with_priority(&a, SMX_PRE_PRIORITY));
    SMX_SHIM_CHECK(smx_stream_create_with_priority(&b, SMX_PRE2_PRIORITY));
  }
  smx_stream_destroy(d->pre_stream); smx_stream_destroy(d->pre_stream2);
  d->pre_stream = a; d->pre_stream2 = b;
  return SMX_OK;
}

int smx_driver_set_fused_tail(smx_driver d, int32_t enabled) {
  if (!d) return fail("null argument");
  d->fuse_tail = enabled != 0;
  return SMX_OK;
}

int smx_driver_work_descs(smx_driver d, smx_buffer_desc* depth, smx_buffer_desc* normals, smx_buffer_desc* radius) {
  if (!d) return fail("null argument");
  if (depth) *depth = *d->last->final_depth->ToCUDA().desc();
  if (normals) *normals = *d->last->normals_buffer.ToCUDA().desc();
  if (radius) *radius = *d->last->radius_buffer.ToCUDA().desc();
  return SMX_OK;
}

int smx_driver_download_frame(smx_driver d, smx_stream s, uint32_t frame_index, uint16_t* depth, uint8_t* color) {
  if (!d) return fail("null argument");
  auto it = d->frames.find(frame_index);
  if (it == d->frames.end()) return fail("frame not resident");
  if (depth) it->second->depth.DownloadAsync(s, depth);
  if (color) it->second->color.DownloadAsync(s, reinterpret_cast<Vec3u8*>(color));
  return smx_stream_synchronize(s);
}

int smx_driver_download_work(smx_driver d, smx_stream s, uint16_t* depth, float* normals, float* radius) {
  if (!d) return fail("null argument");
  { const int rc = wait_for_steps_in_flight(d, s); if (rc != SMX_OK) return rc; }
  if (depth) d->last->final_depth->DownloadAsync(s, depth);
  if (normals) d->last->normals_buffer.DownloadAsync(s, reinterpret_cast<float2_*>(normals));
  if (radius) d->last->radius_buffer.DownloadAsync(s, radius);
  return smx_stream_synchronize(s);
}

}  // extern "C"
