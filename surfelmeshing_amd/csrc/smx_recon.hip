// smx_recon.hip -- the surfel reconstruction object and its gfx950 kernels.
//
// Behaviour: CUDASurfelReconstruction::Integrate / Regularize / TransferAllToCPU /
// ExportVertices of the reference (APP/cuda_surfel_reconstruction.{h,cc},
// APP/cuda_surfel_reconstruction_kernels.{cc,cu}); cited per kernel below.
//
// Structure (DESIGN.md "Frame pipeline") -- this is NOT the reference's launch
// sequence.  The reference scans all N surfel slots in every surfel kernel
// (~140 B/slot/frame) and blocks the host twice per frame.  Here:
//   pass A  k_scan_visible     one streaming pass over stamp,X,Y,Z (16 B/slot), minus
//                              the 1024-slot segments whose bounding box is out of view:
//                              z-buffer atomics + a compacted list of the slots that
//                              project into the image;
//   list kernels               associate / merge-decide / integrate /
//                              update-neighbors (+ creation) run over that list only;
//   pass B  k_neighbor_scan    one streaming pass over the neighbour records
//                              (16 B/slot): detach, which links point into the
//                              regulariser window, list of recently updated slots;
//   k_reg_accumulate           gradient terms of those links (far-term bins / LDS
//                              sums / packed global atomics) and every recent slot's
//                              own smoothness term, needy segments only;
//   k_reg_step                 regulariser step over the recent list only, no gathers.
// Across calls the work is pipelined on two streams (smx_recon_integrate): the
// read-only first kernels of frame f+1 run beside the regulariser of frame f.
// The surfel count, merge count and all list lengths live in device memory;
// no kernel launch needs a host round trip.
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <vector>

#include "smx_common.hpp"
#include <hip/hip_ext.h>

using namespace smx;

namespace {

// Attribute ids = the reference's SoA row numbers, APP/cuda_surfel_reconstruction_kernels.cuh:49-78 (the
// storage itself is grouped differently, see Surfels below)
enum : int {
  kX = 0, kY = 1, kZ = 2, kSmoothX = 3, kSmoothY = 4, kSmoothZ = 5, kConfidence = 6, kRadiusSq = 7,
  kNormalX = 8, kNormalY = 9, kNormalZ = 10, kGradX = 11, kGradY = 12, kGradZ = 13,
  kCreationStamp = 17, kLastUpdateStamp = 18, kNeighbor0 = 19, kGradCount = 23, kColor = 24, kRows = 25
};

struct DevState {
  uint32_t surfel_count;   // slots in use (incl. merged zombies)
  uint32_t merge_count;
  uint32_t create_base_next;  // slot count seen by k_new_flags_scan (which may run beside the previous frame's pass B)
  uint32_t recent_count;   // statistics: slots inside the regulariser window
  uint32_t create_base;
  uint32_t new_count;
  uint32_t capacity_clamped;
  uint32_t n_visible, n_merged, n_edges, n_integrated, n_replaced, n_conflict_hits;
  uint32_t n_window_edges, n_contributors;
  uint32_t n_segments_skipped;
  uint32_t reg_saturated;  // sticky: a regulariser term hit the +-16 m clamp or a sender-class counter came near its byte
  uint32_t n_pairs, n_overflow_pairs, max_tile_pairs;   // statistics of the association tiles' bins
};

// HBM layout of the surfel attributes.  The reference keeps 25 separate rows (SoA, kernels.cuh:49-78); that is
// ideal for its all-slot scans but makes every per-surfel gather touch one cache line per attribute.  Here
// the attributes are grouped into six 16-byte records per slot, chosen by which kernels use them together,
// and each group is its own array (group-major):
//   P  X, Y, Z, LastUpdateStamp      -- exactly what pass A streams (16 B/slot), and what every projection needs
//   S  SmoothX, SmoothY, SmoothZ, -   -- what the regulariser gathers per neighbour (one line instead of three)
//   N  NormalX, NormalY, NormalZ, RadiusSquared
//   T  Neighbor0..3                  -- what pass B streams (16 B/slot)
//   C  Confidence, CreationStamp, Color, -
//   G  GradientX, GradientY, GradientZ, -  (parked next smooth position)
// (Rounds 1-2 kept a copy of T in the second half of a 32-byte S record, so that the regulariser could see with one
// gather whether a neighbour lists the slot back; with the far-term bins nobody asks that question any more.)
// The reference's row order only matters at the boundary (TransferAllToCPU, ExportVertices, the debug row
// accessors); pack/unpack kernels convert there.  Rows 14-16 (Accum*, never used) and 23 (GradientCount,
// replaced by the fixed-point accumulators) have no storage.
enum : int { kGroupP = 0, kGroupS, kGroupN, kGroupT, kGroupC, kGroupG, kGroups };
__host__ __device__ constexpr int row_group(int row) {
  return row <= 2 ? kGroupP : row <= 5 ? kGroupS : row == 6 ? kGroupC : row == 7 ? kGroupN : row <= 10 ? kGroupN
       : row <= 13 ? kGroupG : row <= 16 ? -1 : row == 17 ? kGroupC : row == 18 ? kGroupP : row <= 22 ? kGroupT
       : row == 24 ? kGroupC : -1;
}
__host__ __device__ constexpr int row_sub(int row) {
  return row <= 2 ? row : row <= 5 ? row - 3 : row == 6 ? 0 : row == 7 ? 3 : row <= 10 ? row - 8
       : row <= 13 ? row - 11 : row == 17 ? 1 : row == 18 ? 3 : row <= 22 ? row - 19 : row == 24 ? 2 : -1;
}
// start of each group array in units of pitch x 16 bytes
__host__ __device__ constexpr int group_start(int g) { return g; }
constexpr int kQuadsPerSlot = 6;   // 96 bytes per slot
#ifndef SMX_LAYOUT_NC
#define SMX_LAYOUT_NC 0
#endif
struct Surfels {
  float* base;
  size_t pitch;  // slots per group array (multiple of 64)
  __host__ __device__ __forceinline__ size_t quad(int g, uint32_t i) const {
#if SMX_LAYOUT_NC
    // (A/B, VERDICT r5 item 5: N and C of a slot side by side in one 32-byte record -- a visible slot costs the integration two
    // lines instead of three -- P, S, T, G stay arrays of their own.  Measured: profiles/r6_ab_notes.md)
    if (g == kGroupN) return (size_t)4 * pitch + 2 * (size_t)i;
    if (g == kGroupC) return (size_t)4 * pitch + 2 * (size_t)i + 1;
    return (size_t)(g == kGroupP ? 0 : g == kGroupS ? 1 : g == kGroupT ? 2 : 3) * pitch + (size_t)i;
#else
    return (size_t)group_start(g) * pitch + (size_t)i;
#endif
  }
  __device__ __forceinline__ float& f(int row, uint32_t i) const { return base[quad(row_group(row), i) * 4 + row_sub(row)]; }
  __device__ __forceinline__ uint32_t& u(int row, uint32_t i) const {
    return reinterpret_cast<uint32_t*>(base)[quad(row_group(row), i) * 4 + row_sub(row)];
  }
  // whole 16-byte group of slot i
  __device__ __forceinline__ float4* group(int g, uint32_t i) const { return reinterpret_cast<float4*>(base) + quad(g, i); }
  __device__ __forceinline__ void set_neighbors(uint32_t i, const uint4& t) const { *reinterpret_cast<uint4*>(group(kGroupT, i)) = t; }
  __device__ __forceinline__ void set_neighbor(uint32_t i, int q, uint32_t v) const { u(kNeighbor0 + q, i) = v; }
};

struct FrameCtx {
  Mat34 L;  // local_T_global
  Mat34 G;  // global_T_local
  float fx, fy, cx, cy;
  Unproj up;
  float inv_depth_scaling;
  float sensor_noise_factor;
  float cos_normal_compat;
  float rf2;
  float max_conf;
  int window;      // surfel_integration_active_window_size
  int reg_window;  // regularization_frame_window_size (for the flag table)
  uint32_t frame;
  int W, H;
  int stats;       // collect the value-distribution counters (single-address atomics: off when timing)
  unsigned long long* ts;   // this call's stage-stamp record (GetTimings; see StageStamps), or null
};

// Stage stamps: what GetTimings reports (cuda_surfel_reconstruction.cc:131-319 brackets seven stages with fourteen event
// records in every Integrate call).  An event record is a packet of its own on the stream, and on streams that are never
// idle fourteen of them cost a third of the frame rate (profiles/r28_stage_timing_C2.txt).  Here the kernels stamp the
// device's constant-rate wall clock (s_memrealtime) into a small per-call record, and only a handful of workgroups per
// launch do (a first version let every workgroup's exit raise an "end" word by a fire-and-forget 64-bit atomic max:
// ~10 000 per frame, each behind the 1 - 2 us an s_memrealtime takes to return -- 14 % of the frame rate,
// profiles/r5a_bench_line.json):
//   * a stage BEGINS with the plain store of the first workgroup dispatched (blockIdx 0);
//   * a stage that is followed by another launch on the same in-order stream ENDS with that launch's begin stamp;
//   * where nothing follows on the stream (blend -> hand-over to the internal stream; the last regulariser kernel), and as
//     the fallback where the follower is switched off, the end is the maximum over the LAST workgroups dispatched
//     (kTsTail of them: workgroups are dispatched in order, the last ones in are the last ones out to within a
//     microsecond or two) -- a few dozen atomics per frame.
// Records live in a ring in device memory (kTsRing calls); the first kernel of a call (k_cull_segments) resets the
// call's record and writes its sequence number.  For the non-waiting read the tile kernel's first workgroup copies the
// record of the call before the previous one -- complete by stream order when that kernel runs -- into page-locked host
// memory (with a check word over the record: a reader that meets a record in the middle of being rewritten drops it).
enum : int { kTsSeq = 0, kTsCullBegin, kTsTilesEnd, kTsBlendBegin, kTsBlendEnd, kTsIntBegin, kTsIntEnd, kTsUpdBegin, kTsUpdEnd,
             kTsRegBegin, kTsRegEnd,
             kTsScanBegin, kTsTilesBegin, kTsAccBegin, kTsStepBegin,   // (not part of GetTimings: smx_recon_debug_stamp_ring)
             kTsSeqTail = 15, kTsWords = 16 };
constexpr int kTsRing = 8;
constexpr uint32_t kTsTail = 32;
constexpr unsigned long long kTsCheckSalt = 0x5EED5EED5EED5EEDull;
__device__ __forceinline__ void ts_begin(unsigned long long* ts, int k) {
  if (ts && blockIdx.x == 0 && threadIdx.x == 0) ts[k] = wall_clock64();
}
// (block of n_blocks: the workgroup's place in the dispatch order of the part of the launch that does this work)
__device__ __forceinline__ void ts_end(unsigned long long* ts, int k, uint32_t block, uint32_t n_blocks) {
  if (ts && threadIdx.x == 0 && block < n_blocks && block + kTsTail >= n_blocks) atomicMax(&ts[k], (unsigned long long)wall_clock64());
}

struct Scratch {
  uint32_t* supporting;
  uint32_t* counts;
  long long* depth_sums;
  uint32_t* confl_key;
  float* first_depth;
};

struct Proj { Vec3 g, l; float u, v; int px, py; };

// IsSurfelActiveForIntegration, kernels.cu:77-87
__device__ __forceinline__ bool is_active(uint32_t stamp, uint32_t frame, int window) {
  const int bound = (int)(frame - (uint32_t)window);
  return (int)stamp > bound;
}

// The one projection routine (kernels.cu:1481-1500 == 1722-1741 == 2018-2031 == 1023-1048).
__device__ __forceinline__ bool project_pos(const Vec3& g, const FrameCtx& c, Proj& o) {
  o.g = g;
  o.l = mul(c.L, g);
  if (!(o.l.z > 0)) return false;
  o.u = c.fx * (o.l.x / o.l.z) + c.cx;
  o.v = c.fy * (o.l.y / o.l.z) + c.cy;
  if (!(o.u >= 0 && o.v >= 0 && o.u < (float)c.W && o.v < (float)c.H)) return false;
  o.px = (int)o.u; o.py = (int)o.v;
  return true;
}
// Division-free conservative pre-test for the all-slot scan (93 % of the slots are out of view): a slot that
// fails it is at least one pixel outside the image, whatever the two correctly rounded divisions of
// project_pos would round to; a slot that passes is decided by project_pos itself, so results are unchanged.
__device__ __forceinline__ bool maybe_in_image(const Vec3& g, const FrameCtx& c) {
  const Vec3 l = mul(c.L, g);
  if (!(l.z > 0)) return false;
  const float un = c.fx * l.x + c.cx * l.z, vn = c.fy * l.y + c.cy * l.z;  // u * z, v * z
  return un >= -2.0f * l.z && vn >= -2.0f * l.z && un <= ((float)c.W + 2.0f) * l.z && vn <= ((float)c.H + 2.0f) * l.z;
}
// Can any point of the axis-aligned box [lo, hi] pass maybe_in_image?  The five conditions of that test are affine
// in the point, so a condition violated (with a margin far above float rounding) at all eight corners is violated in
// the whole box.  Returns true when the box is certainly out of view.
__device__ __forceinline__ bool box_out_of_view(const Vec3& lo, const Vec3& hi, const FrameCtx& c) {
  float worst[5] = {-3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f};  // max over the corners of each condition
  float scale = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const Vec3 g = {(k & 1) ? hi.x : lo.x, (k & 2) ? hi.y : lo.y, (k & 4) ? hi.z : lo.z};
    const Vec3 l = mul(c.L, g);
    const float un = c.fx * l.x + c.cx * l.z, vn = c.fy * l.y + c.cy * l.z;
    const float f[5] = {l.z, un + 2.0f * l.z, ((float)c.W + 2.0f) * l.z - un, vn + 2.0f * l.z, ((float)c.H + 2.0f) * l.z - vn};
#pragma unroll
    for (int q = 0; q < 5; ++q) worst[q] = fmaxf(worst[q], f[q]);
    scale = fmaxf(scale, fabsf(l.x) + fabsf(l.y) + fabsf(l.z));
  }
  const float margin = 1.0e-3f * (c.fx + c.fy + (float)c.W + (float)c.H + 4.0f) * (scale + 1.0f);
  return worst[0] < -margin || worst[1] < -margin || worst[2] < -margin || worst[3] < -margin || worst[4] < -margin;
}
__device__ __forceinline__ bool project(const Surfels& S, uint32_t i, const FrameCtx& c, Proj& o) {
  Vec3 g = {S.f(kX, i), S.f(kY, i), S.f(kZ, i)};
  return project_pos(g, c, o);
}

// "triangle quadrant" neighbour pixel, kernels.cu:1078-1120 == 1506-1549 == 1752-1795
__device__ __forceinline__ bool quadrant(const Proj& p, const FrameCtx& c, int& ox, int& oy) {
  const float xf = p.u - (float)p.px, yf = p.v - (float)p.py;
  if (xf < yf) {
    if (xf < 1 - yf) { if (p.px > 1) { ox = p.px - 1; oy = p.py; return true; } return false; }
    else { if (p.py < c.H - 1) { ox = p.px; oy = p.py + 1; return true; } return false; }
  } else {
    if (xf < 1 - yf) { if (p.py > 0) { ox = p.px; oy = p.py - 1; return true; } return false; }
    else { if (p.px < c.W - 1) { ox = p.px + 1; oy = p.py; return true; } return false; }
  }
}

__device__ __forceinline__ float meas_normal_z(float nx, float ny) {
  const float t = 1 - nx * nx - ny * ny;
  return -sqrtf(t > 0.f ? t : 0.f);
}

constexpr int kBlock = 256;
static_assert(kBlock / 64 == 4, "pass A sums four wavefronts' marks");
static_assert(sizeof(uchar3) == 3, "uchar3 must be packed like the reference's Vec3u8");

// Work lists are SEGMENTED: the slots [s*kSeg, (s+1)*kSeg) are scanned by one workgroup, which writes the
// indices it selects to list[s*kSeg ...] (ascending) and their number to seg[s].  No global counter, no
// atomics, and the list order is deterministic.
constexpr int kSeg = 1024;
// Pass B uses larger segments (kSegB slots, one 1024-thread workgroup each): the in-segment regulariser sums
// live in 128 KB of LDS, and the larger the segment the fewer edges leave it (image-row neighbours are a few
// hundred slots apart), i.e. the fewer 64-bit global atomics remain.
constexpr int kSegB = 1024;
constexpr int kBlockB = kSegB / 4;
// The non-empty 256-entry chunks of a segmented list, appended by the pass that builds the list: descriptor = chunk id
// (segment * chunks-per-segment + sub-chunk) | (entries - 1) << 24.  The list kernels walk these instead of probing every
// chunk of every segment: on the frame's binding cycle each dependent memory round trip costs microseconds, and a probe
// that finds an empty chunk is one (profiles/r04_critical_cycle_notes.md).
// The descriptors go to kSubLists INTERLEAVED sub-lists, each with a counter in a cache line of its own: walk step w is
// entry w / kSubLists of sub-list w % kSubLists.  Rounds 1-3 had ONE list behind ONE counter: a returning atomic per
// non-empty segment, 1 300 per launch of pass A at C2 and 5 400 at C3, all arriving within the same few microseconds at
// the end of their workgroups' chains -- and atomics on one address retire at ~12 ns each, so the tail of the launch was
// that queue.  Sixteen counters take a sixteenth each; the producers deal their segments round-robin, so the sub-lists
// stay equally long up to the spread of the descriptors per segment, and a walk that covers kSubLists x the longest one
// meets few empty steps.  A consumer still finds its first descriptor without waiting for any counter (its place does
// not depend on them), which is what a prefix over the sub-lists would have cost.
constexpr uint32_t kCountStride = 32;     // one counter per 128-byte line: atomics on one line serialise, whatever word they hit
constexpr uint32_t kSubLists = 16;
constexpr uint32_t kAccCount = 16;        // acc_chunks.count = rec_chunks.count + kAccCount: the second half of the same lines
struct Chunks { uint32_t* desc; uint32_t* count; uint32_t stride; };   // desc[k * stride + j], count[k * kCountStride]
struct Lists {
  Chunks vis_chunks, rec_chunks;   // (vis_chunks: only the kSubLists counters = the lengths of the visible list's runs)
  Chunks acc_chunks;      // segments the edge kernel has work in (descriptor = segment); its counters share rec_chunks' lines (kAccCount)
  uint32_t* vis_list;     // slots that project into the image this frame: kSubLists dense runs, run k at k * vis_region (vis_*)
  uint32_t vis_region;    // entries per run: every kSubLists-th surviving segment's slots at most
  float* seg_box;         // per pass-A segment: min xyz, max xyz, covered slot count (u32), newest stamp (u32)
  uint32_t* vis_seg;
  uint8_t* seg_streak;      // per pass-A segment: in how many calls in a row pass A has culled it (k_scan_visible, step 2)
  uint32_t* recent_list;  // slots whose last update stamp lies inside the regulariser window
  uint32_t* recent_seg;
  uint32_t* act_list;     // per pass-B segment: the slots the edge kernel has work for (ActEntry), ascending
  uint8_t* flags8;        // per slot: bit 0 = stamp inside the regulariser window, bit 1 = detach request
  uint8_t* dirty8;        // delta tracking (null = off): 1 = a transferred attribute of the slot changed since the
                          // last smx_recon_transfer_changed_to_cpu
  // "Hot" groups (pass B's filter for its far flag gathers, see k_neighbor_scan): per group of slots the number
  // (mod 256) of the last Integrate call in which the group held a slot inside the regulariser window (pass A) or a
  // flag byte / link record of it was written.  `epoch` = this call's number.  (At C2 a group is 2048 slots, at C3 8192.)
  uint8_t* seg_act;       // per pass-A segment: 1 = pass A found visible or recently updated slots in it this frame
  uint32_t descending;    // != 0: this call's all-slot kernels walk the segments downwards (segment_of_block)
  uint8_t* hot_epoch;
  uint16_t* seg_targets;  // per pass-B segment: bitmap of the groups its links point into (see k_neighbor_scan)
  uint32_t n_hot_groups;
  uint32_t epoch;
  int hot_shift;          // slots per group = 1 << hot_shift: the smallest power of two >= 1024 that keeps the table <= 4 KB
};
constexpr int kMaxHotGroups = 4096;
// hot = active in this call or the previous one (or, seen from the pass B that runs beside it, in the next one)
__device__ __forceinline__ bool group_is_hot(uint32_t last, uint32_t epoch) { return ((epoch - last + 1u) & 255u) <= 2u; }

// Far-term bins of the regulariser (k_reg_accumulate -> k_reg_step): one bin per destination segment of kSegB slots.
// A gradient term whose target lies outside the sender's segment is APPENDED to the target segment's bin -- record =
// (target's position in its segment | sender class << 10, gx, gy, gz in 2^-22 fixed point) -- and the segment's
// workgroup of k_reg_step sums its bin in LDS.  The reference pushes these terms with four float atomics each
// (kernels.cu:2176-2182); rounds 1-2 used exclusive "inbox" slots for symmetric links and two packed 64-bit atomics for
// the rest (0.43 M device-scope atomics and 0.39 M random 16-byte stores per frame at C2, and 64 bytes of inbox per
// recent slot read back by the step).  Appending costs one returning atomic per (sender workgroup, destination segment)
// -- 55 k per frame -- and the records of such a pair are adjacent.
struct FarBins {
  uint4* rec;            // [segments][cap]
  uint32_t* count;       // [segments * kCountStride]: word 0 = terms appended since the bin was last consumed (may exceed cap),
                         // word 1 = "some terms for this segment went to grad_acc instead"; both zeroed by the consumer
  uint32_t cap;
  uint32_t hash_mask;    // size - 1 of the sender's LDS table of destinations (kFarHash - 1; tests shrink it)
};
constexpr uint32_t kFarBinCap = 4096;   // records per bin (64 KB per 1024 slots: what the inbox took)
constexpr int kFarHash = 1024;          // destinations one sender workgroup can address through the bins

// Phase stamps inside a kernel (builds with -DSMX_STAMPS only; tools/stamps.py): lane 0 of every workgroup stores the
// shader clock at marked points, the host averages the differences.
#ifdef SMX_STAMPS
#define SMX_STAMP(buf, k) do { if ((buf) && threadIdx.x == 0) (buf)[(size_t)blockIdx.x * 16 + (k)] = __builtin_readcyclecounter(); } while (0)
#else
#define SMX_STAMP(buf, k) do { } while (0)
#endif

// ---- association tiles ------------------------------------------------------------------------------------------
// The per-pixel association state of a frame (z-buffer minimum, supporting surfel, count, depth sum, conflict key) is
// built per image TILE in LDS by k_assoc_tiles instead of with device-scope atomics on five dense images: pass A
// appends one (slot, pixel) PAIR per pixel a visible slot touches (<= 2) to the bin of the pixel's tile; the tile's
// workgroup reduces its pairs with LDS atomics (all of the reductions are order-independent and exact: min, integer
// add) and writes the five images with plain stores.  Nothing is cleared per frame.
constexpr int kTileW = 32, kTileH = 8, kTilePx = kTileW * kTileH;   // one 256-lane workgroup per tile
static_assert(kTilePx == kBlock, "the tile kernel runs choose_segment_direction in one of its workgroups");
// pair code: bits 0-7 pixel inside the tile (row-major, kTileW wide); bit 8: the slot's second ("quadrant") pixel;
// bit 9: the slot is active for integration (kernels.cu:77-87) -- inactive visible slots still take part in the merge phase
constexpr uint32_t kPairSecond = 1u << 8, kPairActive = 1u << 9;
constexpr uint32_t kTileBinCap = 8192;   // pairs per bin (64 KB): 32 per pixel before a tile spills to the overflow list
struct TileBins {
  uint2* pairs;          // [n_tiles][cap]: (slot, code)
  uint32_t* count;       // [n_tiles * kCountStride]: pairs appended in this call (zeroed again by the tile's workgroup); may exceed cap,
  uint4* ovf;            // in which case the rest went to this list: (tile, slot, code, -), room for 2 pairs per slot
  uint32_t* ovf_count;   // this call's overflow counter (two in alternation: the tile kernel zeroes the next call's)
  uint32_t cap;
  int tiles_x;
  uint32_t n_tiles;
};

// Pass A appends its pairs (<= 8 per lane: 4 slots x 2 pixels; key = tile << 10 | code, kNoPair = none) with ONE
// returning device-scope atomic per (workgroup, tile): every pair takes a rank from an LDS counter of its tile (the
// table is open-addressed, keyed by tile number: kPairHash entries), the pair that drew rank 0 reserves the workgroup's run in the tile's bin, and
// after a barrier every pair is stored at base + rank.  The cost does not depend on how many tiles a workgroup touches
// (a first version agreed on one tile after the other by wave ballots: fine on average, 100 us for the wavefronts whose
// 512 pairs fell into a hundred tiles -- profiles/r08_pass_a_append.md).
constexpr uint32_t kNoPair = 0xFFFFFFFFu;
constexpr int kPairHash = 2048;   // entries of pass A's per-workgroup table of tiles (a workgroup has at most 2048 pairs)
__device__ __forceinline__ void pair_store(const TileBins& tb, uint32_t key, uint32_t slot, uint32_t pos) {
  const uint32_t tile = key >> 10, code = key & 1023u;
  if (pos < tb.cap) tb.pairs[(size_t)tile * tb.cap + pos] = make_uint2(slot, code);
  else tb.ovf[atomicAdd(tb.ovf_count, 1u)] = make_uint4(tile, slot, code, 0u);
}

// Work-list entry of the edge kernel (k_reg_accumulate), written by pass B: the slot's position in its segment, which of
// its four links end inside the regulariser window, and whether the slot itself lies inside it.
__device__ __forceinline__ uint32_t act_entry(uint32_t rel, uint32_t window_mask, uint32_t recent) {
  return rel | (window_mask << 10) | ((recent ? 1u : 0u) << 14);
}
constexpr uint32_t kNoActEntry = 0xFFFFFFFFu;
// (one descriptor per segment: the segment number [| (entries - 1) << 22])
__device__ __forceinline__ void emit_segment(const Chunks& ch, uint32_t segment) {
  const uint32_t k = segment % kSubLists;
  ch.desc[(size_t)k * ch.stride + atomicAdd(&ch.count[k * kCountStride], 1u)] = segment;
}

__device__ __forceinline__ bool stamp_outside_window(uint32_t stamp, uint32_t frame, int window) {
  return (int)stamp < (int)(frame - (uint32_t)window);  // kernels.cu:2132
}
__device__ __forceinline__ uint8_t make_flags(uint32_t stamp, uint32_t color, uint32_t frame, int reg_window) {
  return (uint8_t)(((color >> 24) == 1u ? 2u : 0u) | (stamp_outside_window(stamp, frame, reg_window) ? 0u : 1u));
}

// Exclusive prefix sum of one value per thread over a workgroup of kWaves wavefronts (wave64 shuffles + LDS).
template <int kWaves = kBlock / 64>
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t mine, uint32_t* wave_tot /* LDS [kWaves] */, uint32_t& total) {
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t incl = mine;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t t = __shfl_up(incl, off);
    if (lane >= (uint32_t)off) incl += t;
  }
  if (lane == 63) wave_tot[wave] = incl;
  __syncthreads();
  uint32_t wave_off = 0;
  total = 0;
#pragma unroll
  for (int w = 0; w < kWaves; ++w) {
    if ((uint32_t)w < wave) wave_off += wave_tot[w];
    total += wave_tot[w];
  }
  return wave_off + incl - mine;
}

// The same for counts that are zero in most wavefronts (pass B's list ranks): a wavefront without any skips the shuffles.
template <int kWaves = kBlock / 64>
__device__ __forceinline__ uint32_t block_excl_scan_sparse(uint32_t mine, uint32_t* wave_tot /* LDS [kWaves] */, uint32_t& total) {
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t incl = mine;
  if (__ballot(mine != 0) != 0ull) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t t = __shfl_up(incl, off);
      if (lane >= (uint32_t)off) incl += t;
    }
  }
  if (lane == 63) wave_tot[wave] = incl;
  __syncthreads();
  uint32_t wave_off = 0;
  total = 0;
#pragma unroll
  for (int w = 0; w < kWaves; ++w) {
    if ((uint32_t)w < wave) wave_off += wave_tot[w];
    total += wave_tot[w];
  }
  return wave_off + incl - mine;
}

// Segment of a workgroup in the all-slot kernels (pass A, pass B, the edge kernel): workgroups are dispatched in
// blockIdx order, and a launch ends with the latency chains of whatever comes last.  Most segments are trivial in a
// frame (culled, or nothing inside the regulariser's window: the workgroup returns after a test); the work sits in a
// few hundred segments that are dense with visible / recently updated slots -- at the START of the slot range while
// the camera revisits old parts of the map, at its END while it explores.  When that dense block is dispatched FIRST
// it fills the chip, the thousands of trivial workgroups queue behind it and only churn once a heavy round has
// finished; dispatched LAST, the trivial ones are gone by the time it needs the room (edge kernel alone 50 -> 41 us,
// frame +9 %; profiles/r09_dispatch_order.md has the variants, including explicit permutations, which were worse).
// So the launches walk the segments in the direction that ends in the dense block: pass A marks the segments it found
// active, one workgroup of the tile kernel counts the marks in the lower and the upper half of the used range and
// leaves the direction in a word of page-locked HOST memory; the host passes whatever it finds there as a kernel argument
// of the next launches (no synchronisation: the value may be a frame or two old, and any value is correct -- it only
// decides in which order the same workgroups run).
__device__ __forceinline__ uint32_t segment_of_block(uint32_t descending) {
  return descending ? gridDim.x - 1u - blockIdx.x : blockIdx.x;
}
// (one workgroup of kBlock threads; act[s] != 0: active; n_used = segments that hold slots)
__device__ __forceinline__ void choose_segment_direction(const uint8_t* __restrict__ act, uint32_t n_used, uint32_t* __restrict__ descending,
                                                         uint32_t* wave_tot /* LDS [kBlock / 64] */) {
  int balance = 0;   // active segments in the lower half - in the upper half
  const uint32_t half = n_used / 2;
  for (uint32_t k = threadIdx.x; k < n_used; k += kBlock)   // (coalesced, independent loads)
    if (act[k]) balance += k < half ? 1 : -1;
  uint32_t total;
  (void)block_excl_scan((uint32_t)(balance + 4096), wave_tot, total);   // (sum of the biased values: |balance| < 4096 per lane)
  if (threadIdx.x == 0) *descending = ((int)total - 4096 * kBlock) > 0 ? 1u : 0u;
}

// "These loads travel together": pins loaded values at this point of the program, so that every load issued above is in
// flight before the first of them is waited for.  Without it the compiler sinks independent loads below the branches
// that follow (it cannot know that a dependent round trip costs microseconds on the frame's latency chain and a wasted
// 16-byte load nothing): tools/isa_phases.py counts the waits per kernel.
__device__ __forceinline__ void keep(uint32_t v) { asm volatile("" :: "v"(v)); }
__device__ __forceinline__ void keep(int v) { asm volatile("" :: "v"(v)); }
__device__ __forceinline__ void keep(float v) { asm volatile("" :: "v"(v)); }
__device__ __forceinline__ void keep(const float2& v) { keep(v.x); keep(v.y); }
__device__ __forceinline__ void keep(const float4& v) { keep(v.x); keep(v.y); keep(v.z); keep(v.w); }
__device__ __forceinline__ void keep(const uint4& v) { keep(v.x); keep(v.y); keep(v.z); keep(v.w); }

// Which groups of loads are pinned together by keep() (bit mask, A/B: profiles/r6_ab_notes.md section 12): 1 = the walk's first
// descriptor with its counters, 2 = k_integrate's merge mark + records and its pixel loads, 4 = the update kernel's records and
// pixel loads, 8 = the edge kernel's bin reservations, 16 = pass B's first four loads, 32 = the step kernel's bin state + list
// entries, 64 = pass A's records + flag bytes + next list entry.
#ifndef SMX_RT_MASK
#define SMX_RT_MASK 57
#endif
template <int kBit, class T>
__device__ __forceinline__ void keep_if(const T& v) { if (SMX_RT_MASK & kBit) keep(v); }

// Wave priority of the surfel kernels (s_setprio 0 .. 3; 0 = the default every wavefront starts with).  The instruction arbiter
// of a SIMD serves the highest priority first and the OLDEST wavefront among equals -- and the oldest wavefront beside these
// short kernels is the bilateral filter's, which lives for the whole of its tile loop (A/B: profiles/r6_ab_notes.md section 13).
#ifndef SMX_WAVE_PRIO
#define SMX_WAVE_PRIO 1
#endif
#define SMX_SET_WAVE_PRIO() do { if (SMX_WAVE_PRIO) __builtin_amdgcn_s_setprio(SMX_WAVE_PRIO); } while (0)
// Stores of a launch's streaming OUTPUTS -- records no lane of the same launch reads again.  A plain store leaves its line
// dirty in the XCD's L2, and the launch boundary behind the kernel has to write all of them back before the next launch of
// the stream may start (MI355X_MICROARCH.md, price table row "boundary": 1.45 - 1.9 us + dirty bytes / 6 TB/s; the edge
// kernel leaves 30 MB, the frame 79 MB).  Flavours (tools/boundary.hip measures them behind writers of 0 - 64 MB; the
// in-frame A/Bs are in profiles/r6_ab_notes.md): 0 = plain, 1 = nt (non-temporal: kept in the L2, marked for early
// eviction), 2 = sc1 (write-through: leaves the L2 while the kernel still runs, the line is dropped), 3 = sc0 sc1.
// One switch per store site so that each can be judged on its own:
#ifndef SMX_ST_FARBIN
#define SMX_ST_FARBIN 0    // k_reg_accumulate: far-term records into the destination segments' bins (sparse 16-byte stores)
#endif
#ifndef SMX_ST_REGREC
#define SMX_ST_REGREC 0    // k_reg_accumulate: the recent slots' dense records (own term; in-segment sums)
#endif
#ifndef SMX_ST_STEP
#define SMX_ST_STEP 0      // k_reg_step: the new smooth positions (scattered 16-byte records)
#endif
#ifndef SMX_ST_INT
#define SMX_ST_INT 0       // k_integrate: the P / N / C records of the slots it changed
#endif
#ifndef SMX_ST_IMG
#define SMX_ST_IMG 0       // k_assoc_tiles: the five association images
#endif
typedef uint32_t v4u_t __attribute__((ext_vector_type(4)));
template <int kFlavour>
__device__ __forceinline__ void out_store16(void* p, const v4u_t& w) {
  if (kFlavour == 1) __builtin_nontemporal_store(w, reinterpret_cast<v4u_t*>(p));
  else if (kFlavour == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(w) : "memory");
  else if (kFlavour == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(p), "v"(w) : "memory");
  else *reinterpret_cast<v4u_t*>(p) = w;
}
template <int kFlavour>
__device__ __forceinline__ void out_store16(void* p, const uint4& v) { const v4u_t w = {v.x, v.y, v.z, v.w}; out_store16<kFlavour>(p, w); }
template <int kFlavour>
__device__ __forceinline__ void out_store16(void* p, const float4& v) {
  const v4u_t w = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
  out_store16<kFlavour>(p, w);
}
template <int kFlavour>
__device__ __forceinline__ void out_store16(void* p, unsigned long long a, unsigned long long b) {
  const v4u_t w = {(uint32_t)a, (uint32_t)(a >> 32), (uint32_t)b, (uint32_t)(b >> 32)};
  out_store16<kFlavour>(p, w);
}

// ---------------------------------------------------------------------------------------------
// The reference's 5 clears (cuda_surfel_reconstruction.cc:134-138) have no counterpart: k_assoc_tiles writes every pixel
// of the association images with plain stores.  The maps of the multi-launch blend fallback (kernels.cc:165-166):
struct BlendBufs {
  uint8_t* distance_map; uint8_t* new_distance_map; float* deltas; float* new_deltas;
};

// (value-distribution counters: only while statistics are on.  Two steps: the cull step's counter in front of it, the
// others behind the stream's wait for the previous call's second half, whose kernels may still be adding to them)
__global__ void k_reset_frame_stats(DevState* st, int cull_only) {
  if (cull_only) { st->n_segments_skipped = 0; return; }
  st->n_visible = 0; st->n_merged = 0; st->n_integrated = 0; st->n_replaced = 0; st->n_conflict_hits = 0;
  st->n_pairs = 0; st->n_overflow_pairs = 0; st->max_tile_pairs = 0;
}

// ---------------------------------------------------------------------------------------------
// Pass A.  The all-slot half of RenderMinDepthCUDAKernel (kernels.cu:1466-1557): which slots project into the image and
// onto which pixels -- fused with the construction of the visible list and the refresh of the "recent" bit of the flag
// table.  Rows 18,0,1,2 are streamed with 16-byte lane loads (4 slots per lane).  The z-buffer minimum itself (:1463) is
// formed by k_assoc_tiles from the pairs appended here.
//
// Two launches.  k_cull_segments (a lane per segment, a few dozen workgroups) decides which segments have to be read
// at all and compacts them into a SURVIVOR LIST; k_scan_visible, on a chip-sized grid, walks that list, so that the
// three quarters of the segments that are out of view cost neither a workgroup nor a dispatch slot.  (Rounds 1-3: one
// workgroup per segment -- 5 400 at C2, 21 600 at C3, most of which tested a box, copied 1 KB of flag bytes and left,
// each a dependent chain in a slot the segments with work were waiting for.  A first version of this round gave every
// workgroup of a chip-sized grid a strided share of ALL segments and let it cull its own: one launch, but the segments
// with work then queue up behind each other inside the workgroups that happen to own several -- 25 -> 31 us alone at C2,
// profiles/r17_ab_notes.md.)
//
// Segment culling: a segment's box was formed from every slot of it the last time it was read; it is still valid if
// the segment cannot have grown (it was full) and had no visible slot in the previous frame (only visible slots are
// moved, restamped, merged or replaced).  If it is out of view and its newest stamp has left the regulariser window,
// this frame's result for the segment is known without reading its 16 KB of P records: nothing visible, no recent bit.
// The decision needs the new pose and what the PREVIOUS pass A left -- not the slot count (a partial segment is simply
// never culled) and nothing integrate / update / create of the previous call write -- so the launch can run beside those:
// smx_recon_integrate enqueues it in front of the caller's stream's wait for the previous call's map wherever it may.
// The culled segments' flag bytes (recent bit off, detach bit carried over) are copied by k_scan_visible (the copy of the
// table it writes may still be read by the previous call's regulariser when the cull runs) -- and not at all for a segment
// that was already culled in the two previous calls: the two copies of the flag table alternate by call, nothing but
// pass A writes flag bytes of slots that are not visible, so the table written two calls ago already holds this call's
// bytes (seg_streak).
struct SegWork {
  uint32_t* surv_list;   // segments pass A has to read in this call (any order)
  uint32_t* copy_list;   // culled segments whose flag bytes have to be copied
  uint32_t* count;       // [0] survivors, [1] copies; zeroed for the next call by k_assoc_tiles
};
__global__ void __launch_bounds__(kBlock)
k_cull_segments(FrameCtx c, Lists L, SegWork sw, DevState* st, uint32_t nseg_alloc, uint32_t max_new_slots, unsigned long long ts_seq) {
  SMX_SET_WAVE_PRIO();
  if (c.ts && blockIdx.x == 0 && threadIdx.x == 0) {   // (the call's stage-stamp record: reset, sequence number, first stamp)
#pragma unroll
    for (int k = 2; k < kTsWords; ++k) c.ts[k] = 0;
    c.ts[kTsSeq] = ts_seq;
    c.ts[kTsCullBegin] = wall_clock64();
  }
  const uint32_t seg = blockIdx.x * kBlock + threadIdx.x;
  const bool have_seg = seg < nseg_alloc;
  const uint32_t sidx = have_seg ? seg : 0u;   // (segment 0 stands in: no branch around the loads)
  const float4 b0 = *reinterpret_cast<const float4*>(&L.seg_box[8 * (size_t)sidx]);       // lo.xyz, hi.x
  const float4 b1 = *reinterpret_cast<const float4*>(&L.seg_box[8 * (size_t)sidx + 4]);   // hi.yz, covered slots, newest stamp
  const uint32_t vis_prev = L.vis_seg[sidx];
  const uint32_t streak = L.seg_streak[sidx];
  // (the previous call's creation kernel may be advancing the count while this runs: whichever value is read, the count
  // pass A will see is at most max_new_slots -- one per pixel -- above it; k_scan_visible drops entries beyond the end)
  const uint32_t n_bound = st->surfel_count + max_new_slots;
  const bool mine = have_seg && (unsigned long long)seg * kSeg < (unsigned long long)n_bound;
  bool skip = false;
  if (mine && __float_as_uint(b1.z) == (uint32_t)kSeg && vis_prev == 0 &&
      stamp_outside_window(__float_as_uint(b1.w), c.frame, c.reg_window)) {
    const Vec3 lo = {b0.x, b0.y, b0.z}, hi = {b0.w, b1.x, b1.y};
    skip = box_out_of_view(lo, hi, c);
  }
  const bool survive = mine && !skip, copy = skip && streak < 2u;
  if (skip) {
    L.seg_act[seg] = 0;   // (vis_seg stays 0, the box stays as it is)
    if (streak < 255u) L.seg_streak[seg] = (uint8_t)(streak + 1u);
  } else if (mine && streak) {
    L.seg_streak[seg] = 0;
  }
  // compaction: ballot + popcount ranks inside the wavefront, wavefront offsets through LDS, ONE returning atomic per
  // workgroup and list (two dozen per launch at C2)
  __shared__ uint32_t wave_n[2][kBlock / 64], list_base[2];
  const unsigned long long sm = __ballot(survive), cm = __ballot(copy);
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  if (lane == 0) { wave_n[0][wave] = (uint32_t)__popcll(sm); wave_n[1][wave] = (uint32_t)__popcll(cm); }
  __syncthreads();
  if (threadIdx.x < 2) {
    uint32_t total = 0;
#pragma unroll
    for (int w = 0; w < kBlock / 64; ++w) total += wave_n[threadIdx.x][w];
    list_base[threadIdx.x] = total ? atomicAdd(&sw.count[threadIdx.x], total) : 0u;
  }
  __syncthreads();
  uint32_t sbase = list_base[0], cbase = list_base[1];
#pragma unroll
  for (int w = 0; w < kBlock / 64; ++w)
    if ((uint32_t)w < wave) { sbase += wave_n[0][w]; cbase += wave_n[1][w]; }
  const unsigned long long below = (1ull << lane) - 1ull;
  if (survive) sw.surv_list[sbase + (uint32_t)__popcll(sm & below)] = seg;
  if (copy) sw.copy_list[cbase + (uint32_t)__popcll(cm & below)] = seg;
  if (c.stats) {
    const unsigned long long km = __ballot(skip);
    if (lane == 0 && km) atomicAdd(&st->n_segments_skipped, (uint32_t)__popcll(km));
  }
}

__global__ void __launch_bounds__(kBlock, 8)   // (eight workgroups per CU: <= 64 VGPRs)
k_scan_visible(Surfels S, FrameCtx c, Lists L, TileBins tb, SegWork sw, const uint8_t* __restrict__ flags_prev, DevState* st,
               int use_lds_tables) {
  SMX_SET_WAVE_PRIO();
  ts_begin(c.ts, kTsScanBegin);
  __shared__ uint32_t wave_tot[kBlock / 64];
  __shared__ float box_part[kBlock / 64][8];
  // the tiles this workgroup's pairs fall into: open-addressed table keyed by tile number (fixed size: a direct-mapped
  // one -- a word per tile, 38 KB at 1280 x 960 -- capped the chip at four workgroups per CU); pair_key = tile, later the
  // base of the workgroup's run
  __shared__ uint32_t pair_key[kPairHash], pair_cnt[kPairHash];
  __shared__ uint32_t vis_run_pos;
  const bool lds_tables = use_lds_tables != 0;
  const uint32_t G = gridDim.x, wg = blockIdx.x;
  const uint32_t n_surv = sw.count[0], n_copy = sw.count[1];
  uint32_t next_seg = sw.surv_list[wg];   // (the list has room for any index formed here; requested with the counts)
  const uint32_t N = st->surfel_count;
  // ---- the culled segments' flag bytes (one list entry per walk step, all 256 lanes: 1 KB)
  for (uint32_t e = wg; e < n_copy; e += G) {
    const uint32_t i0 = sw.copy_list[e] * (uint32_t)kSeg + threadIdx.x * 4;
    if (i0 < N) {
      const uchar4 of = *reinterpret_cast<const uchar4*>(&flags_prev[i0]);
      *reinterpret_cast<uchar4*>(&L.flags8[i0]) = make_uchar4(of.x & 2u, of.y & 2u, of.z & 2u, of.w & 2u);
    }
  }
  // ---- the surviving segments: a walk over the list, the next entry requested while this one is worked on
  bool first_segment = true;
#pragma unroll 1
  for (uint32_t e = wg; e < n_surv; e += G) {
    {
      const uint32_t seg_id = next_seg;
      // (everything below is per segment: a lane number the optimiser cannot see through keeps it from hoisting the
      // per-lane addresses of a dozen arrays out of the loop and holding them in registers across it -- 96 VGPRs instead
      // of 51, five workgroups per CU instead of eight)
      uint32_t tid = threadIdx.x;
      asm volatile("" : "+v"(tid));
      const uint32_t base = seg_id * (uint32_t)kSeg;
      const uint32_t i0 = base + tid * 4;
      // four consecutive P records (X, Y, Z, stamp) = 64 contiguous bytes per lane, requested first; the group arrays are
      // padded to a multiple of 64 slots, so the loads stay inside the array
      // (no branch around the loads -- a lane beyond the end reads slot 0 and uses nothing of it -- so that nothing has to be
      // merged, i.e. waited for, before the next request goes out)
      const uint32_t il = i0 < N ? i0 : 0u;
      const float4* P = S.group(kGroupP, il);
      const float4 p0 = P[0], p1 = P[1], p2 = P[2], p3 = P[3];
      const uint32_t of_word = *reinterpret_cast<const uint32_t*>(&flags_prev[il]);  // (detach bits carry over)
      next_seg = sw.surv_list[e + G < n_surv ? e + G : e];   // (no branch around the load: this entry again when it is the last)
      // (the records, the old flag bytes and the next list entry: ONE round trip, pinned -- the compiler had asked for the
      // next entry and waited, for the flag bytes and waited, and only then for the records: tools/isa_phases.py)
      keep_if<64>(p0); keep_if<64>(p1); keep_if<64>(p2); keep_if<64>(p3); keep_if<64>(of_word); keep_if<64>(next_seg);
      if (base >= N) continue;   // (the cull step's bound on the slot count was generous)
      const uchar4 of = make_uchar4((uint8_t)(of_word & 255u), (uint8_t)((of_word >> 8) & 255u), (uint8_t)((of_word >> 16) & 255u), (uint8_t)(of_word >> 24));
      const uint32_t in_seg = (N - base < (uint32_t)kSeg) ? N - base : (uint32_t)kSeg;
      float* box = &L.seg_box[8 * (size_t)seg_id];
      if (!first_segment) __syncthreads();   // (the previous segment's readers of the tables and partials are done)
      first_segment = false;
      if (lds_tables) {
#pragma unroll
        for (int k = 0; k < kPairHash / kBlock; ++k) { pair_key[k * kBlock + tid] = kInvalid; pair_cnt[k * kBlock + tid] = 0; }
        __syncthreads();   // (visible before the ranks are drawn)
      }
      uint32_t vis_bits = 0;
      bool lane_recent = false;
      float bmin[3] = {3.0e38f, 3.0e38f, 3.0e38f}, bmax[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
      uint32_t newest = 0;   // in the wrap-around order of stamp_outside_window: compared as signed
      bool have_stamp = false;
      // the pairs of the lane's four slots: entry 2 j = slot j's own pixel, 2 j + 1 = its quadrant pixel
      uint32_t key[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) key[j] = kNoPair;
      if (i0 < N) {
        const uint32_t stamps[4] = {__float_as_uint(p0.w), __float_as_uint(p1.w), __float_as_uint(p2.w), __float_as_uint(p3.w)};
        const float xs[4] = {p0.x, p1.x, p2.x, p3.x};
        const float ys[4] = {p0.y, p1.y, p2.y, p3.y};
        const float zs[4] = {p0.z, p1.z, p2.z, p3.z};
        const uint8_t old_flags[4] = {of.x, of.y, of.z, of.w};
        uint8_t new_flags[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const uint32_t i = i0 + j;
          new_flags[j] = (uint8_t)((old_flags[j] & 2u) | (stamp_outside_window(stamps[j], c.frame, c.reg_window) ? 0u : 1u));
          Proj p;
          const Vec3 g = {xs[j], ys[j], zs[j]};
          if (i < N) {
            bmin[0] = fminf(bmin[0], g.x); bmin[1] = fminf(bmin[1], g.y); bmin[2] = fminf(bmin[2], g.z);
            bmax[0] = fmaxf(bmax[0], g.x); bmax[1] = fmaxf(bmax[1], g.y); bmax[2] = fmaxf(bmax[2], g.z);
            if (!have_stamp || (int)stamps[j] > (int)newest) { newest = stamps[j]; have_stamp = true; }
          }
          if (i < N && maybe_in_image(g, c) && project_pos(g, c, p)) {
            vis_bits |= 1u << j;
            const bool active = is_active(stamps[j], c.frame, c.window);
            if (active && c.stats) atomicAdd(&st->n_visible, 1u);
            const uint32_t act = active ? kPairActive : 0u;
            key[2 * j] = (((uint32_t)(p.py / kTileH) * (uint32_t)tb.tiles_x + (uint32_t)(p.px / kTileW)) << 10) |
                         (uint32_t)((p.py % kTileH) * kTileW + (p.px % kTileW)) | act;
            int ox, oy;
            if (active && quadrant(p, c, ox, oy))   // (:1506-1549: the second z-buffer pixel, active slots only)
              key[2 * j + 1] = (((uint32_t)(oy / kTileH) * (uint32_t)tb.tiles_x + (uint32_t)(ox / kTileW)) << 10) |
                               (uint32_t)((oy % kTileH) * kTileW + (ox % kTileW)) | act | kPairSecond;
          }
        }
        *reinterpret_cast<uchar4*>(&L.flags8[i0]) = make_uchar4(new_flags[0], new_flags[1], new_flags[2], new_flags[3]);
        lane_recent = ((new_flags[0] | new_flags[1] | new_flags[2] | new_flags[3]) & 1u) != 0;
      }
      // ranks inside the workgroup's run of each tile (LDS atomics; the barrier inside the scan below completes them)
      // (rank = table entry << 16 | rank in the entry; kInvalid: no entry found within 16 probes -- that pair reserves its
      // place in the bin on its own)
      uint32_t rank[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        rank[j] = kInvalid;
        if (key[j] != kNoPair && lds_tables) {
          const uint32_t tile = key[j] >> 10;
          uint32_t h = (tile * 2654435761u) >> 16;
          for (int probe = 0; probe < 16; ++probe) {
            h &= (uint32_t)kPairHash - 1u;
            const uint32_t seen = atomicCAS(&pair_key[h], kInvalid, tile);
            if (seen == kInvalid || seen == tile) { rank[j] = (h << 16) | atomicAdd(&pair_cnt[h], 1u); break; }
            ++h;
          }
        }
      }
      // (a slot inside the regulariser window: the group is hot -- same value from every writer, one byte store per wavefront)
      const bool wave_recent = __ballot(lane_recent) != 0;
      if (wave_recent && (tid & 63) == 0) L.hot_epoch[base >> L.hot_shift] = (uint8_t)L.epoch;
      // the segment's bounding box and newest stamp (wave shuffles, then one partial per wavefront through LDS)
      int newest_s = have_stamp ? (int)newest : (int)0x80000000;
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          bmin[k] = fminf(bmin[k], __shfl_xor(bmin[k], off));
          bmax[k] = fmaxf(bmax[k], __shfl_xor(bmax[k], off));
        }
        const int o = __shfl_xor(newest_s, off);
        newest_s = o > newest_s ? o : newest_s;
      }
      if ((tid & 63) == 0) {
        float* bp = box_part[tid >> 6];
        bp[0] = bmin[0]; bp[1] = bmin[1]; bp[2] = bmin[2]; bp[3] = bmax[0]; bp[4] = bmax[1]; bp[5] = bmax[2];
        bp[6] = __int_as_float(newest_s);
        bp[7] = wave_recent ? 1.0f : 0.0f;
      }
      uint32_t total;
      uint32_t off = block_excl_scan((uint32_t)__popc(vis_bits), wave_tot, total);  // (synchronises)
      // the segment's place in its run of the visible list (vis_*): one returning atomic, in flight with the tile reservations below
      uint32_t run_pos = 0;
      if (tid == 0 && total) run_pos = atomicAdd(&L.vis_chunks.count[(e % kSubLists) * kCountStride], total);
      // the pair that drew rank 0 reserves the run of its (workgroup, tile); all reservations of the workgroup are in
      // flight together (the tile number goes through an opaque VGPR: with a visibly uniform address the compiler's atomic
      // optimizer wraps the operation in a wave reduction + readfirstlane, i.e. a wait for each result in turn)
      if (lds_tables) {
        uint32_t got[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          got[j] = 0;
          if (key[j] != kNoPair && (rank[j] & 0xFFFFu) == 0 && rank[j] != kInvalid) {
            uint32_t tv = key[j] >> 10;
            asm volatile("" : "+v"(tv));
            got[j] = atomicAdd(&tb.count[tv * kCountStride], pair_cnt[rank[j] >> 16]);
          }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j)
          if (key[j] != kNoPair && (rank[j] & 0xFFFFu) == 0 && rank[j] != kInvalid) pair_key[rank[j] >> 16] = got[j];   // (every probe is done: the barrier above)
      }
      if (tid == 0) vis_run_pos = run_pos;
      __syncthreads();
      off += (e % kSubLists) * L.vis_region + vis_run_pos;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (vis_bits & (1u << j)) L.vis_list[off++] = i0 + j;
      if (lds_tables) {
#pragma unroll
        for (int j = 0; j < 8; ++j)
          if (key[j] != kNoPair)
            pair_store(tb, key[j], i0 + (uint32_t)(j >> 1),
                       rank[j] != kInvalid ? pair_key[rank[j] >> 16] + (rank[j] & 0xFFFFu) : atomicAdd(&tb.count[(key[j] >> 10) * kCountStride], 1u));
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j)
          if (key[j] != kNoPair) pair_store(tb, key[j], i0 + (uint32_t)(j >> 1), atomicAdd(&tb.count[(key[j] >> 10) * kCountStride], 1u));
      }
      if (tid == 0) {
        L.vis_seg[seg_id] = total;
        L.seg_act[seg_id] = (total != 0 || box_part[0][7] + box_part[1][7] + box_part[2][7] + box_part[3][7] != 0.0f) ? 1 : 0;
        float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
        int ns = (int)0x80000000;
        for (int w = 0; w < kBlock / 64; ++w) {
          for (int k = 0; k < 3; ++k) { lo[k] = fminf(lo[k], box_part[w][k]); hi[k] = fmaxf(hi[k], box_part[w][3 + k]); }
          const int o = __float_as_int(box_part[w][6]);
          ns = o > ns ? o : ns;
        }
        box[0] = lo[0]; box[1] = lo[1]; box[2] = lo[2]; box[3] = hi[0]; box[4] = hi[1]; box[5] = hi[2];
        box[6] = __uint_as_float(in_seg);
        box[7] = __int_as_float(ns);
      }
    }
  }
}

// List kernels walk the list's chunk descriptors (one entry per lane), grid-striding so that the visible slots --
// which cluster in a few segments -- still spread over the whole chip.  In the A/B "scan mode" every slot is visited
// instead, in chunks of kBlock slots.
// number of walk steps; the first descriptor is requested together with the counters (the sub-lists are long enough for
// any index a walk can form).  cntv: lane l holds the length of sub-list l % kSubLists.
template <bool kUseList>
__device__ __forceinline__ uint32_t walk_begin(const Chunks& ch, uint32_t n_slots_scan, uint32_t first, uint32_t& desc, uint32_t& cntv) {
  if (kUseList) {
    desc = ch.desc[(size_t)(first % kSubLists) * ch.stride + first / kSubLists];
    cntv = ch.count[(threadIdx.x % kSubLists) * kCountStride];
    // (pinned: left to itself the compiler moves the descriptor's load behind the walk's first test -- `no step for this
    // workgroup' -- and the first step then begins a round trip late: tools/isa_phases.py)
    keep_if<1>(desc); keep_if<1>(cntv);
    uint32_t longest = cntv;
#pragma unroll
    for (uint32_t off = kSubLists / 2; off > 0; off >>= 1) longest = max(longest, (uint32_t)__shfl_xor((int)longest, (int)off));
    return __builtin_amdgcn_readfirstlane(longest) * kSubLists;
  }
  desc = 0; cntv = 0;
  return (n_slots_scan + kBlock - 1) / kBlock;
}
template <bool kUseList>
__device__ __forceinline__ uint32_t walk_next(const Chunks& ch, uint32_t c, uint32_t n_steps) {
  return (kUseList && c < n_steps) ? ch.desc[(size_t)(c % kSubLists) * ch.stride + c / kSubLists] : 0u;
}
// (is walk step c an entry of its sub-list, or beyond that sub-list's end?)
__device__ __forceinline__ bool walk_step_valid(uint32_t c, uint32_t cntv) {
  const uint32_t k = __builtin_amdgcn_readfirstlane(c % kSubLists);
  return c / kSubLists < (uint32_t)__builtin_amdgcn_readlane((int)cntv, (int)k);
}
// The VISIBLE list is dense: pass A appends a segment's visible slots to one of kSubLists runs (its counter = the run's length;
// no chunk descriptors), walk step c = lanes' entries (c / kSubLists) * kBlock .. of run c % kSubLists.  Rounds 3 - 6 listed them
// per segment in chunks of <= kBlock (a descriptor each): 57 % of the lanes of a step had an entry, and a workgroup found its slot
// numbers two dependent round trips in (descriptor, then entry).  Here the first entries are requested together with the run
// lengths (their address depends on neither), and a step is full except for the last one of each run.
template <bool kUseList>
__device__ __forceinline__ uint32_t vis_load(const Lists& L, uint32_t c) {
  if (!kUseList) return 0u;
  const uint32_t e = min((c / kSubLists) * (uint32_t)kBlock + threadIdx.x, L.vis_region - 1u);   // (in bounds whatever c: formed before the lengths are known)
  return L.vis_list[(size_t)(c % kSubLists) * L.vis_region + e];
}
template <bool kUseList>
__device__ __forceinline__ uint32_t vis_begin(const Lists& L, uint32_t n_slots_scan, uint32_t first, uint32_t& ent, uint32_t& cntv) {
  if (kUseList) {
    ent = vis_load<true>(L, first);
    cntv = L.vis_chunks.count[(threadIdx.x % kSubLists) * kCountStride];
    keep_if<1>(ent); keep_if<1>(cntv);
    uint32_t longest = (cntv + (uint32_t)kBlock - 1u) / (uint32_t)kBlock;
#pragma unroll
    for (uint32_t off = kSubLists / 2; off > 0; off >>= 1) longest = max(longest, (uint32_t)__shfl_xor((int)longest, (int)off));
    return __builtin_amdgcn_readfirstlane(longest) * kSubLists;
  }
  ent = 0; cntv = 0;
  return (n_slots_scan + kBlock - 1) / kBlock;
}
template <bool kUseList>
__device__ __forceinline__ bool vis_entry(uint32_t ent, uint32_t c, uint32_t cntv, uint32_t n_slots_scan, uint32_t lane, uint32_t& i) {
  if (kUseList) {
    const uint32_t k = __builtin_amdgcn_readfirstlane(c % kSubLists);
    const uint32_t len = (uint32_t)__builtin_amdgcn_readlane((int)cntv, (int)k);
    i = ent;
    return (c / kSubLists) * (uint32_t)kBlock + lane < len;
  }
  i = c * kBlock + lane;
  return i < n_slots_scan;
}

// ---------------------------------------------------------------------------------------------
// k_assoc_tiles: the per-pixel halves of RenderMinDepthCUDAKernel (kernels.cu:1463), AssociateSurfelsCUDAKernel /
// ConsiderSurfelAssociationToPixel (:1586-1808) and MergeSurfelsCUDAKernel / ConsiderSurfelMergeAtPixel (:1857-2052),
// one workgroup per image tile over the pairs pass A appended to the tile's bin.  Three phases separated by workgroup
// barriers, as the reference separates them by kernel launches: z-buffer minimum; association (supporting surfel =
// lowest qualifying index, count, exact fixed-point depth sum, associate-phase conflict key); merge decisions (read the
// final supporting surfel; merge-phase conflict key).  All reductions are LDS atomics on order-independent exact
// operations, so the result does not depend on the order of the pairs in the bin.  Merge decisions have snapshot
// semantics: this kernel only records them (merge_flag); the marks (:1987-1989) are applied at the top of k_integrate.
struct PairRec { uint32_t i, code; Vec3 g, l, gn; float r2; };
__device__ __forceinline__ PairRec fetch_pair(const Surfels& S, const FrameCtx& c, uint32_t i, uint32_t code) {
  const float4 p4 = *S.group(kGroupP, i), n4 = *S.group(kGroupN, i);   // both records in flight together
  PairRec r;
  r.i = i; r.code = code;
  r.g = Vec3{p4.x, p4.y, p4.z};
  r.l = mul(c.L, r.g);   // == project_pos(...).l: the z every kernel compares with first_depth
  r.gn = Vec3{n4.x, n4.y, n4.z};
  r.r2 = n4.w;
  return r;
}
struct TileLds {
  int zmin[kTilePx];                  // float bits of the smallest camera-space z (positive floats order like ints)
  uint32_t sup[kTilePx], cnt[kTilePx], confl[kTilePx];
  unsigned long long sum[kTilePx];
  float2 nrm[kTilePx];
  uint16_t depth[kTilePx];
};
// ConsiderSurfelAssociationToPixel, :1586-1695
__device__ __forceinline__ void assoc_stage(TileLds& t, const FrameCtx& c, const PairRec& r) {
  if (!(r.code & kPairActive)) return;
  const uint32_t k = r.code & 255u;
  const float measurement_depth = c.inv_depth_scaling * (float)t.depth[k];
  if (measurement_depth <= 0) return;
  const float first = __int_as_float(t.zmin[k]);
  if (first < (1 - c.sensor_noise_factor) * measurement_depth) {
    // :1615 racing plain store -> deterministic: associate-phase writers carry class bit 31
    if (first == r.l.z) atomicMin(&t.confl[k], 0x80000000u | r.i);
    return;
  }
  const float occlusion_depth = (1 + c.sensor_noise_factor) * measurement_depth;
  if (r.l.z > occlusion_depth) return;
  const float surfel_distance = sqrtf(r.l.x * r.l.x + r.l.y * r.l.y + r.l.z * r.l.z);
  const Vec3 ln = rotate(c.L, r.gn);
  const float dot_angle = (1.0f / surfel_distance) * (r.l.x * ln.x + r.l.y * ln.y + r.l.z * ln.z);
  if (dot_angle > 0) return;
  if (measurement_depth < r.l.z) {
    const float2 n = t.nrm[k];
    const float nz = meas_normal_z(n.x, n.y);
    const float d = ln.x * n.x + ln.y * n.y + ln.z * nz;
    if (d < c.cos_normal_compat) return;
  }
  if (r.r2 <= 0) return;  // :1674
  atomicMin(&t.sup[k], r.i);      // :1688 first-wins CAS -> lowest index
  atomicAdd(&t.cnt[k], 1u);
  atomicAdd(&t.sum[k], (unsigned long long)q_from_float(r.l.z));   // :1694, exact in 2^-32 fixed point
}
// ConsiderSurfelMergeAtPixel, :1857-1990 (the slot's own pixel only; any visible slot with r^2 >= 0, :2017)
// First half: the pixel-side tests; returns the supporting surfel the slot has to be compared with (kInvalid: none).
__device__ __forceinline__ uint32_t merge_candidate(TileLds& t, const FrameCtx& c, const PairRec& r) {
  if (r.code & kPairSecond) return kInvalid;
  if (!(r.r2 >= 0)) return kInvalid;  // :2017
  const uint32_t k = r.code & 255u;
  const float measurement_depth = c.inv_depth_scaling * (float)t.depth[k];
  if (measurement_depth <= 0) return kInvalid;
  const float first = __int_as_float(t.zmin[k]);
  if (first < (1 - c.sensor_noise_factor) * measurement_depth) {
    // :1887 plain store issued after the associate kernel: merge-phase writers win (class 0)
    if (first == r.l.z) atomicMin(&t.confl[k], r.i);
    return kInvalid;
  }
  const float occlusion_depth = (1 + c.sensor_noise_factor) * measurement_depth;
  if (r.l.z > occlusion_depth) return kInvalid;
  const float surfel_distance = sqrtf(r.l.x * r.l.x + r.l.y * r.l.y + r.l.z * r.l.z);
  const Vec3 ln = rotate(c.L, r.gn);
  const float dot_angle = (1.0f / surfel_distance) * (r.l.x * ln.x + r.l.y * ln.y + r.l.z * ln.z);
  if (dot_angle > 0) return kInvalid;
  if (measurement_depth < r.l.z) {
    const float2 n = t.nrm[k];
    const float nz = meas_normal_z(n.x, n.y);
    const float d = ln.x * n.x + ln.y * n.y + ln.z * nz;
    if (d < c.cos_normal_compat) return kInvalid;
  }
  const uint32_t s = t.sup[k];
  if (s == r.i || s == kInvalid) return kInvalid;  // :1950-1953
  return s;
}
// ... second half: the supporting surfel's two records against the slot's, :1955-1990
__device__ __forceinline__ void merge_decide(const PairRec& r, const float4& sp4, const float4& sn4, uint8_t* __restrict__ merge_flag) {
  const float other_r2 = sn4.w;
  const float radius_diff = r.r2 / other_r2;
  const float kT = 1.2f * 1.2f;
  if (radius_diff > kT || radius_diff < 1 / kT) return;
  const float dx = r.g.x - sp4.x, dy = r.g.y - sp4.y, dz = r.g.z - sp4.z;
  const float d2 = dx * dx + dy * dy + dz * dz;
  const float kDist = 0.5f * (0.25f * 0.25f);
  if (d2 > kDist * (r.r2 + other_r2)) return;
  const float dot_angle = r.gn.x * sn4.x + r.gn.y * sn4.y + r.gn.z * sn4.z;
  if (dot_angle < 0.93969f) return;
  merge_flag[r.i] = 1;
}
__device__ __forceinline__ void merge_stage(TileLds& t, const Surfels& S, const FrameCtx& c, const PairRec& r,
                                            uint8_t* __restrict__ merge_flag) {
  const uint32_t s = merge_candidate(t, c, r);
  if (s == kInvalid) return;
  const float4 sp4 = *S.group(kGroupP, s), sn4 = *S.group(kGroupN, s);  // the supported surfel's two records
  keep(sp4); keep(sn4);
  merge_decide(r, sp4, sn4, merge_flag);
}

constexpr int kPairCache = 4;   // pairs per lane kept in registers across the three phases (a tile holds ~600 at C2)
template <class F>
__device__ __forceinline__ void for_uncached_pairs(const Surfels& S, const FrameCtx& c, const TileBins& tb, uint32_t tile,
                                                   uint32_t n_binned, bool overflowed, uint32_t n_ovf, F f) {
  for (uint32_t k = kPairCache * kTilePx + threadIdx.x; k < n_binned; k += kTilePx) {
    const uint2 pr = tb.pairs[(size_t)tile * tb.cap + k];
    f(fetch_pair(S, c, pr.x, pr.y));
  }
  if (overflowed)   // (a bin that ran full: the rest of its pairs is somewhere in the overflow list)
    for (uint32_t e = threadIdx.x; e < n_ovf; e += kTilePx) {
      const uint4 o = tb.ovf[e];
      if (o.x == tile) f(fetch_pair(S, c, o.y, o.z));
    }
}

__global__ void __launch_bounds__(kTilePx)
k_assoc_tiles(Surfels S, FrameCtx c, Scratch sc, Img<const uint16_t> depth, Img<const float2> normals, TileBins tb,
              uint32_t* __restrict__ next_ovf_count, uint8_t* __restrict__ merge_flag, DevState* st,
              const uint8_t* __restrict__ seg_act, uint32_t nseg, uint32_t* __restrict__ direction_out, uint32_t* __restrict__ seg_work_count,
              unsigned long long* stamps, const unsigned long long* __restrict__ ts_ring, volatile unsigned long long* ts_host,
              unsigned long long ts_seq) {
  SMX_SET_WAVE_PRIO();
  __shared__ TileLds t;
  __shared__ uint32_t order_wave_tot[kTilePx / 64];
  ts_begin(c.ts, kTsTilesBegin);
  // side job of the first workgroup's second wavefront: the stage-stamp record of the call before the previous one (complete:
  // that call's regulariser preceded the previous call's integration, whose end this stream has waited for) goes to
  // page-locked host memory for the non-waiting GetTimings
  if (blockIdx.x == 0 && threadIdx.x == 64 && ts_host && ts_seq > 2) {
    const unsigned long long q = ts_seq - 2;
    const unsigned long long* src = ts_ring + (size_t)(q % kTsRing) * kTsWords;
    volatile unsigned long long* dst = ts_host + (size_t)(q % kTsRing) * kTsWords;
    unsigned long long w[kTsWords];
#pragma unroll
    for (int k = 0; k < kTsWords; ++k) w[k] = src[k];
    if (w[kTsSeq] == q) {
      // (no fences: a system-scope fence writes back the L2 of the XCD under every kernel that runs on it.  The record
      // carries a check word instead -- sequence number x all data words -- and a reader that meets a torn record drops it.)
      unsigned long long check = q ^ kTsCheckSalt;
#pragma unroll
      for (int k = 1; k < kTsSeqTail; ++k) check ^= w[k];
      dst[kTsSeq] = q;
#pragma unroll
      for (int k = 1; k < kTsSeqTail; ++k) dst[k] = w[k];
      dst[kTsSeqTail] = check;
    }
  }
  // side job of the first workgroup: the direction in which the launches that follow walk the segments (segment_of_block)
  if (blockIdx.x == 0) {
    const uint32_t n_used = min(nseg, (st->surfel_count + (uint32_t)kSeg - 1u) / (uint32_t)kSeg);
    choose_segment_direction(seg_act, n_used, direction_out, order_wave_tot);
    __syncthreads();
  }
  SMX_STAMP(stamps, 0);
  const uint32_t tile = blockIdx.x, lane = threadIdx.x;
  const int x = (int)(tile % (uint32_t)tb.tiles_x) * kTileW + (int)(lane % kTileW);
  const int y = (int)(tile / (uint32_t)tb.tiles_x) * kTileH + (int)(lane / kTileW);
  const bool in_image = x < c.W && y < c.H;
  const uint32_t n_total = tb.count[tile * kCountStride];
  const uint32_t n_ovf = *tb.ovf_count;
  const uint32_t n_binned = n_total < tb.cap ? n_total : tb.cap;
  const bool overflowed = n_total > tb.cap;
  // the first pairs of every lane, their surfel records and the tile's measurements: all requested up front
  uint2 pr[kPairCache];
  bool have[kPairCache];
#pragma unroll
  for (int q = 0; q < kPairCache; ++q) {
    const uint32_t k = q * kTilePx + lane;
    have[q] = k < n_binned;
    pr[q] = tb.pairs[(size_t)tile * tb.cap + (have[q] ? k : 0u)];   // (no branch around the load: they all travel together)
  }
#pragma unroll
  for (int q = 0; q < kPairCache; ++q) { keep(pr[q].x); keep(pr[q].y); if (!have[q]) pr[q] = make_uint2(0u, 0u); }
  const uint16_t own_depth = in_image ? depth(y, x) : (uint16_t)0;
  const float2 own_normal = in_image ? normals(y, x) : make_float2(0.f, 0.f);
  PairRec rec[kPairCache];
#pragma unroll
  for (int q = 0; q < kPairCache; ++q) rec[q] = fetch_pair(S, c, pr[q].x, pr[q].y);   // (slot 0 stands in for "no pair")
  t.zmin[lane] = 0x7F800000;   // +inf
  t.sup[lane] = kInvalid; t.cnt[lane] = 0; t.confl[lane] = kInvalid; t.sum[lane] = 0;
  t.depth[lane] = own_depth; t.nrm[lane] = own_normal;
  __syncthreads();
  SMX_STAMP(stamps, 1);
  // phase 1: z-buffer minimum over the active pairs (:1463)
  auto zmin_stage = [&](const PairRec& r) { if (r.code & kPairActive) atomicMin(&t.zmin[r.code & 255u], __float_as_int(r.l.z)); };
#pragma unroll
  for (int q = 0; q < kPairCache; ++q) if (have[q]) zmin_stage(rec[q]);
  for_uncached_pairs(S, c, tb, tile, n_binned, overflowed, n_ovf, zmin_stage);
  __syncthreads();
  SMX_STAMP(stamps, 2);
  // (every lane has read the counters: ready for the next call's pass A)
  if (lane == 0) {
    tb.count[tile * kCountStride] = 0;
    if (tile == 0) { *next_ovf_count = 0; seg_work_count[0] = 0; seg_work_count[1] = 0; }   // (pass A of this call has read them: ready for the next call's cull step)
    if (c.stats) {
      atomicAdd(&st->n_pairs, n_total);
      atomicMax(&st->max_tile_pairs, n_total);
      if (tile == 0) st->n_overflow_pairs = n_ovf;
    }
  }
  // phase 2: association
#pragma unroll
  for (int q = 0; q < kPairCache; ++q) if (have[q]) assoc_stage(t, c, rec[q]);
  for_uncached_pairs(S, c, tb, tile, n_binned, overflowed, n_ovf, [&](const PairRec& r) { assoc_stage(t, c, r); });
  __syncthreads();
  SMX_STAMP(stamps, 3);
  // phase 3: merge decisions (the supported surfels' records of all cached pairs are gathered together)
  uint32_t cand[kPairCache];
  float4 cp4[kPairCache], cn4[kPairCache];
#pragma unroll
  for (int q = 0; q < kPairCache; ++q) cand[q] = have[q] ? merge_candidate(t, c, rec[q]) : kInvalid;
#pragma unroll
  for (int q = 0; q < kPairCache; ++q) {
    const uint32_t g = cand[q] == kInvalid ? 0u : cand[q];   // (slot 0 stands in)
    cp4[q] = *S.group(kGroupP, g); cn4[q] = *S.group(kGroupN, g);
  }
#pragma unroll
  for (int q = 0; q < kPairCache; ++q) { keep(cp4[q]); keep(cn4[q]); }
#pragma unroll
  for (int q = 0; q < kPairCache; ++q) if (cand[q] != kInvalid) merge_decide(rec[q], cp4[q], cn4[q], merge_flag);
  for_uncached_pairs(S, c, tb, tile, n_binned, overflowed, n_ovf, [&](const PairRec& r) { merge_stage(t, S, c, r, merge_flag); });
  __syncthreads();
  SMX_STAMP(stamps, 4);
  if (in_image) {
    const size_t k = (size_t)y * c.W + x;
#if SMX_ST_IMG   // (4- and 8-byte stores, full lines per wavefront row: non-temporal only -- a scalar sc1 store is a fabric write of its own)
    __builtin_nontemporal_store(__int_as_float(t.zmin[lane]), &sc.first_depth[k]);
    __builtin_nontemporal_store(t.sup[lane], &sc.supporting[k]);
    __builtin_nontemporal_store(t.cnt[lane], &sc.counts[k]);
    __builtin_nontemporal_store((long long)t.sum[lane], &sc.depth_sums[k]);
    __builtin_nontemporal_store(t.confl[lane], &sc.confl_key[k]);
#else
    sc.first_depth[k] = __int_as_float(t.zmin[lane]);
    sc.supporting[k] = t.sup[lane];
    sc.counts[k] = t.cnt[lane];
    sc.depth_sums[k] = (long long)t.sum[lane];
    sc.confl_key[k] = t.confl[lane];
#endif
  }
  SMX_STAMP(stamps, 5);
  ts_end(c.ts, kTsTilesEnd, blockIdx.x, gridDim.x);
}

// ---------------------------------------------------------------------------------------------
// BlendMeasurementsCUDA, kernels.cc:148-205 + kernels.cu:563-708.
__device__ __forceinline__ float depth_sum_avg(const Scratch& sc, size_t k) {
  return q_to_float(sc.depth_sums[k]) / (float)sc.counts[k];
}

__global__ void __launch_bounds__(kBlock)
k_blend_start(float ds, Img<uint16_t> depth, Scratch sc, BlendBufs b, int W, int H, unsigned long long* ts) {
  // (multi-launch fallback: the stage ends with the last workgroups STARTED -- within a microsecond of the end)
  ts_end(ts, kTsBlendEnd, blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);
  const int x = blockIdx.x * 64 + (threadIdx.x & 63);
  const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (!(x >= 1 && y >= 1 && x < W - 1 && y < H - 1)) return;
  const size_t k = (size_t)y * W + x;
  const uint16_t own = depth(y, x);
  if (own == 0 || sc.supporting[k] == kInvalid) return;
  bool measurement_border = false, surfel_border = false;
  for (int wy = y - 1; wy <= y + 1; ++wy)
    for (int wx = x - 1; wx <= x + 1; ++wx) {
      if (depth(wy, wx) == 0) measurement_border = true;
      else if (sc.supporting[(size_t)wy * W + wx] == kInvalid) surfel_border = true;
    }
  if (surfel_border) {
    b.new_distance_map[k] = 1;
    const float avg = depth_sum_avg(sc, k);
    b.new_deltas[k] = avg - (float)own / ds;
  }
  if (measurement_border) {
    b.distance_map[k] = 1;
    const float avg = depth_sum_avg(sc, k);
    b.deltas[k] = avg - (float)own / ds;
    depth(y, x) = f2u16(ds * avg + 0.5f);  // :610
  } else {
    b.distance_map[k] = 255;
  }
}

__global__ void __launch_bounds__(kBlock)
k_blend_iter(int it, float term, float ds, Img<uint16_t> depth, Scratch sc, BlendBufs b, int W, int H, unsigned long long* ts) {
  ts_end(ts, kTsBlendEnd, blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);
  const int x = blockIdx.x * 64 + (threadIdx.x & 63);
  const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (!(x >= 1 && y >= 1 && x < W - 1 && y < H - 1)) return;
  const size_t k = (size_t)y * W + x;
  if (b.distance_map[k] == 255) {
    float delta_sum = 0; int count = 0;
    for (int wy = y - 1; wy <= y + 1; ++wy)
      for (int wx = x - 1; wx <= x + 1; ++wx) {
        const size_t kk = (size_t)wy * W + wx;
        if (b.distance_map[kk] == it - 1) { delta_sum += b.deltas[kk]; ++count; }
      }
    if (count > 0) {
      b.distance_map[k] = (uint8_t)it;
      const float avg = delta_sum / (float)count;
      b.deltas[k] = avg;
      const float f = (float)(it - 1) * term;
      depth(y, x) = f2u16((float)depth(y, x) + (ds * (1 - f) * avg + 0.5f));  // :681
    }
  }
  if (depth(y, x) != 0 && sc.supporting[k] == kInvalid && b.new_distance_map[k] == 0) {
    float delta_sum = 0; int count = 0;
    for (int wy = y - 1; wy <= y + 1; ++wy)
      for (int wx = x - 1; wx <= x + 1; ++wx) {
        const size_t kk = (size_t)wy * W + wx;
        if (b.new_distance_map[kk] == it - 1) { delta_sum += b.new_deltas[kk]; ++count; }
      }
    if (count > 0) {
      b.new_distance_map[k] = (uint8_t)it;
      const float avg = delta_sum / (float)count;
      b.new_deltas[k] = avg;
      const float f = (float)(it - 1) * term;
      depth(y, x) = f2u16((float)depth(y, x) + (ds * (1 - f) * avg + 0.5f));  // :704
    }
  }
}

// Fused form of the whole BlendMeasurementsCUDA sequence (start kernel + radius-2 iteration kernels,
// 13 launches in the reference incl. the clears): one launch, one workgroup per 32x32 pixel tile.  A ring of
// BFS distance d depends only on pixels within d+1 of it, so a halo of radius-1 pixels makes the tile
// interior exact; the depths and both delta maps of tile + halo live in LDS and only the interior depths are written back.
//
// WHICH cells a ring assigns does not depend on the delta values: the two distance maps of the reference are breadth-
// first fronts over three per-pixel predicates (no measurement / has a supporting surfel / evaluable).  So the fronts are
// computed first, as BIT MASKS -- one 64-bit word per region row (the region is at most 64 cells wide), ring `it` =
// `unassigned & dilate3x3(ring it - 1)`, three shifts and ORs per row -- by ONE wavefront with a lane per row, which
// needs no barrier at all (the rows above and below come by lane shuffles).  The rings of delta values then follow with
// one workgroup barrier each; a lane finds its cells of the ring in the precomputed masks and only those cells read
// their neighbours' deltas, in the reference's window order.  A ring costs what it changes, not nine LDS reads for
// every cell of the region (profiles/r04g: that ring loop was 47 of the old kernel's 66 us), and the loop ends with the
// last non-empty ring.
// Tile edge: 32 or 40 pixels (template parameter), chosen per launch so that the slowest CU has the least to do: the
// kernel is bound by what ONE CU can load and ring through for its tiles, and 300 tiles of 32 x 32 (640 x 480) leave 44
// CUs with two tiles while the others idle after one; 192 tiles of 40 x 40 -- 32 % more cells each, 15 % fewer in total
// -- give every CU at most one (34.5 -> 27.2 us alone, profiles/r13_ab_notes.md).  The region (tile + 2 halos) must stay <= 64 cells wide.
constexpr int kBlendThreads = 1024;
constexpr int kBlendMaxHalo = 16;    // radius <= 17 uses this kernel (region <= 64 x 64), larger radii the multi-launch path
constexpr int kBlendMaxRings = kBlendMaxHalo + 2;
struct BlendMasks {
  unsigned long long zero[64];      // measured depth == 0 (or outside the image)
  unsigned long long supp[64];      // pixel has a supporting surfel
  unsigned long long elig[64];      // kBorder = 1 rule of both kernels (:576-577, :660-661) and not on the region rim
  // cells whose distance is `it` in the measurement-border map (supported cells) / the surfel-border map; it = 1: the
  // start kernel's border cells
  unsigned long long ring_m[kBlendMaxRings][64];
  unsigned long long ring_n[kBlendMaxRings][64];
  int last_ring;                    // the highest non-empty ring (0: nothing to blend in tile + halo)
};
__device__ __forceinline__ unsigned long long dilate_row(unsigned long long m) { return m | (m << 1) | (m >> 1); }
// lane r gets lane r - 1's / r + 1's value, the first / last lane 0: DPP wave_shr:1 / wave_shl:1, no LDS crossbar trip
// (GFX9-family encodings; the whole file is written for wave64 on gfx950 -- DESIGN.md -- and the build says so)
// What the code needs is wave64, the GFX9 DPP controls and packed fp32 -- gfx90a / gfx942 have all three, and
// -DSMX_ALLOW_OTHER_GFX9 lets such a build through (untuned: every grid, tile and register budget here is gfx950's).
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && \
    !(defined(SMX_ALLOW_OTHER_GFX9) && (defined(__gfx90a__) || defined(__gfx940__) || defined(__gfx941__) || defined(__gfx942__)))
#error "libsmx device code is written for gfx950 (wave64, GFX9 DPP controls, packed fp32): build with --offload-arch=gfx950"
#endif
__device__ __forceinline__ unsigned long long shfl_up64(unsigned long long v) {
  const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)v, 0x138, 0xF, 0xF, false);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(v >> 32), 0x138, 0xF, 0xF, false);
  return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ unsigned long long shfl_down64(unsigned long long v) {
  const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)v, 0x130, 0xF, 0xF, false);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(v >> 32), 0x130, 0xF, 0xF, false);
  return ((unsigned long long)hi << 32) | lo;
}
// The 64-bit mask of the region row a lane belongs to (16 lanes per row: lane = 16 * row-in-wavefront + cg, the lane's
// bit b standing for column cg + 16 b) from four wavefront ballots: no LDS atomics, no cleared words.
__device__ __forceinline__ unsigned long long row_mask_from_bits(uint32_t bits) {
  const int sh = (int)(threadIdx.x & 48u);   // 16 * row-in-wavefront
  unsigned long long m = 0;
#pragma unroll
  for (int b = 0; b < 4; ++b) m |= ((__ballot((bits >> b) & 1u) >> sh) & 0xFFFFull) << (16 * b);
  return m;
}

template <int kBlendTile>
__global__ void __launch_bounds__(kBlendThreads)
k_blend_tiles(int radius, float term, float ds, Img<const uint16_t> depth, Img<uint16_t> out, Scratch sc, int W, int H,
              int tiles_x, unsigned long long* stamps, unsigned long long* ts, uint32_t* gate_count,
              int max_ring /* kBlendMaxRings; less: TIMING ONLY (debug_skip bit 8), the blended depths are wrong */) {
  SMX_SET_WAVE_PRIO();
  extern __shared__ __align__(16) unsigned char blend_lds[];
  ts_begin(ts, kTsBlendBegin);
  SMX_STAMP(stamps, 0);
#ifdef SMX_STAMPS
  if (stamps && threadIdx.x == 0) stamps[(size_t)blockIdx.x * 16 + 6] = wall_clock64();   // (entry, 100 MHz wall clock: comparable across workgroups and kernels)
#endif
  const int halo = radius - 1;
  const int rw = kBlendTile + 2 * halo;          // region width == height (<= 64)
  const int cells = rw * rw;
  BlendMasks& M = *reinterpret_cast<BlendMasks*>(blend_lds);
  float* delta = reinterpret_cast<float*>(blend_lds + sizeof(BlendMasks));
  float* ndelta = delta + cells;
  uint16_t* dep = reinterpret_cast<uint16_t*>(ndelta + cells);
  const int tile_x = (int)(blockIdx.x % (uint32_t)tiles_x), tile_y = (int)(blockIdx.x / (uint32_t)tiles_x);
  const int x0 = tile_x * kBlendTile - halo, y0 = tile_y * kBlendTile - halo;
  // ---- load.  This lane's row and its four columns cg, cg + 16, cg + 32, cg + 48 (16 lanes read 16 consecutive cells):
  // depth and "has a supporting surfel"; and the association sums of the four cells the lane owns in the ring phase
  // (needed where such a cell turns out to be a border cell: fetched now, the start ring then waits for nothing) -- all
  // requested before the first use
  const int r = (int)(threadIdx.x >> 4), cg = (int)(threadIdx.x & 15);
  uint32_t dv[4], sv[4];
  bool inside[4];
  long long bs[4]; uint32_t bc[4];
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    const int cx = cg + 16 * b, x = x0 + cx, y = y0 + r;
    inside[b] = r < rw && cx < rw && x >= 0 && y >= 0 && x < W && y < H;
    const int xc = min(max(x, 0), W - 1), yc = min(max(y, 0), H - 1);
    dv[b] = depth(yc, xc);
    sv[b] = sc.supporting[(size_t)yc * W + xc];
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int R = (r + 16 * j) & 63, C = 4 * cg + j;
    const size_t g = (size_t)min(max(y0 + R, 0), H - 1) * W + min(max(x0 + C, 0), W - 1);
    bs[j] = sc.depth_sums[g]; bc[j] = sc.counts[g];
  }
#pragma unroll
  for (int b = 0; b < 4; ++b) { keep(dv[b]); keep(sv[b]); keep((uint32_t)bs[b]); keep((uint32_t)(bs[b] >> 32)); keep(bc[b]); }
  // The four cells' averages at once -- the division the start ring does for the border cells among them (:596 / :605), same
  // operands, same result -- and PINNED, so that four floats stay live through the mask phases instead of twelve registers of sums
  // and counts.  The kernel's 1 024-lane workgroup needs four wave slots and four times its register allocation free on every SIMD
  // of one CU at once, beside kernels that fill the chip: every 8 registers of allocation are worth ~4.5 % of the FRAME
  // (48 -> 56 -> 64: 6 375 -> 6 075 -> 5 947 frames/s, profiles/r6_ab_notes.md section 27).
  float bavg[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) { bavg[j] = q_to_float(bs[j]) / (float)bc[j]; keep(bavg[j]); }
  SMX_STAMP(stamps, 1);
  uint32_t zb = 0, sb0 = 0, eb = 0;
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    const int cx = cg + 16 * b, x = x0 + cx, y = y0 + r;
    const uint32_t d = inside[b] ? dv[b] : 0u;
    if (r < rw && cx < rw) {
      dep[r * rw + cx] = (uint16_t)d;
      delta[r * rw + cx] = 0; ndelta[r * rw + cx] = 0;
    }
    if (d == 0) zb |= 1u << b;
    if (inside[b] && sv[b] != kInvalid) sb0 |= 1u << b;
    // cells on the region rim cannot be evaluated (their 3x3 window leaves the region) and are not needed
    if (inside[b] && x >= 1 && y >= 1 && x < W - 1 && y < H - 1 && cx >= 1 && r >= 1 && cx < rw - 1 && r < rw - 1) eb |= 1u << b;
  }
  {
    const unsigned long long mz = row_mask_from_bits(zb), ms = row_mask_from_bits(sb0), me = row_mask_from_bits(eb);
    if (cg == 0) { M.zero[r] = mz; M.supp[r] = ms; M.elig[r] = me; }
  }
  __syncthreads();
  SMX_STAMP(stamps, 2);
  // ---- the fronts of both maps, by the first wavefront: lane = region row
  if (threadIdx.x < 64) {
    const int row = (int)threadIdx.x;
    const unsigned long long z = M.zero[row], s = M.supp[row], e = M.elig[row];
    const unsigned long long zu = shfl_up64(z), zd = shfl_down64(z), su = shfl_up64(s), sd = shfl_down64(s);
    // start kernel, :563-615 (the decisions read the unmodified depths: so does the zero mask)
    const unsigned long long cand = e & ~z & s;
    unsigned long long fm = cand & (dilate_row(zu) | dilate_row(z) | dilate_row(zd));                      // a window cell without measurement
    unsigned long long fn = cand & (dilate_row(~zu & ~su) | dilate_row(~z & ~s) | dilate_row(~zd & ~sd));  // ... measured, without surfel
    unsigned long long un_m = cand & ~fm;      // distance 255: supported, not reached yet
    unsigned long long un_n = e & ~z & ~s;     // measured cells without a supporting surfel, not reached yet
    M.ring_m[1][row] = fm; M.ring_n[1][row] = fn;
    int last = __ballot((fm | fn) != 0) != 0 ? 1 : 0;
    // iteration kernels, :647-708.  Ring `it` is only needed (and only exact) up to halo - it pixels outside the
    // tile, so the evaluated square shrinks by one pixel per ring.
    if (last)
      for (int it = 2; it < radius; ++it) {
        const int lo = it, hi = rw - it;
        const unsigned long long cm = (row >= lo && row < hi) ? ((1ull << (hi - lo)) - 1ull) << lo : 0ull;
        const unsigned long long mu = shfl_up64(fm), md = shfl_down64(fm), nu = shfl_up64(fn), nd = shfl_down64(fn);
        fm = un_m & (dilate_row(mu) | dilate_row(fm) | dilate_row(md)) & cm;
        fn = un_n & (dilate_row(nu) | dilate_row(fn) | dilate_row(nd)) & cm;
        un_m &= ~fm; un_n &= ~fn;
        // a ring that assigned nothing leaves no frontier: all later rings are empty too
        if (__ballot((fm | fn) != 0) == 0) break;
        M.ring_m[it][row] = fm; M.ring_n[it][row] = fn;
        last = it;
      }
    if (row == 0) M.last_ring = last;
  }
  __syncthreads();
  SMX_STAMP(stamps, 3);
  const int last_ring = min(M.last_ring, max_ring);
  // no measurement / surfel border anywhere in tile + halo: the blend changes nothing here (dep = the input depths)
  if (last_ring >= 1) {
    // From here on a lane owns the cells (r0 + 16 j, 4 cq + j), j = 0..3: four different rows and columns, so that the
    // cells of one ring -- runs along a border, horizontal or vertical -- spread over as many lanes as possible (a lane
    // walks its cells of a ring one after the other).
    const int r0 = (int)(threadIdx.x >> 4), cq = (int)(threadIdx.x & 15);
    // start kernel: the border cells' deltas from the association sums
    {
      uint32_t hit_m = 0, hit_n = 0;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int R = (r0 + 16 * j) & 63, C = 4 * cq + j;
        hit_m |= (uint32_t)((M.ring_m[1][R] >> C) & 1ull) << j;
        hit_n |= (uint32_t)((M.ring_n[1][R] >> C) & 1ull) << j;
      }
      if (hit_m | hit_n) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (!((hit_m | hit_n) & (1u << j))) continue;
          const int R = (r0 + 16 * j) & 63, C = 4 * cq + j, k = R * rw + C;
          const float own = (float)dep[k];
          const float avg = bavg[j];   // depth_sum_avg (formed above)
          if (hit_n & (1u << j)) ndelta[k] = avg - own / ds;
          if (hit_m & (1u << j)) {
            delta[k] = avg - own / ds;
            dep[k] = f2u16(ds * avg + 0.5f);  // :610
          }
        }
      }
    }
    for (int it = 2; it <= last_ring; ++it) {
      const float f = (float)(it - 1) * term;
      uint32_t hit_m = 0, hit_n = 0;   // (the masks are final: read in front of the barrier, not behind it)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int R = (r0 + 16 * j) & 63, C = 4 * cq + j;
        hit_m |= (uint32_t)((M.ring_m[it][R] >> C) & 1ull) << j;
        hit_n |= (uint32_t)((M.ring_n[it][R] >> C) & 1ull) << j;
      }
      __syncthreads();   // the deltas of ring it - 1 are complete
      // (a cell belongs to one map only -- supported cells to the measurement-border map, unsupported ones to the
      // surfel-border map -- so one code path serves both)
      for (uint32_t todo = hit_m | hit_n; todo; todo &= todo - 1) {
        const int j = __ffs((int)todo) - 1;
        const int R = (r0 + 16 * j) & 63, C = 4 * cq + j, k = R * rw + C;
        const bool second = (hit_n >> j) & 1u;
        const unsigned long long* fr = second ? M.ring_n[it - 1] : M.ring_m[it - 1];
        float* dl = second ? ndelta : delta;
        const unsigned long long fu = fr[R - 1], fc = fr[R], fd = fr[R + 1];
        // the nine neighbours' deltas, read unconditionally (no branch per window cell) and added in the reference's
        // window order where the neighbour belongs to the previous ring
        float w[9];
#pragma unroll
        for (int q = 0; q < 9; ++q) w[q] = dl[k + (q / 3 - 1) * rw + (q % 3 - 1)];
        float delta_sum = 0; int count = 0;
#pragma unroll
        for (int q = 0; q < 9; ++q) {
          const unsigned long long fw = q < 3 ? fu : q < 6 ? fc : fd;
          if ((fw >> (C + q % 3 - 1)) & 1ull) { delta_sum += w[q]; ++count; }
        }
        const float avg = delta_sum / (float)count;   // (count > 0: the cell is in the previous ring's dilation)
        dl[k] = avg;
        dep[k] = f2u16((float)dep[k] + (ds * (1 - f) * avg + 0.5f));  // :681 / :704
      }
    }
    __syncthreads();
  }
  SMX_STAMP(stamps, 4);
  for (int k = threadIdx.x; k < kBlendTile * kBlendTile; k += kBlendThreads) {
    const int ty = k / kBlendTile, tx = k - ty * kBlendTile;
    const int x = tile_x * kBlendTile + tx, y = tile_y * kBlendTile + ty;
    if (x < W && y < H) {
      const uint32_t v = dep[(ty + halo) * rw + (tx + halo)];
      // (device-word hand-over: the blended depths are the only thing this launch leaves for the integration, and they leave
      // WRITE-THROUGH -- device-scope stores, complete when the wavefront's store counter says so -- so that the count below
      // needs no release: a release writes back every dirty line of the XCD's L2, and beside this kernel run the ones that
      // write the most, pass B / edges / step of the previous call)
      if (gate_count) asm volatile("global_store_short %0, %1, off sc1" :: "v"(&out(y, x)), "v"(v) : "memory");
      else out(y, x) = (uint16_t)v;
    }
  }
  SMX_STAMP(stamps, 5);
#ifdef SMX_STAMPS
  if (stamps && threadIdx.x == 0) stamps[(size_t)blockIdx.x * 16 + 7] = wall_clock64();   // (exit)
#endif
  ts_end(ts, kTsBlendEnd, blockIdx.x, gridDim.x);
  // The hand-over to the internal stream by a device word instead of an event (smx_recon_set_handover_mode 1): every
  // workgroup releases its output (device scope) and counts itself; k_front_gate, one wavefront in front of the
  // integration on the internal stream, leaves when the count has reached this call's total.
  // (no acquire anywhere -- an acquire drops the clean lines of the XCD's L2 under every kernel that runs beside this one: a
  // first version fenced in every wavefront and polled with acquire loads, 4 300 instead of 6 150 frames/s -- and no release
  // either, see the stores above)
  if (gate_count) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (this wavefront's write-through stores have been acknowledged)
    __syncthreads();
#if SMX_GATE_RELEASE
    if (threadIdx.x == 0) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
#endif
    if (threadIdx.x == 0) __hip_atomic_fetch_add(gate_count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// One wavefront that leaves when `*count` has reached `expected` (compared as a signed difference: the count wraps).  Safe
// as a wait because of WHERE it is enqueued: behind the launch it waits for (smx_recon_integrate enqueues the front of a
// call before anything on the internal stream), like an event wait -- and one wavefront cannot keep the producer's workgroups
// off the chip, which is why the poll is a launch of its own and not the head of the consumer (profiles/r6_ab_notes.md section 9).
// The poll is BOUNDED (kGateTimeoutTicks of the 100 MHz wall clock = 0.25 s, a thousand frames): where something serialises
// kernel dispatches ACROSS queues -- a profiler collecting hardware counters does (rocprofv3 --pmc) -- the gate can be let onto
// the chip in front of the launch it waits for, and an unbounded poll then hangs the process (it did: a whole evidence run of
// this round).  A gate that gives up leaves a sticky mark (count[8]); the map is then wrong, and every entry point that
// synchronises reports it (check_handover).  smx_recon_create switches to the event hand-over by itself when it finds the
// process under a counter-collecting rocprofv3 (ROCPROF_COUNTER_COLLECTION).
constexpr unsigned long long kGateTimeoutTicks = 25000000ull;
__global__ void __launch_bounds__(64)
k_front_gate(uint32_t* count, uint32_t expected, unsigned long long* dbg, uint32_t* host_mark) {
  if (threadIdx.x == 0) {
    const unsigned long long t_in = wall_clock64();
#ifdef SMX_STAMPS
    unsigned long long polls = 0;
#endif
    // (relaxed: the launch boundary behind this kernel is the acquire)
    while ((int32_t)(__hip_atomic_load(count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - expected) < 0) {
      __builtin_amdgcn_s_sleep(2);
      if (wall_clock64() - t_in > kGateTimeoutTicks) { count[8] = 1u; *host_mark = 1u; break; }
#ifdef SMX_STAMPS
      ++polls;
#endif
    }
#ifdef SMX_STAMPS
    if (dbg) { dbg[0] = t_in; dbg[1] = wall_clock64(); dbg[2] = polls; }
#endif
  }
}

// ---------------------------------------------------------------------------------------------
// CreateNewSurfelsCUDA, kernels.cc:37-146.  The u8 flag kernel (kernels.cu:90-111), the CUB
// exclusive scan (kernels.cu:2506-2520) and the two D2H count reads are replaced by:
//   k_new_flags_scan  flags + block-local exclusive ranks (wave64 ballot/popcount + LDS),
//   k_new_create      scan of the <= a few hundred block totals (redundantly per workgroup; workgroup 0 advances
//                     the device-side count) + the creation kernel (kernels.cu:133-231).
constexpr int kScanPxPerThread = 4;
constexpr int kScanPxPerBlock = kBlock * kScanPxPerThread;

// With copy_back the blended depths (k_merge_and_blend's output) are stored into the caller's depth buffer on the way --
// Integrate mutates its depth argument like the reference (kernels.cu:610, 681, 704) -- and used for the flags.
struct NewFlagsArgs {
  Img<const uint16_t> depth; Img<uint16_t> depth_out; int copy_back;
  uint8_t* flags; uint32_t* local_rank; uint32_t* block_sums;
  unsigned long long* dbg;   // (-DSMX_STAMPS: the integration launch's first workgroup leaves the wall clock here)
};
__device__ __forceinline__ void new_flags_scan_body(const NewFlagsArgs& a, const Scratch& sc, int W, int H, DevState* st, uint32_t block) {
  const Img<const uint16_t>& depth = a.depth;
  const Img<uint16_t>& depth_out = a.depth_out;
  const int copy_back = a.copy_back;
  uint8_t* __restrict__ flags = a.flags;
  uint32_t* __restrict__ local_rank = a.local_rank;
  uint32_t* __restrict__ block_sums = a.block_sums;
  __shared__ uint32_t wave_tot[kBlock / 64];
  const int P = W * H;
  const int k0 = (block * kBlock + threadIdx.x) * kScanPxPerThread;
  uint32_t f[kScanPxPerThread];
  uint32_t mine = 0;
  // (all loads of the lane first: written as `d > 0 && supporting == .. && key == ..` per pixel they come one by one)
  uint32_t dv[kScanPxPerThread], sv[kScanPxPerThread], cv[kScanPxPerThread];
#pragma unroll
  for (int j = 0; j < kScanPxPerThread; ++j) {
    const int k = min(k0 + j, P - 1);
    const int y = k / W, x = k - y * W;
    dv[j] = depth(y, x); sv[j] = sc.supporting[k]; cv[j] = sc.confl_key[k];
  }
#pragma unroll
  for (int j = 0; j < kScanPxPerThread; ++j) { keep(dv[j]); keep(sv[j]); keep(cv[j]); }
#pragma unroll
  for (int j = 0; j < kScanPxPerThread; ++j) {
    const int k = k0 + j;
    bool fl = false;
    if (k < P) {
      const int y = k / W, x = k - y * W;
      const uint16_t d = (uint16_t)dv[j];
      if (copy_back) depth_out(y, x) = d;
      fl = x >= 1 && y >= 1 && x < W - 1 && y < H - 1 && d > 0 && sv[j] == kInvalid && cv[j] == kInvalid;
      flags[k] = fl ? 1 : 0;
    }
    f[j] = fl ? 1u : 0u;
    mine += f[j];
  }
  // wave-level inclusive scan of per-lane counts
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t incl = mine;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t t = __shfl_up(incl, off);
    if (lane >= (uint32_t)off) incl += t;
  }
  if (lane == 63) wave_tot[wave] = incl;
  __syncthreads();
  uint32_t wave_off = 0, total = 0;
#pragma unroll
  for (int w = 0; w < kBlock / 64; ++w) {
    if ((uint32_t)w < wave) wave_off += wave_tot[w];
    total += wave_tot[w];
  }
  uint32_t run = wave_off + incl - mine;
#pragma unroll
  for (int j = 0; j < kScanPxPerThread; ++j) {
    const int k = k0 + j;
    if (k < P) local_rank[k] = run;
    run += f[j];
  }
  if (threadIdx.x == 0) {
    block_sums[block] = total;
    // (not create_base itself: the previous frame's pass B may still be reading that one)
    if (block == 0) st->create_base_next = st->surfel_count;  // stable until k_new_create's workgroup 0 adds the new slots
  }
}

// ---------------------------------------------------------------------------------------------
// IntegrateMeasurementsCUDAKernel / IntegrateOrConflictSurfel, kernels.cu:741-1142.  One lane per
// visible surfel; no lock, no block-wide votes (both are redundant in the reference: thread i is
// the only writer of surfel i).
struct FrameIn {
  Img<const uint16_t> depth; Img<const float2> normals; Img<const float> radius; Img<const uchar3> color;
};

// Everything the integration reads at one pixel, requested in one go (7 independent loads) before any of it
// is used -- for both pixels of a surfel at once, see k_integrate.
struct PixelIn {
  uint16_t depth; float first; uint32_t key; uint32_t count; float2 nxy; uchar3 col; float radius;
};
__device__ __forceinline__ PixelIn load_pixel(const FrameCtx& c, const Scratch& sc, const FrameIn& in, int x, int y) {
  const size_t k = (size_t)y * c.W + x;
  PixelIn p;
  p.depth = in.depth(y, x); p.first = sc.first_depth[k]; p.key = sc.confl_key[k]; p.count = sc.counts[k];
  p.nxy = in.normals(y, x); p.col = in.color(y, x); p.radius = in.radius(y, x);
  return p;
}

__device__ __forceinline__ void keep(const PixelIn& p) {
  keep((uint32_t)p.depth); keep(p.first); keep(p.key); keep(p.count); keep(p.nxy); keep(p.radius);
  keep((uint32_t)p.col.x | ((uint32_t)p.col.y << 8) | ((uint32_t)p.col.z << 16));
}

// Register copy of the records of one surfel that the integration may touch.  Thread i is the only reader and
// writer of surfel i in this kernel, so both pixel visits work on the copy (the second one sees what the first
// one changed, as in the reference) and the changed records are written back once, 16 bytes at a time.
struct SurfelRegs {
  float4 P;   // X, Y, Z, LastUpdateStamp
  float4 N;   // normal, RadiusSquared
  float4 C;   // Confidence, CreationStamp, Color, -
  bool dirty;      // P, N or C changed
  bool replaced;   // S and T have to be reset as well (:816-868)
  Vec3 new_smooth;
};

__device__ __forceinline__ void integrate_or_conflict(SurfelRegs& R, const FrameCtx& c, const PixelIn& px,
                                                      bool integrate, int x, int y,
                                                      const Vec3& cam, uint32_t i, DevState* st) {
  if (!integrate) return;
  const float measurement_depth = c.inv_depth_scaling * (float)px.depth;
  if (measurement_depth <= 0) return;
  bool conflicting = false;
  const float first = px.first;
  if (first < (1 - c.sensor_noise_factor) * measurement_depth) {
    if (first == cam.z) {
      const uint32_t key = px.key;
      if (key != kInvalid && (key & 0x7FFFFFFFu) == i) conflicting = true;
    }
    integrate = false;
  }
  if (!integrate && !conflicting) return;
  const float occlusion_depth = (1 + c.sensor_noise_factor) * measurement_depth;
  if (cam.z > occlusion_depth) integrate = false;
  if (!integrate && !conflicting) return;

  const float depth = measurement_depth;
  const Vec3 lp = {depth * (c.up.fx_inv * (float)x + c.up.cx_inv), depth * (c.up.fy_inv * (float)y + c.up.cy_inv), depth};
  const Vec3 gp = mul(c.G, lp);
  const float2 nxy = px.nxy;
  const Vec3 mn = {nxy.x, nxy.y, meas_normal_z(nxy.x, nxy.y)};
  const Vec3 gn = rotate(c.G, mn);
  const uchar3 col = px.col;

  if (conflicting) {  // :816-868
    if (c.stats) atomicAdd(&st->n_conflict_hits, 1u);
    float confidence = R.C.x;
    confidence -= 1;
    R.dirty = true;
    if (confidence <= 0) {
      if (c.stats) atomicAdd(&st->n_replaced, 1u);
      R.P = make_float4(gp.x, gp.y, gp.z, __uint_as_float(c.frame));
      R.new_smooth = gp;
      R.replaced = true;  // smooth position := position, neighbours := none
      R.N = make_float4(gn.x, gn.y, gn.z, px.radius);
      R.C = make_float4(1.0f, __uint_as_float(c.frame),
                        __uint_as_float((uint32_t)col.x | ((uint32_t)col.y << 8) | ((uint32_t)col.z << 16) | (1u << 24)), R.C.w);
    } else {
      R.C.x = confidence;
    }
  }
  if (!integrate) return;

  const float surfel_distance = sqrtf(cam.x * cam.x + cam.y * cam.y + cam.z * cam.z);
  const Vec3 sn = {R.N.x, R.N.y, R.N.z};
  const Vec3 ln = rotate(c.L, sn);
  const float dot_angle = (1.0f / surfel_distance) * (cam.x * ln.x + cam.y * ln.y + cam.z * ln.z);
  if (dot_angle > 0) return;
  if (measurement_depth < cam.z) {
    const float d = sn.x * gn.x + sn.y * gn.y + sn.z * gn.z;  // :898-903
    if (d < c.cos_normal_compat) integrate = false;
  }
  const float old_r2 = R.N.w;
  if (old_r2 < 0) integrate = false;
  if (!integrate) return;

  uint32_t cnt = px.count;  // :933
  if (cnt < 1) cnt = 1;
  const float weight = 1.0f / (float)cnt;
  if (__float_as_uint(R.C.y) < c.frame) {  // :940
    if (c.stats) atomicAdd(&st->n_integrated, 1u);
    const float confidence = R.C.x;
    R.C.x = (confidence + weight < c.max_conf) ? (confidence + weight) : c.max_conf;
    const float nf = 1.0f / (confidence + weight);
    R.P.x = (confidence * R.P.x + weight * gp.x) * nf;
    R.P.y = (confidence * R.P.y + weight * gp.y) * nf;
    R.P.z = (confidence * R.P.z + weight * gp.z) * nf;
    const Vec3 nn = {confidence * sn.x + weight * gn.x, confidence * sn.y + weight * gn.y, confidence * sn.z + weight * gn.z};
    const float inv = 1.0f / sqrtf(nn.x * nn.x + nn.y * nn.y + nn.z * nn.z);
    R.N = make_float4(inv * nn.x, inv * nn.y, inv * nn.z, fminf(old_r2, px.radius));
    const uint32_t oc = __float_as_uint(R.C.z);
    const uint32_t c0 = (uint32_t)(uint8_t)(int)((confidence * (float)(oc & 255u) + weight * (float)col.x) * nf + 0.5f);
    const uint32_t c1 = (uint32_t)(uint8_t)(int)((confidence * (float)((oc >> 8) & 255u) + weight * (float)col.y) * nf + 0.5f);
    const uint32_t c2 = (uint32_t)(uint8_t)(int)((confidence * (float)((oc >> 16) & 255u) + weight * (float)col.z) * nf + 0.5f);
    R.C.z = __uint_as_float(c0 | (c1 << 8) | (c2 << 16));
    R.P.w = __uint_as_float(c.frame);
    R.dirty = true;
  }
}

// The launch is shared with the flag + rank pass of CreateNewSurfelsCUDA (the first n_flag_blocks workgroups): both only
// need the front of the frame (association images, blended depth) and neither reads what the other writes -- the
// integration reads the blended depths from `in.depth`, which is the blend's own output image when the flag pass is the
// one that stores them back into the caller's buffer -- and a launch of its own for 300 small workgroups cost the front
// chain its 11 us plus a launch boundary.
template <bool kUseList>
__global__ void __launch_bounds__(kBlock)
k_integrate(Surfels S, FrameCtx c, Scratch sc, FrameIn in, Lists L,
            uint8_t* __restrict__ merge_flag, DevState* st, NewFlagsArgs nf, uint32_t n_flag_blocks) {
  SMX_SET_WAVE_PRIO();
  // (workgroup 0 is always a flag block: the image has pixels)
#ifdef SMX_STAMPS
  if (nf.dbg && blockIdx.x == 0 && threadIdx.x == 0) nf.dbg[0] = wall_clock64();
#endif
  if (blockIdx.x < n_flag_blocks) { ts_begin(c.ts, kTsIntBegin); new_flags_scan_body(nf, sc, c.W, c.H, st, blockIdx.x); return; }
  const uint32_t block = blockIdx.x - n_flag_blocks, n_blocks = gridDim.x - n_flag_blocks;
  const uint32_t n_scan = kUseList ? 0u : st->surfel_count;
  uint32_t merged_here = 0;
  uint32_t ent, cntv;
  const uint32_t n_steps = vis_begin<kUseList>(L, n_scan, block, ent, cntv);
  for (uint32_t w = block; w < n_steps; w += n_blocks) {
    const uint32_t cur = ent;
    ent = (w + n_blocks < n_steps) ? vis_load<kUseList>(L, w + n_blocks) : 0u;   // (the next step's entries travel while this one is worked on)
    uint32_t i;
    if (!vis_entry<kUseList>(cur, w, cntv, n_scan, threadIdx.x, i)) continue;
    // the merge mark and the three records the integration works on, requested together (pinned: the compiler would ask for
    // the mark, wait, ask for P, wait for the activity test, and only then for N and C -- three round trips on a kernel
    // that sits on both cycles of the frame)
    const uint32_t merge_mark = merge_flag[i];
    SurfelRegs R;
    R.P = *S.group(kGroupP, i); R.N = *S.group(kGroupN, i); R.C = *S.group(kGroupC, i);
    keep_if<2>(merge_mark); keep_if<2>(R.P); keep_if<2>(R.N); keep_if<2>(R.C);
    if (merge_mark) {
      // apply the merge marks, kernels.cu:1987-1989 (decided in k_merge_decide)
      merge_flag[i] = 0;
      S.u(kLastUpdateStamp, i) = 0;
      S.f(kRadiusSq, i) = -1;
      const uint32_t col = (__float_as_uint(R.C.z) & 0x00FFFFFFu) | 0x01000000u;
      S.u(kColor, i) = col;
      L.flags8[i] = make_flags(0u, col, c.frame, c.reg_window);
      L.hot_epoch[i >> L.hot_shift] = (uint8_t)L.epoch;
      if (L.dirty8) L.dirty8[i] = 1;
      ++merged_here;
      continue;  // r^2 < 0: the integrate kernel does nothing for it (:1050-1052)
    }
    R.dirty = false; R.replaced = false;
    const float4 p4 = R.P, n4 = R.N;
    if (!is_active(__float_as_uint(p4.w), c.frame, c.window)) continue;
    Proj p;
    const Vec3 g = {p4.x, p4.y, p4.z};
    if (!project_pos(g, c, p)) continue;
    if (n4.w < 0) continue;
    int ox = p.px, oy = p.py;
    const bool second = quadrant(p, c, ox, oy);
    const PixelIn px0 = load_pixel(c, sc, in, p.px, p.py);
    const PixelIn px1 = load_pixel(c, sc, in, ox, oy);  // (the main pixel again if there is no second one)
    if (SMX_RT_MASK & 2) { keep(px0); keep(px1); }   // (fourteen loads, one round trip: see above)
    integrate_or_conflict(R, c, px0, true, p.px, p.py, p.l, i, st);
    integrate_or_conflict(R, c, px1, second, ox, oy, p.l, i, st);
    if (R.dirty) {
      L.hot_epoch[i >> L.hot_shift] = (uint8_t)L.epoch;   // (stamp / colour mark may have changed, a replacement drops the links)
      if (L.dirty8) L.dirty8[i] = 1;
      out_store16<SMX_ST_INT>(S.group(kGroupP, i), R.P); out_store16<SMX_ST_INT>(S.group(kGroupN, i), R.N);
      out_store16<SMX_ST_INT>(S.group(kGroupC, i), R.C);
      if (R.replaced) {
        *S.group(kGroupS, i) = make_float4(R.new_smooth.x, R.new_smooth.y, R.new_smooth.z, 0.0f);   // (whole records: see k_reg_step)
        S.set_neighbors(i, make_uint4(kInvalid, kInvalid, kInvalid, kInvalid));
      }
    }
    // stamp and detach flag may have changed: refresh the flag table entry
    L.flags8[i] = make_flags(__float_as_uint(R.P.w), __float_as_uint(R.C.z), c.frame, c.reg_window);
  }
  // merge counter: one atomic per wavefront that merged something (kernels.cu:2045-2051 block-reduces)
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) merged_here += __shfl_xor(merged_here, off);
  if ((threadIdx.x & 63) == 0 && merged_here) {
    atomicAdd(&st->merge_count, merged_here);
    if (c.stats) atomicAdd(&st->n_merged, merged_here);   // (per-call statistic: reset by k_reset_frame_stats while statistics are on)
  }
  ts_end(c.ts, kTsIntEnd, block, min(n_blocks, n_steps));   // (the workgroups that walked the last steps of the first round)
}

// ---------------------------------------------------------------------------------------------
// UpdateNeighborsCUDAKernel, kernels.cu:1197-1380.
template <bool kUseList>
__device__ __forceinline__ void update_neighbors_body(const Surfels& S, const FrameCtx& c, const Scratch& sc, const FrameIn& in,
                                                      const Lists& L, const DevState* st, uint32_t block, uint32_t n_blocks) {
  const int kDX[4] = {-1, 1, 0, 0}, kDY[4] = {0, 0, -1, 1};
  // (the slot count BEFORE this frame's creation: the creating workgroups of the same launch advance surfel_count)
  const uint32_t n_scan = kUseList ? 0u : st->create_base_next;
  // (this launch precedes the regulariser's pass B on every path through Integrate: its chunk counter starts at zero)
  if (block == 0 && threadIdx.x < kSubLists) { L.rec_chunks.count[threadIdx.x * kCountStride] = 0; L.acc_chunks.count[threadIdx.x * kCountStride] = 0; }
  uint32_t ent, cntv;
  const uint32_t n_steps = vis_begin<kUseList>(L, n_scan, block, ent, cntv);
  for (uint32_t w = block; w < n_steps; w += n_blocks) {
    const uint32_t cur = ent;
    ent = (w + n_blocks < n_steps) ? vis_load<kUseList>(L, w + n_blocks) : 0u;   // (the next step's entries travel while this one is worked on)
    uint32_t i;
    if (!vis_entry<kUseList>(cur, w, cntv, n_scan, threadIdx.x, i)) continue;
    // the slot's three records in flight together (P: position + stamp, N: normal + r^2, T: neighbour ids)
    const float4 p4 = *S.group(kGroupP, i), n4 = *S.group(kGroupN, i);
    const uint4 t4 = *reinterpret_cast<const uint4*>(S.group(kGroupT, i));
    keep_if<4>(p4); keep_if<4>(n4); keep_if<4>(t4);   // (pinned: N and T would otherwise be asked for behind the tests on P)
    if (!is_active(__float_as_uint(p4.w), c.frame, c.window)) continue;
    const Vec3 g = {p4.x, p4.y, p4.z};
    const Vec3 cam = mul(c.L, g);
    if (!(cam.z > 0)) continue;
    const float u = c.fx * (cam.x / cam.z) + c.cx, v = c.fy * (cam.y / cam.z) + c.cy;
    if (!(u >= 1.0f && v >= 1.0f && u < (float)(c.W - 1) && v < (float)(c.H - 1))) continue;  // :1232-1240
    const int x = (int)u, y = (int)v;
    // the pixel-side reads that do not depend on each other
    const float measurement_depth = c.inv_depth_scaling * (float)in.depth(y, x);
    const float obs_r2 = in.radius(y, x);
    uint32_t cand[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) cand[d] = sc.supporting[(size_t)(y + kDY[d]) * c.W + (x + kDX[d])];
    keep_if<4>(measurement_depth); keep_if<4>(obs_r2);
#pragma unroll
    for (int d = 0; d < 4; ++d) keep_if<4>(cand[d]);
    const float occlusion_depth = (1 + c.sensor_noise_factor) * measurement_depth;
    if (cam.z > occlusion_depth) continue;
    const float surfel_distance = sqrtf(cam.x * cam.x + cam.y * cam.y + cam.z * cam.z);
    const Vec3 gn = {n4.x, n4.y, n4.z};
    const Vec3 ln = rotate(c.L, gn);
    const float dot_angle = (1.0f / surfel_distance) * (cam.x * ln.x + cam.y * ln.y + cam.z * ln.z);
    if (dot_angle > 0) continue;
    const float r2 = n4.w;
    if (r2 < 0) continue;
    if (obs_r2 / r2 > 1.5f * 1.5f) continue;  // :1287-1291

    // gathers of the 4 current neighbours (P) and the 4 candidates (P, N), all issued before the first use;
    // empty slots read the slot's own records
    uint32_t ni[4] = {t4.x, t4.y, t4.z, t4.w};
    float4 np4[4], cp4[4], cn4[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) np4[q] = *S.group(kGroupP, ni[q] == kInvalid ? i : ni[q]);
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      const uint32_t gidx = (cand[d] == kInvalid) ? i : cand[d];
      cp4[d] = *S.group(kGroupP, gidx);
      cn4[d] = *S.group(kGroupN, gidx);
    }
    float nd2[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (ni[q] == kInvalid) nd2[q] = __builtin_inff();
      else {
        const float dx = g.x - np4[q].x, dy = g.y - np4[q].y, dz = g.z - np4[q].z;
        nd2[q] = dx * dx + dy * dy + dz * dz;
      }
    }
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      const uint32_t nb = cand[d];
      if (nb == kInvalid || nb == i) continue;
      const float dx = cp4[d].x - g.x, dy = cp4[d].y - g.y, dz = cp4[d].z - g.z;
      const float d2 = dx * dx + dy * dy + dz * dz;
      if (d2 > c.rf2 * r2) continue;
      const float nd = gn.x * cn4[d].x + gn.y * cn4[d].y + gn.z * cn4[d].z;
      if (nd <= 0) continue;
      int best_n = -1; float best_d2 = -1;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (best_n == -2) continue;
        if (nb == ni[q]) { best_n = -2; }
        else if (nd2[q] > best_d2) { best_n = q; best_d2 = nd2[q]; }
      }
      if (best_n >= 0 && d2 < best_d2) {
#pragma unroll
        for (int q = 0; q < 4; ++q) if (q == best_n) { ni[q] = nb; nd2[q] = d2; }
      }
    }
    if (ni[0] != t4.x || ni[1] != t4.y || ni[2] != t4.z || ni[3] != t4.w) {
      S.set_neighbors(i, make_uint4(ni[0], ni[1], ni[2], ni[3]));
      L.hot_epoch[i >> L.hot_shift] = (uint8_t)L.epoch;   // (new links)
    }
  }
  ts_end(c.ts, kTsUpdEnd, block, min(n_blocks, n_steps));   // (the workgroups that walked the last steps of the first round)
}

// ---------------------------------------------------------------------------------------------
// (creation, second half: the flag + rank pass of CreateNewSurfelsCUDA is further up, next to the launch it shares)
struct CreateArgs {
  const uint8_t* flags; const uint32_t* ranks; const uint32_t* block_sums; uint32_t* block_offsets_out;
  int n_scan_blocks; uint32_t max_surfels; uint8_t* flags8; uint8_t* dirty8;
  uint32_t* next_vis_chunk_count;   // the NEXT call's chunk counter of the visible list: reset by this launch
  int n_pixels;
  uint8_t* hot_epoch; uint32_t epoch; int hot_shift;   // Lists::hot_epoch, epoch, hot_shift
};
__device__ __forceinline__ void new_create_body(const Surfels& S, const FrameCtx& c, const Scratch& sc, const FrameIn& in,
                                                const CreateArgs& a, DevState* st, uint32_t block, uint32_t n_blocks) {
  const uint8_t* __restrict__ flags = a.flags;
  const uint32_t* __restrict__ ranks = a.ranks;
  const uint32_t* __restrict__ block_sums = a.block_sums;
  uint32_t* __restrict__ block_offsets_out = a.block_offsets_out;
  const int n_scan_blocks = a.n_scan_blocks;
  const uint32_t max_surfels = a.max_surfels;
  uint8_t* __restrict__ flags8 = a.flags8;
  // Every workgroup scans the few hundred block totals of k_new_flags_scan itself (cheaper than a launch of its
  // own for one workgroup); workgroup 0 publishes the counts (cc:291) and the offsets (debug decode).
  extern __shared__ uint32_t block_offsets[];  // [n_scan_blocks] exclusive
  __shared__ uint32_t wave_tot[kBlock / 64];
  // (a workgroup whose pixels hold no flagged pixel has nothing to create: most of them, in a map that is not growing.
  // Its pixels lie in ONE scan block -- the grid has one workgroup per kBlock pixels -- whose total says so.  Workgroup 0
  // stays: it publishes the counts.)
  static_assert(kScanPxPerBlock % kBlock == 0, "a creating workgroup's pixels lie inside one scan block");
  if (block != 0 && n_blocks * (uint32_t)kBlock >= (uint32_t)(c.W * c.H) &&
      block_sums[(block * (uint32_t)kBlock) / (uint32_t)kScanPxPerBlock] == 0) return;
  const int per = (n_scan_blocks + kBlock - 1) / kBlock;
  const uint32_t base = st->create_base_next;   // (requested with the totals)
  uint32_t mine = 0, total;
  if (per <= 8) {   // (images up to 2 M pixels) the lane's totals in one round trip, kept for the second loop
    uint32_t v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int bidx = threadIdx.x * per + j;
      v[j] = (j < per && bidx < n_scan_blocks) ? block_sums[bidx] : 0u;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) mine += v[j];
    uint32_t run = block_excl_scan(mine, wave_tot, total);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int bidx = threadIdx.x * per + j;
      if (j < per && bidx < n_scan_blocks) { block_offsets[bidx] = run; run += v[j]; }
    }
  } else {
    for (int j = 0; j < per; ++j) {
      const int bidx = threadIdx.x * per + j;
      if (bidx < n_scan_blocks) mine += block_sums[bidx];
    }
    uint32_t run = block_excl_scan(mine, wave_tot, total);
    for (int j = 0; j < per; ++j) {
      const int bidx = threadIdx.x * per + j;
      if (bidx < n_scan_blocks) { block_offsets[bidx] = run; run += block_sums[bidx]; }
    }
  }
  __syncthreads();
  const uint32_t room = max_surfels - base;
  const uint32_t created = total < room ? total : room;  // cap rule (reference: unchecked, cc:291)
  if (block == 0) {
    for (int bidx = threadIdx.x; bidx < n_scan_blocks; bidx += kBlock) block_offsets_out[bidx] = block_offsets[bidx];
    if (threadIdx.x == 0) {
      st->create_base = base;
      st->new_count = created;
      st->capacity_clamped = (total > room) ? 1u : 0u;
      st->surfel_count = base + created;
    }
  }
  const int kDX[4] = {-1, 1, 0, 0}, kDY[4] = {0, 0, -1, 1};
  const int W = c.W, H = c.H, P = W * H;
  // This launch sits on the frame-to-frame cycle and lasts as long as its slowest lane, and a creating lane used to be
  // that lane: four neighbour directions one after the other, each a chain of dependent gathers (supporting surfel ->
  // its position -> its smooth position), between stores the compiler must not move loads across.  So: every pixel-side
  // value of the pixel and its four neighbours first, all requested together; then the neighbour surfels' P and S
  // records together; then the arithmetic (in the reference's order) and the stores.
  for (int k = block * kBlock + threadIdx.x; k < P; k += n_blocks * kBlock) {
    const uint32_t rank = block_offsets[k / kScanPxPerBlock] + ranks[k];
    // ranks[] keeps the block-local values; global rank = block offset + local rank
    if (flags[k] != 1 || rank >= created) continue;
    const int y = k / W, x = k - y * W;
    const uint32_t i = base + rank;
    const uint32_t depth_u = in.depth(y, x);
    const float2 nxy = in.normals(y, x);
    const uchar3 col = in.color(y, x);
    const float r2 = in.radius(y, x);
    uint32_t nsup[4], nflag[4], nrank_local[4], ndepth_u[4];
    int kks[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      // (a flagged pixel has four neighbours inside the image: the flag pass leaves the border out, kernels.cu:90-111)
      const int yy = min(max(y + kDY[d], 0), H - 1), xx = min(max(x + kDX[d], 0), W - 1);
      kks[d] = yy * W + xx;
      nsup[d] = sc.supporting[kks[d]];
      nflag[d] = flags[kks[d]];
      nrank_local[d] = ranks[kks[d]];
      ndepth_u[d] = in.depth(yy, xx);
    }
    keep(depth_u); keep(nxy); keep(r2); keep((uint32_t)col.x | ((uint32_t)col.y << 8) | ((uint32_t)col.z << 16));
#pragma unroll
    for (int d = 0; d < 4; ++d) { keep(nsup[d]); keep(nflag[d]); keep(nrank_local[d]); keep(ndepth_u[d]); }
    float4 np4[4], ns4[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      const uint32_t gidx = nsup[d] != kInvalid ? nsup[d] : base;   // (unused: any valid slot)
      np4[d] = *S.group(kGroupP, gidx);
      ns4[d] = *S.group(kGroupS, gidx);
    }
#pragma unroll
    for (int d = 0; d < 4; ++d) { keep(np4[d]); keep(ns4[d]); }
    const float depth = c.inv_depth_scaling * (float)depth_u;
    const Vec3 lp = {depth * (c.up.fx_inv * (float)x + c.up.cx_inv), depth * (c.up.fy_inv * (float)y + c.up.cy_inv), depth};
    const Vec3 gp = mul(c.G, lp);
    const Vec3 mn = {nxy.x, nxy.y, meas_normal_z(nxy.x, nxy.y)};
    const Vec3 gn = rotate(c.G, mn);
    Vec3 sum = {0, 0, 0};
    int count_plus_1 = 1;
    uint32_t nbs[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      uint32_t nb = nsup[d];
      if (nb != kInvalid) {
        const float dx = np4[d].x - gp.x, dy = np4[d].y - gp.y, dz = np4[d].z - gp.z;
        const float d2 = dx * dx + dy * dy + dz * dz;
        if (d2 > c.rf2 * r2) nb = kInvalid;
        else {
          sum.x = sum.x + ns4[d].x; sum.y = sum.y + ns4[d].y; sum.z = sum.z + ns4[d].z;
          ++count_plus_1;
        }
      } else if (nflag[d] == 1) {
        const uint32_t nrank = block_offsets[kks[d] / kScanPxPerBlock] + nrank_local[d];
        if (nrank < created) {
          const float od = c.inv_depth_scaling * (float)ndepth_u[d];
          const float ad2 = (depth - od) * (depth - od);
          if (ad2 <= c.rf2 * r2) nb = base + nrank;
        }
      }
      nbs[d] = nb;
    }
    *S.group(kGroupP, i) = make_float4(gp.x, gp.y, gp.z, __uint_as_float(c.frame));                       // X, Y, Z, LastUpdateStamp
    *S.group(kGroupN, i) = make_float4(gn.x, gn.y, gn.z, r2);                                              // normal, RadiusSquared
    // (whole 16-byte records, the unused fourth words included: a sector that is not fully dirty is written back as a slow partial write)
    *S.group(kGroupC, i) = make_float4(1.0f, __uint_as_float(c.frame),
                                       __uint_as_float((uint32_t)col.x | ((uint32_t)col.y << 8) | ((uint32_t)col.z << 16)), 0.0f);
    flags8[i] = make_flags(c.frame, 0u, c.frame, c.reg_window);
    a.hot_epoch[i >> a.hot_shift] = (uint8_t)a.epoch;
    if (a.dirty8) a.dirty8[i] = 1;
    S.set_neighbors(i, make_uint4(nbs[0], nbs[1], nbs[2], nbs[3]));
    *S.group(kGroupS, i) = make_float4((gp.x + sum.x) / (float)count_plus_1, (gp.y + sum.y) / (float)count_plus_1,
                                       (gp.z + sum.z) / (float)count_plus_1, 0.0f);  // :227-229
  }
}

// UpdateNeighborsCUDAKernel and the creation kernel in ONE launch: both only need the integrated surfels, they
// write disjoint slots (visible old ones / new ones), and every launch boundary on the frame-to-frame critical
// path costs a cache write-back across the eight XCDs.  The first n_create_blocks workgroups create.
template <bool kUseList>
__global__ void __launch_bounds__(kBlock)
k_update_and_create(Surfels S, FrameCtx c, Scratch sc, FrameIn in, Lists L, CreateArgs a, uint32_t n_create_blocks,
                    DevState* st) {
  SMX_SET_WAVE_PRIO();
  if (blockIdx.x < n_create_blocks) {
    ts_begin(c.ts, kTsUpdBegin);
    new_create_body(S, c, sc, in, a, st, blockIdx.x, n_create_blocks);
  } else {
    const uint32_t block = blockIdx.x - n_create_blocks, n_blocks = gridDim.x - n_create_blocks;
    // (pass A of the next call, which follows this launch, appends to the other chunk counter)
    if (block == 0 && threadIdx.x < kSubLists) a.next_vis_chunk_count[threadIdx.x * kCountStride] = 0;
    update_neighbors_body<kUseList>(S, c, sc, in, L, st, block, n_blocks);
  }
}

// ---------------------------------------------------------------------------------------------
// Pass B.  UpdateNeighborsCUDARemoveReplacedNeighborsKernel (kernels.cu:1420-1437) +
// RegularizeSurfelsCUDAAccumulateNeighborGradientsKernel (kernels.cu:2115-2195) in one streaming
// pass over the 4 neighbour rows (16 B/slot, 16-byte lane loads), plus construction of the segmented
// list of recently updated slots that the step/update kernels run over.  The per-edge gathers of the
// neighbour's colour word (:1430) and stamp (:2132) are replaced by ONE byte gather from the flag table
// (L2-resident: 1 B/slot).  The gradient clear (kernels.cu:2099-2113) is gone: the fixed-point
// accumulators are zero between calls (k_reg_step zeroes what it consumes).
struct EdgeArgs {
  float rf2, weight;
  long long* grad_acc;
  float4* reg_rec; size_t rec_own_offset;
  FarBins fb;
};
struct EdgeLds {
  unsigned long long* lacc;   // [kSegAcc * 2] per target: (gx | gy), (gz | sender classes)
  uint32_t* hkey; uint32_t* hcnt;   // [kFarHash] far destinations of this workgroup: segment, terms -> base in the bin
  uint16_t* lrank;            // [kSegAcc] per slot of the segment: its rank among the recent slots, or 0xFFFF
  uint32_t* rec_wave;         // [kSegAcc / 64] recent entries per (chunk, wavefront) of the segment
};
constexpr int kSegAcc = 1024;     // slots per accumulation workgroup (16 KB of LDS sums)
constexpr int kBlockAcc = 256;
static_assert(kSegAcc == kSegB && kBlockAcc == kBlockB, "the edge work runs on pass B's segments and workgroups (fused, or over its work lists)");
__device__ __forceinline__ void edge_segment(const Surfels& S, const EdgeArgs& ea, DevState* st, uint32_t base, uint32_t n_act,
                                             uint32_t ent_first, const uint32_t* later_entries, const EdgeLds& lds, uint32_t tid);
#ifndef SMX_FUSE_EDGES
#define SMX_FUSE_EDGES 0   // (default of the run-time switch; smx_recon_set_scan_mode bit 9 turns the fusion ON.  Measured: the fused
                           // launch is shorter alone -- 58 us against 28 + 36 -- and LONGER in the frame, 90 us (heavy segments first) /
                           // 72 us (last) against 34 + 42: C2 6150 / 6249 -> 5852 / 5825 frames/s, profiles/r5_ab_notes.md)
#endif
#ifndef SMX_FUSED_HEAVY_FIRST
#define SMX_FUSED_HEAVY_FIRST 1
#endif
#ifndef SMX_FUSED_WGS_PER_CU
#define SMX_FUSED_WGS_PER_CU 4   // (fused pass B: 36 KB of LDS per workgroup)
#endif
// kFused (A/B, off by default): the workgroup that finds work for the edge kernel in its segment does that work at once, from
// the entries it has just ranked (in LDS instead of the global work list) -- one launch and its boundary less on the internal
// stream, whose five launches a frame spend ~34 of their 154 us between kernels.  It loses in the frame (see SMX_FUSE_EDGES):
// 36 KB of LDS and 86 VGPRs for EVERY segment's workgroup, the streaming ones included, and the chip shared with the front.
template <bool kDetach, bool kAccumulate, bool kFused = false>
__global__ void __launch_bounds__(kBlockB, kFused ? SMX_FUSED_WGS_PER_CU : 1)
k_neighbor_scan(Surfels S, int stats, int use_hot, Lists L, DevState* st, unsigned long long* ts, EdgeArgs ea, uint32_t descending) {
  SMX_SET_WAVE_PRIO();
  ts_begin(ts, kTsRegBegin);
  static_assert(!kFused || kAccumulate, "the fused edge work belongs to the accumulating pass");
  __shared__ unsigned long long e_lacc[kFused ? kSegAcc * 2 : 1];
  __shared__ uint32_t e_hkey[kFused ? kFarHash : 1], e_hcnt[kFused ? kFarHash : 1];
  __shared__ uint16_t e_lrank[kFused ? kSegAcc : 2];
  __shared__ uint32_t e_rec_wave[kSegAcc / 64];
  __shared__ uint32_t lent[kFused ? kSegAcc : 1];   // the segment's work-list entries (fused: LDS instead of L.act_list)
  const uint32_t seg_id = segment_of_block(descending);
  extern __shared__ __align__(16) uint8_t lhot[];   // the hot-group table (n_hot_groups bytes, padded to 16)
  __shared__ uint32_t ltargets[kMaxHotGroups / 32];  // bit g: a link of this segment points into group g (another segment)
  // B1: pure streaming.  Per slot: detach (:1430-1433), which of its neighbours lie inside the regulariser
  // window (4-bit mask -> inwin8), membership in the recent list.  No LDS accumulators here, so the
  // occupancy stays high; the accumulation itself runs in k_reg_accumulate on the few segments that need it.
  __shared__ uint32_t wave_tot[kBlockB / 64];
  __shared__ __attribute__((aligned(4))) uint8_t lflags[kSegB];  // the segment's own flag bytes
  // (the four things every workgroup begins with -- slot count, creation base, its slice of the hot-group table, its slice of
  // the segment's target bitmap -- are requested together and pinned: the compiler had made three round trips in a row of
  // them, which is all a skipped segment's workgroup does, and the launch churns through thousands of those)
  const uint32_t N = st->surfel_count;
  const uint32_t base = seg_id * kSegB;
  // The reference detaches BEFORE it creates new surfels (kernels.cc:333-339 precedes cc:264-286), so
  // slots created in this frame keep links to flagged surfels until the next frame.
  const uint32_t detach_limit = st->create_base;
  uint4 he_first = make_uint4(0u, 0u, 0u, 0u);
  uint32_t reached_first = 0;
  if (use_hot) {
    if (threadIdx.x * 16 < L.n_hot_groups) he_first = *reinterpret_cast<const uint4*>(&L.hot_epoch[threadIdx.x * 16]);
    if (!stats) reached_first = L.seg_targets[(size_t)seg_id * kBlockB + threadIdx.x];
  }
  keep_if<16>(N); keep_if<16>(detach_limit); keep_if<16>(he_first); keep_if<16>(reached_first);
  if (base >= N) return;
  // The far flag gathers (a link that leaves the segment: one random byte from the 5 MB table, 4.6 M per frame at C2,
  // 21 of this kernel's 55 us alone) are skipped where their result is known.  The flag byte of the target matters in
  // two ways: bit 0 (inside the window) -- zero if the target's group is not hot, because pass A found no such slot in
  // it and nothing stamped one since; bit 1 (detach request) -- of consequence only while a link to the marked slot
  // still exists: a mark set in call g removes the links that exist then in pass B of call g (or g + 1 for sources
  // created in g), and a link created in a later call h is examined in pass B of h (or h + 1); in all of those passes
  // the target's or the source's group is hot (mark or link written in the call or the one before) and the gather takes
  // place.  So with BOTH groups cold the byte cannot change anything.  The table (one byte per group: <= 4 KB)
  // is copied to LDS, so the test itself is an LDS read, not another gather.  (use_hot = 0 for two calls whenever
  // flags or links were written outside Integrate.)
  //
  // A whole segment is skipped -- its 17 KB of link records and flag bytes not even read -- when its own group is cold
  // and so is every group one of its links points into: then every flag byte the pass would look at is of no consequence
  // by the argument above, the segment has no recent slot and no edge into the window.  The set of groups a segment's
  // links point into is kept as a bitmap per segment (seg_targets: 16 bits per lane, 512 bytes per segment against 17 KB),
  // rebuilt by every pass that does read the segment.  Links only disappear between such passes (detach, too-far
  // pruning), except in k_integrate / k_update_and_create, which stamp the group of every slot whose links they write:
  // the segment is then hot and the next pass rebuilds its bitmap.  (At C2 two segments out of five are skipped;
  // with groups of 2048 slots a cold segment's thousand far links seldom miss every hot group -- tools/far_terms_hist.py.)
  static_assert(kBlockB * 16 == kMaxHotGroups, "one lane per 16 groups");
  uint32_t hot16 = 0;   // bit k: group 16 * lane + k is hot
  if (use_hot) {
    const uint32_t g = threadIdx.x * 16;
    if (g < L.n_hot_groups) {
      const uint4 he = he_first;
      *reinterpret_cast<uint4*>(&lhot[g]) = he;
      const uint32_t w[4] = {he.x, he.y, he.z, he.w};
#pragma unroll
      for (int k = 0; k < 16; ++k)
        if (g + k < L.n_hot_groups && group_is_hot((w[k >> 2] >> (8 * (k & 3))) & 255u, L.epoch)) hot16 |= 1u << k;
    }
    if (!stats) {   // (the edge statistics count every link of the map)
      const uint32_t own_group = base >> L.hot_shift;
      const uint32_t reached = reached_first | (threadIdx.x == own_group / 16 ? 1u << (own_group % 16) : 0u);
      if (!__syncthreads_or((reached & hot16) != 0)) {
        if (threadIdx.x == 0) {
          L.recent_seg[seg_id] = kInvalid;   // (no recent slot; the mark is what smx_recon_debug_count_skipped_segments counts)
        }
        return;
      }
    }
  }
  const uint32_t i0 = base + threadIdx.x * 4;
  // A QUIET segment -- its own group is cold, it is read because its bitmap reaches a hot group -- has no recent slot
  // (pass A found none in the group) and none of its in-segment links can matter (source and target share the cold
  // group: the argument above), so all there is to do is to look at the far links that end in a hot group.  No own flag
  // bytes, no LDS copy of them, no barrier in front of the link loop, a dozen instructions per link instead of fifty:
  // the kernel's wave-time was the largest of the frame (SQ_WAVE_CYCLES, profiles/r17a_SQ_WAIT_ANY.md) and two fifths of
  // it VALU issue.  The bitmap is not rebuilt either: links only disappear while the own group is cold, so the old
  // bitmap remains a superset of the groups the segment reaches, which is all the skip test above needs.
  if (use_hot && !stats && (kDetach || kAccumulate) && !group_is_hot(lhot[base >> L.hot_shift], L.epoch)) {
    uint4 trec[4];
    uint32_t far_mask = 0;
    if (i0 < N) {
#pragma unroll
      for (int j = 0; j < 4; ++j) trec[j] = *reinterpret_cast<const uint4*>(S.group(kGroupT, i0 + j));
#pragma unroll
      for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const uint32_t nb = q == 0 ? trec[j].x : q == 1 ? trec[j].y : q == 2 ? trec[j].z : trec[j].w;
          if (nb - base >= (uint32_t)kSegB && nb != kInvalid && i0 + j < N && group_is_hot(lhot[nb >> L.hot_shift], L.epoch))
            far_mask |= 1u << (4 * j + q);
        }
      }
    }
    uint8_t inw[4] = {0, 0, 0, 0};
    int need = 0;
    if (__ballot(far_mask != 0) != 0ull) {   // (per wavefront)
      uint32_t far_flag[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        const uint4& t = trec[k >> 2];
        const uint32_t nb = (k & 3) == 0 ? t.x : (k & 3) == 1 ? t.y : (k & 3) == 2 ? t.z : t.w;
        far_flag[k] = L.flags8[((far_mask >> k) & 1u) ? nb : min(i0, N - 1u)];
      }
#pragma unroll
      for (int k = 0; k < 16; ++k) keep(far_flag[k]);
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        if (!((far_mask >> k) & 1u)) continue;
        const uint32_t f = far_flag[k];
        if (kDetach && i0 + (uint32_t)(k >> 2) < detach_limit && (f & 2u)) { S.set_neighbor(i0 + (uint32_t)(k >> 2), k & 3, kInvalid); continue; }  // :1430-1433
        if (kAccumulate && (f & 1u)) { inw[k >> 2] |= (uint8_t)(1u << (k & 3)); need = 1; }
      }
    }
    // the edge kernel's work list of the segment: one entry per slot with a link into the window (ActEntry), at its rank
    uint32_t total_act = 0;
    if (kAccumulate) {
      const uint32_t nact = (uint32_t)((inw[0] != 0) + (inw[1] != 0) + (inw[2] != 0) + (inw[3] != 0));
      uint32_t off = base + block_excl_scan_sparse<kBlockB / 64>(need ? nact : 0u, wave_tot, total_act);
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (inw[j]) { (kFused ? lent[off - base] : L.act_list[off]) = act_entry(threadIdx.x * 4 + j, inw[j], false); ++off; }
    }
    if (threadIdx.x == 0) {
      L.recent_seg[seg_id] = 0;
      if (kAccumulate && !kFused && total_act) emit_segment(L.acc_chunks, seg_id | ((total_act - 1u) << 22));
    }
    if (kFused && total_act) {   // (uniform)
      __syncthreads();
      EdgeLds lds; lds.lacc = e_lacc; lds.hkey = e_hkey; lds.hcnt = e_hcnt; lds.lrank = e_lrank; lds.rec_wave = e_rec_wave;
      edge_segment(S, ea, st, base, total_act, threadIdx.x < total_act ? lent[threadIdx.x] : kNoActEntry, lent, lds, threadIdx.x);
    }
    return;
  }
  if (threadIdx.x < kMaxHotGroups / 32) ltargets[threadIdx.x] = 0;
  uint32_t recent_bits = 0;
  uint32_t inw4 = 0;   // the window masks of the lane's four slots, a byte each
  uchar4 own = make_uchar4(0, 0, 0, 0);
  uint4 trec[4];  // the T records (4 neighbour ids) of the lane's 4 slots: 64 contiguous bytes
  if (i0 < N) {
    own = *reinterpret_cast<const uchar4*>(&L.flags8[i0]);
#pragma unroll
    for (int j = 0; j < 4; ++j) trec[j] = *reinterpret_cast<const uint4*>(S.group(kGroupT, i0 + j));
  }
  *reinterpret_cast<uchar4*>(&lflags[threadIdx.x * 4]) = own;
  __syncthreads();
  const bool quiet = use_hot && !group_is_hot(lhot[base >> L.hot_shift], L.epoch);
  // The far flag bytes of a lane's 16 links are requested TOGETHER (a gather that is not needed reads the lane's own
  // byte): fetched one after the other inside the loop below, as the compiler would arrange it, the 16 dependent round
  // trips of a fully active segment's workgroup are what the whole launch lasts (tools/isa_phases.py: 20 waits -> 5).
  uint32_t far_mask = 0;   // bit 4 j + q: that link leaves the segment and its target's flag byte matters
  uint32_t far_flag[16];
  if ((kDetach || kAccumulate) && i0 < N) {   // (the copy-only pass without detaching looks at no flag but the slot's own)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const uint32_t nb = q == 0 ? trec[j].x : q == 1 ? trec[j].y : q == 2 ? trec[j].z : trec[j].w;
        if (i0 + j < N && nb != kInvalid && nb - base >= (uint32_t)kSegB &&
            !(quiet && !group_is_hot(lhot[nb >> L.hot_shift], L.epoch)))   // (both cold: neither bit is of any consequence)
          far_mask |= 1u << (4 * j + q);
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 16; ++k) far_flag[k] = 0;
  if (__ballot(far_mask != 0) != 0ull) {   // (per wavefront: most wavefronts of the map have nothing to fetch)
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const uint4& t = trec[k >> 2];
      const uint32_t nb = (k & 3) == 0 ? t.x : (k & 3) == 1 ? t.y : (k & 3) == 2 ? t.z : t.w;
      far_flag[k] = L.flags8[((far_mask >> k) & 1u) ? nb : min(i0, N - 1u)];
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) keep(far_flag[k]);
  }
  if (i0 < N) {
    const uint8_t ownf[4] = {own.x, own.y, own.z, own.w};
    uint8_t inw[4] = {0, 0, 0, 0};
    uint32_t edges = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint32_t i = i0 + j;
      if (i >= N) continue;
      if (ownf[j] & 1u) recent_bits |= 1u << j;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const uint32_t nb = q == 0 ? trec[j].x : q == 1 ? trec[j].y : q == 2 ? trec[j].z : trec[j].w;
        if (nb == kInvalid) continue;
        // three of four links stay inside the segment: those flags come from the LDS copy
        const uint32_t rel = nb - base;
        uint32_t f;
        if (rel < (uint32_t)kSegB) f = lflags[rel];
        else f = ((far_mask >> (4 * j + q)) & 1u) ? far_flag[4 * j + q] : 0u;
        if (kDetach && i < detach_limit && (f & 2u)) {  // :1430-1433
          S.set_neighbor(i, q, kInvalid);
          continue;
        }
        if (rel >= (uint32_t)kSegB) { const uint32_t g = nb >> L.hot_shift; atomicOr(&ltargets[g >> 5], 1u << (g & 31u)); }
        ++edges;
        if (kAccumulate && (f & 1u)) inw[j] |= (uint8_t)(1u << q);
      }
    }
    inw4 = (uint32_t)inw[0] | ((uint32_t)inw[1] << 8) | ((uint32_t)inw[2] << 16) | ((uint32_t)inw[3] << 24);
    if (stats && kAccumulate && edges) {
      atomicAdd(&st->n_edges, edges);
      const uint32_t we = __popc(inw[0]) + __popc(inw[1]) + __popc(inw[2]) + __popc(inw[3]);
      if (we) { atomicAdd(&st->n_window_edges, we); atomicAdd(&st->n_contributors, (uint32_t)((inw[0] != 0) + (inw[1] != 0) + (inw[2] != 0) + (inw[3] != 0))); }
    }
  }
  // Two lists per segment, ranks from ONE scan (both counts in one word): the recent slots (the step kernel's list) and the
  // slots the edge kernel has work for -- recent ones (their own term) and those with a link into the window (ActEntry).
  uint32_t act_bits = kAccumulate ? recent_bits : 0u;
  if (kAccumulate) {
#pragma unroll
    for (int j = 0; j < 4; ++j) if ((inw4 >> (8 * j)) & 255u) act_bits |= 1u << j;
  }
  uint32_t total2;
  const uint32_t off2 = block_excl_scan_sparse<kBlockB / 64>((uint32_t)__popc(recent_bits) | ((uint32_t)__popc(act_bits) << 16), wave_tot, total2);
  const uint32_t total = total2 & 0xFFFFu, total_act = total2 >> 16;
  uint32_t off = base + (off2 & 0xFFFFu), off_act = base + (off2 >> 16);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (recent_bits & (1u << j)) L.recent_list[off++] = i0 + j;
    if (act_bits & (1u << j)) {
      (kFused ? lent[off_act - base] : L.act_list[off_act]) = act_entry(threadIdx.x * 4 + j, (inw4 >> (8 * j)) & 15u, (recent_bits >> j) & 1u);
      ++off_act;
    }
  }
  L.seg_targets[(size_t)seg_id * kBlockB + threadIdx.x] = (uint16_t)(ltargets[threadIdx.x >> 1] >> (16 * (threadIdx.x & 1)));
  if (threadIdx.x == 0) {
    L.recent_seg[seg_id] = total;
    if (total) {   // (one walk step per segment)
      const uint32_t k = seg_id % kSubLists;
      L.rec_chunks.desc[(size_t)k * L.rec_chunks.stride + atomicAdd(&L.rec_chunks.count[k * kCountStride], 1u)] = seg_id | ((total - 1u) << 22);
    }
    if (kAccumulate && !kFused && total_act) emit_segment(L.acc_chunks, seg_id | ((total_act - 1u) << 22));
    if (stats && total) atomicAdd(&st->recent_count, total);
  }
  if (kFused && total_act) {   // (uniform)
    __syncthreads();
    EdgeLds lds; lds.lacc = e_lacc; lds.hkey = e_hkey; lds.hcnt = e_hcnt; lds.lrank = e_lrank; lds.rec_wave = e_rec_wave;
    edge_segment(S, ea, st, base, total_act, threadIdx.x < total_act ? lent[threadIdx.x] : kNoActEntry, lent, lds, threadIdx.x);
  }
}

// B2: RegularizeSurfelsCUDAAccumulateNeighborGradientsKernel (kernels.cu:2115-2195) on the segments that have
// a recent slot or an edge into the regulariser window; 512-lane workgroups, two slots per lane.  It also forms the
// slot's OWN smoothness term (the neighbour loop of RegularizeSurfelsCUDAKernel, :2238-2256) from the same
// gathered neighbour positions and parks it in the G record, so that k_reg_step gathers nothing.
//
// The reference pushes every edge's gradient term to the neighbour with four float atomicAdds.  Device-scope
// atomics are the slowest thing this chip does, so the terms travel two ways, both of which end in the same
// exact integer sum (the three gradient components quantised to 2^-22 m, the weights counted per sender class and
// multiplied out in 2^-32 fixed point by the reader -- integer addition: the split cannot change the result):
//   1. target inside the workgroup's own segment (3 of 4 edges): summed in LDS, stored once per target, coalesced
//      (grad_local); the words hold (gx | gy) and (gz | sender class counts) as signed 32-bit halves (pack_pair);
//   2. otherwise: appended to the bin of the target's segment (FarBins).  The term draws a rank from an LDS counter of
//      its destination (a small open-addressed table keyed by segment number), one lane per destination reserves the
//      workgroup's run in the bin with ONE returning atomic, and after a barrier every term is stored at base + rank
//      (the scheme of pass A's tile bins).  k_reg_step sums the bin in LDS.
//   A term that finds neither room in the table nor in the bin goes to grad_acc with two packed 64-bit atomics (the
//   path every far term of an asymmetric link took in rounds 1-2) and leaves a mark beside the bin's counter: the
//   reader looks into grad_acc only then.
// two packed words per term: (gx | gy) and (gz | one count in the byte of the sender's class)
__device__ __forceinline__ void far_term_spill(long long* __restrict__ grad_acc, const FarBins& fb, uint32_t target,
                                               int qx, int qy, int qz, int neighbor_count) {
  unsigned long long* a = reinterpret_cast<unsigned long long*>(&grad_acc[2 * (size_t)target]);
  atomicAdd(&a[0], pack_pair(qx, qy));
  atomicAdd(&a[1], pack_pair(qz, 1 << (8 * (neighbor_count - 1))));
  fb.count[(size_t)(target / kSegB) * kCountStride + 1] = 1u;   // (the reader of that segment looks into grad_acc)
}
#ifndef SMX_LIST_WGS_PER_CU
#define SMX_LIST_WGS_PER_CU 12   // (8 .. 32 measured: profiles/r17_ab_notes.md r26; the chunk lists hold ~3 000 steps at C2, ~10 000 at C3)
#endif
#ifndef SMX_PASS_A_WGS_PER_CU
#define SMX_PASS_A_WGS_PER_CU 8
#endif
#ifndef SMX_EXT_STOP_EVENTS
#define SMX_EXT_STOP_EVENTS 1   // (0: event records as packets of their own, the arrangement up to r22)
#endif
#ifndef SMX_GATE_RELEASE
#define SMX_GATE_RELEASE 0   // (1: a device-scope release in front of the count, on top of the write-through stores -- A/B)
#endif
#ifndef SMX_HANDOVER_FLAGS
#define SMX_HANDOVER_FLAGS 1   // (default of smx_recon_set_handover_mode)
#endif
#ifndef SMX_REG_PRIORITY_HIGH
#define SMX_REG_PRIORITY_HIGH 1
#endif
#ifndef SMX_READY_WAIT_EARLY
#define SMX_READY_WAIT_EARLY 2   // (where the call waits for its input images: 0 = between pass A and the tile kernel (rounds 3-4), 1 = next
                                 // to the wait for the previous map (+0.5 %), 2 = at the very front of the call, in front of the cull step
                                 // (+0.9 % on top: ONE barrier packet between the previous update + create and pass A); profiles/r5_ab_notes.md)
#endif
#ifndef SMX_ACC_WGS_PER_CU
#define SMX_ACC_WGS_PER_CU 5   // (256-lane workgroups, 26 KB of LDS each; <= 96 VGPRs without scratch. 4 / 5 / 6 measured: profiles/r5_ab_notes.md)
#endif
#ifndef SMX_EDGE_ONEPASS
#define SMX_EDGE_ONEPASS 1   // (0: the chunk-by-chunk form of rounds 5-6, kept for A/B -- profiles/r6_ab_notes.md section 11)
#endif
#if SMX_EDGE_ONEPASS
// The edge work of ONE segment (k_reg_accumulate's step; also the tail of the fused pass B): n_act entries, the lane's
// entry of the first chunk in ent_first, the entries of later chunks in later_entries[e] (global work list or LDS).
//
// A segment with more than 256 entries takes several chunks of 256 (they share the LDS sums).  Rounds 5-6 worked the chunks
// one after the other, each with the whole chain -- link records, neighbour gathers, a barrier, the returning atomics that
// reserve the workgroup's runs in the destination bins, a barrier, the far stores -- and the workgroup stamps of
// tools/stamps.py showed what that costs (profiles/r8c_stamps_alone.txt, C2): every workgroup of the launch starts within
// 0.4 us, the 320 + 139 segments with up to 256 entries are through after 7 - 10 us, and the 393 dense segments (769 - 1024
// entries: four chunks) after 24 - 30 us -- the launch lasts as long as four chains in a row while the chip empties.
// Here the chains of a segment's chunks are taken apart and laid side by side:
//   1. entries and link records of ALL chunks requested together (one round trip); the recent slots' ranks;
//   2. the far terms are COUNTED per destination from the link records and the window masks alone (no positions needed) in
//      the LDS table, and the workgroup's runs in the bins are reserved ONCE per segment -- while the first chunk's
//      position gathers are already travelling;
//   3. the chunks' terms are formed one chunk after the other with the next chunk's gathers in flight, and every far term
//      is stored at once at the place it draws from its destination's run (an LDS counter): no barrier between chunks, no
//      far term waits in registers.
// Four barriers per segment instead of three per chunk; the dependent round trips of a dense segment fall from eight to
// two plus what the gathers of four chunks cannot hide behind each other.  The sums are integer sums and the order of the
// records inside a bin is of no consequence (k_reg_step adds them up in LDS): results unchanged bit for bit.
__device__ __forceinline__ void edge_segment(const Surfels& S, const EdgeArgs& ea, DevState* st, uint32_t base, uint32_t n_act,
                                             uint32_t ent_first, const uint32_t* later_entries, const EdgeLds& lds, uint32_t tid) {
  constexpr int kCh = kSegAcc / kBlockAcc;   // chunks of a full segment
  unsigned long long* const lacc = lds.lacc;
  uint32_t* const hkey = lds.hkey; uint32_t* const hcnt = lds.hcnt;
  uint16_t* const lrank = lds.lrank; uint32_t* const rec_wave = lds.rec_wave;
  const float rf2 = ea.rf2, weight = ea.weight;
  long long* const grad_acc = ea.grad_acc;
  float4* const reg_rec = ea.reg_rec;
  const size_t rec_own_offset = ea.rec_own_offset;
  const FarBins& fb = ea.fb;
  const uint32_t n_chunks = (n_act + kBlockAcc - 1u) / kBlockAcc;   // (uniform: 1 .. kCh)
#pragma unroll
  for (int k = 0; k < kSegAcc * 2 / kBlockAcc; ++k) lacc[k * kBlockAcc + tid] = 0;
#pragma unroll
  for (int k = 0; k < kSegAcc / 2 / kBlockAcc; ++k) reinterpret_cast<uint32_t*>(lrank)[k * kBlockAcc + tid] = 0xFFFFFFFFu;
#pragma unroll
  for (int k = 0; k < kFarHash / kBlockAcc; ++k) { hkey[k * kBlockAcc + tid] = kInvalid; hcnt[k * kBlockAcc + tid] = 0; }
  // 1. the lane's entry of every chunk and its link record (idle lanes and chunks: the segment's first slot, no links used)
  uint32_t entq[kCh];
  uint4 tq[kCh];
  entq[0] = tid < n_act ? ent_first : kNoActEntry;
#pragma unroll
  for (int c = 1; c < kCh; ++c)
    entq[c] = ((uint32_t)c < n_chunks && (uint32_t)c * kBlockAcc + tid < n_act) ? later_entries[(uint32_t)c * kBlockAcc + tid] : kNoActEntry;
#pragma unroll
  for (int c = 0; c < kCh; ++c)
    tq[c] = ((uint32_t)c < n_chunks) ? *reinterpret_cast<const uint4*>(S.group(kGroupT, base + (entq[c] != kNoActEntry ? (entq[c] & 1023u) : 0u)))
                                     : make_uint4(kInvalid, kInvalid, kInvalid, kInvalid);
  // (what a chunk's terms need: the slot's own S and N records and one S record per link -- a recent slot every valid
  // neighbour (its own step term, :2238-2256), any other slot only the neighbours inside the window (the terms it pushes);
  // unused links read the slot's own record)
  auto gather_mask = [](uint32_t ent, const uint4& t) -> uint32_t {
    if (ent == kNoActEntry) return 0u;
    uint32_t g = (ent >> 10) & 15u;
    if ((ent >> 14) & 1u) g |= (t.x != kInvalid ? 1u : 0u) | (t.y != kInvalid ? 2u : 0u) | (t.z != kInvalid ? 4u : 0u) | (t.w != kInvalid ? 8u : 0u);
    return g;
  };
  float4 cur_s, cur_n, cur_ts[4];
  {
    const uint32_t i = base + (entq[0] != kNoActEntry ? (entq[0] & 1023u) : 0u);
    const uint32_t g = gather_mask(entq[0], tq[0]);
    const uint32_t nb[4] = {tq[0].x, tq[0].y, tq[0].z, tq[0].w};
    cur_s = *S.group(kGroupS, i);
    cur_n = *S.group(kGroupN, i);
#pragma unroll
    for (int q = 0; q < 4; ++q) cur_ts[q] = *S.group(kGroupS, (g & (1u << q)) ? nb[q] : i);
  }
  // A recent slot's results -- the in-segment sums and its own term -- go to two dense arrays at the slot's RANK among the
  // segment's recent slots, which is its place in the recent list pass B wrote (both lists ascend by slot): the step
  // kernel reads the records of a segment as coalesced runs.  (Two arrays of 16-byte records, not one of 32-byte records:
  // the two halves are ready at different times, and a 16-byte store into a 32-byte record is a partial sector write --
  // WRITE_SIZE booked 32 bytes for each, 30 MB a frame at C2 instead of 15, profiles/r40_WRITE_SIZE.md.)
  uint32_t rank_in_wave[kCh];
#pragma unroll
  for (int c = 0; c < kCh; ++c) {
    const bool rec = entq[c] != kNoActEntry && ((entq[c] >> 14) & 1u);
    const unsigned long long bal = __ballot(rec);
    rank_in_wave[c] = (uint32_t)__popcll(bal & ((1ull << (tid & 63u)) - 1ull));
    if ((tid & 63u) == 0) rec_wave[c * (kBlockAcc / 64) + (tid >> 6)] = (uint32_t)__popcll(bal);
  }
  __syncthreads();   // (the tables are clear, the recent counts of every (chunk, wavefront) in place)
  {
    uint32_t before = 0;   // recent entries in the wavefronts in front of this lane's, chunk-major
#pragma unroll
    for (int c = 0; c < kCh; ++c) {
      uint32_t mine = before;
#pragma unroll
      for (int wv = 0; wv < kBlockAcc / 64; ++wv) {
        const uint32_t n = rec_wave[c * (kBlockAcc / 64) + wv];
        if ((uint32_t)wv < (tid >> 6)) mine += n;
        before += n;
      }
      if (entq[c] != kNoActEntry && ((entq[c] >> 14) & 1u)) lrank[entq[c] & 1023u] = (uint16_t)(mine + rank_in_wave[c]);
    }
  }
  // 2. how many far terms this workgroup has for each destination segment: find (or claim) the destination's entry in the
  // LDS table and count (the same probe sequence finds the entry again in step 3; a destination that finds no room in the
  // table -- 16 probes -- is not in it then either, and its terms go to grad_acc)
#pragma unroll
  for (int c = 0; c < kCh; ++c) {
    if ((uint32_t)c >= n_chunks) break;   // (uniform)
    if (entq[c] == kNoActEntry) continue;
    const uint32_t mask = (entq[c] >> 10) & 15u;
    const uint32_t nb[4] = {tq[c].x, tq[c].y, tq[c].z, tq[c].w};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (!(mask & (1u << q)) || nb[q] - base < (uint32_t)kSegAcc) continue;
      const uint32_t dseg = nb[q] / kSegB;
      uint32_t h = (dseg * 2654435761u) >> 16;
      for (int probe = 0; probe < 16; ++probe) {
        h &= fb.hash_mask;
        const uint32_t seen = atomicCAS(&hkey[h], kInvalid, dseg);
        if (seen == kInvalid || seen == dseg) { atomicAdd(&hcnt[h], 1u); break; }
        ++h;
      }
    }
  }
  __syncthreads();
  // one lane per destination reserves the workgroup's run in that bin; the table then holds the next free place of the run
  // (the lane's reservations are in flight together: written as `hcnt[e] = atomicAdd(..)` in one loop each of them is
  // waited for before the next is sent -- four round trips per segment for a wavefront that has one destination per pass)
  {
    uint32_t dsegk[kFarHash / kBlockAcc], cntk[kFarHash / kBlockAcc], got[kFarHash / kBlockAcc];
#pragma unroll
    for (int k = 0; k < kFarHash / kBlockAcc; ++k) { dsegk[k] = hkey[k * kBlockAcc + tid]; cntk[k] = hcnt[k * kBlockAcc + tid]; }
#pragma unroll
    for (int k = 0; k < kFarHash / kBlockAcc; ++k) {
      got[k] = 0;
      if (dsegk[k] != kInvalid) got[k] = atomicAdd(&fb.count[(size_t)dsegk[k] * kCountStride], cntk[k]);
    }
#pragma unroll
    for (int k = 0; k < kFarHash / kBlockAcc; ++k) keep_if<8>(got[k]);
#pragma unroll
    for (int k = 0; k < kFarHash / kBlockAcc; ++k) if (dsegk[k] != kInvalid) hcnt[k * kBlockAcc + tid] = got[k];
  }
  __syncthreads();
  // 3. the chunks' terms
#pragma unroll 1
  for (uint32_t c = 0; c < n_chunks; ++c) {
    const uint32_t ent = entq[0];
    const uint4 own_t = tq[0];
    // the next chunk's records travel while this chunk's terms are formed
    float4 nxt_s = cur_s, nxt_n = cur_n, nxt_ts[4] = {cur_ts[0], cur_ts[1], cur_ts[2], cur_ts[3]};
    if (c + 1 < n_chunks) {   // (uniform)
      const uint32_t i1 = base + (entq[1] != kNoActEntry ? (entq[1] & 1023u) : 0u);
      const uint32_t g1 = gather_mask(entq[1], tq[1]);
      const uint32_t nb1[4] = {tq[1].x, tq[1].y, tq[1].z, tq[1].w};
      nxt_s = *S.group(kGroupS, i1);
      nxt_n = *S.group(kGroupN, i1);
#pragma unroll
      for (int q = 0; q < 4; ++q) nxt_ts[q] = *S.group(kGroupS, (g1 & (1u << q)) ? nb1[q] : i1);
    }
    const bool act = ent != kNoActEntry;
    const uint32_t rel_own = act ? (ent & 1023u) : 0u;
    const uint32_t mask = act ? ((ent >> 10) & 15u) : 0u;
    const bool rec = act && ((ent >> 14) & 1u);
    const uint32_t i = base + rel_own;
    const uint32_t nb[4] = {own_t.x, own_t.y, own_t.z, own_t.w};
    const uint32_t gmask = gather_mask(ent, own_t);
    if (act) {
      const Vec3 sp = {cur_s.x, cur_s.y, cur_s.z};
      const Vec3 nrm = {cur_n.x, cur_n.y, cur_n.z};
      const float r2 = cur_n.w;
      const int neighbor_count = mask ? __popc(mask) : 1;
      const float factor = 2 * weight / (float)neighbor_count;  // :2153
      int own_count = 0;
      Vec3 rg = {0, 0, 0};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (!(gmask & (1u << q))) continue;
        const Vec3 t = {cur_ts[q].x - sp.x, cur_ts[q].y - sp.y, cur_ts[q].z - sp.z};
        const float nd = nrm.x * t.x + nrm.y * t.y + nrm.z * t.z;
        bool pruned = false;
        if (mask & (1u << q)) {
          const float f = factor * nd;
          const Vec3 term = {f * nrm.x, f * nrm.y, f * nrm.z};   // (the weight term, :2182, travels as the sender's class)
          // the fixed-point channel carries |component| < 16 m (q22_from_float clamps): a huge regularizer_weight or a
          // corrupt position is reported instead of silently bending the gradient
          if (!(fabsf(term.x) < 16.0f && fabsf(term.y) < 16.0f && fabsf(term.z) < 16.0f)) st->reg_saturated = 1u;
          const int qx = q22_from_float(term.x), qy = q22_from_float(term.y), qz = q22_from_float(term.z);
          const uint32_t rel = nb[q] - base;
          if (rel < (uint32_t)kSegAcc) {
            // component-major LDS layout: consecutive lanes (consecutive targets) hit consecutive banks
            atomicAdd(&lacc[rel], pack_pair(qx, qy));
            atomicAdd(&lacc[kSegAcc + rel], pack_pair(qz, 1 << (8 * (neighbor_count - 1))));
          } else {
            // the destination's entry in the table (step 2 put it there, or found no room), the term's place in the run
            const uint32_t dseg = nb[q] / kSegB;
            uint32_t h = (dseg * 2654435761u) >> 16;
            uint32_t pos = kInvalid;
            for (int probe = 0; probe < 16; ++probe) {
              h &= fb.hash_mask;
              if (hkey[h] == dseg) { pos = atomicAdd(&hcnt[h], 1u); break; }
              ++h;
            }
            if (pos < fb.cap)
              out_store16<SMX_ST_FARBIN>(&fb.rec[(size_t)dseg * fb.cap + pos],
                                         make_uint4((nb[q] % kSegB) | ((uint32_t)(neighbor_count - 1) << 10), (uint32_t)qx, (uint32_t)qy, (uint32_t)qz));
            else
              far_term_spill(grad_acc, fb, nb[q], qx, qy, qz, neighbor_count);
          }
          const float d2 = t.x * t.x + t.y * t.y + t.z * t.z;
          if (d2 > rf2 * r2) { S.set_neighbor(i, q, kInvalid); pruned = true; }  // :2190-2192
        }
        // the slot's own regulariser term (RegularizeSurfelsCUDAKernel :2238-2256 sees the row after the pruning
        // above): same neighbour positions, same n.t product, so it is formed here and k_reg_step gathers nothing
        if (rec && !pruned) {
          ++own_count;
          rg.x = rg.x - nd * nrm.x; rg.y = rg.y - nd * nrm.y; rg.z = rg.z - nd * nrm.z;
        }
      }
      // (second half of the slot's dense record; the first half -- the in-segment sums -- follows when the segment is through)
      if (rec) out_store16<SMX_ST_REGREC>(&reg_rec[rec_own_offset + (size_t)(base + lrank[rel_own])], make_float4(rg.x, rg.y, rg.z, __int_as_float(own_count)));
    }
#pragma unroll
    for (int k = 0; k + 1 < kCh; ++k) { entq[k] = entq[k + 1]; tq[k] = tq[k + 1]; }
    cur_s = nxt_s; cur_n = nxt_n;
#pragma unroll
    for (int q = 0; q < 4; ++q) cur_ts[q] = nxt_ts[q];
  }
  __syncthreads();   // (every chunk's in-segment terms are in)
  // The in-segment sums of the recent slots: first half of their dense records.  (Rounds 3-4 stored the sums of EVERY slot
  // of the segment, zeros included -- 16 KB of full lines per workgroup, 39 MB a frame at C2, of which the step kernel read
  // the recent slots' 7 MB: profiles/r31_WRITE_SIZE.md.)
#pragma unroll
  for (int k = 0; k < kSegAcc / kBlockAcc; ++k) {
    const uint32_t rel = k * kBlockAcc + tid;
    const uint32_t rank = lrank[rel];
    if (rank == 0xFFFFu) continue;
    const unsigned long long v0 = lacc[rel], v1 = lacc[kSegAcc + rel];
    out_store16<SMX_ST_REGREC>(&reg_rec[(size_t)(base + rank)], v0, v1);
  }
}
#else
// The edge work of ONE segment (k_reg_accumulate's step; also the tail of the fused pass B): n_act entries, the lane's
// entry of the first chunk in ent_first, the entries of later chunks in later_entries[e] (global work list or LDS).
__device__ __forceinline__ void edge_segment(const Surfels& S, const EdgeArgs& ea, DevState* st, uint32_t base, uint32_t n_act,
                                             uint32_t ent_first, const uint32_t* later_entries, const EdgeLds& lds, uint32_t tid) {
  unsigned long long* const lacc = lds.lacc;
  uint32_t* const hkey = lds.hkey; uint32_t* const hcnt = lds.hcnt;
  uint16_t* const lrank = lds.lrank; uint32_t* const rec_wave = lds.rec_wave;
  const float rf2 = ea.rf2, weight = ea.weight;
  long long* const grad_acc = ea.grad_acc;
  float4* const reg_rec = ea.reg_rec;
  const size_t rec_own_offset = ea.rec_own_offset;
  const FarBins& fb = ea.fb;
#pragma unroll
  for (int k = 0; k < kSegAcc * 2 / kBlockAcc; ++k) lacc[k * kBlockAcc + tid] = 0;
#pragma unroll
  for (int k = 0; k < kSegAcc / 2 / kBlockAcc; ++k) reinterpret_cast<uint32_t*>(lrank)[k * kBlockAcc + tid] = 0xFFFFFFFFu;
  uint32_t rec_before = 0;   // recent entries in the chunks in front of this one
  // A segment with more than 256 entries takes several chunks, one after the other (they share the LDS sums): the next
  // chunk's entries are requested at the top of this one, and its slots' own records behind this chunk's first barrier,
  // so that a chunk waits for its link records and its bin reservations only -- two dependent round trips instead of four
  // (the dense segments, four chunks each, are what the launch lasts: 40 us alone without this, profiles/r5_ab_notes.md).
  uint32_t ent = tid < n_act ? ent_first : kNoActEntry;
  uint4 own_t = *reinterpret_cast<const uint4*>(S.group(kGroupT, base + (ent != kNoActEntry ? (ent & 1023u) : 0u)));
  float4 own_s = *S.group(kGroupS, base + (ent != kNoActEntry ? (ent & 1023u) : 0u));
  float4 own_n = *S.group(kGroupN, base + (ent != kNoActEntry ? (ent & 1023u) : 0u));
#pragma unroll 1
  for (uint32_t c0 = 0; c0 < n_act; c0 += kBlockAcc) {
  const bool more = c0 + kBlockAcc < n_act;   // (uniform)
  uint32_t ent_following = kNoActEntry;
  if (more && c0 + kBlockAcc + tid < n_act) ent_following = later_entries[c0 + kBlockAcc + tid];
  if (c0) __syncthreads();   // (the previous chunk's far stores have read the table)
#pragma unroll
  for (int k = 0; k < kFarHash / kBlockAcc; ++k) { hkey[k * kBlockAcc + tid] = kInvalid; hcnt[k * kBlockAcc + tid] = 0; }
  const bool act = ent != kNoActEntry;
  const uint32_t rel_own = act ? (ent & 1023u) : 0u;
  const uint32_t mask = act ? ((ent >> 10) & 15u) : 0u;
  const bool rec = act && ((ent >> 14) & 1u);
  // A recent slot's results -- the in-segment sums and its own term -- go to two dense arrays at the slot's RANK among the
  // segment's recent slots, which is its place in the recent list pass B wrote (both lists ascend by slot): the step
  // kernel reads the records of a segment as coalesced runs.  (Two arrays of 16-byte records, not one of 32-byte records:
  // the two halves are ready at different times, and a 16-byte store into a 32-byte record is a partial sector write --
  // WRITE_SIZE booked 32 bytes for each, 30 MB a frame at C2 instead of 15, profiles/r40_WRITE_SIZE.md.)
  const unsigned long long bal = __ballot(rec);
  uint32_t rec_rank = rec_before + (uint32_t)__popcll(bal & ((1ull << (tid & 63u)) - 1ull));
  if ((tid & 63u) == 0) rec_wave[tid >> 6] = (uint32_t)__popcll(bal);
  // (the slot's own records: requested a chunk ahead; idle lanes hold the segment's first slot's)
  const uint32_t i = base + rel_own;
  __syncthreads();
#pragma unroll
  for (int wv = 0; wv < kBlockAcc / 64; ++wv) {
    const uint32_t n = rec_wave[wv];
    if ((uint32_t)wv < (tid >> 6)) rec_rank += n;
    rec_before += n;
  }
  if (rec) lrank[rel_own] = (uint16_t)rec_rank;
  const uint32_t nb[4] = {own_t.x, own_t.y, own_t.z, own_t.w};
  // a recent slot needs every valid neighbour (its own step term, :2238-2256), any other slot only the neighbours inside
  // the window (the terms it pushes): one 16-byte record per link, all requested before the first is used
  uint32_t gmask = mask;
  if (rec) {
#pragma unroll
    for (int q = 0; q < 4; ++q) if (nb[q] != kInvalid) gmask |= 1u << q;
  }
  float4 ts[4];
#pragma unroll
  for (int q = 0; q < 4; ++q)   // the neighbour's smooth position; unused links read the slot's own record
    ts[q] = *S.group(kGroupS, (gmask & (1u << q)) ? nb[q] : i);
  // far terms wait in registers for their place in the destination's bin: (table entry | rank << 10, target, q22 x 3)
  uint32_t far_where[4], far_target[4];
  int far_q[4][3];
#pragma unroll
  for (int q = 0; q < 4; ++q) far_where[q] = kInvalid;
  if (act) {
    const Vec3 sp = {own_s.x, own_s.y, own_s.z};
    const Vec3 nrm = {own_n.x, own_n.y, own_n.z};
    const float r2 = own_n.w;
    const int neighbor_count = mask ? __popc(mask) : 1;
    const float factor = 2 * weight / (float)neighbor_count;  // :2153
    int own_count = 0;
    Vec3 rg = {0, 0, 0};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (!(gmask & (1u << q))) continue;
      const Vec3 t = {ts[q].x - sp.x, ts[q].y - sp.y, ts[q].z - sp.z};
      const float nd = nrm.x * t.x + nrm.y * t.y + nrm.z * t.z;
      bool pruned = false;
      if (mask & (1u << q)) {
        const float f = factor * nd;
        const Vec3 term = {f * nrm.x, f * nrm.y, f * nrm.z};   // (the weight term, :2182, travels as the sender's class)
        // the fixed-point channel carries |component| < 16 m (q22_from_float clamps): a huge regularizer_weight or a
        // corrupt position is reported instead of silently bending the gradient
        if (!(fabsf(term.x) < 16.0f && fabsf(term.y) < 16.0f && fabsf(term.z) < 16.0f)) st->reg_saturated = 1u;
        const int qx = q22_from_float(term.x), qy = q22_from_float(term.y), qz = q22_from_float(term.z);
        const uint32_t rel = nb[q] - base;
        if (rel < (uint32_t)kSegAcc) {
          // component-major LDS layout: consecutive lanes (consecutive targets) hit consecutive banks
          atomicAdd(&lacc[rel], pack_pair(qx, qy));
          atomicAdd(&lacc[kSegAcc + rel], pack_pair(qz, 1 << (8 * (neighbor_count - 1))));
        } else {
          // find (or claim) the destination's entry in the LDS table, then draw a rank
          const uint32_t dseg = nb[q] / kSegB;
          uint32_t h = (dseg * 2654435761u) >> 16;
          uint32_t where = kInvalid;
          for (int probe = 0; probe < 16; ++probe) {
            h &= fb.hash_mask;
            const uint32_t seen = atomicCAS(&hkey[h], kInvalid, dseg);
            if (seen == kInvalid || seen == dseg) { where = h | (atomicAdd(&hcnt[h], 1u) << 10); break; }
            ++h;
          }
          if (where == kInvalid) {
            far_term_spill(grad_acc, fb, nb[q], qx, qy, qz, neighbor_count);
          } else {
            far_where[q] = where; far_target[q] = nb[q] | ((uint32_t)(neighbor_count - 1) << 30);
            far_q[q][0] = qx; far_q[q][1] = qy; far_q[q][2] = qz;
          }
        }
        const float d2 = t.x * t.x + t.y * t.y + t.z * t.z;
        if (d2 > rf2 * r2) { S.set_neighbor(i, q, kInvalid); pruned = true; }  // :2190-2192
      }
      // the slot's own regulariser term (RegularizeSurfelsCUDAKernel :2238-2256 sees the row after the pruning
      // above): same neighbour positions, same n.t product, so it is formed here and k_reg_step gathers nothing
      if (rec && !pruned) {
        ++own_count;
        rg.x = rg.x - nd * nrm.x; rg.y = rg.y - nd * nrm.y; rg.z = rg.z - nd * nrm.z;
      }
    }
    // (second half of the slot's dense record; the first half -- the in-segment sums -- follows when the segment is through)
    if (rec) out_store16<SMX_ST_REGREC>(&reg_rec[rec_own_offset + (size_t)(base + rec_rank)], make_float4(rg.x, rg.y, rg.z, __int_as_float(own_count)));
  }
  __syncthreads();
  // one lane per destination reserves the workgroup's run in that bin; the table then holds the run's start
#pragma unroll
  for (int k = 0; k < kFarHash / kBlockAcc; ++k) {
    const uint32_t e = k * kBlockAcc + tid;
    const uint32_t dseg = hkey[e];
    if (dseg != kInvalid) hcnt[e] = atomicAdd(&fb.count[(size_t)dseg * kCountStride], hcnt[e]);
  }
  // the next chunk's own records travel while the reservations return and the far records are stored
  uint4 nxt_t = own_t; float4 nxt_s = own_s, nxt_n = own_n;
  if (more) {
    const uint32_t in = base + (ent_following != kNoActEntry ? (ent_following & 1023u) : 0u);
    nxt_t = *reinterpret_cast<const uint4*>(S.group(kGroupT, in));
    nxt_s = *S.group(kGroupS, in);
    nxt_n = *S.group(kGroupN, in);
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const uint32_t where = far_where[q];
    if (where == kInvalid) continue;
    const uint32_t target = far_target[q] & 0x3FFFFFFFu, cls = far_target[q] >> 30;
    const uint32_t pos = hcnt[where & 1023u] + (where >> 10);
    if (pos < fb.cap)
      out_store16<SMX_ST_FARBIN>(&fb.rec[(size_t)(target / kSegB) * fb.cap + pos],
                                 make_uint4((target % kSegB) | (cls << 10), (uint32_t)far_q[q][0], (uint32_t)far_q[q][1], (uint32_t)far_q[q][2]));
    else
      far_term_spill(grad_acc, fb, target, far_q[q][0], far_q[q][1], far_q[q][2], (int)cls + 1);
  }
  ent = ent_following; own_t = nxt_t; own_s = nxt_s; own_n = nxt_n;
  }   // chunks
  // The in-segment sums of the recent slots: first half of their dense records, once every chunk's terms are in (the
  // barriers above).  (Rounds 3-4 stored the sums of EVERY slot of the segment, zeros included -- 16 KB of full lines per
  // workgroup, 39 MB a frame at C2, of which the step kernel read the recent slots' 7 MB: profiles/r31_WRITE_SIZE.md.)
#pragma unroll
  for (int k = 0; k < kSegAcc / kBlockAcc; ++k) {
    const uint32_t rel = k * kBlockAcc + tid;
    const uint32_t rank = lrank[rel];
    if (rank == 0xFFFFu) continue;
    const unsigned long long v0 = lacc[rel], v1 = lacc[kSegAcc + rel];
    out_store16<SMX_ST_REGREC>(&reg_rec[(size_t)(base + rank)], v0, v1);
  }
}

#endif   // SMX_EDGE_ONEPASS
__global__ void __launch_bounds__(kBlockAcc, SMX_ACC_WGS_PER_CU)   // (second argument: wavefronts per SIMD = workgroups per CU here)
k_reg_accumulate(Surfels S, float rf2, float weight, long long* __restrict__ grad_acc,
                 float4* __restrict__ reg_rec, size_t rec_own_offset, FarBins fb,
                 const uint32_t* __restrict__ act_list, Chunks acc, DevState* st, unsigned long long* ts, unsigned long long* stamps,
                 uint32_t max_entries /* kSegAcc; less: TIMING ONLY (debug_skip bit 7), the map is wrong afterwards */) {
  SMX_SET_WAVE_PRIO();
  ts_begin(ts, kTsAccBegin);
#ifdef SMX_STAMPS
  // (diagnosis, tools/acc_stamps.py: per workgroup the wall clock at entry and exit, steps, entries, the largest step)
  const unsigned long long t_in = wall_clock64();
  uint32_t dbg_steps = 0, dbg_entries = 0, dbg_max = 0;
#endif
  __shared__ unsigned long long lacc[kSegAcc * 2];  // per target: (gx | gy), (gz | sender classes)
  __shared__ uint32_t hkey[kFarHash], hcnt[kFarHash];   // far destinations of this workgroup: segment, terms -> base in the bin
  __shared__ uint16_t lrank[kSegAcc];                // per slot of the segment: its rank among the recent slots, or 0xFFFF
  __shared__ uint32_t rec_wave[kSegAcc / 64];        // recent entries per (chunk, wavefront) of the segment
  // A walk over the segments pass B listed (acc_chunks: a recent slot or an edge into the window), on a grid the size
  // of the chip.  Round 5: a step no longer visits the segment's 1024 slots (512 lanes x 2, four fifths of them idle in
  // the typical listed segment) but the ENTRIES of the work list pass B wrote for it -- the slots that have work, dense,
  // one per lane, 256 at a time -- so the kernel needs half the registers per lane and three times as many steps are in
  // flight per compute unit; the mask and flag bytes (2 KB per step) shrink to 4 bytes per entry.  The next step's
  // descriptor and the first entries of ITS list travel while this step is worked on.
  uint32_t desc, cntv;
  const uint32_t n_steps = walk_begin<true>(acc, 0u, blockIdx.x, desc, cntv);
  // (an unused descriptor is 0 or an old one: any segment of the map, any entry count -- whatever is read is ignored)
  uint32_t ent_next = (blockIdx.x < n_steps && threadIdx.x <= (desc >> 22))
                          ? act_list[(size_t)(desc & 0x003FFFFFu) * kSegAcc + threadIdx.x] : kNoActEntry;
  bool lds_used = false;
#pragma unroll 1
  for (uint32_t w = blockIdx.x; w < n_steps; w += gridDim.x) {
  const uint32_t cur = desc;
  desc = walk_next<true>(acc, w + gridDim.x, n_steps);
  // (per step: a lane number the optimiser cannot see through keeps the per-lane addresses out of loop-carried registers)
  uint32_t tid = threadIdx.x;
  asm volatile("" : "+v"(tid));
  const uint32_t ent_first = ent_next;
  ent_next = (w + gridDim.x < n_steps && tid <= (desc >> 22)) ? act_list[(size_t)(desc & 0x003FFFFFu) * kSegAcc + tid] : kNoActEntry;
  if (!walk_step_valid(w, cntv)) continue;
  const uint32_t seg_id = cur & 0x003FFFFFu, n_act = min((cur >> 22) + 1u, max_entries);
  const uint32_t base = seg_id * kSegAcc;
#ifdef SMX_STAMPS
  ++dbg_steps; dbg_entries += n_act; dbg_max = max(dbg_max, n_act);
#endif
  if (lds_used) __syncthreads();   // (the previous step's readers of the tables are done)
  lds_used = true;
  {
    EdgeArgs ea; ea.rf2 = rf2; ea.weight = weight; ea.grad_acc = grad_acc; ea.reg_rec = reg_rec; ea.rec_own_offset = rec_own_offset; ea.fb = fb;
    EdgeLds lds; lds.lacc = lacc; lds.hkey = hkey; lds.hcnt = hcnt; lds.lrank = lrank; lds.rec_wave = rec_wave;
    edge_segment(S, ea, st, base, n_act, ent_first, act_list + (size_t)base, lds, tid);
  }
  }
#ifdef SMX_STAMPS
  if (stamps && threadIdx.x == 0 && blockIdx.x < 8192) {
    unsigned long long* o = stamps + (size_t)blockIdx.x * 16;
    o[0] = t_in; o[1] = wall_clock64(); o[2] = dbg_steps; o[3] = dbg_entries; o[4] = dbg_max; o[5] = n_steps;
  }
#endif
}

// Rebuilds the flag table for an arbitrary (frame, window): used when Regularize() is called with other
// parameters than the last Integrate(), and after a state upload.
__global__ void __launch_bounds__(kBlock)
k_rebuild_flags(Surfels S, uint32_t frame, int reg_window, uint8_t* __restrict__ flags8, const DevState* st) {
  const uint32_t N = st->surfel_count;
  for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < N; i += gridDim.x * kBlock)
    flags8[i] = make_flags(S.u(kLastUpdateStamp, i), S.u(kColor, i), frame, reg_window);
}

// RegularizeSurfelsCUDAKernel, kernels.cu:2197-2290, over the recent list: one walk step per segment that holds recent
// slots (descriptor = segment | (recent slots - 1) << 22), four slots per lane.  The step's workgroup first sums the
// far terms of the segment's bin in LDS (FarBins; same packed words as k_reg_accumulate's in-segment sums) and empties
// the bin.
constexpr int kStepSub = kSegB / kBlock;
__global__ void __launch_bounds__(kBlock)
k_reg_step(Surfels S, float weight, long long* __restrict__ grad_acc, const float4* __restrict__ reg_rec, size_t rec_own_offset,
           FarBins fb, Lists L, DevState* st, unsigned long long* ts) {
  SMX_SET_WAVE_PRIO();
  __shared__ unsigned long long lfar[kSegB * 2];   // per target of the segment: (gx | gy), (gz | sender classes)
  ts_begin(ts, kTsStepBegin);
  uint32_t desc, cntv;
  const uint32_t n_steps = walk_begin<true>(L.rec_chunks, 0u, blockIdx.x, desc, cntv);
  bool lds_used = false;
  for (uint32_t w = blockIdx.x; w < n_steps; w += gridDim.x) {
    const uint32_t cur = desc;
    desc = walk_next<true>(L.rec_chunks, w + gridDim.x, n_steps);   // (the next step's descriptor travels while this one is worked on)
    if (!walk_step_valid(w, cntv)) continue;
    const uint32_t seg = cur & 0x003FFFFFu, total = (cur >> 22) + 1u;
    const uint32_t seg_base = seg * kSegB;
    const uint2 bin_state = *reinterpret_cast<const uint2*>(&fb.count[(size_t)seg * kCountStride]);
    const uint32_t n_far = min(bin_state.x, fb.cap);
    const bool spilled = bin_state.y != 0;
    // the slots' own records: requested before the bin is summed
    uint32_t idx[kStepSub];
    bool on[kStepSub];
#pragma unroll
    for (int sub = 0; sub < kStepSub; ++sub) {
      const uint32_t e = sub * kBlock + threadIdx.x;
      on[sub] = e < total;
      idx[sub] = on[sub] ? L.recent_list[seg_base + e] : seg_base;
    }
    keep_if<32>(bin_state.x); keep_if<32>(bin_state.y);   // (the bin's state and the list entries: one round trip, not two)
#pragma unroll
    for (int sub = 0; sub < kStepSub; ++sub) keep_if<32>(idx[sub]);
    float4 rp[kStepSub], rs[kStepSub], rn[kStepSub], rgr[kStepSub];
    ulonglong2 rl[kStepSub];
#pragma unroll
    for (int sub = 0; sub < kStepSub; ++sub) {
      const uint32_t i = idx[sub];
      rp[sub] = *S.group(kGroupP, i); rs[sub] = *S.group(kGroupS, i); rn[sub] = *S.group(kGroupN, i);
      // the slot's dense record (k_reg_accumulate): in-segment sums, own term + neighbour count -- entry e of the segment
      const size_t e = (size_t)seg_base + (on[sub] ? (uint32_t)(sub * kBlock) + threadIdx.x : 0u);
      rl[sub] = *reinterpret_cast<const ulonglong2*>(&reg_rec[e]);
      rgr[sub] = reg_rec[rec_own_offset + e];
    }
    if (n_far) {
      if (lds_used) __syncthreads();   // (the previous step's readers are done)
#pragma unroll
      for (int k = 0; k < kSegB * 2 / kBlock; ++k) lfar[k * kBlock + threadIdx.x] = 0;
      __syncthreads();
      const uint4* bin = fb.rec + (size_t)seg * fb.cap;
      for (uint32_t k0 = 0; k0 < n_far; k0 += 4 * kBlock) {
        uint4 t[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const uint32_t k = k0 + u * kBlock + threadIdx.x; t[u] = bin[k < n_far ? k : 0]; }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (k0 + u * kBlock + threadIdx.x >= n_far) continue;
          const uint32_t rel = t[u].x & 1023u, cls = t[u].x >> 10;
          atomicAdd(&lfar[rel], pack_pair((int)t[u].y, (int)t[u].z));
          atomicAdd(&lfar[kSegB + rel], pack_pair((int)t[u].w, 1 << (8 * cls)));
        }
      }
      __syncthreads();
      lds_used = true;
    }
    // (n_far is uniform: word 0 only changes behind the barriers above.  With an empty bin nothing has ordered the other
    // wavefronts' loads of bin_state before the reset below -- they drift freely across barrier-free steps -- and a
    // wavefront that read the spill mark late would find it gone and drop its slots' grad_acc terms.)
    if (!n_far) __syncthreads();
    if ((bin_state.x | bin_state.y) && threadIdx.x == 0)
      *reinterpret_cast<uint2*>(&fb.count[(size_t)seg * kCountStride]) = make_uint2(0u, 0u);
#pragma unroll
    for (int sub = 0; sub < kStepSub; ++sub) {
      if (!on[sub]) continue;
      const uint32_t i = idx[sub];
      const Vec3 mp = {rp[sub].x, rp[sub].y, rp[sub].z};
      const Vec3 sp = {rs[sub].x, rs[sub].y, rs[sub].z};
      // exact fixed-point sums: terms from the own segment (summed in LDS by k_reg_accumulate, stored plainly), from
      // other segments (the bin) and, rarely, from grad_acc; integer addition, so the split does not matter
      ulonglong2 acc2 = rl[sub];
      if (n_far) { acc2.x += lfar[i - seg_base]; acc2.y += lfar[kSegB + (i - seg_base)]; }
      if (spilled) {
        ulonglong2* ap = reinterpret_cast<ulonglong2*>(&grad_acc[2 * (size_t)i]);
        const ulonglong2 a = *ap;
        if (a.x | a.y) *ap = make_ulonglong2(0, 0);
        acc2.x += a.x; acc2.y += a.y;
      }
      long long sum[3];
      int ylo, cls;
      unpack_pair((long long)acc2.x, sum[0], ylo);
      unpack_pair((long long)acc2.y, sum[2], cls);
      sum[1] = ylo;
      const uint32_t senders[4] = {(uint32_t)cls & 255u, ((uint32_t)cls >> 8) & 255u, ((uint32_t)cls >> 16) & 255u, (uint32_t)cls >> 24};
      // the packed class counters are bytes (the top one signed): far more senders than any real map produces, but a
      // count near the limit may already have carried -- reported, never silent
      if (senders[0] >= 100u || senders[1] >= 100u || senders[2] >= 100u || senders[3] >= 100u) st->reg_saturated = 1u;
      // sum over the senders of weight / (sender's neighbour count) (:2182), exact: per class, count x 2^-32 quotient
      long long wsum_q = 0;
#pragma unroll
      for (int cnt = 1; cnt <= 4; ++cnt)
        if (senders[cnt - 1]) wsum_q += (long long)senders[cnt - 1] * q_from_float(weight / (float)cnt);
      const float acc[4] = {q22_to_float(sum[0]), q22_to_float(sum[1]), q22_to_float(sum[2]), q_to_float(wsum_q)};
      Vec3 grad = {2 * (sp.x - mp.x) + acc[0], 2 * (sp.y - mp.y) + acc[1], 2 * (sp.z - mp.z) + acc[2]};
      // own term and neighbour count: formed by k_reg_accumulate from the pre-step smooth positions
      const Vec3 rg = {rgr[sub].x, rgr[sub].y, rgr[sub].z};
      const int neighbor_count = __float_as_int(rgr[sub].w);
      if (neighbor_count > 0) {
        const float factor = 2 * weight / (float)neighbor_count;
        grad.x = grad.x + factor * rg.x; grad.y = grad.y + factor * rg.y; grad.z = grad.z + factor * rg.z;
      }
      const float wsum = 1 + weight + acc[3];  // :2267
      const float kStep = 0.5f / wsum;
      const float max_step = 1.0f * sqrtf(rn[sub].w);
      const float step_len = kStep * sqrtf(grad.x * grad.x + grad.y * grad.y + grad.z * grad.z);
      float step = kStep;
      if (step_len > max_step) step = max_step / step_len * kStep;
      // No kernel reads another slot's smooth position after k_reg_accumulate, so the result goes straight to the
      // S record: the reference's parking rows and RegularizeSurfelsCUDAUpdateKernel (:2283-2308) are not needed.
      if (L.dirty8) L.dirty8[i] = 1;
      // (ONE 16-byte store, the unused fourth word included: three scalar stores leave four bytes of every record clean, so
      // no 64-byte sector of the S array is ever fully dirty and every write-back is a partial one -- sparse stores retire
      // at 21 G/s = 0.7 TB/s on this chip against 5 TB/s for full sectors, profiles/r17_counter_calibration.md)
      out_store16<SMX_ST_STEP>(S.group(kGroupS, i), make_float4(sp.x - step * grad.x, sp.y - step * grad.y, sp.z - step * grad.z, rs[sub].w));
    }
  }
  ts_end(ts, kTsRegEnd, blockIdx.x, min(gridDim.x, n_steps));   // (the workgroups that walked the last steps of the first round)
}

// RegularizeSurfelsCUDACopyOnlyKernel (:2310-2327), over the recent list.
__global__ void __launch_bounds__(kBlock)
k_reg_copy_raw(Surfels S, Lists L, const DevState* st, unsigned long long* ts) {
  uint32_t desc, cntv;
  const uint32_t n_steps = walk_begin<true>(L.rec_chunks, 0u, blockIdx.x, desc, cntv);
  ts_end(ts, kTsRegEnd, blockIdx.x, min(gridDim.x, n_steps));   // (copy-only regulariser: the last workgroups STARTED)
  for (uint32_t w = blockIdx.x; w < n_steps; w += gridDim.x) {
    desc = walk_next<true>(L.rec_chunks, w, n_steps);
    if (!walk_step_valid(w, cntv)) continue;
    const uint32_t seg = desc & 0x003FFFFFu, total = (desc >> 22) + 1u;
    for (uint32_t e = threadIdx.x; e < total; e += kBlock) {
      const uint32_t i = L.recent_list[seg * kSegB + e];
      if (L.dirty8) L.dirty8[i] = 1;
      const float4 p4 = *S.group(kGroupP, i);
      *S.group(kGroupS, i) = make_float4(p4.x, p4.y, p4.z, 0.0f);
    }
  }
}

__global__ void k_reset_recent(DevState* st, int stats, uint32_t* rec_chunk_count) {   // (one workgroup of kSubLists threads)
  if (stats && threadIdx.x == 0) { st->recent_count = 0; st->n_edges = 0; st->n_window_edges = 0; st->n_contributors = 0; }
  if (rec_chunk_count) { rec_chunk_count[threadIdx.x * kCountStride] = 0; rec_chunk_count[threadIdx.x * kCountStride + kAccCount] = 0; }
}

// ---- changed-surfel delta for the mesher ---------------------------------------------------------
// Three small kernels over the dirty bytes: per-segment counts, one-workgroup scan of the counts, gather of
// (slot index, the eight transferred attributes) into packed rows + clearing of the marks.
__global__ void __launch_bounds__(kBlock)
k_delta_count(const uint8_t* __restrict__ dirty8, uint32_t* __restrict__ seg_count, const DevState* st) {
  __shared__ uint32_t wave_tot[kBlock / 64];
  const uint32_t N = st->surfel_count, i0 = blockIdx.x * kSeg + threadIdx.x * 4;
  uint32_t mine = 0;
  if (i0 < N) {
    const uchar4 d = *reinterpret_cast<const uchar4*>(&dirty8[i0]);
    mine = (d.x != 0) + (i0 + 1 < N && d.y != 0) + (i0 + 2 < N && d.z != 0) + (i0 + 3 < N && d.w != 0);
  }
  uint32_t total;
  (void)block_excl_scan(mine, wave_tot, total);
  if (threadIdx.x == 0) seg_count[blockIdx.x] = total;
}
__global__ void __launch_bounds__(1024)
k_delta_scan(uint32_t* __restrict__ seg_count, int nseg, uint32_t* __restrict__ total_out) {
  // exclusive scan in place (the segment count is a few thousand)
  __shared__ uint32_t wave_tot[1024 / 64];
  __shared__ uint32_t carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < nseg; base += 1024) {
    const int b = base + threadIdx.x;
    const uint32_t v = (b < nseg) ? seg_count[b] : 0;
    uint32_t total;
    const uint32_t excl = block_excl_scan<1024 / 64>(v, wave_tot, total);
    if (b < nseg) seg_count[b] = carry + excl;
    __syncthreads();
    if (threadIdx.x == 0) carry += total;
    __syncthreads();
  }
  if (threadIdx.x == 0) *total_out = carry;
}
__global__ void __launch_bounds__(kBlock)
k_delta_gather(Surfels S, uint8_t* __restrict__ dirty8, const uint32_t* __restrict__ seg_offset, float* __restrict__ out,
               uint32_t total, const DevState* st) {
  __shared__ uint32_t wave_tot[kBlock / 64];
  const uint32_t N = st->surfel_count, i0 = blockIdx.x * kSeg + threadIdx.x * 4;
  uint32_t bits = 0;
  if (i0 < N) {
    const uchar4 d = *reinterpret_cast<const uchar4*>(&dirty8[i0]);
    bits = (d.x != 0 ? 1u : 0u) | ((i0 + 1 < N && d.y != 0) ? 2u : 0u) | ((i0 + 2 < N && d.z != 0) ? 4u : 0u) |
           ((i0 + 3 < N && d.w != 0) ? 8u : 0u);
    if (bits) *reinterpret_cast<uchar4*>(&dirty8[i0]) = make_uchar4(0, 0, 0, 0);
  }
  uint32_t seg_total;
  uint32_t off = seg_offset[blockIdx.x] + block_excl_scan((uint32_t)__popc(bits), wave_tot, seg_total);
  const int want[8] = {kSmoothX, kSmoothY, kSmoothZ, kRadiusSq, kNormalX, kNormalY, kNormalZ, kLastUpdateStamp};  // cc:348-358
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (bits & (1u << j)) {
      const uint32_t i = i0 + j;
      out[off] = __uint_as_float(i);
#pragma unroll
      for (int k = 0; k < 8; ++k) out[(size_t)(k + 1) * total + off] = S.f(want[k], i);
      ++off;
    }
}

// Boundary conversion between the grouped records and the reference's row layout: out[k][i] = row rows[k] of
// slot i (pack) and back (unpack).  Rows without storage read as 0.
struct RowList { int n; int rows[kRows]; };
__global__ void __launch_bounds__(kBlock)
k_pack_rows(Surfels S, RowList rl, float* __restrict__ out, uint32_t count) {
  for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < count; i += gridDim.x * kBlock)
    for (int k = 0; k < rl.n; ++k) {
      const int g = row_group(rl.rows[k]), sub = row_sub(rl.rows[k]);
      out[(size_t)k * count + i] = g < 0 ? 0.0f : S.base[S.quad(g, i) * 4 + sub];
    }
}
__global__ void __launch_bounds__(kBlock)
k_unpack_rows(Surfels S, RowList rl, const float* __restrict__ in, uint32_t count) {
  for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < count; i += gridDim.x * kBlock)
    for (int k = 0; k < rl.n; ++k) {
      const int g = row_group(rl.rows[k]), sub = row_sub(rl.rows[k]);
      if (g >= 0) S.base[S.quad(g, i) * 4 + sub] = in[(size_t)k * count + i];
    }
}

// ExportVerticesCUDAKernel, kernels.cu:2412-2433
__global__ void __launch_bounds__(kBlock)
k_export(Surfels S, float* __restrict__ pos, uint8_t* __restrict__ col, const DevState* st) {
  const uint32_t N = st->surfel_count;
  for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < N; i += gridDim.x * kBlock) {
    const bool merged = S.f(kRadiusSq, i) < 0;
    const float nanv = __builtin_nanf("");
    pos[3 * (size_t)i + 0] = merged ? nanv : S.f(kSmoothX, i);
    pos[3 * (size_t)i + 1] = merged ? nanv : S.f(kSmoothY, i);
    pos[3 * (size_t)i + 2] = merged ? nanv : S.f(kSmoothZ, i);
    const uint32_t c = S.u(kColor, i);
    col[3 * (size_t)i + 0] = (uint8_t)(c & 255u);
    col[3 * (size_t)i + 1] = (uint8_t)((c >> 8) & 255u);
    col[3 * (size_t)i + 2] = (uint8_t)((c >> 16) & 255u);
  }
}

// Candidate lists for the mesher: the rows the neighbour index is built from (smooth position,
// NaN for merged slots so that the index leaves them out) and the per-query (position, radius^2) of a list of slots.
__global__ void __launch_bounds__(kBlock)
k_index_rows(Surfels S, float* __restrict__ out, uint32_t count) {
  const float nanv = __builtin_nanf("");
  for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < count; i += gridDim.x * kBlock) {
    const float4 s = *S.group(kGroupS, i);
    const bool merged = S.f(kRadiusSq, i) < 0;
    out[i] = merged ? nanv : s.x;
    out[(size_t)count + i] = merged ? nanv : s.y;
    out[(size_t)2 * count + i] = merged ? nanv : s.z;
  }
}
__global__ void __launch_bounds__(kBlock)
k_candidate_queries(Surfels S, const uint32_t* __restrict__ slots, uint32_t nq, const DevState* st,
                    float radius_factor_sq, float* __restrict__ q /* [4][nq]: x, y, z, r^2 */) {
  const uint32_t N = st->surfel_count;
  for (uint32_t k = blockIdx.x * kBlock + threadIdx.x; k < nq; k += gridDim.x * kBlock) {
    const uint32_t i = slots[k];
    float4 s = make_float4(0, 0, 0, 0);
    float r2 = -1.0f;  // out of range or merged: an empty ball
    if (i < N) {
      s = *S.group(kGroupS, i);
      const float rs = S.f(kRadiusSq, i);
      if (!(rs < 0)) r2 = radius_factor_sq * rs;  // surfel_meshing.cc:359-360
    }
    q[k] = s.x; q[(size_t)nq + k] = s.y; q[(size_t)2 * nq + k] = s.z; q[(size_t)3 * nq + k] = r2;
  }
}

// The per-triangle tests of SurfelMeshing::CheckRemeshing (APP/surfel_meshing.cc:590-650) over the device-resident
// map: one thread per triangle, three (S, N) record gathers.  Flag bits: see smx.h.
__global__ void __launch_bounds__(kBlock)
k_check_triangles(Surfels S, const uint32_t* __restrict__ tri, uint32_t n_tri, const DevState* st,
                  float factor_sq, uint8_t* __restrict__ flags) {
  const uint32_t N = st->surfel_count;
  for (uint32_t t = blockIdx.x * kBlock + threadIdx.x; t < n_tri; t += gridDim.x * kBlock) {
    const uint32_t v[3] = {tri[3 * (size_t)t], tri[3 * (size_t)t + 1], tri[3 * (size_t)t + 2]};
    if (v[0] >= N || v[1] >= N || v[2] >= N) { flags[t] = 16; continue; }
    Vec3 p[3], nrm[3];
    float maxsq[3];
    uint32_t f = 0;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const float4 s = *S.group(kGroupS, v[k]);
      const float4 n = *S.group(kGroupN, v[k]);
      p[k] = Vec3{s.x, s.y, s.z};
      nrm[k] = Vec3{n.x, n.y, n.z};
      maxsq[k] = factor_sq * n.w;   // :556-557, 593-596
      if (n.w < 0) f |= 16u;        // :559
    }
    float e[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int b = (k + 1) % 3;
      const float dx = p[b].x - p[k].x, dy = p[b].y - p[k].y, dz = p[b].z - p[k].z;
      e[k] = dx * dx + dy * dy + dz * dz;
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {  // :605-617
      const int b = (k + 1) % 3, c = (k + 2) % 3;
      if (e[k] > maxsq[k] && e[k] > maxsq[b] && (e[b] > maxsq[c] || e[c] > maxsq[c])) f |= 1u;
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {  // :632-635, pivot k
      const int r = (k + 1) % 3, l = (k + 2) % 3;
      const float rx = p[r].x - p[k].x, ry = p[r].y - p[k].y, rz = p[r].z - p[k].z;
      const float lx = p[l].x - p[k].x, ly = p[l].y - p[k].y, lz = p[l].z - p[k].z;
      const float cx = ry * lz - rz * ly, cy = rz * lx - rx * lz, cz = rx * ly - ry * lx;
      const float d0 = cx * nrm[k].x + cy * nrm[k].y + cz * nrm[k].z;
      const float d1 = cx * nrm[r].x + cy * nrm[r].y + cz * nrm[r].z;
      const float d2 = cx * nrm[l].x + cy * nrm[l].y + cz * nrm[l].z;
      if (d0 <= 0 && d1 <= 0 && d2 <= 0) f |= (2u << k);
    }
    flags[t] = (uint8_t)f;
  }
}

// The loop-closure hook the reference describes but does not ship (README.md:152-176, main.cc:1194-1200): a rigid
// correction per creation frame.  Streams the C records (creation stamp); only moved slots touch P, S, N.
__global__ void __launch_bounds__(kBlock)
k_deform_by_creation_frame(Surfels S, const float* __restrict__ frame_T, uint32_t n_frames,
                           const uint8_t* __restrict__ reactivate, uint32_t frame_index, uint8_t* __restrict__ dirty8,
                           const DevState* st) {
  const uint32_t N = st->surfel_count;
  for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < N; i += gridDim.x * kBlock) {
    const uint32_t c = S.u(kCreationStamp, i);
    if (c >= n_frames) continue;
    float4 nr = *S.group(kGroupN, i);
    if (nr.w < 0) continue;  // merged
    Mat34 T;
#pragma unroll
    for (int k = 0; k < 12; ++k) T.m[k] = frame_T[12 * (size_t)c + k];
    float4 pr = *S.group(kGroupP, i);
    float4 sr = *S.group(kGroupS, i);
    const Vec3 p = {pr.x, pr.y, pr.z};
    const Vec3 q = mul(T, p);
    const float ox = q.x - p.x, oy = q.y - p.y, oz = q.z - p.z;  // README.md:160-165: one offset for both positions
    const Vec3 nn = rotate(T, Vec3{nr.x, nr.y, nr.z});            // README.md:166-168
    const bool restamp = reactivate != nullptr && reactivate[c] && __float_as_uint(pr.w) != frame_index;
    // a correction that leaves the slot as it is (identity rows) is not a change for the delta hand-off
    if (ox == 0 && oy == 0 && oz == 0 && nn.x == nr.x && nn.y == nr.y && nn.z == nr.z && !restamp) continue;
    pr.x = p.x + ox; pr.y = p.y + oy; pr.z = p.z + oz;
    sr.x = sr.x + ox; sr.y = sr.y + oy; sr.z = sr.z + oz;
    nr.x = nn.x; nr.y = nn.y; nr.z = nn.z;
    if (restamp) pr.w = __uint_as_float(frame_index);             // README.md:172-174
    *S.group(kGroupP, i) = pr;
    *S.group(kGroupS, i) = sr;
    *S.group(kGroupN, i) = nr;
    if (dirty8) dirty8[i] = 1;
  }
}

__global__ void __launch_bounds__(kBlock)
k_decode_conflicting(const uint32_t* __restrict__ key, uint32_t* __restrict__ out, int P) {
  const int k = blockIdx.x * kBlock + threadIdx.x;
  if (k < P) out[k] = (key[k] == kInvalid) ? kInvalid : (key[k] & 0x7FFFFFFFu);
}

__global__ void __launch_bounds__(kBlock)
k_global_ranks(const uint32_t* __restrict__ local_rank, const uint32_t* __restrict__ block_offsets,
               uint32_t* __restrict__ out, int P) {
  const int k = blockIdx.x * kBlock + threadIdx.x;
  if (k < P) out[k] = block_offsets[k / kScanPxPerBlock] + local_rank[k];
}

}  // namespace

// =============================================================================================
struct smx_recon_s {
  int device;               // the HIP device the object lives on (every entry point runs on it)
  uint32_t max_surfels;
  int W, H;
  float fx, fy, cx, cy;
  Surfels S;
  long long* grad_acc;      // [slots][2] packed fixed point (see pack_pair), cross-segment contributions (atomics)
  float4* reg_rec;          // two dense arrays [slots + kSegAcc] of the recent slots' results, by (segment, rank in its recent list):
                            // the in-segment sums, then the own terms (each written by ONE store instruction per chunk: full sectors)
  FarBins fb;               // far terms of the regulariser, per destination segment (see FarBins)
  Lists L;
  int nseg;                 // number of kSeg-slot segments (= workgroups of pass A)
  int nsegB;                // number of kSegB-slot segments (= workgroups of pass B)
  uint8_t* merge_flag;
  bool table_valid;         // flag table's "recent" bits correspond to (table_frame, table_window)
  uint32_t table_frame;
  int table_window;
  int stats_enabled;
  hipEvent_t hook_consumed, hook_chain;   // smx_recon_integrate_hooks: one-shot, taken by the next Integrate call
  hipEvent_t hook_ready;                  // smx_recon_integrate_inputs_ready: likewise
  int blend_multi_launch;   // A/B switch: 1 = the reference's start + iteration launches instead of the fused kernel
  Scratch sc;               // the association images (every pixel is rewritten by k_assoc_tiles in every call)
  TileBins tb;              // pass A's pairs, binned by association tile
  SegWork sw;               // pass A's work lists (k_cull_segments -> k_scan_visible)
  bool sw_dirty;            // a call failed between the cull step and the tile kernel: the lists' counters are not zero
  hipEvent_t pending_mark;  // != null: the caller's stream has not waited for the previous call's update + create yet (smx_recon_integrate)
  uint32_t* ovf_count_set[2];   // overflow counters, alternating by call (the tile kernel zeroes the next call's)
  unsigned long long* stamps;   // -DSMX_STAMPS builds: [3][8192 workgroups][16] shader clocks (tile kernel, blend kernel), wall clocks + counts (edge kernel)
  int no_lds_tables;        // A/B switch (scan mode bit 4)
  int blend_other_tile;     // A/B switch (scan mode bit 7)
  int cu_count;             // compute units of the object's device
  uint32_t bin_cap_full;    // the bins' allocated capacity (tb.cap is lowered by the A/B switch that forces overflows)
  uint32_t* vis_count_set[2];   // chunk counters of the visible list, alternating by call (k_update_and_create zeroes the next call's)
  uint32_t* dir_host;           // page-locked word: the direction the tile kernel last chose for the all-slot kernels
  uint32_t* dir_dev;            // (segment_of_block); its device alias
  int sc_cur;
  int hot_holdoff;          // > 0: pass B does not use the hot-group table (decremented per Integrate call)
  int hot_filter_enabled;   // A/B switch (smx_recon_set_scan_mode bit 2 clears it)
  int fuse_edges;           // pass B does the edge work of its segment itself (one launch less); scan mode bit 9 sets it
  uint16_t* blended_depth;  // [H][W] output of the fused blend (stored into the caller's depth by k_new_flags_scan)
  BlendBufs bb;
  uint8_t* new_flags;
  uint32_t* new_ranks;
  uint32_t* block_sums;     // per k_new_flags_scan workgroup: flagged pixels
  uint32_t* block_offsets;  // exclusive scan of block_sums (written by k_new_create's workgroup 0)
  uint32_t* tmp_u32;  // [W*H] debug decode target
  int n_scan_blocks;
  DevState* st;
  int scan_mode;
  int timing_enabled;       // bit 0: the reference's 14 stage events, bit 1: events around every kernel, bit 2: stage stamps (default)
  bool have_timings;
  unsigned long long* ts_ring;   // [kTsRing][kTsWords] stage stamps of the last kTsRing Integrate calls (StageStamps)
  unsigned long long* ts_host;   // page-locked copy target of the ring (the waiting read)
  volatile unsigned long long* ts_mapped;   // page-locked ring the tile kernel copies complete records into (the non-waiting read)
  unsigned long long* ts_mapped_dev;        // ... its device alias
  unsigned long long ts_seq;     // Integrate calls with stamps so far (a call's record: ts_ring[seq % kTsRing])
  int wall_khz;                  // rate of the device's wall clock (hipDeviceAttributeWallClockRate)
  hipEvent_t ev[14];
  // per-kernel instrumentation (timing_enabled bit 1) and single-kernel profiling over many frames
  hipEvent_t kev[2 * 16];
  bool kev_recorded[16];
  int prof_slot;
  int prof_cap, prof_n;
  hipEvent_t* prof_ev;
  uint32_t* delta_seg;   // delta hand-off: per-segment counts / offsets, and the total
  uint32_t* delta_total;
  float* staging;    // row-layout staging for the boundary conversions (TransferAllToCPU, debug rows)
  size_t staging_floats;
  float* cand_q; uint32_t* cand_slots; uint8_t* cand_state;   // workspace of smx_recon_neighbor_candidates
  uint32_t cand_cap, cand_state_cap;
  hipEvent_t ev_staging;  // recorded after the last enqueued read of `staging`: TransferAllToCPU returns with its
  bool staging_busy;      // row downloads in flight, and the next user may come on another stream
  int grid_surfels;  // persistent grid for the grid-stride all-slot kernels
  int grid_list;     // persistent grid of the chunked list kernels
  int grid_acc;      // ... of the edge kernel (512-lane workgroups)
  int grid_list_full; // (grid_list is lowered by the A/B switch that forces long walks)
  int debug_skip;    // smx_recon_debug_set_skip (timing only)
  // Frame pipelining: the regulariser of frame f runs on an internal stream while the caller's stream already
  // executes clear / pass A / associate / merge / blend of frame f+1 (those read only P and N records, which the
  // regulariser does not write, and a second copy of the flag table).  Every entry point first orders the
  // caller's stream after the pending regulariser, so the API keeps its one-stream semantics.
  int overlap_enabled;
  hipStream_t reg_stream;     // high priority: the frame-to-frame critical path (integrate .. regulariser)
  uint32_t* gate_count;       // blend workgroups that have released their output, all calls (k_front_gate)
  uint32_t gate_expected;     // ... the value after the last call enqueued with the device-word hand-over
  int handover_mode;          // smx_recon_set_handover_mode: 1 = device word + gate kernel for front -> integration, 0 = event
  hipEvent_t ev_front;        // caller's stream: pass A .. flags of a call are enqueued
  hipEvent_t ev_upd;          // internal stream: update + create of a call are enqueued (the inputs are consumed, the map is ready for the next pass A)
  hipEvent_t ev_reg;          // internal stream: end of the work enqueued so far (recorded on demand by join_regularizer)
  bool reg_pending;
  hipStream_t last_stream;    // the caller's stream of the last Integrate call
  uint8_t* flags_buf[2];    // the flag table is double-buffered by frame (L.flags8 = the current frame's)
  bool have_frame;          // an Integrate call has been made since creation / the last state upload
  uint32_t last_frame;      // its frame_index: the segment culling of pass A presumes that it never decreases
};

// kernel slots of one Integrate call (launch order)
enum : int {
  kSlotScanVisible = 0, kSlotAssocTiles, kSlotBlend, kSlotIntegrate,
  kSlotUpdateNeighbors, kSlotNeighborScan,
  kSlotRegAccumulate, kSlotRegStep,
  kSlotRegUpdate,
  kSlotCull,    // pass A's cull step (enqueued in front of the stream wait for the previous call's map)
  kSlotEmpty,   // nothing between its two time stamps: what a pair of stamps costs (subtracted by bench.py)
  kSlotCount
};
static const char* const kSlotNames[kSlotCount] = {
  "scan_visible", "assoc_tiles", "blend", "integrate+new_flags", "update_neighbors+create",
  "neighbor_scan", "reg_accumulate", "reg_step", "reg_update", "cull_segments", "empty_slot"};

// launch_carries: the slot's kernel is launched with hipExtLaunchKernelGGL and takes start() / stop() as its own start and
// completion events -- the profile (smx_recon_profile_begin: bench.py times the frame's longest kernel inside its timed
// region) then puts no packet of its own on the stream.  Two event records around a kernel on the internal stream cost the
// frame what a hand-off costs (profiles/r17_ab_notes.md, r23).
struct SlotTimer {
  smx_recon r; hipStream_t st; int slot; bool kev, prof, by_launch;
  SlotTimer(smx_recon r_, hipStream_t st_, int slot_, bool launch_carries = false) : r(r_), st(st_), slot(slot_) {
    kev = (r->timing_enabled & 2) != 0;
    prof = (r->prof_slot == slot) && r->prof_ev && r->prof_n < r->prof_cap;
    by_launch = prof && launch_carries && SMX_EXT_STOP_EVENTS != 0;
    if (kev) (void)hipEventRecord(r->kev[2 * slot], st);
    if (prof && !by_launch) (void)hipEventRecord(r->prof_ev[2 * r->prof_n], st);
  }
  hipEvent_t start() const { return by_launch ? r->prof_ev[2 * r->prof_n] : nullptr; }
  hipEvent_t stop() const { return by_launch ? r->prof_ev[2 * r->prof_n + 1] : nullptr; }
  ~SlotTimer() {
    if (kev) { (void)hipEventRecord(r->kev[2 * slot + 1], st); r->kev_recorded[slot] = true; }
    if (prof) { if (!by_launch) (void)hipEventRecord(r->prof_ev[2 * r->prof_n + 1], st); r->prof_n++; }
  }
};

namespace {

// Orders stream st after the regulariser that may still run on the internal stream.
int join_regularizer(smx_recon r, hipStream_t st) {
  // (the flag stays set: a later call may come with another stream, which has to be ordered as well; waiting on a
  // completed event costs nothing on the device.  The mark is recorded here, on demand: the frame loop itself never
  // needs it, and every event operation on the internal stream sits on the frame-to-frame critical chain.)
  if (r->reg_pending) {
    SMX_HIP(hipEventRecord(r->ev_reg, r->reg_stream));
    SMX_HIP(hipStreamWaitEvent(st, r->ev_reg, 0));
    if (st == r->last_stream) r->pending_mark = nullptr;   // (that stream now waits for more than the deferred mark covers)
  }
  return SMX_OK;
}

// zero_chunks: pass B appends to the recent list's chunk descriptors; inside Integrate the launch in front of it
// (k_update_and_create) has reset their counter, everywhere else it is done here.
int enqueue_regularize(smx_recon r, hipStream_t st, uint32_t frame, float rf, float weight, int window,
                       bool detach, bool copy_only, bool zero_chunks, unsigned long long* ts = nullptr, hipEvent_t acc_done = nullptr) {
  const dim3 g(r->nsegB), bB(kBlockB), gl(r->grid_list), b(kBlock);
  const float rf2 = rf * rf;
  if (!r->table_valid || r->table_frame != frame || r->table_window != window) {
    hipLaunchKernelGGL(k_rebuild_flags, dim3(r->grid_surfels), b, 0, st, r->S, frame, window, r->L.flags8, r->st);
    r->table_valid = true; r->table_frame = frame; r->table_window = window;
    r->hot_holdoff = 3;   // (the flags were rewritten outside pass A: this pass and those of the next two calls gather everything)
    // (... and pass A copies the flag bytes of every segment it culls again: the table it would otherwise rely on is this one)
    SMX_HIP(hipMemsetAsync(r->L.seg_streak, 0, (size_t)r->nseg, st));
  }
  unsigned long long* ts_first = zero_chunks ? nullptr : ts;   // (the first pass B of an Integrate call begins the stage)
  const int stats = r->stats_enabled;
  const int use_hot = (r->hot_holdoff == 0 && !r->scan_mode && r->hot_filter_enabled) ? 1 : 0;
  const size_t hot_lds = ((size_t)r->L.n_hot_groups + 15) & ~(size_t)15;
  {
    SlotTimer t(r, st, kSlotNeighborScan, true);
    if (stats || zero_chunks) hipLaunchKernelGGL(k_reset_recent, dim3(1), dim3(kSubLists), 0, st, r->st, stats, zero_chunks ? r->L.rec_chunks.count : nullptr);
    EdgeArgs ea;
    ea.rf2 = rf2; ea.weight = weight; ea.grad_acc = r->grad_acc; ea.reg_rec = r->reg_rec;
    ea.rec_own_offset = (size_t)r->S.pitch + kSegAcc; ea.fb = r->fb;
    const uint32_t lds_b = (uint32_t)hot_lds;
    const bool fused = !copy_only && r->fuse_edges != 0;
    // (fused: the heavy segments carry their edge work, the longest chains of the launch -- they go FIRST, the thousands of
    // workgroups that only stream or skip fill in behind them; not fused: last, see segment_of_block)
    const uint32_t dir = fused && SMX_FUSED_HEAVY_FIRST ? (r->L.descending ? 0u : 1u) : r->L.descending;
#define SMX_LAUNCH_PASS_B(D, A, F) hipExtLaunchKernelGGL((k_neighbor_scan<D, A, F>), g, bB, lds_b, st, t.start(), t.stop(), 0, r->S, stats, use_hot, r->L, r->st, ts_first, ea, dir)
    if (copy_only) { if (detach) SMX_LAUNCH_PASS_B(true, false, false); else SMX_LAUNCH_PASS_B(false, false, false); }
    else if (fused) { if (detach) SMX_LAUNCH_PASS_B(true, true, true); else SMX_LAUNCH_PASS_B(false, true, true); }
    else { if (detach) SMX_LAUNCH_PASS_B(true, true, false); else SMX_LAUNCH_PASS_B(false, true, false); }
#undef SMX_LAUNCH_PASS_B
  }
  if (!copy_only && !r->fuse_edges) {
    SlotTimer t(r, st, kSlotRegAccumulate, true);
    const bool done_by_launch = acc_done && !t.stop();
    hipExtLaunchKernelGGL(k_reg_accumulate, dim3(r->grid_acc), dim3(kBlockAcc), 0, st, t.start(), done_by_launch ? acc_done : t.stop(), 0, r->S, rf2, weight, r->grad_acc, r->reg_rec, (size_t)r->S.pitch + kSegAcc,
                       r->fb, r->L.act_list, r->L.acc_chunks, r->st, ts_first, (r->stamps && (frame & 63u) == 31u) ? r->stamps + 2 * 16 * 8192 : nullptr,
                       (r->debug_skip & 128) ? 256u : (uint32_t)kSegAcc);
    if (acc_done && !done_by_launch) SMX_HIP(hipEventRecord(acc_done, st));
  }
  if (copy_only) {
    SlotTimer t(r, st, kSlotRegUpdate);
    hipLaunchKernelGGL(k_reg_copy_raw, gl, b, 0, st, r->S, r->L, r->st, ts);
  } else {
    SlotTimer t(r, st, kSlotRegStep, true);
    hipExtLaunchKernelGGL(k_reg_step, gl, b, 0, st, t.start(), t.stop(), 0, r->S, weight, r->grad_acc, r->reg_rec, (size_t)r->S.pitch + kSegAcc, r->fb, r->L, r->st, ts);
  }
  SMX_LAUNCH_CHECK();
  return SMX_OK;
}

int ensure_staging(smx_recon r, size_t floats) {
  if (r->staging_floats >= floats) return SMX_OK;
  if (r->staging) { SMX_HIP(hipDeviceSynchronize()); SMX_HIP(hipFree(r->staging)); r->staging = nullptr; r->staging_floats = 0; }
  SMX_HIP(hipMalloc(reinterpret_cast<void**>(&r->staging), floats * sizeof(float)));
  r->staging_floats = floats;
  return SMX_OK;
}

// Every user of the shared staging buffer first orders its stream after the previous user's last read.
int acquire_staging(smx_recon r, hipStream_t st, size_t floats) {
  if (r->staging_busy) SMX_HIP(hipStreamWaitEvent(st, r->ev_staging, 0));
  return ensure_staging(r, floats);
}
int release_staging(smx_recon r, hipStream_t st) {
  SMX_HIP(hipEventRecord(r->ev_staging, st));
  r->staging_busy = true;
  return SMX_OK;
}

// (for the create functions, whose failure paths have to release what exists so far instead of returning on the spot)
int hip_rc(hipError_t e, const char* what) {
  if (e == hipSuccess) return SMX_OK;
  set_error("%s failed: %s", what, hipGetErrorString(e));
  return SMX_ERR_HIP;
}

template <typename T>
int dev_alloc(T** p, size_t count, bool zero) {
  SMX_HIP(hipMalloc(reinterpret_cast<void**>(p), count * sizeof(T)));
  if (zero) SMX_HIP(hipMemset(*p, 0, count * sizeof(T)));
  return SMX_OK;
}

}  // namespace

extern "C" {

int smx_recon_create(uint32_t max_surfel_count, int32_t width, int32_t height,
                     float fx, float fy, float cx, float cy, int32_t device_id, smx_recon* out) {
  // (slot indices travel with two class bits on top in k_reg_accumulate's pending far terms: far_target)
  SMX_CHECK_ARG(out != nullptr && max_surfel_count > 0 && max_surfel_count <= (1u << 30));
  SMX_CHECK_ARG(width >= 3 && height >= 3);
  int device = 0;
  { const int rcd = resolve_device(device_id, &device); if (rcd != SMX_OK) return rcd; }
  SMX_ON_DEVICE(device);
  {   // (once per process: what the application should have set before its first HIP call -- smx_runtime_advice)
    static std::atomic<int> advised{0};
    char text[512];
    if (!advised.exchange(1) && smx_runtime_advice(text, sizeof(text)) > 0) {
      const char* quiet = getenv("SMX_QUIET");
      if (!(quiet && quiet[0] == '1')) fprintf(stderr, "libsmx: %s\n", text);
    }
  }
  smx_recon_s* r = new smx_recon_s();
  memset(r, 0, sizeof(*r));
  r->device = device;
  r->max_surfels = max_surfel_count;
  r->W = width; r->H = height; r->fx = fx; r->fy = fy; r->cx = cx; r->cy = cy;
  r->S.pitch = ((size_t)max_surfel_count + 63) / 64 * 64;
  const size_t P = (size_t)width * height;
  int rc;
#define SMX_TRY(x) do { rc = (x); if (rc != SMX_OK) { (void)smx_recon_destroy(r); return rc; } } while (0)  /* (no leak on a failed allocation) */
  // cuda_surfel_reconstruction.cc:59 -- 25 rows x max_surfel_count (zero-filled here so that the
  // padded tail of every row is defined)
  SMX_TRY(dev_alloc(&r->S.base, (size_t)kQuadsPerSlot * 4 * r->S.pitch, true));
  SMX_TRY(dev_alloc(&r->grad_acc, 2 * ((size_t)r->S.pitch + kSegAcc), true));
  SMX_TRY(dev_alloc(&r->reg_rec, 2 * ((size_t)r->S.pitch + kSegAcc), true));
  r->nseg = div_up((long long)r->S.pitch, kSeg);
  r->nsegB = div_up((long long)r->S.pitch, kSegB);
  r->fb.cap = kFarBinCap; r->fb.hash_mask = (uint32_t)kFarHash - 1u;
  SMX_TRY(dev_alloc(&r->fb.rec, (size_t)r->nsegB * kFarBinCap, false));
  SMX_TRY(dev_alloc(&r->fb.count, ((size_t)r->nsegB + 1) * kCountStride, true));
  r->L.vis_region = (uint32_t)(div_up(r->nseg, kSubLists) * kSeg);
  SMX_TRY(dev_alloc(&r->L.vis_list, (size_t)kSubLists * r->L.vis_region, true));
  SMX_TRY(dev_alloc(&r->L.recent_list, (size_t)r->nsegB * kSegB, false));
  SMX_TRY(dev_alloc(&r->L.vis_seg, (size_t)r->nseg, true));
  SMX_TRY(dev_alloc(&r->L.seg_box, (size_t)r->nseg * 8, true));
  SMX_TRY(dev_alloc(&r->L.seg_act, (size_t)r->nseg, true));
  SMX_TRY(dev_alloc(&r->L.seg_streak, (size_t)r->nseg, true));
  SMX_TRY(dev_alloc(&r->sw.surv_list, (size_t)r->nseg + 65536, true));   // (+ room for the index a walk forms first)
  SMX_TRY(dev_alloc(&r->sw.copy_list, (size_t)r->nseg + 65536, true));
  SMX_TRY(dev_alloc(&r->sw.count, 2, true));
  // (the direction word of segment_of_block: written by the tile kernel, read by the host without synchronisation)
  SMX_TRY(hip_rc(hipHostMalloc(reinterpret_cast<void**>(&r->dir_host), 2 * sizeof(uint32_t), hipHostMallocMapped), "hipHostMalloc"));
  r->dir_host[0] = 0;
  r->dir_host[1] = 0;   // (a front gate that gave up leaves its mark here as well: smx_recon_integrate reads it without a synchronisation)
  SMX_TRY(hip_rc(hipHostGetDevicePointer(reinterpret_cast<void**>(&r->dir_dev), r->dir_host, 0), "hipHostGetDevicePointer"));
  SMX_TRY(dev_alloc(&r->L.recent_seg, (size_t)r->nsegB, true));
  // chunk descriptors: every chunk of every segment in the worst case, + room for the index a walk forms first
  // chunk descriptors: kSubLists interleaved sub-lists; one holds at most every kSubLists-th segment's chunks, + room for
  // the index a walk forms before it knows the lengths
  r->L.vis_chunks.stride = 0; r->L.vis_chunks.desc = nullptr;   // (the visible list is dense: counters only, vis_begin)
  r->L.rec_chunks.stride = (uint32_t)(div_up(r->nsegB, kSubLists) + 8192);
  SMX_TRY(dev_alloc(&r->L.rec_chunks.desc, (size_t)kSubLists * r->L.rec_chunks.stride, true));
  SMX_TRY(dev_alloc(&r->L.rec_chunks.count, (size_t)kSubLists * kCountStride, true));
  static_assert(kSegAcc == kSegB && kAccCount < kCountStride, "pass B lists the edge kernel's segments by its own segment numbers");
  r->L.acc_chunks.stride = r->L.rec_chunks.stride;
  r->L.acc_chunks.count = r->L.rec_chunks.count + kAccCount;
  SMX_TRY(dev_alloc(&r->L.acc_chunks.desc, (size_t)kSubLists * r->L.acc_chunks.stride, true));
  SMX_TRY(dev_alloc(&r->flags_buf[0], (size_t)r->nsegB * kSegB, true));
  SMX_TRY(dev_alloc(&r->flags_buf[1], (size_t)r->nsegB * kSegB, true));
  r->L.flags8 = r->flags_buf[0];
  r->L.hot_shift = 10;
  while ((((size_t)r->nseg * kSeg) >> r->L.hot_shift) + 1 > (size_t)kMaxHotGroups) ++r->L.hot_shift;
  r->L.n_hot_groups = (uint32_t)((((size_t)r->nseg * kSeg) >> r->L.hot_shift) + 1);
  SMX_TRY(dev_alloc(&r->L.hot_epoch, (size_t)r->L.n_hot_groups + 64, true));   // (zeros: "last active in call 0")
  SMX_TRY(dev_alloc(&r->L.seg_targets, (size_t)r->nsegB * kBlockB, true));   // (written by the unfiltered passes of the hold-off calls before anyone reads it)
  r->L.epoch = 128;   // (far from the zeros)
  r->hot_filter_enabled = 1;
  r->fuse_edges = SMX_FUSE_EDGES;
  r->hot_holdoff = 3;
  SMX_TRY(dev_alloc(&r->merge_flag, r->S.pitch, true));
  SMX_TRY(dev_alloc(&r->gate_count, 32, true));   // (a line of its own)
  r->gate_expected = 0;
  r->handover_mode = SMX_HANDOVER_FLAGS;
  {
    // (a rocprofv3 that collects hardware counters serialises the dispatches of all queues: see k_front_gate)
    const char* cc = getenv("ROCPROF_COUNTER_COLLECTION");
    if (cc && cc[0] && strcmp(cc, "0") != 0) r->handover_mode = 0;
  }
  SMX_TRY(dev_alloc(&r->L.act_list, (size_t)r->nsegB * kSegB, true));
  SMX_TRY(dev_alloc(&r->sc.supporting, P, true));
  SMX_TRY(dev_alloc(&r->sc.counts, P, true));
  SMX_TRY(dev_alloc(&r->sc.depth_sums, P, true));
  SMX_TRY(dev_alloc(&r->sc.confl_key, P, true));
  SMX_TRY(dev_alloc(&r->sc.first_depth, P, true));
  for (int k = 0; k < 2; ++k) {
    SMX_TRY(dev_alloc(&r->vis_count_set[k], (size_t)kSubLists * kCountStride, true));
    SMX_TRY(dev_alloc(&r->ovf_count_set[k], 1, true));
  }
  r->L.vis_chunks.count = r->vis_count_set[0];
  // Association tiles: one bin of kTileBinCap pairs per tile (a tile of 256 pixels holds ~600 pairs at C2; pairs
  // beyond the capacity go to the overflow list, which has room for every pair a call can produce: 2 per slot).
  r->tb.tiles_x = div_up(width, kTileW);
  r->tb.n_tiles = (uint32_t)(r->tb.tiles_x * div_up(height, kTileH));
  r->tb.cap = r->bin_cap_full = kTileBinCap;
  SMX_TRY(dev_alloc(&r->tb.pairs, (size_t)r->tb.n_tiles * kTileBinCap, false));
  SMX_TRY(dev_alloc(&r->tb.count, (size_t)r->tb.n_tiles * kCountStride, true));
  SMX_TRY(dev_alloc(&r->tb.ovf, 2 * (size_t)r->S.pitch + 64, false));
  r->tb.ovf_count = r->ovf_count_set[0];
#ifdef SMX_STAMPS
  SMX_TRY(dev_alloc(&r->stamps, (size_t)3 * 16 * 8192, true));
#endif
  SMX_TRY(dev_alloc(&r->blended_depth, P, true));
  SMX_TRY(dev_alloc(&r->bb.distance_map, P, true));
  SMX_TRY(dev_alloc(&r->bb.new_distance_map, P, true));
  SMX_TRY(dev_alloc(&r->bb.deltas, P, true));
  SMX_TRY(dev_alloc(&r->bb.new_deltas, P, true));
  SMX_TRY(dev_alloc(&r->new_flags, P, true));
  SMX_TRY(dev_alloc(&r->new_ranks, P, true));
  SMX_TRY(dev_alloc(&r->tmp_u32, P, true));
  r->n_scan_blocks = div_up((long long)P, kScanPxPerBlock);
  SMX_TRY(dev_alloc(&r->block_sums, (size_t)r->n_scan_blocks, true));
  SMX_TRY(dev_alloc(&r->block_offsets, (size_t)r->n_scan_blocks, true));
  SMX_TRY(dev_alloc(&r->st, 1, true));
  SMX_TRY(dev_alloc(&r->ts_ring, (size_t)kTsRing * kTsWords, true));
  SMX_TRY(hip_rc(hipHostMalloc(reinterpret_cast<void**>(&r->ts_host), sizeof(unsigned long long) * kTsRing * kTsWords, hipHostMallocDefault), "hipHostMalloc"));
  {
    void* m = nullptr;
    SMX_TRY(hip_rc(hipHostMalloc(&m, sizeof(unsigned long long) * kTsRing * kTsWords, hipHostMallocMapped), "hipHostMalloc"));
    memset(m, 0, sizeof(unsigned long long) * kTsRing * kTsWords);
    r->ts_mapped = static_cast<volatile unsigned long long*>(m);
    SMX_TRY(hip_rc(hipHostGetDevicePointer(reinterpret_cast<void**>(&r->ts_mapped_dev), m, 0), "hipHostGetDevicePointer"));
  }
  SMX_TRY(hip_rc(hipDeviceGetAttribute(&r->wall_khz, hipDeviceAttributeWallClockRate, device), "hipDeviceGetAttribute"));
  if (r->wall_khz <= 0) r->wall_khz = 100000;   // (s_memrealtime: 100 MHz)
  for (int i = 0; i < 14; ++i) SMX_TRY(hip_rc(hipEventCreate(&r->ev[i]), "hipEventCreate"));
  for (int i = 0; i < 2 * 16; ++i) SMX_TRY(hip_rc(hipEventCreate(&r->kev[i]), "hipEventCreate"));
  {
    // the regulariser is on the frame-to-frame critical path, the work it overlaps with is not
    int lo = 0, hi = 0;
    SMX_TRY(hip_rc(hipDeviceGetStreamPriorityRange(&lo, &hi), "hipDeviceGetStreamPriorityRange"));
    SMX_TRY(hip_rc(hipStreamCreateWithPriority(&r->reg_stream, hipStreamNonBlocking, SMX_REG_PRIORITY_HIGH ? hi : 0), "hipStreamCreateWithPriority"));
  }
  // (device-scope release: these events order GPU streams, the host never reads data behind them)
  const unsigned evf = hipEventDisableTiming | hipEventReleaseToDevice;
  SMX_TRY(hip_rc(hipEventCreateWithFlags(&r->ev_front, evf), "hipEventCreateWithFlags"));
  SMX_TRY(hip_rc(hipEventCreateWithFlags(&r->ev_upd, evf), "hipEventCreateWithFlags"));
  SMX_TRY(hip_rc(hipEventCreateWithFlags(&r->ev_reg, evf), "hipEventCreateWithFlags"));
  SMX_TRY(hip_rc(hipEventCreateWithFlags(&r->ev_staging, hipEventDisableTiming), "hipEventCreateWithFlags"));
  r->overlap_enabled = 1;
  r->prof_slot = -1;
  r->timing_enabled = 4;   // (GetTimings is served by the stage stamps: on from the first call, like the reference's events)
  hipDeviceProp_t prop;
  SMX_TRY(hip_rc(hipGetDeviceProperties(&prop, device), "hipGetDeviceProperties"));
#undef SMX_TRY
  const int cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  r->grid_surfels = cus * 8;  // 8 x 256-thread workgroups per CU: full occupancy, >> 256 workgroups
  r->cu_count = cus;
  r->grid_acc = cus * SMX_ACC_WGS_PER_CU;
  r->grid_list = r->grid_list_full = cus * SMX_LIST_WGS_PER_CU;  // (a walk step per listed chunk; workgroups without a step only cost)
  r->stats_enabled = 1;
  *out = r;
  return SMX_OK;
}

int smx_recon_destroy(smx_recon r) {
  if (!r) return SMX_OK;
  SMX_ON_DEVICE(r->device);
  void* ptrs[] = {r->sc.supporting, r->sc.counts, r->sc.depth_sums, r->sc.confl_key, r->sc.first_depth,
                  r->tb.pairs, r->tb.count, r->tb.ovf, r->ovf_count_set[0], r->ovf_count_set[1],
                  r->vis_count_set[0], r->vis_count_set[1], r->L.seg_act, r->L.seg_streak, r->sw.surv_list, r->sw.copy_list, r->sw.count, r->blended_depth, r->cand_q, r->cand_slots, r->cand_state, r->L.dirty8, r->delta_seg, r->delta_total, r->staging, r->S.base, r->grad_acc, r->reg_rec, r->fb.rec, r->fb.count, r->L.vis_list, r->L.recent_list, r->L.vis_seg, r->L.seg_box, r->L.recent_seg, r->L.vis_chunks.desc, r->L.rec_chunks.desc, r->L.acc_chunks.desc, r->L.rec_chunks.count, r->flags_buf[0], r->flags_buf[1], r->L.hot_epoch, r->L.seg_targets,
                  r->merge_flag, r->L.act_list, r->bb.distance_map, r->bb.new_distance_map,
                  r->bb.deltas, r->bb.new_deltas, r->new_flags, r->new_ranks, r->tmp_u32, r->block_sums, r->block_offsets, r->st};
  if (r->reg_stream) { (void)hipStreamSynchronize(r->reg_stream); (void)hipStreamDestroy(r->reg_stream); }
  if (r->dir_host) { (void)hipDeviceSynchronize(); (void)hipHostFree(r->dir_host); }
  if (r->ts_host) (void)hipHostFree(r->ts_host);
  if (r->ts_mapped) { (void)hipDeviceSynchronize(); (void)hipHostFree(const_cast<unsigned long long*>(r->ts_mapped)); }
  if (r->ts_ring) (void)hipFree(r->ts_ring);
  if (r->ev_front) (void)hipEventDestroy(r->ev_front);
  if (r->ev_upd) (void)hipEventDestroy(r->ev_upd);
  if (r->ev_reg) (void)hipEventDestroy(r->ev_reg);
  if (r->ev_staging) (void)hipEventDestroy(r->ev_staging);
  for (void* p : ptrs) if (p) (void)hipFree(p);
  for (int i = 0; i < 14; ++i) if (r->ev[i]) (void)hipEventDestroy(r->ev[i]);
  for (int i = 0; i < 2 * 16; ++i) if (r->kev[i]) (void)hipEventDestroy(r->kev[i]);
  if (r->prof_ev) { for (int i = 0; i < 2 * r->prof_cap; ++i) (void)hipEventDestroy(r->prof_ev[i]); delete[] r->prof_ev; }
  delete r;
  return SMX_OK;
}

int smx_recon_set_timing_enabled(smx_recon r, int32_t enabled) {
  SMX_CHECK_ARG(r != nullptr && enabled >= 0 && enabled <= 7);
  SMX_ON_DEVICE(r->device);
  // (bit 1 alone -- events around every kernel, a measurement mode -- keeps the stage stamps on: GetTimings must not go
  // silent because a tool asked for per-kernel times; 0 switches everything off)
  if ((enabled & 3) == 2) enabled |= 4;
  r->timing_enabled = enabled;
  if (!(enabled & 1)) r->have_timings = false;
  return SMX_OK;
}

int smx_recon_kernel_slot_count(void) { return kSlotCount; }
const char* smx_recon_kernel_slot_name(int32_t slot) { return (slot >= 0 && slot < kSlotCount) ? kSlotNames[slot] : ""; }

int smx_recon_get_kernel_timings(smx_recon r, float* out_ms, int32_t capacity) {
  SMX_CHECK_ARG(r != nullptr && out_ms != nullptr && capacity >= kSlotCount);
  SMX_ON_DEVICE(r->device);
  for (int i = 0; i < kSlotCount; ++i) {
    out_ms[i] = 0;
    if (!r->kev_recorded[i]) continue;
    SMX_HIP(hipEventSynchronize(r->kev[2 * i + 1]));
    SMX_HIP(hipEventElapsedTime(&out_ms[i], r->kev[2 * i], r->kev[2 * i + 1]));
  }
  return SMX_OK;
}

int smx_recon_profile_begin(smx_recon r, int32_t slot, int32_t max_frames) {
  SMX_CHECK_ARG(r != nullptr && slot >= 0 && slot < kSlotCount && max_frames > 0);
  SMX_ON_DEVICE(r->device);
  if (r->prof_ev) { for (int i = 0; i < 2 * r->prof_cap; ++i) (void)hipEventDestroy(r->prof_ev[i]); delete[] r->prof_ev; }
  r->prof_ev = new hipEvent_t[2 * max_frames];
  for (int i = 0; i < 2 * max_frames; ++i) SMX_HIP(hipEventCreate(&r->prof_ev[i]));
  r->prof_cap = max_frames; r->prof_n = 0; r->prof_slot = slot;
  return SMX_OK;
}

int smx_recon_profile_end(smx_recon r, float* avg_ms, int32_t* frames) {
  SMX_CHECK_ARG(r != nullptr && avg_ms != nullptr && frames != nullptr);
  SMX_ON_DEVICE(r->device);
  double sum = 0;
  for (int i = 0; i < r->prof_n; ++i) {
    float ms = 0;
    SMX_HIP(hipEventSynchronize(r->prof_ev[2 * i + 1]));
    SMX_HIP(hipEventElapsedTime(&ms, r->prof_ev[2 * i], r->prof_ev[2 * i + 1]));
    sum += ms;
  }
  *frames = r->prof_n;
  *avg_ms = r->prof_n ? (float)(sum / r->prof_n) : 0.0f;
  r->prof_slot = -1;
  return SMX_OK;
}

int smx_recon_set_stats_enabled(smx_recon r, int32_t enabled) {
  SMX_CHECK_ARG(r != nullptr);
  SMX_ON_DEVICE(r->device);
  r->stats_enabled = enabled ? 1 : 0;
  return SMX_OK;
}

int smx_recon_set_overlap(smx_recon r, int32_t enabled) {
  SMX_CHECK_ARG(r != nullptr && (enabled == 0 || enabled == 1));
  SMX_ON_DEVICE(r->device);
  if (r->reg_pending) { SMX_HIP(hipStreamSynchronize(r->reg_stream)); r->reg_pending = false; }
  r->overlap_enabled = enabled;
  return SMX_OK;
}

int smx_recon_set_handover_mode(smx_recon r, int32_t mode) {
  SMX_CHECK_ARG(r != nullptr && (mode == 0 || mode == 1));
  r->handover_mode = mode;   // (takes effect with the next call: each call's wait matches the signal of its own front)
  return SMX_OK;
}

int smx_recon_get_handover_mode(smx_recon r, int32_t* mode) {
  SMX_CHECK_ARG(r != nullptr && mode != nullptr);
  *mode = r->handover_mode;
  return SMX_OK;
}

int smx_recon_set_scan_mode(smx_recon r, int32_t mode) {
  SMX_CHECK_ARG(r != nullptr && mode >= 0 && mode <= 1023);
  SMX_ON_DEVICE(r->device);
  r->grid_list = ((mode >> 8) & 1) ? 4 : r->grid_list_full;   // four workgroups walk every list: many steps each
  r->scan_mode = mode & 1;
  r->blend_multi_launch = (mode >> 1) & 1;
  r->hot_filter_enabled = ((mode >> 2) & 1) ? 0 : 1;
  r->tb.cap = ((mode >> 3) & 1) ? 16u : r->bin_cap_full;   // 16 pairs per bin: most pairs travel through the overflow list
  r->no_lds_tables = (mode >> 4) & 1;                     // pass A reserves bin space per pair (the path of a pair that finds no entry in the workgroup's table)
  r->fb.cap = ((mode >> 5) & 1) ? 4u : kFarBinCap;        // 4 records per far-term bin: most far terms spill to grad_acc
  r->fb.hash_mask = ((mode >> 6) & 1) ? 1u : (uint32_t)kFarHash - 1u;   // 2 destinations per sender workgroup: the rest spills
  r->fuse_edges = ((mode >> 9) & 1) ? 1 : SMX_FUSE_EDGES;   // one launch: the segment's workgroup of pass B does the segment's edge work itself
  r->blend_other_tile = (mode >> 7) & 1;                  // the blend's other tile size (40 x 40 where it would take 32 x 32 and vice versa)
  return SMX_OK;
}

int smx_recon_debug_set_skip(smx_recon r, int32_t mask) {
  SMX_CHECK_ARG(r != nullptr && mask >= 0 && mask <= 511);
  r->debug_skip = mask;
  return SMX_OK;
}

int smx_recon_integrate(smx_recon r, smx_stream s, uint32_t frame_index, float depth_scaling,
                        const smx_buffer_desc* depth, const smx_buffer_desc* normals,
                        const smx_buffer_desc* radius, const smx_buffer_desc* color,
                        const float global_T_local[12], const smx_integrate_params* p) {
  SMX_CHECK_ARG(r != nullptr);
  const hipEvent_t hook_consumed = r->hook_consumed, hook_chain = r->hook_chain, hook_ready = r->hook_ready;   // one-shot, also when the call fails
  r->hook_consumed = nullptr; r->hook_chain = nullptr; r->hook_ready = nullptr;
  SMX_CHECK_ARG(depth && normals && radius && color && global_T_local && p);
  SMX_ON_DEVICE(r->device);
  SMX_CHECK_ARG(depth->width == r->W && depth->height == r->H && normals->width == r->W && normals->height == r->H);
  SMX_CHECK_ARG(radius->width == r->W && radius->height == r->H && color->width == r->W && color->height == r->H);
  // (the radius is only read when blending is on: do_blending is an independent flag, APP/main.cc:348-354)
  SMX_CHECK_ARG(!p->do_blending || (p->measurement_blending_radius >= 2 && p->measurement_blending_radius <= 255));
  hipStream_t st = (hipStream_t)s;
  if (r->have_frame && (int32_t)(frame_index - r->last_frame) < 0) {
    // Time moves backwards (a sequence replayed on a live object, a loop-closure re-integration): the reference accepts
    // any index.  What presumes forward time here is derived state only -- the segments' newest-stamp cache behind
    // pass A's culling and the hot-group table behind pass B's filter -- so it is dropped: this call reads every segment.
    { const int rcj = join_regularizer(r, st); if (rcj != SMX_OK) return rcj; }
    SMX_HIP(hipMemsetAsync(r->L.vis_seg, 0, (size_t)r->nseg * 4, st));
    SMX_HIP(hipMemsetAsync(r->L.seg_box, 0, (size_t)r->nseg * 8 * sizeof(float), st));
    r->hot_holdoff = 3;
  }
  r->have_frame = true; r->last_frame = frame_index;
  // (a front gate that gave up -- kernel dispatches serialised across queues: k_front_gate -- has left its mark in page-locked
  // memory: no more gates, every one of them would sit out its bound; the map is invalid and smx_recon_counts / _get_stats say so)
  if (r->handover_mode == 1 && reinterpret_cast<volatile uint32_t*>(r->dir_host)[1] != 0u) r->handover_mode = 0;
  FrameCtx c;
  memcpy(c.G.m, global_T_local, sizeof(float) * 12);
  c.L = se3_inverse(global_T_local);  // cc:144
  c.fx = r->fx; c.fy = r->fy; c.cx = r->cx; c.cy = r->cy;
  c.up = make_unproj(r->fx, r->fy, r->cx, r->cy);
  c.inv_depth_scaling = 1.0f / depth_scaling;
  c.sensor_noise_factor = p->sensor_noise_factor;
  c.cos_normal_compat = cosf((float)(M_PI / 180.0f * p->normal_compatibility_threshold_deg));  // kernels.cc:261
  c.rf2 = p->radius_factor_for_regularization_neighbors * p->radius_factor_for_regularization_neighbors;
  c.max_conf = p->max_surfel_confidence;
  c.window = p->surfel_integration_active_window_size;
  c.reg_window = p->regularization_frame_window_size;
  c.frame = frame_index;
  c.W = r->W; c.H = r->H;
  c.stats = r->stats_enabled;
  c.ts = nullptr;
  if (r->timing_enabled & 4) { ++r->ts_seq; c.ts = r->ts_ring + (size_t)(r->ts_seq % kTsRing) * kTsWords; }
  const int P = r->W * r->H;
  const dim3 b(kBlock), gpx(div_up(P, kBlock)), gimg(div_up(r->W, 64), div_up(r->H, 4));
  const dim3 gs(r->nseg), gl(r->grid_list);
  const bool tm = (r->timing_enabled & 1) != 0;
  FrameIn in;
  in.depth = as_img<const uint16_t>(depth); in.normals = as_img<const float2>(normals);
  in.radius = as_img<const float>(radius); in.color = as_img<const uchar3>(color);
  const Img<uint16_t> depth_rw = as_img<uint16_t>(depth);

  // the flag table of the previous frame stays readable for the regulariser that may still be running
  const uint8_t* flags_prev = r->L.flags8;
  r->L.flags8 = (r->L.flags8 == r->flags_buf[0]) ? r->flags_buf[1] : r->flags_buf[0];
  r->L.epoch = (r->L.epoch + 1u) & 255u;
  if (r->hot_holdoff > 0) --r->hot_holdoff;
  // Streams.  The frame-to-frame critical cycle is integrate(f) -> update + create(f) -> pass B -> edges -> step ->
  // integrate(f + 1): the regulariser of frame f has to finish before frame f + 1 rewrites positions and links, and it
  // is longer than the front of frame f + 1 (pass A, association tiles, blend, flags), which only reads P / N records
  // and the other copy of the flag table.  That cycle runs on the INTERNAL stream, in order, with no event wait of its
  // own inside it; the front runs on the caller's stream beside the previous call's regulariser and hands over to the
  // internal stream once (ev_front).  The caller's stream then waits for update + create (ev_upd): its buffers are
  // consumed, and the next call's pass A finds the map complete.  Both hand-offs sit on the shorter chain.  Every other
  // entry point first orders its stream after the internal one (join_regularizer), which keeps the reference's
  // one-stream semantics for anything that goes through the API.  Not pipelined: the same launches, all on the caller's
  // stream.  (Rounds 1-2 kept integrate and update + create on the caller's stream and forked the regulariser: two
  // hand-offs on the critical cycle, which then was about as long as the front chain -- profiles/r03c_matrix.txt.)
  const bool pipelined = r->overlap_enabled != 0;
  const hipStream_t sF = st;
  const hipStream_t sR = pipelined ? r->reg_stream : st;
  // TIMING EXPERIMENT (debug_skip bit 4; results NOT exact -- the step kernel of the previous call still reads P / N records
  // that this call's integration rewrites): integrate + update stay on the caller's stream behind the blend (no hand-over
  // there), waiting only for the previous call's EDGE kernel; the internal stream runs pass B / edges / step and waits for
  // update + create.  The step kernel then sits on no cycle.
  const bool split = pipelined && (r->debug_skip & 16) != 0;
  // (a caller that comes with another stream than last time: that stream has not waited for the previous call's map yet)
  if (r->reg_pending && r->last_stream != st) { const int rcj = join_regularizer(r, st); if (rcj != SMX_OK) return rcj; }
  r->last_stream = st;
  if (r->stats_enabled) hipLaunchKernelGGL(k_reset_frame_stats, dim3(1), dim3(1), 0, sF, r->st, 1);
  if (tm) SMX_HIP(hipEventRecord(r->ev[0], sF));
  // two chunk counters / overflow counters in alternation: the kernels of this call read theirs while they reset the
  // next call's (k_update_and_create / k_assoc_tiles)
  r->L.descending = *reinterpret_cast<volatile uint32_t*>(r->dir_host);
  r->sc_cur ^= 1;
  r->L.vis_chunks.count = r->vis_count_set[r->sc_cur];
  r->tb.ovf_count = r->ovf_count_set[r->sc_cur];
#if SMX_READY_WAIT_EARLY == 2
  // (the wait for the input images at the very front of the call -- in a running pipeline they were ready long ago --
  // so that ONE barrier packet stands between the previous call's update + create and this call's pass A: 12.5 -> 8 us
  // from update's end to pass A's begin)
  if (hook_ready) SMX_HIP(hipStreamWaitEvent(sF, hook_ready, 0));
#endif
  { // the cull step: needs the pose and what the previous pass A left, nothing the previous call's second half writes --
    // so it goes in FRONT of this stream's wait for that half wherever the wait could be deferred (below)
    SlotTimer t(r, sF, kSlotCull);
    if (r->sw_dirty) SMX_HIP(hipMemsetAsync(r->sw.count, 0, 2 * sizeof(uint32_t), sF));
    r->sw_dirty = true;
    hipLaunchKernelGGL(k_cull_segments, dim3((unsigned)div_up(r->nseg, kBlock)), b, 0, sF, c, r->L, r->sw, r->st, (uint32_t)r->nseg, (uint32_t)P, r->ts_seq); }
  if (r->pending_mark) { SMX_HIP(hipStreamWaitEvent(sF, r->pending_mark, 0)); r->pending_mark = nullptr; }
#if SMX_READY_WAIT_EARLY == 1
  // (the wait for the input images next to the wait for the map: pass A and the tile kernel then follow each other without
  // a barrier packet between them -- 38 -> 31 us from pass A's first workgroup to the tile kernel's at C2)
  if (hook_ready) SMX_HIP(hipStreamWaitEvent(sF, hook_ready, 0));
#endif
  if (r->stats_enabled) hipLaunchKernelGGL(k_reset_frame_stats, dim3(1), dim3(1), 0, sF, r->st, 0);
  { SlotTimer t(r, sF, kSlotScanVisible, true);
    const bool lds_tables = !r->no_lds_tables;
    // chip-sized grid: as many workgroups as the chip holds at once (8 per CU) walk the survivor list
    const dim3 ga((unsigned)(r->cu_count * SMX_PASS_A_WGS_PER_CU));
    hipExtLaunchKernelGGL(k_scan_visible, ga, b, 0, sF, t.start(), t.stop(), 0, r->S, c, r->L, r->tb, r->sw, flags_prev, r->st, lds_tables ? 1 : 0);
    r->table_valid = true; r->table_frame = frame_index; r->table_window = c.reg_window; }
  // (smx_recon_integrate_inputs_ready) from here on the input images are read
#if !SMX_READY_WAIT_EARLY
  if (hook_ready) SMX_HIP(hipStreamWaitEvent(sF, hook_ready, 0));
#endif
  // (the tile kernel also leaves the direction for later launches' segment_of_block in host memory, see there)
  { SlotTimer t(r, sF, kSlotAssocTiles, true);
    hipExtLaunchKernelGGL(k_assoc_tiles, dim3(r->tb.n_tiles), dim3(kTilePx), 0, sF, t.start(), t.stop(), 0, r->S, c, r->sc, in.depth, in.normals, r->tb,
                       r->ovf_count_set[r->sc_cur ^ 1], r->merge_flag, r->st, r->L.seg_act, (uint32_t)r->nseg, r->dir_dev, r->sw.count,
                       r->stamps ? r->stamps : nullptr, r->ts_ring, c.ts ? r->ts_mapped_dev : nullptr, r->ts_seq);
    r->sw_dirty = false; }
  // (the stage times of GetTimings: data association = pass A + the tile kernel, which also decides the merges)
  if (tm) { SMX_HIP(hipEventRecord(r->ev[1], sF)); SMX_HIP(hipEventRecord(r->ev[2], sF)); SMX_HIP(hipEventRecord(r->ev[3], sF)); SMX_HIP(hipEventRecord(r->ev[4], sF)); }
  const int halo = p->measurement_blending_radius - 1;
  const bool fused_blend = p->do_blending && halo <= kBlendMaxHalo && !r->blend_multi_launch;
  bool front_by_launch = false, front_by_gate = false;
  const float ds = 1.0f / c.inv_depth_scaling;  // kernels.cc:179
  const float term = p->do_blending ? 1.0f / ((float)p->measurement_blending_radius - 1.0f) : 0.0f;  // kernels.cc:196
  const Img<uint16_t> blended = {r->blended_depth, r->H, r->W, (size_t)r->W * sizeof(uint16_t)};
  if (fused_blend) {
    // (the two measurement modes that bracket kernels with event records of their own keep the event hand-over)
    front_by_gate = pipelined && !split && r->handover_mode == 1 && !(r->debug_skip & 4) && !tm && !(r->timing_enabled & 2);
    SlotTimer t(r, sF, kSlotBlend, front_by_gate);   // (device-word hand-over: the launch's event slots are the timer's)
    // tile edge: the one that leaves the busiest CU the fewest cells (see k_blend_tiles)
    const int cus = r->cu_count;
    int tile = 32;
    if (40 + 2 * halo <= 64) {
      auto cost = [&](int t) { const long long n = (long long)div_up(r->W, t) * div_up(r->H, t); return ((n + cus - 1) / cus) * (long long)(t + 2 * halo) * (t + 2 * halo); };
      if ((cost(40) < cost(32)) != (r->blend_other_tile != 0)) tile = 40;
    }
    const int rw = tile + 2 * halo;
    const size_t lds = sizeof(BlendMasks) + (size_t)rw * rw * 10;
    const int tiles_x = div_up(r->W, tile);
    const uint32_t n_blend = (uint32_t)(tiles_x * div_up(r->H, tile));
    // (-DSMX_STAMPS builds: one call in 64 is recorded, so that what is read back after a run is a frame from the middle of the
    // pipeline and not its last one, which drains)
    unsigned long long* stamps = (r->stamps && (frame_index & 63u) == 32u) ? r->stamps + 16 * 8192 : nullptr;
    // (the hand-over to the internal stream as this launch's own completion event)
    // (With the device-word hand-over the launch's event slots are free for whoever times this kernel -- a profile of the
    // blend used to push the hand-over onto an event record of its own behind the launch, and the profiled frame loop ran a
    // tenth slower than the one it was meant to describe: the "slow mode" of rounds 5 - 6 was bench.py profiling this kernel,
    // profiles/r6_ab_notes.md section 14.)
    front_by_launch = !front_by_gate && SMX_EXT_STOP_EVENTS && pipelined && !split && !tm && !(r->timing_enabled & 2) && r->prof_slot != kSlotBlend;
    uint32_t* const gate = front_by_gate ? r->gate_count : nullptr;
    if (front_by_gate) r->gate_expected += n_blend;
    const hipEvent_t start = front_by_gate ? t.start() : nullptr;
    const hipEvent_t stop = front_by_gate ? t.stop() : front_by_launch ? r->ev_front : nullptr;
    if (tile == 40)
      hipExtLaunchKernelGGL(k_blend_tiles<40>, dim3(n_blend), dim3(kBlendThreads), (uint32_t)lds, sF, start, stop, 0, p->measurement_blending_radius, term, ds,
                            in.depth, blended, r->sc, r->W, r->H, tiles_x, stamps, c.ts, gate, (r->debug_skip & 256) ? 1 : kBlendMaxRings);
    else
      hipExtLaunchKernelGGL(k_blend_tiles<32>, dim3(n_blend), dim3(kBlendThreads), (uint32_t)lds, sF, start, stop, 0, p->measurement_blending_radius, term, ds,
                            in.depth, blended, r->sc, r->W, r->H, tiles_x, stamps, c.ts, gate, (r->debug_skip & 256) ? 1 : kBlendMaxRings);
  } else if (p->do_blending) {
    // the reference's own sequence (2 clears + start + iterations, kernels.cc:165-205), in place on the caller's depth
    SlotTimer t(r, sF, kSlotBlend);
    SMX_HIP(hipMemsetAsync(r->bb.distance_map, 0, (size_t)P, sF));
    SMX_HIP(hipMemsetAsync(r->bb.new_distance_map, 0, (size_t)P, sF));
    hipLaunchKernelGGL(k_blend_start, gimg, b, 0, sF, ds, depth_rw, r->sc, r->bb, r->W, r->H, c.ts);
    for (int it = 2; it < p->measurement_blending_radius; ++it)
      hipLaunchKernelGGL(k_blend_iter, gimg, b, 0, sF, it, term, ds, depth_rw, r->sc, r->bb, r->W, r->H, c.ts);
  }
  if (tm) SMX_HIP(hipEventRecord(r->ev[5], sF));
  // Which pixels spawn a surfel depends only on the association images and the blended depth: the flag + rank pass of
  // CreateNewSurfelsCUDA rides in the integration launch below; with the fused blend it also stores the blended depths
  // into the caller's buffer, and the integration kernel reads them from the blend's output image.
  NewFlagsArgs nf;
  nf.depth = fused_blend ? Img<const uint16_t>{r->blended_depth, r->H, r->W, (size_t)r->W * sizeof(uint16_t)} : in.depth;
  nf.depth_out = depth_rw; nf.copy_back = fused_blend ? 1 : 0;
  nf.flags = r->new_flags; nf.local_rank = r->new_ranks; nf.block_sums = r->block_sums;
  nf.dbg = (r->stamps && (frame_index & 63u) == 32u) ? r->stamps + (size_t)(2 * 8192 + 8189) * 16 : nullptr;
  FrameIn in_integrate = in;
  in_integrate.depth = nf.depth;
  // Everything up to here only read P and N records; from here on they (and T, S) are written, so the previous
  // call's regulariser has to be done: it is, by stream order -- the rest of the call follows it on the internal stream.
  const hipStream_t sI = split ? sF : sR;   // integrate, update + create
  if (split) {
    if (r->reg_pending) SMX_HIP(hipStreamWaitEvent(sF, r->ev_front, 0));   // (here: "the previous call's edge kernel is done")
    r->reg_pending = true;
  } else if (pipelined) {
    // (work on the internal stream from here on: whatever happens below, later entry points order themselves behind it)
    r->reg_pending = true;
    if (front_by_gate) {
      // (debug_skip bit 5, test only: the gate is told to wait for one workgroup more than the blend has -- it gives up after its
      // bound, and the next synchronising entry point reports it)
      hipLaunchKernelGGL(k_front_gate, dim3(1), dim3(64), 0, sR, r->gate_count, r->gate_expected + ((r->debug_skip & 32) ? 1u : 0u),
                         (r->stamps && (frame_index & 63u) == 32u) ? r->stamps + (size_t)(2 * 8192 + 8190) * 16 : nullptr, r->dir_dev + 1);
    } else {
      if (!front_by_launch) SMX_HIP(hipEventRecord(r->ev_front, sF));
      if (!(r->debug_skip & 4)) SMX_HIP(hipStreamWaitEvent(sR, r->ev_front, 0));   // (bit 2: timing only -- what is the hand-over worth?)
    }
  }
  if (tm) SMX_HIP(hipEventRecord(r->ev[6], sR));
  const bool front_only = (r->debug_skip & 2) != 0, skip_reg = (r->debug_skip & 3) != 0;   // (timing only)
  bool mark_by_launch = false;
  if (front_only) SMX_HIP(hipMemsetAsync(r->vis_count_set[r->sc_cur ^ 1], 0, sizeof(uint32_t) * kSubLists * kCountStride, sR));   // (k_update_and_create's side job)
  if (!front_only) { SlotTimer t(r, sI, kSlotIntegrate, true);
    const uint32_t nfb = (uint32_t)r->n_scan_blocks;
    const dim3 gi(nfb + (uint32_t)r->grid_list);
    if (r->scan_mode) hipExtLaunchKernelGGL((k_integrate<false>), gi, b, 0, sI, t.start(), t.stop(), 0, r->S, c, r->sc, in_integrate, r->L, r->merge_flag, r->st, nf, nfb);
    else hipExtLaunchKernelGGL((k_integrate<true>), gi, b, 0, sI, t.start(), t.stop(), 0, r->S, c, r->sc, in_integrate, r->L, r->merge_flag, r->st, nf, nfb); }
  if (tm) { SMX_HIP(hipEventRecord(r->ev[7], sR)); SMX_HIP(hipEventRecord(r->ev[8], sR)); }
  // (debug_skip bit 6, TIMING ONLY -- the map is wrong afterwards: the caller's stream is released behind the INTEGRATION launch
  // instead of behind update + create, an upper bound for "pass A needs the new slots, not the new links")
  if ((r->debug_skip & 64) && pipelined) SMX_HIP(hipEventRecord(r->ev_front, sI));
  if (!front_only) { SlotTimer t(r, sI, kSlotUpdateNeighbors);
    CreateArgs ca;
    ca.flags = r->new_flags; ca.ranks = r->new_ranks; ca.block_sums = r->block_sums; ca.block_offsets_out = r->block_offsets;
    ca.n_scan_blocks = r->n_scan_blocks; ca.max_surfels = r->max_surfels; ca.flags8 = r->L.flags8; ca.dirty8 = r->L.dirty8;
    ca.next_vis_chunk_count = r->vis_count_set[r->sc_cur ^ 1];
    ca.n_pixels = P; ca.hot_epoch = r->L.hot_epoch; ca.epoch = r->L.epoch; ca.hot_shift = r->L.hot_shift;
    const uint32_t ncb = (uint32_t)div_up(P, kBlock);
    const dim3 guc(ncb + (uint32_t)r->grid_list);
    const size_t lds = (size_t)r->n_scan_blocks * sizeof(uint32_t);
    // (the "map complete / inputs consumed" mark below as this launch's own completion event: no packet of its own between
    // this kernel and pass B on the internal stream)
    mark_by_launch = SMX_EXT_STOP_EVENTS && !tm && !(r->timing_enabled & 2) && (pipelined || hook_consumed) && r->prof_slot != kSlotUpdateNeighbors;
    const hipEvent_t stop = mark_by_launch ? (hook_consumed ? hook_consumed : r->ev_upd) : nullptr;
    if (r->scan_mode) hipExtLaunchKernelGGL((k_update_and_create<false>), guc, b, (uint32_t)lds, sI, nullptr, stop, 0, r->S, c, r->sc, in, r->L, ca, ncb, r->st);
    else hipExtLaunchKernelGGL((k_update_and_create<true>), guc, b, (uint32_t)lds, sI, nullptr, stop, 0, r->S, c, r->sc, in, r->L, ca, ncb, r->st); }
  // (the detach half of UpdateNeighborsCUDA runs fused into pass B below)
  if (tm) { SMX_HIP(hipEventRecord(r->ev[9], sR)); SMX_HIP(hipEventRecord(r->ev[10], sR)); }
  if (tm) { SMX_HIP(hipEventRecord(r->ev[11], sR)); SMX_HIP(hipEventRecord(r->ev[12], sR)); }
  if (r->timing_enabled & 2) { SlotTimer t(r, sR, kSlotEmpty); }
  SMX_LAUNCH_CHECK();
  int rc = SMX_OK;
  const int iters = p->regularization_iterations_per_integration_iteration;
  // The input images are free from here on, and the map is ready for the next call's pass A: the caller's stream
  // continues behind this point (a device-side wait; the host does not block).  The caller's "inputs consumed" event
  // (smx_recon_integrate_hooks) marks the same point: one record serves both (every event operation on the internal
  // stream sits on the frame-to-frame critical chain).
  {
    const hipEvent_t mark = hook_consumed ? hook_consumed : r->ev_upd;
    if ((pipelined || hook_consumed) && !mark_by_launch) SMX_HIP(hipEventRecord(mark, sI));
    // A caller that asked for the "inputs consumed" event orders the reuse of its images itself; its stream then only has
    // to wait before the next call's pass A reads the map -- and that call's cull step, which does not, may go first.
    // (Every other entry point orders its stream after the whole internal stream: join_regularizer.)
    if (split) SMX_HIP(hipStreamWaitEvent(sR, mark, 0));   // (the internal stream's pass B waits; the caller's stream carries on)
    else if (r->debug_skip & 8) { }   // (bit 3: timing only -- the caller's stream does not wait for update + create)
    else if ((r->debug_skip & 64) && pipelined) { if (hook_consumed) r->pending_mark = r->ev_front; else SMX_HIP(hipStreamWaitEvent(sF, r->ev_front, 0)); }
    else if (pipelined && hook_consumed) r->pending_mark = mark;
    else if (pipelined) SMX_HIP(hipStreamWaitEvent(sF, mark, 0));
  }
  if (skip_reg) {
  } else if (iters == 0) {
    rc = enqueue_regularize(r, sR, frame_index, p->radius_factor_for_regularization_neighbors, p->regularizer_weight,
                            p->regularization_frame_window_size, true, true, false, c.ts);
  } else {
    for (int k = 0; k < iters && rc == SMX_OK; ++k)
      rc = enqueue_regularize(r, sR, frame_index, p->radius_factor_for_regularization_neighbors, p->regularizer_weight,
                              p->regularization_frame_window_size, k == 0, false, k > 0, c.ts, (split && k == iters - 1) ? r->ev_front : nullptr);
  }
  if (rc != SMX_OK) return rc;
  if (tm) { SMX_HIP(hipEventRecord(r->ev[13], sR)); r->have_timings = true; }
  // (smx_recon_integrate_hooks) whatever this event covers is complete before the regulariser counts as complete,
  // i.e. before the second half of the next call and all of the call after it
  if (hook_chain) SMX_HIP(hipStreamWaitEvent(sR, hook_chain, 0));
  return SMX_OK;
}

#ifdef SMX_STAMPS
int smx_recon_debug_download_stamps(smx_recon r, unsigned long long* out) {   // [3][8192][16]
  SMX_CHECK_ARG(r != nullptr && out != nullptr && r->stamps != nullptr);
  SMX_ON_DEVICE(r->device);
  SMX_HIP(hipDeviceSynchronize());
  SMX_HIP(hipMemcpy(out, r->stamps, sizeof(unsigned long long) * 3 * 16 * 8192, hipMemcpyDeviceToHost));
  return SMX_OK;
}
#endif

int smx_recon_set_internal_cu_mask(smx_recon r, const uint32_t* mask_words, uint32_t n_words) {
  SMX_CHECK_ARG(r != nullptr && (n_words == 0 || mask_words != nullptr) && n_words <= 32);
  SMX_ON_DEVICE(r->device);
  SMX_HIP(hipStreamSynchronize(r->reg_stream));
  r->reg_pending = false;
  hipStream_t s = nullptr;
  if (n_words) {
    SMX_HIP(hipExtStreamCreateWithCUMask(&s, n_words, mask_words));
  } else {
    int lo = 0, hi = 0;
    SMX_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
    SMX_HIP(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, SMX_REG_PRIORITY_HIGH ? hi : 0));
  }
  (void)hipStreamDestroy(r->reg_stream);
  r->reg_stream = s;
  return SMX_OK;
}

int smx_recon_debug_internal_stream(smx_recon r, smx_stream* out) {   // (measurement: smx_debug_handover_probe)
  SMX_CHECK_ARG(r != nullptr && out != nullptr);
  *out = (smx_stream)r->reg_stream;
  return SMX_OK;
}

int smx_recon_integrate_hooks(smx_recon r, smx_event inputs_consumed, smx_event chain_after) {
  SMX_CHECK_ARG(r != nullptr);
  r->hook_consumed = (hipEvent_t)inputs_consumed;
  r->hook_chain = (hipEvent_t)chain_after;
  return SMX_OK;
}

int smx_recon_integrate_inputs_ready(smx_recon r, smx_event inputs_ready) {
  SMX_CHECK_ARG(r != nullptr);
  r->hook_ready = (hipEvent_t)inputs_ready;
  return SMX_OK;
}

int smx_recon_regularize(smx_recon r, smx_stream s, uint32_t frame_index, float regularizer_weight,
                         float radius_factor_for_regularization_neighbors, int32_t regularization_frame_window_size) {
  SMX_CHECK_ARG(r != nullptr);
  SMX_ON_DEVICE(r->device);
  { const int rcj = join_regularizer(r, (hipStream_t)s); if (rcj != SMX_OK) return rcj; }
  return enqueue_regularize(r, (hipStream_t)s, frame_index, radius_factor_for_regularization_neighbors,
                            regularizer_weight, regularization_frame_window_size, false, false, true);
}

// (behind a synchronisation of the object's work) did a gate kernel give up?  Sticky: the map is wrong from that call on.
static int check_handover(smx_recon r, hipStream_t st) {
  uint32_t mark = 0;
  SMX_HIP(hipMemcpyAsync(&mark, r->gate_count + 8, sizeof(mark), hipMemcpyDeviceToHost, st));
  SMX_HIP(hipStreamSynchronize(st));
  if (mark) {
    set_error("the front -> integration hand-over timed out: kernel dispatches of different queues are being serialised (a profiler "
              "in counter mode?) -- the map is invalid; use smx_recon_set_handover_mode(r, 0) in such an environment");
    return SMX_ERR_UNSUPPORTED;
  }
  return SMX_OK;
}

int smx_recon_counts(smx_recon r, smx_stream s, uint32_t* surfel_count, uint32_t* surfels_size) {
  SMX_CHECK_ARG(r != nullptr);
  SMX_ON_DEVICE(r->device);
  { const int rcj = join_regularizer(r, (hipStream_t)s); if (rcj != SMX_OK) return rcj; }
  DevState h;
  SMX_HIP(hipMemcpyAsync(&h, r->st, sizeof(h), hipMemcpyDeviceToHost, (hipStream_t)s));
  SMX_HIP(hipStreamSynchronize((hipStream_t)s));
  { const int rch = check_handover(r, (hipStream_t)s); if (rch != SMX_OK) return rch; }
  if (surfel_count) *surfel_count = h.surfel_count - h.merge_count;  // .h:125-128
  if (surfels_size) *surfels_size = h.surfel_count;
  return SMX_OK;
}

int smx_recon_get_stats(smx_recon r, smx_stream s, smx_recon_stats* out) {
  SMX_CHECK_ARG(r != nullptr && out != nullptr);
  SMX_ON_DEVICE(r->device);
  { const int rcj = join_regularizer(r, (hipStream_t)s); if (rcj != SMX_OK) return rcj; }
  DevState h;
  SMX_HIP(hipMemcpyAsync(&h, r->st, sizeof(h), hipMemcpyDeviceToHost, (hipStream_t)s));
  SMX_HIP(hipStreamSynchronize((hipStream_t)s));
  { const int rch = check_handover(r, (hipStream_t)s); if (rch != SMX_OK) return rch; }
  out->surfels_size = h.surfel_count; out->merge_count = h.merge_count;
  out->n_visible = h.n_visible; out->n_new = h.new_count; out->n_merged = h.n_merged;
  out->n_recent = h.recent_count; out->n_edges = h.n_edges;
  out->n_integrated = h.n_integrated; out->n_replaced = h.n_replaced; out->n_conflict_hits = h.n_conflict_hits;
  out->capacity_clamped = h.capacity_clamped;
  out->n_window_edges = h.n_window_edges; out->n_contributors = h.n_contributors;
  out->n_segments_skipped = h.n_segments_skipped;
  out->regularizer_saturated = h.reg_saturated;
  out->n_pairs = h.n_pairs; out->n_overflow_pairs = h.n_overflow_pairs; out->max_tile_pairs = h.max_tile_pairs;
  return SMX_OK;
}

int smx_recon_transfer_all_to_cpu(smx_recon r, smx_stream s, uint32_t frame_index, smx_surfel_buffers_cpu* buf) {
  SMX_CHECK_ARG(r != nullptr && buf != nullptr);
  SMX_ON_DEVICE(r->device);
  { const int rcj = join_regularizer(r, (hipStream_t)s); if (rcj != SMX_OK) return rcj; }
  hipStream_t st = (hipStream_t)s;
  uint32_t n = 0;
  SMX_HIP(hipMemcpyAsync(&n, &r->st->surfel_count, sizeof(n), hipMemcpyDeviceToHost, st));
  SMX_HIP(hipStreamSynchronize(st));
  buf->frame_index = frame_index;  // cc:345-346
  buf->surfel_count = n;
  if (n == 0) return SMX_OK;
  const size_t bytes = (size_t)n * 4;
  // the 8 rows are packed out of the grouped records into a row-layout staging buffer, then copied row by row
  int rc = acquire_staging(r, st, (size_t)8 * n);
  if (rc != SMX_OK) return rc;
  RowList rl;
  rl.n = 8;
  const int want[8] = {kSmoothX, kSmoothY, kSmoothZ, kRadiusSq, kNormalX, kNormalY, kNormalZ, kLastUpdateStamp};
  for (int k = 0; k < 8; ++k) rl.rows[k] = want[k];
  hipLaunchKernelGGL(k_pack_rows, dim3(r->grid_surfels), dim3(kBlock), 0, st, r->S, rl, r->staging, n);
  SMX_LAUNCH_CHECK();
  struct { int row; void* dst; } rows[8] = {
      {kSmoothX, buf->surfel_x_buffer}, {kSmoothY, buf->surfel_y_buffer}, {kSmoothZ, buf->surfel_z_buffer},
      {kRadiusSq, buf->surfel_radius_squared_buffer},
      {kNormalX, buf->surfel_normal_x_buffer}, {kNormalY, buf->surfel_normal_y_buffer}, {kNormalZ, buf->surfel_normal_z_buffer},
      {kLastUpdateStamp, buf->surfel_last_update_stamp_buffer}};  // cc:348-358
  int k = 0;
  for (auto& q : rows) {
    SMX_CHECK_ARG(q.dst != nullptr);
    SMX_HIP(hipMemcpyAsync(q.dst, r->staging + (size_t)k * n, bytes, hipMemcpyDeviceToHost, st));
    ++k;
  }
  return release_staging(r, st);  // (the copies are still in flight: the caller synchronises, main.cc:1266-1267)
}

int smx_recon_set_delta_tracking(smx_recon r, smx_stream s, int32_t enabled) {
  SMX_CHECK_ARG(r != nullptr && (enabled == 0 || enabled == 1));
  SMX_ON_DEVICE(r->device);
  hipStream_t st = (hipStream_t)s;
  { const int rcj = join_regularizer(r, st); if (rcj != SMX_OK) return rcj; }
  SMX_HIP(hipStreamSynchronize(st));  // no kernel may be using the pointer that changes here
  if (enabled && !r->L.dirty8) {
    int rc = dev_alloc(&r->L.dirty8, (size_t)r->nseg * kSeg, false);
    if (rc == SMX_OK && !r->delta_seg) rc = dev_alloc(&r->delta_seg, (size_t)r->nseg, true);
    if (rc == SMX_OK && !r->delta_total) rc = dev_alloc(&r->delta_total, 1, true);
    if (rc != SMX_OK) return rc;
    SMX_HIP(hipMemset(r->L.dirty8, 1, (size_t)r->nseg * kSeg));  // everything that exists counts as changed
  } else if (!enabled && r->L.dirty8) {
    SMX_HIP(hipFree(r->L.dirty8));
    r->L.dirty8 = nullptr;
  }
  return SMX_OK;
}

int smx_recon_transfer_changed_to_cpu(smx_recon r, smx_stream s, uint32_t frame_index, smx_surfel_delta_cpu* d) {
  SMX_CHECK_ARG(r != nullptr && d != nullptr);
  SMX_ON_DEVICE(r->device);
  if (!r->L.dirty8) { set_error("delta tracking is off (smx_recon_set_delta_tracking)"); return SMX_ERR_INVALID_ARGUMENT; }
  hipStream_t st = (hipStream_t)s;
  { const int rcj = join_regularizer(r, st); if (rcj != SMX_OK) return rcj; }
  hipLaunchKernelGGL(k_delta_count, dim3(r->nseg), dim3(kBlock), 0, st, r->L.dirty8, r->delta_seg, r->st);
  hipLaunchKernelGGL(k_delta_scan, dim3(1), dim3(1024), 0, st, r->delta_seg, r->nseg, r->delta_total);
  SMX_LAUNCH_CHECK();
  uint32_t total = 0, n = 0;
  SMX_HIP(hipMemcpyAsync(&total, r->delta_total, 4, hipMemcpyDeviceToHost, st));
  SMX_HIP(hipMemcpyAsync(&n, &r->st->surfel_count, 4, hipMemcpyDeviceToHost, st));
  SMX_HIP(hipStreamSynchronize(st));
  d->frame_index = frame_index;
  d->surfel_count = n;
  d->count = total;
  if (total > d->capacity) {  // (the marks are untouched: call again with larger arrays)
    set_error("delta of %u slots does not fit the capacity %u", total, d->capacity);
    return SMX_ERR_INVALID_ARGUMENT;
  }
  if (total == 0) return SMX_OK;
  SMX_CHECK_ARG(d->surfel_index && d->x && d->y && d->z && d->radius_squared && d->normal_x && d->normal_y &&
                d->normal_z && d->last_update_stamp);
  int rc = acquire_staging(r, st, (size_t)9 * total);
  if (rc != SMX_OK) return rc;
  hipLaunchKernelGGL(k_delta_gather, dim3(r->nseg), dim3(kBlock), 0, st, r->S, r->L.dirty8, r->delta_seg, r->staging, total,
                     r->st);
  SMX_LAUNCH_CHECK();
  void* dst[9] = {d->surfel_index, d->x, d->y, d->z, d->radius_squared, d->normal_x, d->normal_y, d->normal_z,
                  d->last_update_stamp};
  for (int k = 0; k < 9; ++k)
    SMX_HIP(hipMemcpyAsync(dst[k], r->staging + (size_t)k * total, (size_t)total * 4, hipMemcpyDeviceToHost, st));
  SMX_HIP(hipStreamSynchronize(st));
  return SMX_OK;
}

int smx_recon_export_vertices(smx_recon r, smx_stream s, const smx_buffer_desc* position_buffer,
                              const smx_buffer_desc* color_buffer) {
  SMX_CHECK_ARG(r && position_buffer && color_buffer);
  SMX_ON_DEVICE(r->device);
  { const int rcj = join_regularizer(r, (hipStream_t)s); if (rcj != SMX_OK) return rcj; }
  hipLaunchKernelGGL(k_export, dim3(r->grid_surfels), dim3(kBlock), 0, (hipStream_t)s, r->S,
                     (float*)position_buffer->address, (uint8_t*)color_buffer->address, r->st);
  SMX_LAUNCH_CHECK();
  return SMX_OK;
}

int smx_recon_build_neighbor_index(smx_recon r, smx_stream s, smx_nn nn, float cell_size) {
  SMX_CHECK_ARG(r != nullptr && nn != nullptr && cell_size > 0);
  SMX_ON_DEVICE(r->device);
  hipStream_t st = (hipStream_t)s;
  { const int rcj = join_regularizer(r, st); if (rcj != SMX_OK) return rcj; }
  uint32_t n = 0;
  SMX_HIP(hipMemcpyAsync(&n, &r->st->surfel_count, sizeof(n), hipMemcpyDeviceToHost, st));
  SMX_HIP(hipStreamSynchronize(st));
  if (n == 0) return smx_nn_build(nn, s, nullptr, nullptr, nullptr, 0, cell_size, 1);
  int rc = acquire_staging(r, st, (size_t)3 * n);
  if (rc != SMX_OK) return rc;
  hipLaunchKernelGGL(k_index_rows, dim3(r->grid_surfels), dim3(kBlock), 0, st, r->S, r->staging, n);
  SMX_LAUNCH_CHECK();
  rc = smx_nn_build(nn, s, r->staging, r->staging + n, r->staging + (size_t)2 * n, n, cell_size, 1);
  if (rc != SMX_OK) return rc;
  return release_staging(r, st);
}

int smx_recon_neighbor_candidates(smx_recon r, smx_stream s, smx_nn nn, const uint32_t* surfel_indices,
                                  uint32_t n_indices, float radius_factor_squared, int32_t k,
                                  const uint8_t* state, uint8_t skip_mask, int32_t inputs_on_device,
                                  uint32_t* out_idx, float* out_d2, int32_t* out_count, int32_t outputs_on_device) {
  SMX_CHECK_ARG(r != nullptr && nn != nullptr && radius_factor_squared >= 0 && k >= 1 && k <= 64);
  SMX_ON_DEVICE(r->device);
  SMX_CHECK_ARG(n_indices == 0 || (surfel_indices && out_idx && out_d2 && out_count));
  if (n_indices == 0) return SMX_OK;
  hipStream_t st = (hipStream_t)s;
  { const int rcj = join_regularizer(r, st); if (rcj != SMX_OK) return rcj; }
  // workspace owned by the object, grown only when a batch is larger than any before it (then, and only then, the
  // device is synchronised: earlier batches may still be reading the old buffers)
  if (n_indices > r->cand_cap) {
    SMX_HIP(hipDeviceSynchronize());
    if (r->cand_q) { SMX_HIP(hipFree(r->cand_q)); r->cand_q = nullptr; }
    if (r->cand_slots) { SMX_HIP(hipFree(r->cand_slots)); r->cand_slots = nullptr; }
    r->cand_cap = 0;
    const size_t cap = (size_t)n_indices + n_indices / 8 + 1024;
    int rca = dev_alloc(&r->cand_q, 4 * cap, false);
    if (rca == SMX_OK) rca = dev_alloc(&r->cand_slots, cap, false);
    if (rca != SMX_OK) return rca;
    r->cand_cap = (uint32_t)cap;
  }
  const uint8_t* dstate = state;
  const uint32_t* dslots = surfel_indices;
  if (!inputs_on_device) {
    SMX_HIP(hipMemcpyAsync(r->cand_slots, surfel_indices, (size_t)n_indices * 4, hipMemcpyHostToDevice, st));
    dslots = r->cand_slots;
    if (state) {
      uint32_t n = 0;
      SMX_HIP(hipMemcpyAsync(&n, &r->st->surfel_count, sizeof(n), hipMemcpyDeviceToHost, st));
      SMX_HIP(hipStreamSynchronize(st));
      if (n > r->cand_state_cap) {
        SMX_HIP(hipDeviceSynchronize());
        if (r->cand_state) { SMX_HIP(hipFree(r->cand_state)); r->cand_state = nullptr; }
        r->cand_state_cap = 0;
        const int rca = dev_alloc(&r->cand_state, (size_t)r->S.pitch, false);
        if (rca != SMX_OK) return rca;
        r->cand_state_cap = (uint32_t)r->S.pitch;
      }
      if (n > 0) SMX_HIP(hipMemcpyAsync(r->cand_state, state, n, hipMemcpyHostToDevice, st));
      dstate = r->cand_state;
    }
  }
  float* q = r->cand_q;
  const unsigned blocks = (unsigned)std::min<size_t>(((size_t)n_indices + kBlock - 1) / kBlock, 4096);
  hipLaunchKernelGGL(k_candidate_queries, dim3(blocks), dim3(kBlock), 0, st, r->S, dslots, n_indices, r->st,
                     radius_factor_squared, q);
  SMX_LAUNCH_CHECK();
  // (device inputs and outputs: nothing below allocates or synchronises either)
  return smx_nn_query_batch(nn, s, n_indices, q, q + n_indices, q + (size_t)2 * n_indices, q + (size_t)3 * n_indices, k,
                            dstate, skip_mask, 1, out_idx, out_d2, out_count, outputs_on_device);
}

int smx_recon_check_triangles(smx_recon r, smx_stream s, const uint32_t* triangles, uint32_t n_triangles,
                              float long_edge_total_factor_squared, uint8_t* flags, int32_t on_device) {
  SMX_CHECK_ARG(r != nullptr && (n_triangles == 0 || (triangles && flags)));
  SMX_ON_DEVICE(r->device);
  if (n_triangles == 0) return SMX_OK;
  hipStream_t st = (hipStream_t)s;
  { const int rcj = join_regularizer(r, st); if (rcj != SMX_OK) return rcj; }
  uint32_t* dtri = nullptr;
  uint8_t* dflags = nullptr;
  int rc = SMX_OK;
  hipError_t e = hipSuccess;
  auto fail = [&](hipError_t err) { set_error("check triangles failed: %s", hipGetErrorString(err)); rc = SMX_ERR_HIP; };
  if (!on_device) {
    if ((e = hipMalloc(reinterpret_cast<void**>(&dtri), (size_t)n_triangles * 12)) != hipSuccess) fail(e);
    else if ((e = hipMalloc(reinterpret_cast<void**>(&dflags), n_triangles)) != hipSuccess) fail(e);
    else if ((e = hipMemcpyAsync(dtri, triangles, (size_t)n_triangles * 12, hipMemcpyHostToDevice, st)) != hipSuccess) fail(e);
  }
  if (rc == SMX_OK) {
    const unsigned blocks = (unsigned)std::min<size_t>(((size_t)n_triangles + kBlock - 1) / kBlock, 8192);
    hipLaunchKernelGGL(k_check_triangles, dim3(blocks), dim3(kBlock), 0, st, r->S, on_device ? triangles : dtri,
                       n_triangles, r->st, long_edge_total_factor_squared, on_device ? flags : dflags);
    if ((e = hipGetLastError()) != hipSuccess) fail(e);
  }
  if (!on_device) {
    if (rc == SMX_OK && (e = hipMemcpyAsync(flags, dflags, n_triangles, hipMemcpyDeviceToHost, st)) != hipSuccess) fail(e);
    if ((e = hipStreamSynchronize(st)) != hipSuccess && rc == SMX_OK) fail(e);
    if (dtri) (void)hipFree(dtri);
    if (dflags) (void)hipFree(dflags);
  }
  return rc;
}

namespace {
// stage times of one stamp record (ms); false if the record is not that call's or the call did not get through
bool stage_ms_from_stamps(const unsigned long long* t, unsigned long long seq, int khz, float out_ms[7]) {
  // (a call whose second half was left out -- smx_recon_debug_set_skip front-only -- has no integration stamp: the stages that
  // ran are reported, the others stay 0)
  if (t[kTsSeq] != seq || t[kTsCullBegin] == 0) return false;
  auto ms = [&](unsigned long long a, unsigned long long b) { return (b > a && a != 0) ? (float)((double)(b - a) / (double)khz) : 0.0f; };
  auto first = [](unsigned long long a, unsigned long long b, unsigned long long c) { return a ? a : (b ? b : c); };
  // a stage ends where the next launch of its stream begins; the tail workgroups' maximum where there is no such launch
  const unsigned long long tiles_end = first(t[kTsBlendBegin], t[kTsTilesEnd], t[kTsCullBegin]);
  const unsigned long long int_end = first(t[kTsUpdBegin], t[kTsIntEnd], first(t[kTsIntBegin], t[kTsBlendEnd], tiles_end));
  const unsigned long long upd_end = first(t[kTsRegBegin], t[kTsUpdEnd], int_end);
  out_ms[0] = ms(t[kTsCullBegin], tiles_end);        // data association: cull step, pass A, association tiles
  out_ms[1] = 0.0f;                                  // surfel merging: decided inside the tile kernel, applied by k_integrate
  out_ms[2] = ms(tiles_end, t[kTsBlendEnd]);         // measurement blending
  out_ms[3] = ms(t[kTsIntBegin], int_end);           // integration (+ the new-surfel flag and rank pass)
  out_ms[4] = ms(int_end, upd_end);                  // neighbour update (+ creation, same launch)
  out_ms[5] = 0.0f;                                  // new surfel creation: inside the neighbour-update launch
  out_ms[6] = ms(upd_end, t[kTsRegEnd]);             // regularisation: pass B, edges, step
  return true;
}
// (a plain synchronous copy: both callers have waited for the work they ask about, and the object creates no stream it
// does not need -- every queue more raises the odds of the slow mode of smx_buffer.hip: smx_runtime_defaults)
int copy_stamp_ring(smx_recon r) {
  SMX_HIP(hipMemcpy(r->ts_host, r->ts_ring, sizeof(unsigned long long) * kTsRing * kTsWords, hipMemcpyDeviceToHost));
  return SMX_OK;
}
}  // namespace

int smx_recon_get_timings(smx_recon r, float out_ms[7]) {
  SMX_CHECK_ARG(r != nullptr && out_ms != nullptr);
  SMX_ON_DEVICE(r->device);
  for (int i = 0; i < 7; ++i) out_ms[i] = 0;
  if (r->timing_enabled & 1) {
    // (measurement / comparison mode: the reference's own 14 event records, cc:131-319 -- every record a packet between two
    // kernels of a stream that is never idle: 4 000 instead of 6 000 frames/s at 640 x 480, profiles/r28_stage_timing_C2.txt)
    if (!r->have_timings) return SMX_OK;
    SMX_HIP(hipEventSynchronize(r->ev[13]));  // cc:420
    for (int i = 0; i < 7; ++i) SMX_HIP(hipEventElapsedTime(&out_ms[i], r->ev[2 * i], r->ev[2 * i + 1]));
    return SMX_OK;
  }
  if (!(r->timing_enabled & 4) || r->ts_seq == 0) return SMX_OK;
  // Like the reference (cudaEventSynchronize(regularization_end_event_), cc:420): wait until the last call is through --
  // all of its work precedes the end of the internal stream's queue (or of the caller's stream's, not pipelined).
  if (r->reg_pending) { SMX_HIP(hipEventRecord(r->ev_reg, r->reg_stream)); SMX_HIP(hipEventSynchronize(r->ev_reg)); }
  if (!r->reg_pending || !r->overlap_enabled) SMX_HIP(hipStreamSynchronize(r->last_stream));
  { const int rc = copy_stamp_ring(r); if (rc != SMX_OK) return rc; }
  (void)stage_ms_from_stamps(r->ts_host + (size_t)(r->ts_seq % kTsRing) * kTsWords, r->ts_seq, r->wall_khz, out_ms);
  return SMX_OK;
}

int smx_recon_debug_stamp_ring(smx_recon r, uint64_t* out, int32_t capacity_words, int32_t* wall_clock_khz) {
  SMX_CHECK_ARG(r != nullptr && out != nullptr && capacity_words >= kTsRing * kTsWords);
  SMX_ON_DEVICE(r->device);
  // (measurement: the raw stamp records of the last calls, read from the device ring -- the caller has synchronised)
  { const int rc = copy_stamp_ring(r); if (rc != SMX_OK) return rc; }
  for (int k = 0; k < kTsRing * kTsWords; ++k) out[k] = r->ts_host[k];
  if (wall_clock_khz) *wall_clock_khz = r->wall_khz;
  return SMX_OK;
}

int smx_recon_get_timings_nowait(smx_recon r, float out_ms[7], uint64_t* call_number) {
  SMX_CHECK_ARG(r != nullptr && out_ms != nullptr);
  for (int i = 0; i < 7; ++i) out_ms[i] = 0;
  if (call_number) *call_number = 0;
  if (!(r->timing_enabled & 4) || r->ts_seq == 0 || !r->ts_mapped) return SMX_OK;
  // No device operation at all: the records the tile kernel has copied into page-locked memory (every call copies the
  // record of the call before the previous one), newest whole one.
  unsigned long long best = 0, rec[kTsWords];
  for (int k = 0; k < kTsRing; ++k) {
    volatile unsigned long long* t = r->ts_mapped + (size_t)k * kTsWords;
    unsigned long long w[kTsWords];
    unsigned long long check = kTsCheckSalt;
    for (int j = 0; j < kTsWords; ++j) w[j] = t[j];
    for (int j = 0; j < kTsSeqTail; ++j) check ^= w[j];
    if (w[kTsSeq] == 0 || check != w[kTsSeqTail] || w[kTsSeq] <= best) continue;   // (empty, or torn: being rewritten)
    best = w[kTsSeq];
    memcpy(rec, w, sizeof(rec));
  }
  if (best && stage_ms_from_stamps(rec, best, r->wall_khz, out_ms) && call_number) *call_number = best;
  return SMX_OK;
}

int smx_recon_debug_download_surfels(smx_recon r, smx_stream s, float* rows, uint32_t count) {
  SMX_CHECK_ARG(r != nullptr && rows != nullptr && count <= r->max_surfels);
  SMX_ON_DEVICE(r->device);
  if (count == 0) return SMX_OK;
  { const int rcj = join_regularizer(r, (hipStream_t)s); if (rcj != SMX_OK) return rcj; }
  int rc = acquire_staging(r, (hipStream_t)s, (size_t)kRows * count);
  if (rc != SMX_OK) return rc;
  RowList rl;
  rl.n = kRows;
  for (int k = 0; k < kRows; ++k) rl.rows[k] = k;
  hipLaunchKernelGGL(k_pack_rows, dim3(r->grid_surfels), dim3(kBlock), 0, (hipStream_t)s, r->S, rl, r->staging, count);
  SMX_HIP(hipMemcpyAsync(rows, r->staging, (size_t)kRows * count * 4, hipMemcpyDeviceToHost, (hipStream_t)s));
  SMX_HIP(hipStreamSynchronize((hipStream_t)s));
  return SMX_OK;
}

// After surfel attributes were changed from outside the frame loop (state upload, deformation): drop the work
// lists and segment boxes (count 0 = no box, the segment is scanned) and rebuild the flag table -- its detach bits
// come from the colour words, the recent bits are refreshed per frame.
static int invalidate_derived(smx_recon r, hipStream_t st) {
  SMX_HIP(hipMemsetAsync(r->L.vis_seg, 0, (size_t)r->nseg * 4, st));
  SMX_HIP(hipMemsetAsync(r->L.seg_box, 0, (size_t)r->nseg * 8 * sizeof(float), st));
  SMX_HIP(hipMemsetAsync(r->L.recent_seg, 0, (size_t)r->nsegB * 4, st));
  SMX_HIP(hipMemsetAsync(r->vis_count_set[0], 0, sizeof(uint32_t) * kSubLists * kCountStride, st));
  SMX_HIP(hipMemsetAsync(r->vis_count_set[1], 0, sizeof(uint32_t) * kSubLists * kCountStride, st));
  SMX_HIP(hipMemsetAsync(r->L.rec_chunks.count, 0, sizeof(uint32_t) * kSubLists * kCountStride, st));
  hipLaunchKernelGGL(k_rebuild_flags, dim3(r->grid_surfels), dim3(kBlock), 0, st, r->S, 0u, 0x7FFFFFFF, r->L.flags8, r->st);
  SMX_LAUNCH_CHECK();
  r->table_valid = false;
  r->hot_holdoff = 3;     // (flags and links may have been rewritten: pass B gathers everything for two calls)
  r->have_frame = false;  // (the boxes are gone, so any frame index may follow)
  return SMX_OK;
}

int smx_recon_debug_upload_surfels(smx_recon r, smx_stream s, const float* rows, uint32_t count, uint32_t merge_count) {
  SMX_CHECK_ARG(r != nullptr && count <= r->max_surfels && (rows != nullptr || count == 0));
  SMX_ON_DEVICE(r->device);
  { const int rcj = join_regularizer(r, (hipStream_t)s); if (rcj != SMX_OK) return rcj; }
  hipStream_t st = (hipStream_t)s;
  if (count) {
    int rc = acquire_staging(r, st, (size_t)kRows * count);
    if (rc != SMX_OK) return rc;
    SMX_HIP(hipMemcpyAsync(r->staging, rows, (size_t)kRows * count * 4, hipMemcpyHostToDevice, st));
    RowList rl;
    rl.n = kRows;
    for (int k = 0; k < kRows; ++k) rl.rows[k] = k;
    hipLaunchKernelGGL(k_unpack_rows, dim3(r->grid_surfels), dim3(kBlock), 0, st, r->S, rl, r->staging, count);
  }
  DevState h;
  memset(&h, 0, sizeof(h));
  h.surfel_count = count; h.merge_count = merge_count;
  SMX_HIP(hipMemcpyAsync(r->st, &h, sizeof(h), hipMemcpyHostToDevice, st));
  SMX_HIP(hipMemsetAsync(r->grad_acc, 0, 2 * r->S.pitch * sizeof(long long), st));
  SMX_HIP(hipMemsetAsync(r->fb.count, 0, (size_t)r->nsegB * kCountStride * sizeof(uint32_t), st));
  SMX_HIP(hipMemsetAsync(r->merge_flag, 0, r->S.pitch, st));
  if (r->L.dirty8) SMX_HIP(hipMemsetAsync(r->L.dirty8, 1, (size_t)r->nseg * kSeg, st));
  int rc = invalidate_derived(r, st);
  if (rc != SMX_OK) return rc;
  SMX_HIP(hipStreamSynchronize(st));
  return SMX_OK;
}

int smx_recon_deform_by_creation_frame(smx_recon r, smx_stream s, const float* frame_T, uint32_t n_frames,
                                       const uint8_t* reactivate, uint32_t frame_index, int32_t inputs_on_device) {
  SMX_CHECK_ARG(r != nullptr && (n_frames == 0 || frame_T != nullptr));
  SMX_ON_DEVICE(r->device);
  if (n_frames == 0) return SMX_OK;
  hipStream_t st = (hipStream_t)s;
  { const int rcj = join_regularizer(r, st); if (rcj != SMX_OK) return rcj; }
  float* dT = nullptr;
  uint8_t* dre = nullptr;
  int rc = SMX_OK;
  hipError_t e = hipSuccess;
  auto fail = [&](hipError_t err) { set_error("deform failed: %s", hipGetErrorString(err)); rc = SMX_ERR_HIP; };
  if (!inputs_on_device) {
    if ((e = hipMalloc(reinterpret_cast<void**>(&dT), (size_t)n_frames * 48)) != hipSuccess) fail(e);
    else if ((e = hipMemcpyAsync(dT, frame_T, (size_t)n_frames * 48, hipMemcpyHostToDevice, st)) != hipSuccess) fail(e);
    if (rc == SMX_OK && reactivate) {
      if ((e = hipMalloc(reinterpret_cast<void**>(&dre), n_frames)) != hipSuccess) fail(e);
      else if ((e = hipMemcpyAsync(dre, reactivate, n_frames, hipMemcpyHostToDevice, st)) != hipSuccess) fail(e);
    }
  }
  if (rc == SMX_OK) {
    hipLaunchKernelGGL(k_deform_by_creation_frame, dim3(r->grid_surfels), dim3(kBlock), 0, st, r->S,
                       inputs_on_device ? frame_T : dT, n_frames, inputs_on_device ? reactivate : dre, frame_index,
                       r->L.dirty8, r->st);
    if ((e = hipGetLastError()) != hipSuccess) fail(e);
  }
  // positions and stamps changed behind the work lists, segment boxes and the flag table
  if (rc == SMX_OK) rc = invalidate_derived(r, st);
  if (!inputs_on_device) {
    if ((e = hipStreamSynchronize(st)) != hipSuccess && rc == SMX_OK) fail(e);
    if (dT) (void)hipFree(dT);
    if (dre) (void)hipFree(dre);
  }
  return rc;
}

int smx_recon_debug_download_scratch(smx_recon r, smx_stream s, int32_t which, void* dst) {
  SMX_CHECK_ARG(r != nullptr && dst != nullptr);
  SMX_ON_DEVICE(r->device);
  { const int rcj = join_regularizer(r, (hipStream_t)s); if (rcj != SMX_OK) return rcj; }
  hipStream_t st = (hipStream_t)s;
  const size_t P = (size_t)r->W * r->H;
  const void* src = nullptr; size_t bytes = 0;
  switch (which) {
    case SMX_SCRATCH_SUPPORTING: src = r->sc.supporting; bytes = P * 4; break;
    case SMX_SCRATCH_SUPPORT_COUNTS: src = r->sc.counts; bytes = P * 4; break;
    case SMX_SCRATCH_DEPTH_SUMS: src = r->sc.depth_sums; bytes = P * 8; break;
    case SMX_SCRATCH_FIRST_DEPTH: src = r->sc.first_depth; bytes = P * 4; break;
    case SMX_SCRATCH_NEW_FLAGS: src = r->new_flags; bytes = P; break;
    case SMX_SCRATCH_CONFLICTING:
      hipLaunchKernelGGL(k_decode_conflicting, dim3(div_up((long long)P, kBlock)), dim3(kBlock), 0, st, r->sc.confl_key, r->tmp_u32, (int)P);
      src = r->tmp_u32; bytes = P * 4; break;
    case SMX_SCRATCH_NEW_INDICES:
      hipLaunchKernelGGL(k_global_ranks, dim3(div_up((long long)P, kBlock)), dim3(kBlock), 0, st, r->new_ranks, r->block_offsets, r->tmp_u32, (int)P);
      src = r->tmp_u32; bytes = P * 4; break;
    default: set_error("unknown scratch id %d", which); return SMX_ERR_INVALID_ARGUMENT;
  }
  SMX_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, st));
  SMX_HIP(hipStreamSynchronize(st));
  return SMX_OK;
}

int smx_recon_debug_count_skipped_segments(smx_recon r, smx_stream s, uint32_t* out) {
  SMX_CHECK_ARG(r != nullptr && out != nullptr);
  SMX_ON_DEVICE(r->device);
  { const int rcj = join_regularizer(r, (hipStream_t)s); if (rcj != SMX_OK) return rcj; }
  std::vector<uint32_t> marks((size_t)r->nsegB);
  SMX_HIP(hipMemcpyAsync(marks.data(), r->L.recent_seg, marks.size() * 4, hipMemcpyDeviceToHost, (hipStream_t)s));
  SMX_HIP(hipStreamSynchronize((hipStream_t)s));
  uint32_t n = 0;
  for (uint32_t m : marks) n += m == kInvalid ? 1u : 0u;
  *out = n;
  return SMX_OK;
}

}  // extern "C"
