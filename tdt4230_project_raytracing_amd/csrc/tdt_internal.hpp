// Internal definitions shared by the translation units of libtdtrt.so (tdt_rt.hip: context, trace, helpers;
// tdt_multi.hip: the multi-device context; tdt_build.hip: the GPU octree builder; tdt_edit.hip: voxel edits).
// Nothing here is part of the C ABI (include/tdt_rt.h).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>
#include <vector>

#include "tdt_rt.h"

constexpr int kNumSlots = 8;

// What a trace dispatch depends on besides the sample range — compared with what the recorded pixel costs were measured on
// (see launch()).  A plain struct, zero-filled before it is written, so that a field added later is part of the comparison
// by construction (it used to be a byte string that silently dropped what did not fit).
struct CostSig {
  int32_t cam_i[4]; float cam_f[12]; int32_t part[2];
  float octree_f[7]; int32_t octree_i[3];
  int32_t cover[2], image[2];
  struct { const void *buffer; unsigned long long version; } slot[kNumSlots];
};

namespace tdt { struct Multi; struct EditScratch; }

struct tdt_buffer {
  tdt_ctx *ctx;
  void *dev;
  size_t bytes;
  unsigned long long version;   // bumped by every write: invalidates derived data (LDS table image)
  unsigned char shadow[64];   // first bytes, host side: the octree uniform blocks are read from here
  std::vector<tdt_buffer *> replicas;   // multi-device context: the per-device buffers behind this handle (dev is null)
};

struct tdt_image {
  tdt_ctx *ctx;
  float *dev;
  int w, h;
  bool owned;
  tdt_image *full;              // multi-device context: the assembled frame on the first device (dev aliases its memory)
};

struct tdt_ctx {
  int device;
  hipStream_t stream;
  bool own_stream;
  std::string err;
  tdt_buffer *ssbo[kNumSlots];
  tdt_buffer *atomic0;
  tdt_image *image0;
  unsigned long long *counters;
  unsigned int *queue; unsigned queue_parity, order_parity;   // two pixel-queue heads / two sets of sort counters, used alternately (each launch zeroes the other)
  uint16_t *packed;             // LDS-table image of the bound cells buffer
  const tdt_buffer *packed_of;  // which buffer/version `packed` was built from
  unsigned long long packed_version;
  uint32_t *slot_cost, *slot_acc, *slot_order, *order_hist;   // per queue slot: cost feedback of the last trace dispatch, the hand-out order derived from it; 2 x 256 sort counters
  uint32_t cost_dispatches;            // dispatches summed into slot_cost so far
  uint32_t acc_samples, last_launch_samples;   // samples per pixel behind slot_acc / traced by the last launch
  int two_phase_min_spp;               // TDT_TWO_PHASE_MIN_SPP: frames with fewer samples per pixel take one pass (16)
  int probe_div;                       // TDT_PROBE_DIV: probe samples of a two-phase frame = spp / probe_div (16)
  float order_blend;                   // TDT_ORDER_BLEND: weight of the 8x8-tile mean in a thin (probe) cost estimate
  uint32_t tile_capacity, cost_tiles;  // allocation size (work-groups); number of work-groups slot_cost holds the last dispatch's costs for (0: none)
  int cost_range[2]; bool order_exact, no_order_reuse;   // sample range of the launch that recorded slot_cost; slot_order was sorted from the costs of that very launch repeated (TDT_NO_ORDER_REUSE=1: sort every frame)
  CostSig cost_sig;                    // what those costs were measured on (camera, octree parameters, buffer versions, partition)
  // miss pre-pass (cameras outside the octree: miss_prepass_kernel): done flag per queue slot, the filtered hand-out order, scratch
  uint8_t *slot_done; uint32_t *slot_live, *filter_counts; uint32_t done_capacity; bool use_done, no_prepass;   // use_done: set for the launches of a frame whose pre-pass ran (TDT_NO_PREPASS=1: never)
  bool no_cost_order;           // TDT_NO_COST_ORDER=1: always hand work-groups out in image order
  uint32_t *scan;               // device scratch of scan_cells_kernel
  uint32_t max_parent_value, max_any_value, live_nodes;   // its result for `packed_of` (live_nodes: one past the last node that is not all zeros)
  float *thr; int32_t thr_cc; uint32_t thr_ic_bits, thr_n; float thr_f0max; bool thr_ok, no_table_form;   // FORM_TABLE builds: per-cell x-index thresholds for (cell_count, inv_cell_count) over thr_n cells (TDT_NO_TABLE_FORM=1: off)
  int num_cus;
  bool force_generic;   // TDT_FORCE_GENERIC=1: always run the literal-arithmetic kernel (A/B testing)
  int event_threshold;  // TDT_EVENT_THRESHOLD=n fixes the event threshold (experiments); 0 = adaptive
  int force_smooth; bool no_cost_accum; float max_share;   // TDT_ORDER_SMOOTH / TDT_NO_COST_ACCUM / TDT_MAX_SHARE (diagnostics)
  int event_clamp;      // TDT_EVENT_CLAMP: upper clamp of the adaptive event threshold
  float event_k;        // TDT_EVENT_K overrides the adaptive threshold's r (0: chosen from the tree size)
  void *frame_carry; size_t frame_carry_bytes;   // hit-record carry between the two phases of a frame (tdt_dispatch_compute)
  bool no_two_phase;                             // TDT_NO_TWO_PHASE=1
  uint16_t *full_grid; const tdt_buffer *full_of; unsigned long long full_version; int full_depth; bool full_ok, no_full;   // whole-depth lookup table of small resident trees (TDT_NO_FULL_GRID=1: off)
  uint32_t *brick_grid; void *bricks; size_t bricks_bytes; const tdt_buffer *brick_of; unsigned long long brick_version; int brick_depth; bool brick_ok, no_bricks;   // BRICK builds (depth-8 / 9 trees that are not LDS-resident; TDT_NO_BRICKS=1: off)
  bool carry_final;                              // set around the last launch of a two-phase frame: its records need not be stored
  bool probe_launch;                             // set around the probe launch of a two-phase frame (kernel name only)
  bool phase_timing; hipEvent_t phase_ev[4]; int phase_n;   // tdt_debug_phase_timing: events around the launches of the last frame
  uint32_t *present; size_t present_bytes;   // staging of tdt_image_read_rgba8
  uint32_t *pixel_log; size_t pixel_log_u32;   // TDT_PIXEL_LOG diagnostics (instrumented dispatches only)
  unsigned long long *stats;    // tdt_debug_stats: pass statistics of -DTDT_STATS builds (null otherwise)
  int last_variant[6];  // tdt_debug_last_variant: the build the last trace launch ran
  bool no_specialise;   // TDT_NO_SPECIALISE=1: never pick a scene-specialised kernel (A/B testing)
  tdt::Multi *multi;            // non-null: this is a multi-device context (tdt_ctx_create_multi); see tdt_multi.hip
  tdt::EditScratch *edit;       // scratch of the parallel voxel-edit path (tdt_edit.hip), allocated on first use
  std::vector<tdt_buffer *> buffers;
  std::vector<tdt_image *> images;
  std::vector<tdt_compute *> computes;
};

struct tdt_compute {
  tdt_ctx *ctx;
  int kind;
  // `uniform Camera camera` raytracer.comp:133-146; GL initialises uniforms to 0
  int32_t image_width, image_height, samples_per_pixel, max_bounce;
  float horizontal[3], vertical[3], lower_left_corner[3], origin[3];
  int part_rank, part_world;
  std::vector<tdt_compute *> replicas;  // multi-device context: the per-device programs behind this handle
};

namespace tdt {

int fail(tdt_ctx *ctx, int code, const std::string &msg);           // records the message, returns the code
int hip_fail(tdt_ctx *ctx, hipError_t e, const char *what);
#define TDT_HIP(ctx, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return tdt::hip_fail((ctx), e_, #call); } while (0)

template <class T> void erase_from(std::vector<T *> &v, T *p) {
  for (size_t i = 0; i < v.size(); i++) if (v[i] == p) { v.erase(v.begin() + i); return; }
}

// ComputeShader::dispatch_compute's group arithmetic (compute_shader.rs:30-32) and what it covers
struct Cover { int groups_x, groups_y, cover_w, cover_h; };
Cover cover_of(const tdt_compute *c, int width, int height);
// 32x32 work-groups of the covered image and the ones this rank owns (t % world == rank)
struct Tiles { int tiles_x, tiles_y, total, owned; };
Tiles tiles_of(const tdt_compute *c, const Cover &k);

// a tdt_buffer over device memory some kernel of this library filled: `dev` comes from hipMalloc(bytes + 16) with the
// 16 bytes of slack zeroed, and belongs to the buffer from here on (tdt_rt.hip)
int adopt_device_buffer(tdt_ctx *ctx, void *dev, size_t bytes, tdt_buffer **out);

// ---- tdt_multi.hip: every public entry point forwards here when the handle belongs to a multi-device context ----
void multi_destroy(tdt_ctx *ctx);
int multi_finish(tdt_ctx *ctx);
int multi_compute_create(tdt_ctx *ctx, int kind, tdt_compute **out);
void multi_compute_destroy(tdt_compute *c);
int multi_set_i32(tdt_compute *c, const char *name, int32_t v);
int multi_set_vec3f(tdt_compute *c, const char *name, float x, float y, float z);
int multi_buffer_create(tdt_ctx *ctx, const void *data, size_t bytes, tdt_buffer **out);
void multi_buffer_destroy(tdt_buffer *b);
int multi_bind_buffer_base(tdt_ctx *ctx, int target, unsigned slot, tdt_buffer *b);
int multi_buffer_sub_data(tdt_buffer *b, size_t offset, size_t bytes, const void *data);
int multi_image_create(tdt_ctx *ctx, void *device_ptr, int width, int height, tdt_image **out);
void multi_image_destroy(tdt_image *img);
int multi_dispatch_compute(tdt_compute *c, int width, int height, int depth);
int multi_dispatch_accumulate(tdt_compute *c, int width, int height, int depth, int spp_begin, int spp_count, void *carry);
int multi_dispatch_resolve(tdt_compute *c, int width, int height, int depth, int total_spp);
int multi_dispatch_counted(tdt_compute *c, int width, int height, int depth, uint64_t counts[8]);
int multi_forget_costs(tdt_ctx *ctx);
tdt_ctx *multi_first_member(tdt_ctx *front);

// ---- tdt_edit.hip ----
int launch_update(tdt_compute *c, int width, int height, int depth);
void edit_scratch_destroy(tdt_ctx *ctx);

}  // namespace tdt
