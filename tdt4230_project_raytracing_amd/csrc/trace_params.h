// Kernel argument block of the trace kernels (passed by value: lands in SGPRs via s_load).
#pragma once
#include <stdint.h>

struct TraceParams {
  // `uniform Camera camera` raytracer.comp:133-146
  int32_t image_width, image_height;
  float hor[3], ver[3], llc[3], org[3];
  int32_t samples_per_pixel, max_bounce;
  // OctreeFloats / OctreeInts raytracer.comp:150-166 (uniform data: kept in scalar registers
  // instead of being re-read from the SSBO every traversal step)
  float min_x, min_y, min_z, scale, inv_scale, inv_cell_count;
  int32_t max_depth, max_iter, cell_count;
  // SSBO payloads (device pointers) and their sizes in dwords (reads past the end return 0)
  const uint32_t *cells;      uint32_t cells_dwords;
  const uint32_t *materials;  uint32_t materials_dwords;
  const uint32_t *albedos;    uint32_t albedos_dwords;
  const uint32_t *metal;      uint32_t metal_dwords;
  const uint32_t *dielectric; uint32_t dielectric_dwords;
  // output image (RGBA32F, row 0 = bottom) and optional per-pixel carry (16 floats / pixel)
  float *image;
  float *carry;
  uint32_t *pixel_log;          // TDT_PIXEL_LOG diagnostics (8 u32 per queue slot), or null
  unsigned long long *counters; // instrumented launches only: event totals (see Counters)
  unsigned int *queue;          // global pixel queue head (zero at launch)
  unsigned int *queue_next;     // the next launch's head: zeroed by this launch (two heads alternate), or null
  const uint32_t *slot_order;   // queue slots (work-group * 1024 + pixel), most expensive first (from the previous dispatch), or null
  uint32_t *slot_cost;          // per queue slot: pixel time of THIS dispatch (feeds the next one), or null
  const uint16_t *packed;       // cells [0, lds_cells) re-encoded as 16 bits per node: value << 2 | code
  uint32_t lds_nodes;           // number of nodes (8 per cell) staged in LDS by every block
  const uint16_t *full_grid;    // FULL builds: one entry per finest-level voxel position (8^max_depth), see build_full_grid_kernel
  const uint32_t *brick_grid;   // BRICK builds: the 5-level jump table with brick headers (32^3 x 32 bit), see build_bricks_kernel
  const float *thr; uint32_t thr_cells; float thr_f0max;   // FORM_TABLE builds: (F1, F2) per cell index below thr_cells and the largest F0, see x_thresholds (trace_device.hpp)
  const void *bricks;           // BRICK builds: per level-5 position 27 x 64 (depth 8) or 81 x 256 (depth 9) 16-bit entries: what the levels below 5 end on, by decision sequence
  int32_t compact;             // 1: image is this rank's tile buffer [owned tile k][32][32] RGBA
  int32_t cover_w, cover_h;    // pixels a dispatch covers: min(32*groups, image size)
  int32_t tiles_x;             // 32x32 work-groups per row = ceil(cover_w / 32)
  uint32_t tiles_x_magic;      // floor(2^32 / tiles_x) + 1 when every work-group index t of the dispatch has t * tiles_x < 2^32 (then t / tiles_x = mulhi(t, magic)); else 0
  int32_t owned_tiles;         // work-groups this rank traces
  int32_t part_rank, part_world;  // work-group partition: tile t (row-major) belongs to rank t % world
  int32_t spp_begin, spp_count;   // sample range of this launch
  const uint32_t *plan;           // device word written by order_plan_kernel: 1 = draw queue slots exactly as asked, 0 / null = in batches of 64
  int32_t carry_final;            // 1: read the records in `carry` at the start of a pixel but do not store them at its end
  int32_t accumulate;             // 0: main() as written (resolve and store); 1: add the samples to the running sums in `image`
  int32_t mode;                // 0 render (sum, sqrt, clamp, store), 1 accumulate into image, 2 resolve
  int32_t total_spp;           // resolve divisor
  int32_t event_threshold;     // > 0: fixed number of lanes that must wait for the event code; 0: adaptive
  float event_clamp;           // upper clamp of the adaptive threshold
  float event_k;               // adaptive threshold: r = C_t / (2 C_e) of the model in trace_kernel
#ifdef TDT_STATS
  unsigned long long *stats;   // -DTDT_STATS builds only (tools/loss_budget.py): pass / lane statistics of the product kernels, see TDT_ST in trace_kernel
#endif
};
