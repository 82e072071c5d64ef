// libtdtrt.so — the C ABI of include/tdt_rt.h over hand-written gfx950 kernels.
//
// Host part: a one-for-one stand-in for the reference's `src/renderer` GL wrappers (context,
// buffer upload + binding, RGBA32F image, uniform-by-name, dispatch_compute); device part: the
// per-pixel trace of assets/shaders/raytracer.comp.  There is no CPU path in this library.
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "tdt_rt.h"
#include "trace_device.hpp"
#include "trace_params.h"

#ifndef TDT_BLOCK
#define TDT_BLOCK 1024      // threads per persistent block; TDT_BLOCKS_PER_CU of them share a CU
#define TDT_BLOCKS_PER_CU 1
#endif
// ============================================================================ kernels ======
namespace tdt {

// ---- work decomposition ------------------------------------------------------------------
// Persistent 1024-thread blocks, one per CU (the block's LDS holds the top of the octree, see
// NodeSource).  Pixels are NOT bound to threads: the covered image is a single global queue of
// pixel slots q = k * 1024 + p — k-th 32x32 work-group owned by this rank (row-major group index
// t = rank + k * world: SURVEY §8e), p-th pixel inside it in 8x8-tile-major order — and every lane
// pulls its next pixel from it (one atomic per wave: ballot + prefix count) when it has finished
// all samples of its current one.  A lane that drew cheap (sky) pixels keeps working instead of
// idling behind the most expensive pixel of its tile and all CUs stay busy until the queue is empty.
// The queue is handed out in slot order — then the chip works on one narrow band of the image at a time and
// neighbouring lanes share octree nodes and cache lines — or through P.slot_order, a permutation of the slots
// built from the previous dispatch's per-pixel work counts ("Cost-feedback scheduling" below).
TDT_DEV void decode_pixel(const TraceParams &P, int k, uint32_t p, int &x, int &y, size_t &pix, bool &inside) {
  const int t = P.part_rank + k * P.part_world;
  int gx, gy;
  if (P.tiles_x_magic) { gy = (int)__umulhi((uint32_t)t, P.tiles_x_magic); gx = t - gy * P.tiles_x; }      // (uniform; see TraceParams)
  else { gx = t % P.tiles_x; gy = t / P.tiles_x; }
  const int tile = (int)(p >> 6), w = (int)(p & 63);
  const int lx = (tile & 3) * 8 + (w & 7), ly = (tile >> 2) * 8 + (w >> 3);
  x = gx * 32 + lx;
  y = gy * 32 + ly;
  pix = P.compact ? ((size_t)k * 1024 + (size_t)(ly * 32 + lx)) : ((size_t)y * (size_t)P.image_width + (size_t)x);
  inside = x < P.cover_w && y < P.cover_h;
}

// pixel addressing of the one-thread-per-pixel helper kernels (resolve)
TDT_DEV bool pixel_of_thread(const TraceParams &P, int &x, int &y, size_t &pix) {
  const int k = blockIdx.x >> 2, sub = blockIdx.x & 3;
  bool inside;
  decode_pixel(P, k, (uint32_t)(sub * 256 + threadIdx.x), x, y, pix, inside);
  return inside;
}

TDT_DEV uint32_t wave_sum(uint32_t v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}


// Lane states of the flattened path tracer.  The reference's loop nest
//   for sample { while bounce { for traversal-step { for level } } }        (rc:238,271,410,372)
// is run per lane as a state machine, so lanes of one wave can be in different samples /
// bounces / steps at the same time; every lane still executes exactly the reference's sequence
// of operations for its own pixel, in the same order (bit-identical sums).
enum : int { ST_DONE = -1, ST_TRAVERSE = 0, ST_HIT = 1, ST_END = 2, ST_FETCH = 3, ST_PRIMARY = 4, ST_NEWRAY = 5 };   // events are the positive states

template <bool AS_MASKS> struct HitOwed;             // (see its use in trace_kernel)
template <> struct HitOwed<true> {
  bool use_leaf = false, leaf_rec = false;
  TDT_DEV void set(bool later_step, bool cube_ok) { use_leaf = later_step; leaf_rec = later_step && cube_ok; }
  TDT_DEV bool leaf_site() const { return use_leaf; }
  TDT_DEV bool new_record() const { return leaf_rec; }
};
template <> struct HitOwed<false> {
  uint32_t bits = 0u;
  TDT_DEV void set(bool later_step, bool cube_ok) { bits = later_step ? (cube_ok ? 3u : 1u) : 0u; }
  TDT_DEV bool leaf_site() const { return (bits & 1u) != 0u; }
  TDT_DEV bool new_record() const { return (bits & 2u) != 0u; }
};

// -DTDT_STATS builds (tools/loss_budget.py; never the product library): what the passes of the product kernels carry — how many
// traversal / event passes a wave runs and how many lanes are live in each code region — summed per wave in LDS (one lane writes its
// wave's row) and added to P.stats when the wave ends.  Together with the static instruction counts of the regions
// (tools/isa_regions.py) this splits the SQ counters' VALU wave-instructions and active lanes by region: profiles/r04_loss_budget.json.
enum : int { STAT_TRAV_PASS = 0, STAT_TRAV_LANES, STAT_INSIDE_LANES, STAT_ALIVE_LANES, STAT_EVENT_PASS, STAT_EVENT_LANES, STAT_HIT_PASS, STAT_HIT_LANES,
             STAT_LAMB_PASS, STAT_LAMB_LANES, STAT_METAL_PASS, STAT_METAL_LANES, STAT_DIEL_PASS, STAT_DIEL_LANES, STAT_END_PASS, STAT_END_LANES,
             STAT_PIXEL_END_PASS, STAT_PIXEL_END_LANES, STAT_FETCH_PASS, STAT_FETCH_LANES, STAT_PRIMARY_PASS, STAT_PRIMARY_LANES,
             STAT_NEWRAY_PASS, STAT_NEWRAY_LANES, STAT_GATE_WAIT_LANES, STAT_DRAINED_TRAV_PASS, STAT_DRAINED_EVENT_PASS, STAT_LOOP_PASS, STAT_WAVE_TICKS, STAT_WAVES,
             STAT_T_FIRST, STAT_T_LAST, STAT_COUNT = 32 };
#ifdef TDT_STATS
TDT_DEV void stat_add(uint32_t *row, int i, unsigned long long mask) {      // mask: the lanes (of those executing this) that are live in the region
  const unsigned long long m = __ballot(1);
  if (mask != 0ull && (threadIdx.x & 63) == (uint32_t)__builtin_ctzll(m)) { row[i] += 1u; row[i + 1] += (uint32_t)__popcll(mask); }
}
#define TDT_ST(i, mask) stat_add(s_stat_row, (i), (mask))
#define TDT_ST1(i, v) do { const unsigned long long m_ = __ballot(1); if ((threadIdx.x & 63) == (uint32_t)__builtin_ctzll(m_)) s_stat_row[i] += (uint32_t)(v); } while (0)
#else
#define TDT_ST(i, mask) do {} while (0)
#define TDT_ST1(i, v) do {} while (0)
#endif

// P.accumulate == 0: the whole of main() rc:234-252; 1: only the sample loop, adding to running sums (a uniform run-time
// flag, looked at once per pixel, so the scene-specialised variants serve progressive passes too).
// COUNT: instrumented build that also totals the events defining the algorithmic bytes.
// FORM: how treeLookup's x index is computed — FORM_LITERAL the float formula as written; FORM_POW2 the exact-comparison form
// for cell_count = 2^k (see tree_lookup_pow2); FORM_TABLE the same walk with per-cell thresholds for any other cell_count (the
// reference's own 100000), see build_thresholds_kernel.
template <bool COUNT, int FORM, int DEPTH = 0, bool RESIDENT = false, bool SAFEV = false, bool FULL = false, bool UNIT = false, bool BRICK = false>
__global__ __launch_bounds__(TDT_BLOCK) void trace_kernel(const TraceParams P) {
  constexpr bool POW2 = FORM != FORM_LITERAL;        // the exact forms (integer digits, one-compare box test)
  // round 4's second batch of step slimming (see the uses): the scene-specialised builds; the general kernel, with its nine-level node
  // memo, sits at the register cap and keeps the plain forms
  constexpr bool kSlim2 = DEPTH > 0;
#ifdef TDT_THRESHOLD_EVERY
  constexpr uint32_t kThresholdEvery = kSlim2 ? TDT_THRESHOLD_EVERY : 1u;
#else
  constexpr uint32_t kThresholdEvery = kSlim2 ? 8u : 1u;      // (a power of two; see the threshold update.  The general kernel has no register for the counter)
#endif
  constexpr bool kSharedRand = !BRICK;                // (scatter<>: one Rand(hit.xy) for metal and dielectric lanes; in the brick builds -0.8 % on 4K/256^3 on its own, +0.6 % on 512^3 together with kColdArgs: not there)
  constexpr bool kColdArgs = BRICK;                   // (see material_source's use)
#ifdef TDT_NO_STUCK_CUT
  constexpr bool kStuckCut = false;
#else
  // (rays that stop advancing leave the traversal loop at once: see the step's tail.  The brick builds = trees of depth 6-9 outside the LDS
  // table, where it was found and measured; in every build of depth >= 7 it costs the demo frame and the monument 0.5-1 % and wins nothing
  // on the held-out set)
  constexpr bool kStuckCut = BRICK && !COUNT;
#endif
#ifdef TDT_NO_SHARED_NORM
  constexpr bool kSharedNorm = false;
#else
  constexpr bool kSharedNorm = kSlim2;                // (scatter<>: one normalize() behind the three material branches)
#endif
  __shared__ __attribute__((aligned(16))) uint16_t s_nodes[(BRICK ? kBrickLdsCells : kLdsCells) * 8 + 8];   // + the sentinel slot (BRICK: the host stages no more than fit)
  for (uint32_t i = threadIdx.x * 8u; i < P.lds_nodes; i += (uint32_t)TDT_BLOCK * 8u)      // one cell (8 x u16) per lane and trip
    *reinterpret_cast<uint4 *>(&s_nodes[i]) = *reinterpret_cast<const uint4 *>(&P.packed[i]);
  // sentinel: the rest of a partial last cell and one more cell (RESIDENT: all EMPTY, see tree_lookup_pow2), or the escape code
  if (threadIdx.x < 16 && P.lds_nodes + threadIdx.x < ((P.lds_nodes + 7u) & ~7u) + 8u)
    s_nodes[P.lds_nodes + threadIdx.x] = (uint16_t)((RESIDENT && POW2) ? 0u : kPackedEscape);
  __syncthreads();
  constexpr bool TABLE = FORM == FORM_TABLE;
  static_assert(!TABLE || (SAFEV && !BRICK && !FULL), "per-cell thresholds: the resident walk, or the jump table's bands of a tree outside the LDS table");
  constexpr uint32_t kThrCells = !TABLE ? 0u : (RESIDENT ? kLdsCells : kThrTopCells);
  __shared__ __attribute__((aligned(8))) float2 s_thr[kThrCells + 1];      // FORM_TABLE: x_thresholds of the cells PARENT nodes point at (trees outside the LDS table: of the top cells)
  __shared__ uint32_t s_band;
  if (TABLE) {
    for (uint32_t i = threadIdx.x; i <= kThrCells; i += (uint32_t)TDT_BLOCK)
      s_thr[i] = i < P.thr_cells ? reinterpret_cast<const float2 *>(P.thr)[i] : make_float2(2.0f, 2.0f);
    __syncthreads();
  }
  constexpr int GL = top_grid_levels(RESIDENT, TABLE);                 // levels of the top-level jump table (see Grid<GL>)
  constexpr bool kUseGrid = !FULL && !BRICK && !COUNT && POW2 && SAFEV && DEPTH >= GL;   // see build_top_grid (FULL: the whole-depth table in global memory instead; BRICK: the 32-bit table below)
  __shared__ typename Grid<GL>::Entry s_grid[kUseGrid ? Grid<GL>::kEntries : 1];
  __shared__ int s_grid_ok;
  if (kUseGrid) build_top_grid<GL, TABLE>(s_nodes, P.lds_nodes, DEPTH, s_grid, &s_grid_ok, s_thr, P.thr_cells, P.thr_f0max, &s_band);
  __shared__ __attribute__((aligned(16))) uint32_t s_grid32[BRICK ? (1 << 15) : 4];     // BRICK: build_bricks_kernel's table, copied as it is
  if (BRICK) {
    for (uint32_t i = threadIdx.x * 4u; i < (1u << 15); i += (uint32_t)TDT_BLOCK * 4u)
      *reinterpret_cast<uint4 *>(&s_grid32[i]) = *reinterpret_cast<const uint4 *>(&P.brick_grid[i]);
    __syncthreads();
  }
  NodeSource ns;
  ns.lds = s_nodes; ns.lds_nodes = P.lds_nodes; ns.lds_cells = (P.lds_nodes + 7u) >> 3;
  ns.grid = s_grid; ns.grid_ok = kUseGrid ? (__builtin_amdgcn_readfirstlane(s_grid_ok) != 0) : false;
  ns.grid_band = FULL ? Grid<5>::kBand : (ns.grid_ok ? (TABLE && kUseGrid ? __uint_as_float((uint32_t)__builtin_amdgcn_readfirstlane((int)s_band)) : Grid<GL>::kBand) : 2.0f);
  ns.thr = s_thr; ns.thr_f0max = P.thr_f0max;
  ns.full = P.full_grid;
  ns.grid32 = s_grid32; ns.bricks = P.bricks;
  if (BRICK) {
    // the bricks' base address in a VGPR pair: as a scalar pair it was the one value the compiler spilled to a VGPR lane and read back
    // (2 v_readlane + the hazard wait) in EVERY traversal step — the brick builds run out of SGPRs, not of VGPRs
    uint32_t lo, hi;
    asm volatile("v_mov_b32 %0, %2\n\tv_mov_b32 %1, %3" : "=v"(lo), "=v"(hi) : "s"((uint32_t)(uintptr_t)P.bricks), "s"((uint32_t)((uintptr_t)P.bricks >> 32)));
    ns.bricks = reinterpret_cast<const void *>(((uintptr_t)hi << 32) | (uintptr_t)lo);
  }
  ns.cells = __builtin_amdgcn_make_buffer_rsrc((void *)P.cells, 0, (int)((P.cells_dwords >> 1) << 3), 0x00020000);
  // (kColdArgs) the material tables' descriptors — 16 SGPRs — are built where they are used, from scalar loads of the kernel-argument segment
  // through a pointer the compiler cannot hoist out of the loop, instead of living in SGPRs (or, spilled, in VGPR lanes: one v_readlane per
  // dword in every event pass) across the traversal loop.  Brick builds: 17 / 18 spilled SGPRs -> 6 / 7, 4K/256^3 -0.4 %, 512^3 -0.6 %; the
  // whole-depth build has no spills to lose and pays 1.4 % for the exposed scalar-load latency: it keeps the descriptors in registers
  typedef const __attribute__((address_space(4))) TraceParams *KArg;
  const KArg Pk = (KArg)__builtin_amdgcn_kernarg_segment_ptr();
  MatSource ms_regs;
  if (!kColdArgs) ms_regs = material_source(P);
#ifdef TDT_STATS
  __shared__ uint32_t s_stats[(TDT_BLOCK / 64) * STAT_COUNT];
  uint32_t *const s_stat_row = &s_stats[(threadIdx.x >> 6) * STAT_COUNT];
  if ((threadIdx.x & 63) < STAT_COUNT) s_stat_row[threadIdx.x & 63] = 0u;
  const unsigned long long stat_t0 = __builtin_amdgcn_s_memrealtime();
  bool stat_drained = false;
#endif
  const uint32_t total_slots = (uint32_t)P.owned_tiles * 1024u;
  if (blockIdx.x == 0 && threadIdx.x == 0 && P.queue_next) *P.queue_next = 0u;     // the NEXT launch's queue head (two heads alternate: no memset per launch)

  const float inf = __builtin_inff();
#ifdef TDT_STATS
  uint32_t stat_pix_t0 = 0u;
#endif
  unsigned long long pixel_rt0 = 0ull; bool wave_drained = false; uint32_t lane_S = 0, lane_E = 0, pass_no = 0, pixel_pass0 = 0, evpass_no = 0, pixel_evpass0 = 0;
  const unsigned long long wave_t0 = COUNT ? __builtin_amdgcn_s_memrealtime() : 0ull;
  if (COUNT && (threadIdx.x & 63) == 0) atomicCAS(&P.counters[23], 0ull, wave_t0);   // time base of the pixel log   // 100 MHz, same on every XCD
  Counters cnt = {};
  uint32_t n_pixels = 0;
  // node-memo levels (2 VGPRs each): the brick builds' level walk is the fallback of the few per cent of their steps that have a lane in a
  // band — three levels' worth of memo instead of nine pays for the octree corner in VGPRs and the hit flags in a VGPR there too
  // (4K/256^3 -1.9 %, 1080p/512^3 -3.3 %; one or two levels: -1.0 / -2.9 %)
  constexpr int kMemo = BRICK ? kBrickMemoLevels : kMemoLevels;
  NodeMemo<kMemo> memo;
#pragma unroll
  for (int l = 0; l < kMemo; l++) { memo.key[l] = 0x3FFFFFFFu; memo.val[l] = 0u; }

  int state = ST_FETCH;
  int x = 0, y = 0; size_t pix = 0;
  uint32_t pixel_slot = 0;                          // queue slot (work-group * 1024 + pixel) of the current pixel
  const bool exact_draw = P.plan != nullptr && __builtin_amdgcn_readfirstlane((int)P.plan[0]) != 0;
  uint32_t buf_slot = 0xFFFFFFFFu, buf_next = 0u, buf_count = 0u;   // the wave's batch of queue slots: lane i holds the i-th; dealt / size
  int s = 0, loop_count = 0;
  Ray r = {0.f, 0.f, 0.f, 0.f, 0.f, 1.f};
  float ix = 0.f, iy = 0.f, iz = 0.f;                // 1 / direction (ray-invariant, rc:319)
  float ar = 1.f, ag = 1.f, ab = 1.f;                // accumulative_attenuation rc:267
  float sr = 0.f, sg = 0.f, sb = 0.f;                // color rc:237
  float t_stride = 0.f, t_octree_max = 0.f, inv_pow_depth = 0.5f;
  int it = 0;                                        // OctreeHit's i rc:410
  // what the hit found by the traversal step still owes the event code: bit 0 = the record is the leaf call site's (it > 0), bit 1 = that
  // call site writes a new record (its slab test hit).  As two bools the compiler keeps them as lane masks in scalar register pairs and
  // updates those with a dozen s_and / s_andn2 / s_or per pass; as an integer in a VGPR the pass is 16 scalar instructions shorter — which
  // is worth 0.8 % where registers are to spare (round 3: not in the brick builds; round 4 freed theirs by shortening the node memo)
  HitOwed<DEPTH == 0> owed;      // (as lane masks in the general kernel, which sits at the register cap)
  uint32_t hit_index = 0;
  float leaf_box_x = 0.f, leaf_box_y = 0.f, leaf_box_z = 0.f;
  Carry pc;
  pc.root = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, false}; pc.leaf = pc.root; pc.root_t = 0.f;
  const int s_end = P.spp_begin + P.spp_count;

  // adaptive event threshold (wave-uniform).  Model: a traversal pass costs C_t issue slots, an event pass C_e
  // whatever the number of lanes it serves; with threshold T about T/2 lanes idle through the traversal
  // passes, and rays of n steps arrive at the event code at a rate of (64 - T/2) / n per pass.  Rays served
  // per issue slot peak at  T = 64 / (1/2 + sqrt(r n)),  r = C_t / (2 C_e);  n is measured per wave over a
  // sliding window of kEventWindow rays (in cost order a wave meets the expensive pixels first and the sky
  // last).  r (P.event_k) is fitted: 0.2 while the tree fits one XCD's L2, rising to 0.5 beyond — there the
  // node loads' latency, not issue slots, bounds a traversal pass and event passes come almost free.  (Round 2, after the
  // traversal step had become cheaper: 0.12 ... 0.35.)
  uint32_t w_steps = 0, w_rays = 0, lane_work = 0;
  uint32_t ev_no = 0;
  int threshold = P.event_threshold > 0 ? P.event_threshold : 24;

  // region timers of the instrumented build (s_memtime, wave-uniform): 0 traversal step, 1 gate, 2 hit + scatter, 3 end of path,
  // 4 pixel fetch, 5 primary ray + threshold, 6 new-ray prologue -> counters[24..30]
  unsigned long long tacc[7] = {0, 0, 0, 0, 0, 0, 0}, tlast = COUNT ? __builtin_amdgcn_s_memtime() : 0ull;
#define TDT_TICK(i) do { if (COUNT) { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); tacc[i] += n_ - tlast; tlast = n_; } } while (0)
  // The octree's minimum corner in VGPRs (the scene-specialised builds): a VALU instruction with an SGPR operand issues at half rate on
  // gfx950 (tools/micro/pipe_model.hip: v_add_f32 with an SGPR source 4.2 cycles, with VGPR sources 2.7), and the traversal step reads
  // these three twice each.  (asm: the compiler would fold a plain copy back into the scalar operand.)
  float vmin_x, vmin_y, vmin_z;
  if (DEPTH == 0) { vmin_x = P.min_x; vmin_y = P.min_y; vmin_z = P.min_z; }      // (the general kernel has no register to spare)
  else asm volatile("v_mov_b32 %0, %3\n\tv_mov_b32 %1, %4\n\tv_mov_b32 %2, %5" : "=v"(vmin_x), "=v"(vmin_y), "=v"(vmin_z) : "s"(P.min_x), "s"(P.min_y), "s"(P.min_z));
  // The lanes in ST_TRAVERSE and in an event state as wave masks, carried from pass to pass: a traversal step only moves lanes out of
  // the first set into the second, so the masks after the step are mask algebra on the step's own conditions — no compare of `state`
  // in a pass that runs no event code (each v_cmp is a half-rate instruction, and the pass is bound by those: tools/micro/pipe_model.hip).
  // th_now: the event threshold scaled by the lanes still alive — they only retire in the fetch code, so it is computed after an event
  // pass, not before every gate.
  unsigned long long k_trav = 0ull, k_event = ~0ull;       // (every lane starts in ST_FETCH)
  int th_now = threshold;
  for (;;) {
    if (COUNT) pass_no++;
    TDT_ST1(STAT_LOOP_PASS, 1);
    // ------------------------------------------------------------ one traversal step rc:410-447
    TDT_MARK(traversal);
    {
      // Flat form: every lane evaluates the loop condition and the position (the values of lanes that are not traversing
      // are never used), so the step is ONE exec region instead of three nested ones.
      const bool trav = __builtin_amdgcn_inverse_ballot_w64(k_trav);
      const unsigned long long k_trav0 = k_trav;
      if (COUNT && trav) { lane_S++; cnt.trav_slots += slot64(); cnt.trav_active++; }
      const bool go = trav && (it < P.max_iter) && (t_stride < t_octree_max);      // else: OctreeHit returns false rc:449
      // rc:412 max(1e-4 (inv_pow_depth + 0.1), 1e-6): in the exact forms inv_pow_depth of a traversing lane is 2^-levels > 0, so the first
      // operand is >= 1e-5 and the max returns it (lanes in other states carry other things in that register; their value is never used)
      const float adv = POW2 ? 0.0001f * (inv_pow_depth + 0.1f) : f_max(0.0001f * (inv_pow_depth + 0.1f), 0.000001f);
      const float tt = t_stride + adv;
      const float wx = tt * r.dx + r.ox, wy = tt * r.dy + r.oy, wz = tt * r.dz + r.oz;
      // UNIT: scale and 1 / scale are exactly 1.0f (the reference's own scene, main.rs:457, and every scene here), and x * 1.0f is
      // x for every x — seven multiplications of the step that need not be issued (64^3 -1.6 %, 256^3 -1.2 %)
      const float lx = UNIT ? (wx + -vmin_x) : (wx + -vmin_x) * P.inv_scale, ly = UNIT ? (wy + -vmin_y) : (wy + -vmin_y) * P.inv_scale,
                  lz = UNIT ? (wz + -vmin_z) : (wz + -vmin_z) * P.inv_scale;
      bool in_box;
      if (POW2) {
        // rc:417 "fract(p) - p != vec3(0)": fract(p) - p is zero exactly for p in [0,1) (and -0).  As ONE unsigned compare:
        // the bit pattern of a float in [+0, 1) is below that of 1.0f; negative numbers carry the sign bit, NaN / inf / p >= 1
        // are larger; and p + 0.0f is p for every p except -0, which it turns into +0 (so -0 passes, as it does in the shader).
        // 3 adds + v_max3_u32 + 1 compare instead of 6 compares and 6 mask ANDs.
        // (UNIT builds: the host also checked that no component of the corner is a zero — a sum x + y is -0 only when both terms are, so
        // lx = wx + -min cannot be -0 and the three additions of +0 are not needed)
        constexpr bool kNoNegZero = UNIT && kSlim2;
        const uint32_t ux = __float_as_uint(kNoNegZero ? lx : lx + 0.0f), uy = __float_as_uint(kNoNegZero ? ly : ly + 0.0f), uz = __float_as_uint(kNoNegZero ? lz : lz + 0.0f);
        const uint32_t um = ux > uy ? ux : uy;
        in_box = (um > uz ? um : uz) < 0x3F800000u;
      } else {
        const float ex = f_fract(lx) + -lx, ey = f_fract(ly) + -ly, ez = f_fract(lz) + -lz;
        in_box = !((__builtin_fabsf(ez) + __builtin_fabsf(ey)) != -__builtin_fabsf(ex));
      }
      const bool inside = go && in_box;
      TDT_ST(STAT_TRAV_PASS, __ballot(trav)); TDT_ST1(STAT_INSIDE_LANES, __popcll(__ballot(inside))); TDT_ST1(STAT_ALIVE_LANES, __popcll(__ballot(state != ST_DONE)));
#ifdef TDT_STATS
      TDT_ST1(STAT_DRAINED_TRAV_PASS, stat_drained ? 1 : 0);
#endif
      if (inside) {
        float ugx, ugy, ugz; uint32_t value;
        if (COUNT) cnt.iterations++;
        const float ipd_in = inv_pow_depth;          // (kStuckCut)
        const bool leaf = POW2 ? tree_lookup_pow2<COUNT, kMemo, DEPTH, RESIDENT, SAFEV, FULL, BRICK, TABLE>(P, ns, lx, ly, lz, inv_pow_depth, ugx, ugy, ugz, value, memo, cnt)
                               : tree_lookup<COUNT>(P, ns, lx, ly, lz, inv_pow_depth, ugx, ugy, ugz, value, cnt);
        TDT_MARK(traversal_b);
        if (!kSlim2) lane_work += kCostStep + (127u - (__float_as_uint(inv_pow_depth) >> 23));   // + tree levels visited (inv_pow_depth = 2^-levels)
        const float bx = (UNIT ? ugx : ugx * P.scale) + vmin_x, by = (UNIT ? ugy : ugy * P.scale) + vmin_y, bz = (UNIT ? ugz : ugz * P.scale) + vmin_z;
        const float cs0 = UNIT ? inv_pow_depth : P.scale * inv_pow_depth;
        // leaf: the exact cell (rc:427-428); empty: padded by -1e-5 / +2e-5 (rc:441-442)
        // (x + -0.0f is x, bit for bit, for every x: one select on the pad instead of one per coordinate)
        const float pad = leaf ? -0.0f : -0.00001f;
        // (kSlim2: the cell's corner and what it holds straight into the registers a hit carries to the event code: they mean nothing while a
        // lane traverses, so every lane in the octree writes them — no copies in the leaf branch)
        if (kSlim2) { leaf_box_x = bx + pad; leaf_box_y = by + pad; leaf_box_z = bz + pad; hit_index = value; }
        const float cx = kSlim2 ? leaf_box_x : bx + pad, cy = kSlim2 ? leaf_box_y : by + pad, cz = kSlim2 ? leaf_box_z : bz + pad;
        const float cs = leaf ? cs0 : cs0 + 0.00002f;
        float t_enter, t_exit;
        cube_slabs(r, ix, iy, iz, cx, cy, cz, cs, t_stride, t_octree_max, t_enter, t_exit);
        const bool cube_ok = !(t_exit < t_enter);
        if (leaf) {
          // CubeHit's record (rc:336-354) is deferred to the event code, where the lanes that hit
          // are batched: only a few lanes per step reach a leaf.  The traversal registers are
          // dead from here on, so they carry the cube.
          if (!kSlim2) { leaf_box_x = cx; leaf_box_y = cy; leaf_box_z = cz; hit_index = value; }
          inv_pow_depth = cs; t_stride = t_enter;
          owed.set(it > 0, cube_ok);
          state = ST_HIT;
        } else {
          const float ts_new = cube_ok ? t_exit : t_octree_max;
          if (kStuckCut) {
            // A step that leaves (t_stride, inv_pow_depth) as it found them will find them so again: the loop body is a function of these two
            // and of the ray, so every remaining iteration up to max_iter repeats this one and OctreeHit returns false (rc:449).  It happens:
            // at the finest levels of a 256^3 / 512^3 tree treeLookup's float index arithmetic (rc:376-378) hands back a cell the sample
            // point is not in, whose slab interval along the ray is empty at t_stride — 54 % of all iterations of the 4K/256^3 frame and
            // 86 % of the 1080p/512^3 frame are such repeats (counted with the oracle; the sums with and without them are the same bits).
            // The lane leaves the loop the way the last iteration would: nothing but `false` comes out of it.
            const bool stuck = __float_as_uint(ts_new) == __float_as_uint(t_stride) && __float_as_uint(inv_pow_depth) == __float_as_uint(ipd_in);
            t_stride = stuck ? t_octree_max : ts_new;
          } else t_stride = ts_new;
          it++;
        }
      }
      state = (trav && !inside) ? ST_END : state;       // left the octree / ran out of iterations
      // (ONE compare of `state` per pass: who still traverses; whoever left has an event to be served.  A ballot of the step's own conditions
      //  — inside && !leaf — would need none, but the compiler lowers the ballot of a boolean that is not a compare to v_cndmask + v_cmp)
      k_trav &= __ballot(state == ST_TRAVERSE);
      k_event = ~k_trav & (k_event | k_trav0);
    }

    TDT_TICK(0);
    // ------------------------------------------------------------ path events
    TDT_MARK(gate);
    const unsigned long long m_trav = k_trav, m_event = k_event;
    if (m_trav == 0ull && m_event == 0ull) break;
    w_steps += (uint32_t)__popcll(m_trav);            // lanes that take the next traversal step
    // run the (long, material-divergent) event code only when enough lanes wait for it (a separate
    // threshold for scatter alone was measured: worse at every setting)
    // once the queue has run dry lanes retire (ST_DONE) and only latency is left to win: scale the threshold
    // with the lanes still alive so that the survivors do not wait for company that will never come
    TDT_ST1(STAT_GATE_WAIT_LANES, __popcll(m_event));      // (every pass: lanes parked at the gate or about to be served)
    if ((int)__popcll(m_event) < th_now && m_trav != 0ull) { TDT_TICK(1); continue; }
    TDT_TICK(1);
    TDT_ST(STAT_EVENT_PASS, m_event); TDT_ST(STAT_HIT_PASS, __ballot(state == ST_HIT));
#ifdef TDT_STATS
    TDT_ST1(STAT_DRAINED_EVENT_PASS, stat_drained ? 1 : 0);
#endif

    TDT_MARK(hit_prologue);
    if (COUNT) evpass_no++;
    if (COUNT) { cnt.event_slots += slot64(); cnt.event_active += (state > ST_TRAVERSE); }
    // a pixel's cost for the hand-out order (a schedule: no pixel depends on it): 64 per path event + (kSlim2) kCostRayStep per traversal step
    // of the ray that ends here — `it` counted them (the step that finds a leaf does not advance it: + 1) — added once per ray, in the event
    // pass, instead of three instructions in every traversal step.  (The tree levels a step visited used to be part of it; with one table or
    // brick read per step whatever the depth they no longer say what a step costs.)
    if (kSlim2) lane_work += (state == ST_HIT || state == ST_END) ? kCostEvent + kCostRayStep * ((uint32_t)it + 1u) : ((state > ST_TRAVERSE) ? kCostEvent : 0u);
    else lane_work += (state > ST_TRAVERSE) ? kCostEvent : 0u;
    if (COUNT) lane_E += (state > ST_TRAVERSE) ? 1u : 0u;
    if (state == ST_HIT) {                            // RayColor loop body rc:272-295
      if (COUNT) { cnt.scatter_slots += slot64(); cnt.scatter_active++; }
      MatSource ms;
      if (kColdArgs) {
        KArg pk = Pk; asm volatile("" : "+s"(pk));
        ms.materials = table_rsrc(pk->materials, pk->materials_dwords); ms.albedos = table_rsrc(pk->albedos, pk->albedos_dwords);
        ms.metal = table_rsrc(pk->metal, pk->metal_dwords); ms.dielectric = table_rsrc(pk->dielectric, pk->dielectric_dwords);
      } else ms = ms_regs;
      const MatRef mat = material_fetch(ms, hit_index);
      TDT_ST(STAT_LAMB_PASS, __ballot(mat.type == 0u)); TDT_ST(STAT_METAL_PASS, __ballot(mat.type == 1u)); TDT_ST(STAT_DIEL_PASS, __ballot(mat.type == 2u));
      if (owed.new_record()) { cube_hit_record(r, t_stride, leaf_box_x, leaf_box_y, leaf_box_z, inv_pow_depth, pc.leaf); if (COUNT) cnt.leaf_records++; }
      loop_count += 1;
      const HitTmp &src = owed.leaf_site() ? pc.leaf : pc.root;
      Hit h;
      h.px = src.px; h.py = src.py; h.pz = src.pz; h.nx = src.nx; h.ny = src.ny; h.nz = src.nz; h.ff = src.ff;
      h.index = hit_index;
      Ray nr; float tr, tg, tb;
      const bool scattered = scatter<COUNT, kSharedRand, kSharedNorm>(ms, r, h, mat, nr, tr, tg, tb, cnt);
      TDT_MARK(hit_epilogue);
      if (scattered) {
        ar = ar * tr; ag = ag * tg; ab = ab * tb;
        r = nr;
        // rc:271: the bounce limit ends the path here and now — ST_END is handled further down in this same pass; going
        // through ST_NEWRAY first would make the lane wait for another event pass
        state = loop_count < P.max_bounce ? ST_NEWRAY : ST_END;
      } else {
        state = ST_END;
      }
    }
    TDT_TICK(2);
    TDT_MARK(end_of_path);
    TDT_ST(STAT_END_PASS, __ballot(state == ST_END));
    if (state == ST_END) {                            // rc:297-301, rc:246
      float cr, cg, cb;
      if (loop_count > 0) { cr = ar; cg = ag; cb = ab; }
      else {
        const float yp = r.dy + 1.0f;
        const float w = 1.0f + -(0.5f * yp);
        cr = w + 0.25f * yp; cg = w + 0.35f * yp; cb = 1.0f;
      }
      sr = sr + cr; sg = sg + cg; sb = sb + cb;
      s++;
      if (s < s_end) state = ST_PRIMARY;
      else {
        TDT_MARK(pixel_end);
        TDT_ST(STAT_PIXEL_END_PASS, __ballot(1));
        float4 *dst = reinterpret_cast<float4 *>(P.image) + pix;
        if (P.accumulate) {
          *dst = make_float4(sr, sg, sb, 0.f);
          if (P.carry && !P.carry_final) {           // (the last launch of a two-phase frame: nobody will read the records again)
            float4 *c = reinterpret_cast<float4 *>(P.carry) + pix * 4;
            c[0] = make_float4(pc.root.nx, pc.root.ny, pc.root.nz, pc.root.px);
            c[1] = make_float4(pc.root.py, pc.root.pz, pc.root.ff ? 1.f : 0.f, pc.root_t);
            c[2] = make_float4(pc.leaf.nx, pc.leaf.ny, pc.leaf.nz, pc.leaf.px);
            c[3] = make_float4(pc.leaf.py, pc.leaf.pz, pc.leaf.ff ? 1.f : 0.f, 0.f);
          }
        } else {
          const float n = (float)P.samples_per_pixel;   // rc:249-251
          float4 o;
          o.x = f_min(f_max(__builtin_sqrtf(sr / n), 0.f), 1.f);
          o.y = f_min(f_max(__builtin_sqrtf(sg / n), 0.f), 1.f);
          o.z = f_min(f_max(__builtin_sqrtf(sb / n), 0.f), 1.f);
          o.w = 1.0f;
          *dst = o;
        }
#ifdef TDT_STATS
        if (P.stats && P.pixel_log) {                 // (-DTDT_STATS builds with TDT_PIXEL_LOG set: when each pixel of a PRODUCT launch started and ended, and on which wave)
          uint32_t *L = P.pixel_log + (size_t)pixel_slot * 8;
          L[0] = lane_work; L[3] = stat_pix_t0; L[4] = (uint32_t)__builtin_amdgcn_s_memrealtime(); L[5] = blockIdx.x * 16 + (threadIdx.x >> 6); L[7] = (uint32_t)threshold;
        }
#endif
        if (COUNT && P.pixel_log) {
          uint32_t *L = P.pixel_log + (size_t)pixel_slot * 8;
          L[0] = lane_S; L[1] = lane_E; L[2] = pass_no - pixel_pass0; L[3] = (uint32_t)(pixel_rt0 - P.counters[23]);
          L[4] = (uint32_t)(__builtin_amdgcn_s_memrealtime() - P.counters[23]); L[5] = blockIdx.x * 16 + (threadIdx.x >> 6); L[6] = evpass_no - pixel_evpass0; L[7] = (uint32_t)threshold;
        }
        if (COUNT) {
          const unsigned long long d = (__builtin_amdgcn_s_memrealtime() - pixel_rt0) / 10000ull;   // 0.1 ms bins
          atomicAdd(&P.counters[32 + 16384 + (wave_drained ? 128 : 0) + (d > 127ull ? 127ull : d)], 1ull);
        }
        if (P.slot_cost) P.slot_cost[pixel_slot] = lane_work | 1u;   // a store: nothing to wait for (the sort adds it up)
        state = ST_FETCH;
      }
    }
    TDT_TICK(3);
    TDT_MARK(fetch);
    {                                                 // next pixel from the block queue
      // Slots are drawn from the queue up to 64 at a time — one atomic and one coalesced read of the hand-out order per wave
      // and refill — into buf_slot (lane i holds the i-th slot of the batch) and dealt to the lanes that ask, in lane
      // order, through a wave-uniform cursor.  (A lane-by-lane atomic + dependent order read stalled the whole wave for
      // the round trip every time ONE lane finished a pixel: 9 % of the frame.)
      bool want = state == ST_FETCH;
      unsigned long long m = __ballot(want);
      TDT_ST(STAT_FETCH_PASS, m);
      while (m != 0ull) {
        if (buf_next >= buf_count) {                  // refill: guided self-scheduling — 64 slots while plenty are left,
          // fewer towards the end of the queue (slots parked in one wave's batch are out of reach of idle lanes elsewhere)
          // 64 at a time, unless the frame has pixels so long (order_plan_kernel) that a slot parked in one wave's batch
          // while its other lanes are still busy would start too late: then exactly as many as are asked for
          const uint32_t chunk = exact_draw ? (uint32_t)__popcll(m) : 64u;
          uint32_t base = 0;
          if ((threadIdx.x & 63) == 0) base = atomicAdd(P.queue, chunk);
          base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
          const uint32_t bq = base + (threadIdx.x & 63u);
          buf_slot = ((threadIdx.x & 63u) < chunk && bq < total_slots) ? (P.slot_order ? P.slot_order[bq] : bq) : 0xFFFFFFFFu;
          buf_next = 0u; buf_count = chunk;
          if (COUNT && base + chunk > total_slots) wave_drained = true;
#ifdef TDT_STATS
          if (base + chunk > total_slots && !stat_drained) { stat_drained = true; if (P.stats && (threadIdx.x & 63) == 0) atomicMin(&P.stats[STAT_COUNT], __builtin_amdgcn_s_memrealtime()); }      // [STAT_COUNT]: when the first wave met the end of the queue
#endif
        }
        const uint32_t rank = (uint32_t)__popcll(m & ((1ull << (threadIdx.x & 63)) - 1ull));
        const uint32_t avail = buf_count - buf_next;
        const uint32_t got = (uint32_t)__shfl((int)buf_slot, (int)((buf_next + rank) & 63u), 64);
        const bool served = want && rank < avail;
        buf_next += (uint32_t)__popcll(m) < avail ? (uint32_t)__popcll(m) : avail;
        m = __ballot(want && !served);
        if (served) {
          want = false;
          const uint32_t q = got;
          if (q == 0xFFFFFFFFu) { state = ST_DONE; if (COUNT) atomicMin(&P.counters[22], __builtin_amdgcn_s_memrealtime()); }
          else {
#ifdef TDT_STATS
            stat_pix_t0 = (uint32_t)__builtin_amdgcn_s_memrealtime();
#endif
            if (COUNT) { pixel_rt0 = __builtin_amdgcn_s_memrealtime(); lane_S = 0; lane_E = 0; pixel_pass0 = pass_no; pixel_evpass0 = evpass_no; }
            bool inside;
            pixel_slot = q;
            decode_pixel(P, (int)(pixel_slot >> 10), pixel_slot & 1023u, x, y, pix, inside);
            lane_work = 0u;
            if (inside) {                             // outside the covered image: ask again next time
              n_pixels++;
              sr = 0.f; sg = 0.f; sb = 0.f; s = P.spp_begin;
              pc.root = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, false}; pc.leaf = pc.root; pc.root_t = 0.f;
              if (P.accumulate && P.spp_begin != 0) {    // (a range that starts at sample 0 starts from nothing)
                const float4 acc = *(reinterpret_cast<const float4 *>(P.image) + pix);
                sr = acc.x; sg = acc.y; sb = acc.z;
                if (P.carry) {
                  const float4 *c = reinterpret_cast<const float4 *>(P.carry) + pix * 4;
                  const float4 c0 = c[0], c1 = c[1], c2 = c[2], c3 = c[3];
                  pc.root = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z != 0.f}; pc.root_t = c1.w;
                  pc.leaf = {c2.x, c2.y, c2.z, c2.w, c3.x, c3.y, c3.z != 0.f};
                }
              }
              if (s < s_end) state = ST_PRIMARY;
              else if (!P.accumulate) {               // zero samples: main() still stores sqrt(0/0) clamped
                const float n = (float)P.samples_per_pixel;
                const float v0 = f_min(f_max(__builtin_sqrtf(0.f / n), 0.f), 1.f);
                *(reinterpret_cast<float4 *>(P.image) + pix) = make_float4(v0, v0, v0, 1.0f);
              }
            }
          }
        }
      }
    }
    TDT_TICK(4);
    TDT_MARK(primary);
    TDT_ST(STAT_PRIMARY_PASS, __ballot(state == ST_PRIMARY));
    if (state == ST_PRIMARY) {                        // rc:240-245
      r = primary_ray(P, x, y, s);
      loop_count = 0; ar = 1.f; ag = 1.f; ab = 1.f;
      state = ST_NEWRAY;
    }
    TDT_MARK(threshold);
    w_rays += (uint32_t)__popcll(__ballot(state == ST_NEWRAY));
    // (the estimate moves slowly — a sliding window of kEventWindow rays — and its update is 21 VALU instructions, three of them
    // transcendental, on wave-uniform values: every kThresholdEvery-th event pass; bench frame -1.2 %, 4K/256^3 -1.5 %, 512^3 -1.9 %)
    if (P.event_threshold <= 0 && (ev_no++ & (kThresholdEvery - 1u)) == 0u) {
      if (w_rays > kEventWindow) { w_steps >>= 1; w_rays >>= 1; }   // sliding window: the mix of pixels a wave sees changes over a frame
      // (a schedule, not arithmetic of the image: the raw hardware rcp / sqrt do, 3 instructions instead of ~35)
      const float est = 64.0f * __builtin_amdgcn_rcpf(0.5f + __builtin_amdgcn_sqrtf(P.event_k * (float)w_steps * __builtin_amdgcn_rcpf((float)w_rays + 1.0f)));
      // (the upper clamp in the float domain: an integer min against a kernel argument moves the whole threshold update to the
      // scalar unit through v_readfirstlane, and that VALU -> SALU hand-over stalls every pass: measured 2 %)
      const int th = (int)__builtin_fminf(est, P.event_clamp);
      threshold = w_rays < 256u ? 24 : (th < 2 ? 2 : th);
    }
    TDT_TICK(5);
    TDT_MARK(newray);
    TDT_ST(STAT_NEWRAY_PASS, __ballot(state == ST_NEWRAY));
    if (state == ST_NEWRAY) {                         // while-condition rc:271 + OctreeHit prologue rc:399-408
      if (!(loop_count < P.max_bounce)) state = ST_END;
      else {
        if (COUNT) cnt.octree_hit_calls++;
        q_rcp3(r.dx, r.dy, r.dz, ix, iy, iz);
        const float lx = (P.min_x + -r.ox) * ix, ly = (P.min_y + -r.oy) * iy, lz = (P.min_z + -r.oz) * iz;
        const float ux = ((P.min_x + P.scale) + -r.ox) * ix, uy = ((P.min_y + P.scale) + -r.oy) * iy,
                    uz = ((P.min_z + P.scale) + -r.oz) * iz;
        const float mnx = hw_min(lx, ux), mny = hw_min(ly, uy), mnz = hw_min(lz, uz);
        const float mxx = hw_max(lx, ux), mxy = hw_max(ly, uy), mxz = hw_max(lz, uz);
        const float t_enter = hw_max(hw_max(hw_max(mnx, 0.0003f), mny), mnz);
        const float t_exit = hw_min(hw_min(hw_min(mxx, inf), mxy), mxz);
        t_octree_max = inf;
        if (t_exit >= t_enter) {
          cube_hit_record(r, t_enter, P.min_x, P.min_y, P.min_z, P.scale, pc.root);
          pc.root_t = t_enter;
          t_octree_max = t_exit;
        }
        t_stride = pc.root_t;
        inv_pow_depth = 0.5f;
        it = 0;
        state = ST_TRAVERSE;
      }
    }
    TDT_TICK(6);
    k_trav = __ballot(state == ST_TRAVERSE); k_event = __ballot(state > ST_TRAVERSE);
    { const int n_alive = __popcll(k_trav | k_event); th_now = n_alive == 64 ? threshold : ((threshold * n_alive) >> 6) + 1; }
    TDT_MARK(loop_tail);
  }
  TDT_MARK(after_loop);
#undef TDT_TICK
#ifdef TDT_STATS
  if (P.stats) {
    const unsigned long long stat_t1 = __builtin_amdgcn_s_memrealtime();
    const uint32_t lane = threadIdx.x & 63u;
    if (lane < (uint32_t)STAT_WAVE_TICKS) { const uint32_t v = s_stat_row[lane]; if (v) atomicAdd(&P.stats[lane], (unsigned long long)v); }
    if (lane == 0) { atomicAdd(&P.stats[STAT_WAVE_TICKS], stat_t1 - stat_t0); atomicAdd(&P.stats[STAT_WAVES], 1ull);
                     atomicMin(&P.stats[STAT_T_FIRST], stat_t0); atomicMax(&P.stats[STAT_T_LAST], stat_t1);
                     const uint32_t w = blockIdx.x * (TDT_BLOCK / 64) + (threadIdx.x >> 6);      // per-wave end times (100 MHz ticks) of the last launch
                     if (w < 8192u) P.stats[STAT_COUNT + 1 + w] = stat_t1; }
  }
#endif

  if (COUNT) {
    // wave timeline (diagnostics): [18] earliest start, [19] latest end, [20] sum of wave end times, [21] waves
    if ((threadIdx.x & 63) == 0) {
      const unsigned long long t_end = __builtin_amdgcn_s_memrealtime();
      atomicMin(&P.counters[18], wave_t0); atomicMax(&P.counters[19], t_end);
      atomicAdd(&P.counters[20], t_end - wave_t0); atomicAdd(&P.counters[21], 1ull);
      P.counters[32 + blockIdx.x * (TDT_BLOCK / 64) + (threadIdx.x >> 6)] = t_end;   // per-wave end time (grid <= 256 blocks... see kWaveLog)
    }
    if ((threadIdx.x & 63) == 0) for (int i = 0; i < 7; i++) atomicAdd(&P.counters[24 + i], tacc[i]);
    uint32_t v[18] = {n_pixels, cnt.octree_hit_calls, cnt.iterations, cnt.node_loads,
                      cnt.lambertian, cnt.metal, cnt.dielectric, cnt.unknown,
                      cnt.trav_slots, cnt.trav_active, cnt.level_slots, cnt.level_active, cnt.event_slots, cnt.event_active,
                      cnt.scatter_slots, cnt.scatter_active, cnt.memo_miss, cnt.leaf_records};
    for (int i = 0; i < 18; i++) {
      uint32_t tot = wave_sum(v[i]);
      if ((threadIdx.x & 63) == 0 && tot) atomicAdd(&P.counters[i], (unsigned long long)tot);
    }
  }
}

__global__ __launch_bounds__(256) void resolve_kernel(const TraceParams P) {
  int x, y; size_t pix;
  if (!pixel_of_thread(P, x, y, pix)) return;
  float4 *dst = reinterpret_cast<float4 *>(P.image) + pix;
  float4 a = *dst;
  if (P.carry_final && a.w != 0.f) return;            // (frames whose miss pre-pass ran: launch() sets the flag) running sums carry alpha 0; a pixel the pre-pass finished holds its colour, alpha 1, and is left alone
  const float n = (float)P.total_spp;
  float4 o;
  o.x = f_min(f_max(__builtin_sqrtf(a.x / n), 0.f), 1.f);
  o.y = f_min(f_max(__builtin_sqrtf(a.y / n), 0.f), 1.f);
  o.z = f_min(f_max(__builtin_sqrtf(a.z / n), 0.f), 1.f);
  o.w = 1.0f;
  *dst = o;
}

// Miss pre-pass (cameras OUTSIDE the octree).  A pixel all of whose primary rays miss the root cube — and whose first traversal
// position, origin + 6e-5 d, lies outside it too — never enters the tree: every sample runs PRIMARY -> NEWRAY -> one traversal step
// that leaves at once -> END with the sky colour (rc:297-301), and since no call site ever writes a record, nothing is carried from
// sample to sample.  For a camera that looks at a model from outside that is most of the image, and in the trace kernel those pixels
// cost a lane slot, a queue draw and (in a two-phase frame) 64 B of carry each way.  One thread per pixel evaluates exactly the
// arithmetic those states evaluate — primary ray, reciprocals, root slab test, the in-octree test of the first step, the sky
// polynomial, the sum in sample order, main()'s sqrt / clamp — and, if EVERY sample misses, stores the pixel's final colour and
// marks its queue slot done; the first sample that would enter the tree leaves the pixel to the trace kernel, untouched.  The hand-
// out order of the frame's launches is then filtered (filter_*_kernel), so the trace kernel never sees a finished pixel.  Same bits:
// tests/test_gpu_prepass.py (outside / grazing / far cameras against the oracle, with and without the pre-pass).
__global__ __launch_bounds__(256) void miss_prepass_kernel(const TraceParams P, uint8_t *__restrict__ done) {
  int x, y; size_t pix;
  const bool covered = pixel_of_thread(P, x, y, pix);
  const size_t slot = (size_t)(blockIdx.x >> 2) * 1024u + (size_t)((blockIdx.x & 3) * 256 + threadIdx.x);
  bool all_miss = covered;
  float sr = 0.f, sg = 0.f, sb = 0.f;
  const float inf = __builtin_inff();
  for (int s = 0; s < P.samples_per_pixel && __ballot(all_miss) != 0ull; s++) {
    const Ray r = primary_ray(P, x, y, s);            // rc:240-245 (PRIMARY)
    float ix, iy, iz;
    q_rcp3(r.dx, r.dy, r.dz, ix, iy, iz);             // NEWRAY: OctreeHit's root test rc:399-408
    const float lx = (P.min_x + -r.ox) * ix, ly = (P.min_y + -r.oy) * iy, lz = (P.min_z + -r.oz) * iz;
    const float ux = ((P.min_x + P.scale) + -r.ox) * ix, uy = ((P.min_y + P.scale) + -r.oy) * iy, uz = ((P.min_z + P.scale) + -r.oz) * iz;
    const float mnx = hw_min(lx, ux), mny = hw_min(ly, uy), mnz = hw_min(lz, uz);
    const float mxx = hw_max(lx, ux), mxy = hw_max(ly, uy), mxz = hw_max(lz, uz);
    const float t_enter = hw_max(hw_max(hw_max(mnx, 0.0003f), mny), mnz);
    const float t_exit = hw_min(hw_min(hw_min(mxx, inf), mxy), mxz);
    const bool root_hit = t_exit >= t_enter;
    // the first traversal step of a ray whose root test missed: t_stride = the root call site's t, still 0 for this pixel;
    // inv_pow_depth = 0.5 (rc:412); it leaves through the in-octree test rc:417 (written as the literal form of the test)
    const float adv = f_max(0.0001f * (0.5f + 0.1f), 0.000001f);
    const float tt = 0.f + adv;
    const float wx = tt * r.dx + r.ox, wy = tt * r.dy + r.oy, wz = tt * r.dz + r.oz;
    const float px = (wx + -P.min_x) * P.inv_scale, py = (wy + -P.min_y) * P.inv_scale, pz = (wz + -P.min_z) * P.inv_scale;
    const float ex = f_fract(px) + -px, ey = f_fract(py) + -py, ez = f_fract(pz) + -pz;
    const bool in_box = !((__builtin_fabsf(ez) + __builtin_fabsf(ey)) != -__builtin_fabsf(ex));
    const bool steps_in = (0 < P.max_iter) && (0.f < inf) && in_box;
    if (root_hit || steps_in) all_miss = false;
    const float yp = r.dy + 1.0f;                     // END with loop_count == 0: the sky rc:299-301
    const float w = 1.0f + -(0.5f * yp);
    sr = sr + (w + 0.25f * yp); sg = sg + (w + 0.35f * yp); sb = sb + 1.0f;
  }
  if (covered && all_miss) {                          // main() rc:249-251
    const float n = (float)P.samples_per_pixel;
    float4 o;
    o.x = f_min(f_max(__builtin_sqrtf(sr / n), 0.f), 1.f);
    o.y = f_min(f_max(__builtin_sqrtf(sg / n), 0.f), 1.f);
    o.z = f_min(f_max(__builtin_sqrtf(sb / n), 0.f), 1.f);
    o.w = 1.0f;
    *(reinterpret_cast<float4 *>(P.image) + pix) = o;
  }
  done[slot] = (!covered || all_miss) ? 1u : 0u;      // (slots outside the covered image have nothing to trace either)
}

// The hand-out order without the slots the pre-pass finished: a stable compaction in two passes over chunks of kFilterChunk entries
// (per-chunk counts; then every block sums the counts before its chunk and writes).  order == nullptr: image order (the identity).
constexpr uint32_t kFilterChunk = 4096;
TDT_DEV uint32_t filter_rank(bool live, uint32_t *s_wave, uint32_t &block_total) {      // this thread's rank among the live entries of its round
  const unsigned long long m = __ballot(live);
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  if (lane == 0) s_wave[wave] = (uint32_t)__popcll(m);
  __syncthreads();
  uint32_t before = 0, total = 0;
  for (uint32_t w = 0; w < 16u; w++) { const uint32_t c = s_wave[w]; before += w < wave ? c : 0u; total += c; }
  __syncthreads();
  block_total = total;
  return before + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
}
__global__ __launch_bounds__(1024) void filter_count_kernel(const uint32_t *__restrict__ order, const uint8_t *__restrict__ done, uint32_t n, uint32_t *__restrict__ counts) {
  __shared__ uint32_t s_wave[16];
  uint32_t total = 0;
  for (uint32_t r = 0; r < kFilterChunk / 1024u; r++) {
    const uint32_t i = blockIdx.x * kFilterChunk + r * 1024u + threadIdx.x;
    const bool live = i < n && done[order ? order[i] : i] == 0u;
    uint32_t t;
    (void)filter_rank(live, s_wave, t);
    total += t;
  }
  if (threadIdx.x == 0) counts[blockIdx.x] = total;
}
__global__ __launch_bounds__(1024) void filter_write_kernel(const uint32_t *__restrict__ order, const uint8_t *__restrict__ done, uint32_t n, const uint32_t *__restrict__ counts,
                                                            uint32_t *__restrict__ out) {
  // live entries of the chunks before this one, and of all chunks: where this block writes, and where the padding starts.  The
  // finished slots' places at the end of the list are filled with 0xFFFFFFFF, which the trace kernel's fetch reads as "the queue is
  // empty" (the value it hands to lanes beyond the last slot anyway) — so the kernel needs no count and no code of its own for this
  __shared__ uint32_t s_wave[16], s_sum[2];
  uint32_t part = 0, all = 0;
  for (uint32_t c = threadIdx.x; c < gridDim.x; c += 1024u) { const uint32_t v = counts[c]; all += v; part += c < blockIdx.x ? v : 0u; }
  for (int o = 32; o > 0; o >>= 1) { part += (uint32_t)__shfl_xor((int)part, o, 64); all += (uint32_t)__shfl_xor((int)all, o, 64); }
  if (threadIdx.x < 2) s_sum[threadIdx.x] = 0u;
  __syncthreads();
  if ((threadIdx.x & 63u) == 0) { atomicAdd(&s_sum[0], part); atomicAdd(&s_sum[1], all); }
  __syncthreads();
  uint32_t base = s_sum[0];
  uint32_t pad = s_sum[1] + (blockIdx.x * kFilterChunk - base);      // finished entries of the chunks before this one come first in the padding
  __syncthreads();
  for (uint32_t r = 0; r < kFilterChunk / 1024u; r++) {
    const uint32_t i = blockIdx.x * kFilterChunk + r * 1024u + threadIdx.x;
    const uint32_t slot = i < n ? (order ? order[i] : i) : 0u;
    const bool live = i < n && done[slot] == 0u;
    uint32_t t;
    const uint32_t rank = filter_rank(live, s_wave, t);
    if (live) out[base + rank] = slot;
    else if (i < n) out[pad + (threadIdx.x - rank)] = 0xFFFFFFFFu;      // (entries of one round: valid ones first, so t - rank finished ones precede thread t)
    const uint32_t first = blockIdx.x * kFilterChunk + r * 1024u, valid = first >= n ? 0u : (n - first < 1024u ? n - first : 1024u);
    base += t; pad += valid - t;
  }
}

// Re-encode the first n nodes of the cells payload as one dword each for the LDS table (NodeSource).
__global__ __launch_bounds__(256) void pack_cells_kernel(const uint32_t *__restrict__ cells, uint32_t cells_dwords,
                                                        uint16_t *__restrict__ packed, uint32_t n_nodes) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n_nodes) return;
  uint32_t value = 0, type = 0;
  if (2u * i + 1u < cells_dwords) { value = cells[2u * i]; type = cells[2u * i + 1u]; }   // 8-byte granules, as fetch_node
  const uint32_t code = (type == 0u) ? 0u : (type == 2u ? 2u : 1u);
  packed[i] = (uint16_t)((value <= kPackedMaxValue) ? ((value << 2) | code) : kPackedEscape);
}

// One pass over the whole cells payload: out[0] = largest PARENT value (used as a cell index), out[1] = largest value of any
// node, out[2] = one past the last node that is not all zeros (a host that pre-allocates its cell buffer — the reference does,
// main.rs:339-341 — leaves a tail of zero nodes, which read exactly as nodes past the end of the buffer do: EMPTY, value 0).
// Lets the host pick the specialised lookups (SAFEV / RESIDENT).
__global__ __launch_bounds__(256) void scan_cells_kernel(const uint32_t *__restrict__ cells, uint32_t n_nodes, uint32_t *__restrict__ out) {
  uint32_t mp = 0, ma = 0, live = 0;
  for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n_nodes; i += gridDim.x * 256u) {
    const uint2 n = *reinterpret_cast<const uint2 *>(cells + 2u * (size_t)i);
    ma = n.x > ma ? n.x : ma;
    if (n.y != 0u && n.y != 2u) mp = n.x > mp ? n.x : mp;
    if ((n.x | n.y) != 0u) live = i + 1u;
  }
  for (int o = 32; o > 0; o >>= 1) {
    uint32_t a = (uint32_t)__shfl_xor((int)mp, o, 64), b = (uint32_t)__shfl_xor((int)ma, o, 64), c = (uint32_t)__shfl_xor((int)live, o, 64);
    mp = a > mp ? a : mp; ma = b > ma ? b : ma; live = c > live ? c : live;
  }
  if ((threadIdx.x & 63) == 0) { atomicMax(&out[0], mp); atomicMax(&out[1], ma); atomicMax(&out[2], live); }
}

// FORM_TABLE builds: (F1, F2) of every cell index below n and, in aux[1], the bits of the largest F0 (see x_thresholds in
// trace_device.hpp); aux[0] != 0 when some cell's x index is not of the three-threshold shape (then the literal kernel runs).
// all != nullptr: the three thresholds of every cell, for the exhaustive check.  Once per (cell_count, inv_cell_count, n).
__global__ __launch_bounds__(256) void build_thresholds_kernel(float inv_cell_count, int32_t cell_count, uint32_t n, float2 *__restrict__ thr, uint32_t *__restrict__ aux,
                                                               float4 *__restrict__ all) {
  const uint32_t v = blockIdx.x * 256u + threadIdx.x;
  if (v >= n) return;
  const float4 F = x_thresholds(v, inv_cell_count, cell_count, &aux[0]);
  if (thr) thr[v] = make_float2(F.x, F.y);
  if (all) all[v] = F;
  atomicMax(&aux[1], __float_as_uint(F.z));              // (non-negative floats order as their bit patterns)
}

// Exhaustive check of x_thresholds' claim for one cell_count: for every cell index v < n_cells and EVERY f in [0, 1) the literal
// formula's index equals 2v - 1 + (f >= F0(v)) + (f >= F1(v)) + (f >= F2(v)).  shift != 0 checks the harness: thresholds moved by
// that many ulps must fail.  (v in the grid's y dimension, f bit patterns grid-strided in x.)
__global__ __launch_bounds__(256) void selftest_index_kernel(float inv_cell_count, int32_t cell_count, const float4 *__restrict__ thr, int shift,
                                                             unsigned long long *mismatches) {
  const uint32_t v = blockIdx.y;
  const float two_cc = (float)(int32_t)((uint32_t)cell_count << 1);
  float4 F = thr[v];
  if (shift) { F.x = __uint_as_float(__float_as_uint(F.x) + (uint32_t)shift); F.y = __uint_as_float(__float_as_uint(F.y) + (uint32_t)shift);
               if (F.z != 0.0f) F.z = __uint_as_float(__float_as_uint(F.z) + (uint32_t)shift); }
  unsigned long long bad = 0;
  for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < 0x3F800000u; i += gridDim.x * 256u) {
    const float f = __uint_as_float(i);
    const int32_t lit = x_index_literal(v, f, inv_cell_count, two_cc);
    const int32_t tab = (int32_t)(2u * v) - 1 + (f >= F.z ? 1 : 0) + (f >= F.x ? 1 : 0) + (f >= F.y ? 1 : 0);
    bad += lit != tab ? 1u : 0u;
  }
  for (int o = 32; o > 0; o >>= 1) bad += __shfl_xor((long long)bad, o, 64);
  if ((threadIdx.x & 63) == 0 && bad) atomicAdd(mismatches, bad);
}

// The whole-depth table of FULL builds (see tree_lookup_pow2): entry (x, y, z digits of a finest-level voxel position) = what
// treeLookup's descent with those child digits ends on.  16 bits: (depth - levels) << 2 | code, and for a LEAF its value << 5 (the only
// value a traversal step uses; a PARENT can only be what the last level holds).  *bad is raised when the tree does not fit the
// claim the table rests on (a PARENT of a level that feeds a later x decision at or above grid_v_bound) or a material index
// does not fit 11 bits.
__global__ __launch_bounds__(256) void build_full_grid_kernel(const uint32_t *__restrict__ cells, uint32_t cells_dwords, int depth,
                                                             uint16_t *__restrict__ grid, uint32_t *__restrict__ bad) {
  const uint32_t e = blockIdx.x * 256u + threadIdx.x;
  if (e >= (1u << (3 * depth))) return;
  const uint32_t xg = e >> (2 * depth), yg = (e >> depth) & ((1u << depth) - 1u), zg = e & ((1u << depth) - 1u);
  uint32_t v = 0, code = 1u, m = 0;
  bool ok = true;
  for (int l = 1; l <= depth && code == 1u; l++) {
    const int sh = depth - l;
    const uint32_t idx = ((2u * v + ((xg >> sh) & 1u)) << 2) + (((yg >> sh) & 1u) << 1) + ((zg >> sh) & 1u);
    uint32_t value = 0, type = 0;
    if (2u * idx + 1u < cells_dwords) { value = cells[2u * idx]; type = cells[2u * idx + 1u]; }     // reads past the end are 0 (robust access)
    code = (type == 0u) ? 0u : (type == 2u ? 2u : 1u);
    v = value; m = (uint32_t)l;
    if (code == 1u && l < depth && v >= grid_v_bound(l)) ok = false;
  }
  uint32_t enc = (((uint32_t)depth - m) << 2) | code;  // depth - levels: what the lookup shifts the digits by, and the exponent of the cell size above 2^-depth (a PARENT: 0)
  if (code == 2u) { enc |= v << 5; ok = ok && v < 2048u; }
  grid[e] = (uint16_t)enc;
  if (!ok) atomicOr(bad, 1u);
}

// The bricks of BRICK builds (see tree_lookup_pow2): one block per level-5 position e = x5 << 10 | y5 << 5 | z5.  Every thread walks
// the five levels above it as treeLookup would (digits of e; the bands keep the lanes away from coordinates where that is not
// what the reference does); a position that holds a PARENT gets its 27 x 64 (depth 8) or 81 x 256 (depth 9) entries by walking
// on, once per decision sequence, reading the real nodes (past the end of the buffer: zeros) — the nodes of all brick levels but
// the last through LDS.  grid32[e] = what tree_lookup_pow2 decodes; the first exponent is 31 when the cells of some level
// reachable from here do not share floor(log2(index)), an index is 0 or >= 2^22, or a LEAF value does not fit the entry: waves
// that meet such a position walk.  *bad: a PARENT above level 5 at or beyond grid_v_bound (the table's own claim does not hold:
// no BRICK build for this tree).
__device__ __forceinline__ void brick_node(const uint32_t *__restrict__ cells, uint32_t cells_dwords, uint32_t idx, uint32_t &value, uint32_t &code) {
  idx &= 0x1FFFFFFFu;
  uint32_t type = 0; value = 0;
  if (2u * idx + 1u < cells_dwords) { value = cells[2u * idx]; type = cells[2u * idx + 1u]; }     // reads past the end are 0 (robust access)
  code = (type == 0u) ? 0u : (type == 2u ? 2u : 1u);
}
template <int BL>                                     // levels a brick covers: depth - 5 for depths 6-9; 0: no bricks, the table alone (depth 10)
__global__ __launch_bounds__(256) void build_bricks_kernel(const uint32_t *__restrict__ cells, uint32_t cells_dwords, uint32_t *__restrict__ grid32,
                                                           uint16_t *__restrict__ bricks, uint32_t *__restrict__ bad) {
  constexpr uint32_t kEntries = brick_entries(5 + BL);
  constexpr uint32_t kStored = BL <= 1 ? 1u : (BL == 2 ? 12u : (BL == 3 ? 12u + 144u : 12u + 144u + 1728u));   // nodes of all brick levels but the last: kept in LDS
  const uint32_t e = blockIdx.x, xg = e >> 10, yg = (e >> 5) & 31u, zg = e & 31u, tid = threadIdx.x;
  uint32_t v = 0, code = 1u, m = 0;
  bool ok = true;
  int band = -18;                                     // exponent of this position's band (see tree_lookup_pow2 BRICK): at least 2^-18
  for (int l = 1; l <= 5 && code == 1u; l++) {
    const int sh = 5 - l;
    const int need = brick_band_exp(l, v);            // the decision of this level adds the coordinate to v
    band = need > band ? need : band;
    brick_node(cells, cells_dwords, ((2u * v + ((xg >> sh) & 1u)) << 2) + (((yg >> sh) & 1u) << 1) + ((zg >> sh) & 1u), v, code);
    m = (uint32_t)l;
    if (code == 1u && l < 5 && v >= grid_v_bound(l)) ok = false;          // this v feeds the next level's x decision (bounds: band <= 2^-11)
  }
  if (!ok && tid == 0) atomicOr(bad, 1u);
  const uint32_t k = band >= -11 ? 0u : (uint32_t)(-11 - band);           // band 2^-(11 + k) >= 2^band, k in 0..7
  if (code != 1u) {                                   // EMPTY / LEAF within five levels
    if (code == 2u && v >= (1u << 23) && tid == 0) atomicOr(bad, 1u);
    // (builds with bricks: depth - levels instead of the levels, as in the brick entries — what the lookup shifts the digits by)
    if (tid == 0) grid32[e] = (code == 2u ? v << 6 : 0u) | ((BL == 0 ? m : (uint32_t)(5 + BL) - m) << 2) | code | (k << 29);
    return;
  }
  if (BL == 0) {                                      // no bricks: the entry hands the level-6 cell to the walk
    if (tid == 0) grid32[e] = 1u | (v << 2) | (k << 29);
    if (v >= (1u << 22) && tid == 0) atomicOr(bad, 1u);
    return;
  }
  // PARENT: v is the level-6 cell.  s_v / s_c: the nodes of brick level j = 0.. (level 6 + j): 12 ((a + b) x y x z) below each PARENT
  // of the level above, all levels but the last
  __shared__ uint32_t s_v[kStored], s_c[kStored], s_lo[4], s_hi[4], s_inv;
  if (tid < 4) { s_lo[tid] = 31u; s_hi[tid] = 0u; }
  if (tid == 0) s_inv = 0u;
  __syncthreads();
  auto note = [&](int level_j, uint32_t w) {          // w: a cell index the decision of brick level level_j adds the coordinate to
    if (w == 0u || w >= (1u << 22)) atomicOr(&s_inv, 1u);
    else { const uint32_t ex = 31u - (uint32_t)__builtin_clz(w); atomicMin(&s_lo[level_j], ex); atomicMax(&s_hi[level_j], ex); }
  };
  if (tid == 0) note(0, v);
  uint32_t above = 0u, above_count = 1u, here = 0u;   // where the level above sits in s_v (level -1: the level-5 PARENT itself), where this one goes
  for (int j = 0; j < BL - 1; j++) {
    const uint32_t count = above_count * 12u;
    for (uint32_t i = tid; i < count; i += 256u) {
      const uint32_t up = i / 12u, sub = i % 12u;
      const uint32_t pv = j == 0 ? v : s_v[above + up], pc = j == 0 ? 1u : s_c[above + up];
      uint32_t w = 0, c = 0;
      if (pc == 1u) {
        brick_node(cells, cells_dwords, ((2u * pv + (sub >> 2)) << 2) + (sub & 3u), w, c);
        if (c == 1u) note(j + 1, w);
      }
      s_v[here + i] = w; s_c[here + i] = c;
    }
    __syncthreads();
    above = here; above_count = count; here += count;
  }
  for (uint32_t s = tid; s < kEntries; s += 256u) {
    const uint32_t ci = s >> (2 * BL), yb = (s >> BL) & ((1u << BL) - 1u), zb = s & ((1u << BL) - 1u);
    uint32_t val = v, cd = 1u, mm = 5u, at = 0u, base = 0u, count = 12u, div = kEntries >> (2 * BL);      // div: 3^BL
    for (int j = 0; j < BL && cd == 1u; j++) {        // level 6 + j
      div /= 3u;
      const uint32_t c = (ci / div) % 3u, y = (yb >> (BL - 1 - j)) & 1u, z = (zb >> (BL - 1 - j)) & 1u, sub = (c << 2) | (y << 1) | z;
      if (j < BL - 1) {                               // a stored level
        at = at * 12u + sub;
        val = s_v[base + at]; cd = s_c[base + at];
        base += count; count *= 12u;
      } else {
        brick_node(cells, cells_dwords, ((2u * val + c) << 2) + (y << 1) + z, val, cd);      // (val: the PARENT above, i.e. this level's cell)
      }
      mm = 6u + (uint32_t)j;
    }
    if (cd == 2u && val >= 1024u) atomicOr(&s_inv, 1u);
    bricks[(size_t)e * kEntries + s] = (uint16_t)((cd == 2u ? val << 6 : 0u) | (((uint32_t)(5 + BL) - mm) << 2) | cd);      // depth - levels | code, a LEAF's value above
  }
  __syncthreads();
  if (tid == 0) {
    bool shared = s_inv == 0u;
    uint32_t word = 1u | (k << 29);
    for (int j = 0; j < BL; j++) {
      shared = shared && (s_lo[j] == 31u || s_lo[j] == s_hi[j]);          // (31: no PARENT leads to that level)
      word |= (s_lo[j] == 31u ? 0u : s_lo[j]) << (2 + 5 * j);
    }
    grid32[e] = shared ? word : (1u | (31u << 2) | (k << 29));
  }
}

// Exhaustive check of the short correctly-rounded forms against the IEEE expressions: every one of
// the 2^32 float bit patterns (NaN results compare equal to NaN results).  which: 0 rcp, 1 sqrt, 2 rsq.
__global__ __launch_bounds__(256) void selftest_kernel(int which, unsigned long long *mismatches) {
  unsigned long long bad = 0;
  const uint32_t stride = gridDim.x * 256u;
  uint32_t i = blockIdx.x * 256u + threadIdx.x;
  for (uint32_t k = 0; k < (0xFFFFFFFFu / stride) + 1u; k++, i += stride) {
    if (k > 0 && i < stride) break;                   // wrapped
    const float x = __uint_as_float(i);
    if (which >= 5 && which <= 8) {
      // build_top_grid's claim, for EVERY coordinate c in [0,1) and every cell index below grid_v_bound: outside the
      // bands (|2^L c - rint(2^L c)| > Grid<L>::kBand) the x decision of levels 1..L is the plain binary digit of c and never
      // 2v+2.  5: the 4-level table, 7: the 5-level table; 6 / 8 check the harness: with the band test removed the claim must fail.
      if (!(x >= 0.0f && x < 1.0f)) continue;
      const int L = which >= 7 ? 5 : 4;
      const float band = which >= 7 ? Grid<5>::kBand : Grid<4>::kBand;
      const float tg = x * (float)(1 << L);
      if ((which == 5 || which == 7) && !(__builtin_fabsf(tg - __builtin_rintf(tg)) > band)) continue;
      const uint32_t xg = (uint32_t)tg;
      for (int l = 1; l <= L; l++) {
        const float f = l == 1 ? x : f_fract_nonneg(x * (float)(1 << (l - 1)));
        const uint32_t digit = (xg >> (L - l)) & 1u;
        const uint32_t vmax = l == 1 ? 1u : grid_v_bound(l - 1);          // the cell index this level's decision uses
        for (uint32_t v = 0; v < vmax; v++) {
          const float fv = (float)v, q = (fv + f) - fv;
          const uint32_t qa = q > 0.5f ? 1u : 0u, qb = q == 1.0f ? 1u : 0u;
          bad += (qa != digit || qb != 0u) ? 1u : 0u;
        }
      }
      continue;
    }
    if (which == 9 || which == 10) {
      // cube_normal_fast's claim: wherever cube_normal_fast_ok holds, it returns the bits of the literal sequence.  x is the
      // component on axis k; the two others run over zeros, smaller values, ties and non-finite values; the ray direction
      // over the sign combinations and zero / denormal / inf / NaN components.  10 checks the harness: without the guard the
      // claim must fail (ties, NaNs, the ends of the exponent range).
      const float nan = __uint_as_float(0x7FC00000u), inf = __builtin_inff();
      const float others[8][2] = {{0.f, 0.f}, {-0.f, 0.f}, {0.5f * x, -0.25f * x}, {-0.75f * x, 0.f}, {x, 0.f}, {0.f, -x}, {1e-30f, -1e-30f}, {nan, 0.f}};
      const float dirs[12][3] = {{1.f, 1.f, 1.f}, {-1.f, 1.f, 1.f}, {1.f, -1.f, 1.f}, {1.f, 1.f, -1.f}, {-0.3f, -0.5f, -0.8f}, {0.6f, -0.0f, 0.8f}, {0.f, 0.f, 1.f},
                                 {0.f, -1.f, 0.f}, {1e-40f, -1e-40f, 1.f}, {inf, 1.f, -1.f}, {1.f, nan, 1.f}, {-0.f, 0.f, -0.f}};
      for (int k = 0; k < 3; k++) for (int o = 0; o < 8; o++) for (int d = 0; d < 12; d++) {
        const float u = others[o][0], w = others[o][1];
        const float nx = k == 0 ? x : u, ny = k == 1 ? x : (k == 0 ? u : w), nz = k == 2 ? x : w;
        const Ray r = {0.f, 0.f, 0.f, dirs[d][0], dirs[d][1], dirs[d][2]};
        bool sx, sy, sz;
        const bool ok = cube_normal_fast_ok(nx, ny, nz, sx, sy, sz);
        if (!ok && which == 9) continue;
        HitTmp f, l;
        cube_normal_fast(r, nx, ny, nz, sx, sy, sz, f);
        cube_normal_literal(r, nx, ny, nz, l);
        const bool same = (__float_as_uint(f.nx) == __float_as_uint(l.nx) || (f.nx != f.nx && l.nx != l.nx)) &&
                          (__float_as_uint(f.ny) == __float_as_uint(l.ny) || (f.ny != f.ny && l.ny != l.ny)) &&
                          (__float_as_uint(f.nz) == __float_as_uint(l.nz) || (f.nz != f.nz && l.nz != l.nz)) && f.ff == l.ff;
        bad += same ? 0u : 1u;
      }
      continue;
    }
    if (which == 15 || which == 16) {
      // the per-position bands of the bricks' level-5 table: outside 2^max(-18, 5 - l + floor(log2 v) - 23) around the integers of
      // 32 c the x decision of level l with cell index v is the coordinate's binary digit and never 2v + 2 — every coordinate x
      // every level x every cell index below the level's bound, each with ITS band (mode 7 checks one band for all).  16: the
      // harness — with half the band the claim must fail.
      if (!(x >= 0.0f && x < 1.0f)) continue;
      const float tg = x * 32.0f, dist = __builtin_fabsf(tg - __builtin_rintf(tg));
      const uint32_t xg = (uint32_t)tg;
      for (int l = 1; l <= 5; l++) {
        const float f = l == 1 ? x : f_fract_nonneg(x * (float)(1 << (l - 1)));
        const uint32_t digit = (xg >> (5 - l)) & 1u;
        const uint32_t vmax = l == 1 ? 1u : grid_v_bound(l - 1);
        for (uint32_t v = 0; v < vmax; v++) {
          int be = brick_band_exp(l, v); be = be < -18 ? -18 : be;
          const float band = __builtin_ldexpf(1.0f, which == 15 ? be : be - 1);
          if (!(dist > band)) continue;
          const float fv = (float)v, q = (fv + f) - fv;
          bad += ((q > 0.5f ? 1u : 0u) != digit || q == 1.0f) ? 1u : 0u;
        }
      }
      continue;
    }
    if (which == 13 || which == 14) {
      // the bricks' claim: fl(v + f) - v depends on an integer v < 2^22 only through e = floor(log2 v) — for every f in [0, 1) and
      // the ends, the middle and the neighbours of the ends of every binade.  14: the harness (the next binade's value must differ).
      if (!(x >= 0.0f && x < 1.0f)) continue;
      for (uint32_t e = 0; e < 22u; e++) {
        const float q = brick_q(which == 13 ? e : e + 1u, x);
        const uint32_t lo = 1u << e, vs[5] = {lo, lo + 1u, lo + (lo >> 1) + (lo >> 3), 2u * lo - 2u, 2u * lo - 1u};
        for (int k = 0; k < 5; k++) {
          const uint32_t v = vs[k] < lo ? lo : (vs[k] > 2u * lo - 1u ? 2u * lo - 1u : vs[k]);
          const float fv = (float)v;
          bad += (__float_as_uint((fv + x) - fv) == __float_as_uint(q)) ? 0u : 1u;
        }
      }
      continue;
    }
    if (which == 11 || which == 12) {
      // the two one-parameter divisions done as reciprocal + residual step (div_core): pow_poly's (m - 1) / (m + 1) for every
      // mantissa, and reflectance's (1 - x) / (1 + x) for every x whose operands lie in the window.  12: the harness — without
      // the residual step the quotients must differ somewhere.
      const float m = __uint_as_float((i & 0x007FFFFFu) | 0x3F800000u);
      const float n1 = m - 1.0f, d1 = m + 1.0f, n2 = 1.0f + -x, d2 = 1.0f + x;
      const float s1 = which == 11 ? div_core(n1, d1) : n1 * rcp_core(d1), l1 = n1 / d1;
      bad += (__float_as_uint(s1) == __float_as_uint(l1)) ? 0u : 1u;
      if (exp_in_window(n2) && exp_in_window(d2)) {
        const float s2 = which == 11 ? div_core(n2, d2) : n2 * rcp_core(d2), l2 = n2 / d2;
        bad += (__float_as_uint(s2) == __float_as_uint(l2)) ? 0u : 1u;
      }
      continue;
    }
    float a, b;
    if (which == 0) { a = q_rcp(x); b = 1.0f / x; }
    else if (which == 1) { a = q_sqrt(x); b = __builtin_sqrtf(x); }
    else if (which == 2) { a = q_rsq(x); b = 1.0f / __builtin_sqrtf(x); }
    else if (which == 3) { a = __builtin_amdgcn_rcpf(x); b = 1.0f / x; }      // harness check: the raw hardware seed must NOT pass
    else { b = x - __builtin_floorf(x); a = (x >= 0.0f) ? __builtin_amdgcn_fractf(x) : b; }   // v_fract_f32 vs x - floor(x), x >= 0
    const bool same = (__float_as_uint(a) == __float_as_uint(b)) || (a != a && b != b);
    bad += same ? 0u : 1u;
  }
  for (int o = 32; o > 0; o >>= 1) bad += __shfl_xor((long long)bad, o, 64);
  if ((threadIdx.x & 63) == 0 && bad) atomicAdd(mismatches, bad);
}

// Cost-feedback scheduling.  A lane owns a pixel for all of its samples (they are sequential) and a frame is
// only ~8 pixels per lane, so in image order a frame ends with a long tail (the queue runs dry after 60-85 %
// of the kernel) and its waves mix sky pixels with deep ones.  The trace kernel therefore records the work of
// each of its pixels — tree levels visited + kCostStep per traversal step + kCostEvent per path event, counts
// that depend on the pixel alone, not on which wave ran it (a pixel's wall time does, and ordering by it never
// settles) — and these kernels turn the costs into the pixel hand-out order of the NEXT dispatch, most
// expensive first: the frame ends on the cheapest pixels, and the pixels of a wave are alike, which also lets
// the adaptive event threshold fit them.  Like a renderer's render loop (main.rs:486-601) the assumption is
// that consecutive frames cost about the same per pixel; the first dispatch of a context, or one with a
// different number of work-groups, runs in image order.  Only the schedule changes: every pixel is computed
// exactly as before.  Measured (MI355X, steady state): 1080p/64 spp 64^3 38.1 -> 34.5 ms, 4K/64 spp 256^3
// 437 -> 369 ms, 1080p/64 spp 512^3 288 -> 201 ms.
// Counting sort by a 512-bin logarithmic cost key (16 bins per octave), descending, in two passes over chunks
// of kOrderChunk slots: order_hist_kernel sums per-chunk LDS histograms into hist[512]; order_scatter_kernel
// reserves each chunk's range of every bin with one atomic on cursor[bin] and scatters through LDS counters.
constexpr uint32_t kOrderChunk = 8192, kOrderBits = 4;   // 2^kOrderBits bins per octave of cost (16: 512 bins)
__device__ __forceinline__ uint32_t order_key(uint32_t c, uint32_t g) {   // larger cost -> smaller key; never-run slots last
  const uint32_t last = (32u << g) - 1u;
  if (c == 0) return last;
  const uint32_t e = 31u - (uint32_t)__builtin_clz(c);
  const uint32_t frac = e >= g ? (c >> (e - g)) & ((1u << g) - 1u) : (c << (g - e)) & ((1u << g) - 1u);
  return last - ((e << g) + frac);
}
// smooth != 0: every aligned run of 64 slots — one 8x8 screen tile of a work-group (decode_pixel) — is keyed by the sum
// of its costs and stays together, in image order, in the output (used when the inputs changed since the costs were taken).
__device__ __forceinline__ uint32_t run_cost(uint32_t c) {          // wave-wide sum, saturating
  unsigned long long t = c;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
  return t > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)t;
}
// blend > 0 (per-pixel order from a THIN history: the few probe samples of a two-phase frame): a pixel's estimate is shrunk
// towards the mean of its 8x8 tile — neighbours see the same surfaces, so their 64 x more samples say more about what this
// pixel's remaining samples will cost than its own few do.  What it buys is the END of the frame: a pixel whose probe samples
// happened to be cheap no longer starts last and finishes alone (probe order: the last wave ended 8 % after the first).
__device__ __forceinline__ uint32_t blended_cost(uint32_t a, float blend) {
  // (blend < 0: experiment — shrink towards the tile's MAXIMUM instead of its mean, weight |blend|)
  float tile_stat;
  if (blend < 0.0f) {
    uint32_t m = a;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const uint32_t q = (uint32_t)__shfl_xor((int)m, o, 64); m = q > m ? q : m; }
    tile_stat = (float)m; blend = -blend;
  } else tile_stat = (float)run_cost(a) * (1.0f / 64.0f);
  const float v = (1.0f - blend) * (float)a + blend * tile_stat;
  return a == 0u ? 0u : (uint32_t)(v < 1.0f ? 1.0f : (v > 4.0e9f ? 4.0e9f : v));     // (0 = never run: stays last)
}
__global__ __launch_bounds__(1024) void order_hist_kernel(const uint32_t *__restrict__ cost, uint32_t *__restrict__ acc, uint32_t n, uint32_t *__restrict__ hist, uint32_t g, int smooth, int keep, float blend) {
  __shared__ uint32_t s_bin[512];
  if (threadIdx.x < 512) s_bin[threadIdx.x] = 0;
  __syncthreads();
  const uint32_t lo = blockIdx.x * kOrderChunk, hi = lo + kOrderChunk < n ? lo + kOrderChunk : n;   // n is a multiple of 1024
  for (uint32_t i = lo + threadIdx.x; i < hi; i += 1024u) {
    // acc = the cost estimate the order is built from: this dispatch's costs, added to those of the earlier dispatches
    // that traced the same inputs (keep) — every pass sharpens the estimate — or on their own
    const uint32_t c = cost[i], before = keep ? acc[i] : 0u, a = before + c < before ? 0xFFFFFFFFu : before + c;
    acc[i] = a;
    if (smooth) { const uint32_t k = order_key(run_cost(a), g); if ((threadIdx.x & 63u) == 0) atomicAdd(&s_bin[k], 64u); }
    else atomicAdd(&s_bin[order_key(blend != 0.0f ? blended_cost(a, blend) : a, g)], 1u);
  }
  __syncthreads();
  if (threadIdx.x < 512 && s_bin[threadIdx.x]) atomicAdd(&hist[threadIdx.x], s_bin[threadIdx.x]);
}
__global__ __launch_bounds__(1024) void order_scatter_kernel(uint32_t *__restrict__ cost, uint32_t *__restrict__ acc, uint32_t n, const uint32_t *__restrict__ prefix,
                                                             uint32_t *__restrict__ cursor, uint32_t *__restrict__ order, uint32_t g, int smooth, float blend) {
  __shared__ uint32_t s_bin[512], s_base[512];
  if (threadIdx.x < 512) s_bin[threadIdx.x] = 0;
  __syncthreads();
  const uint32_t lo = blockIdx.x * kOrderChunk, hi = lo + kOrderChunk < n ? lo + kOrderChunk : n;
  for (uint32_t i = lo + threadIdx.x; i < hi; i += 1024u) {
    if (smooth) { const uint32_t k = order_key(run_cost(acc[i]), g); if ((threadIdx.x & 63u) == 0) atomicAdd(&s_bin[k], 64u); }
    else atomicAdd(&s_bin[order_key(blend != 0.0f ? blended_cost(acc[i], blend) : acc[i], g)], 1u);
  }
  __syncthreads();
  if (threadIdx.x < 512) {
    const uint32_t mine = s_bin[threadIdx.x];
    s_base[threadIdx.x] = prefix[threadIdx.x] + (mine ? atomicAdd(&cursor[threadIdx.x], mine) : 0u);
  }
  __syncthreads();
  for (uint32_t i = lo + threadIdx.x; i < hi; i += 1024u) {
    if (smooth) {
      const uint32_t k = order_key(run_cost(acc[i]), g);
      uint32_t pos = 0;
      if ((threadIdx.x & 63u) == 0) pos = atomicAdd(&s_base[k], 64u);
      pos = (uint32_t)__shfl((int)pos, 0, 64);
      order[pos + (threadIdx.x & 63u)] = i;
    } else {
      order[atomicAdd(&s_base[order_key(blend != 0.0f ? blended_cost(acc[i], blend) : acc[i], g)], 1u)] = i;
    }
    cost[i] = 0;
    // tile-sum mode = the inputs changed: these costs served once, as a prior for this dispatch's order; the estimate for
    // what follows (the second phase of this frame) starts over from what this dispatch measures
    if (smooth) acc[i] = 0;
  }
}

// How the next dispatch draws slots from the queue (see the fetch code of trace_kernel): plan[0] = 1 -> exactly as asked,
// 0 -> in batches of 64.  Batches make a wave's 64 pixels alike (4K/256^3: -9 %) and spare most queue round trips — an exact
// draw stalls the wave for an atomic and a dependent read of the hand-out order every time ONE lane finishes a pixel — but a
// slot parked in a batch starts late, which must cost more than it gains once a single pixel is long against the frame.  From
// the cost histogram: f = c_hi * lanes / sum(cost) is the share of the frame a pixel of the 99.9th cost percentile occupies its
// lane for; exact drawing when f > max_share = 1 (such a pixel alone outlasts the average lane's whole frame).  Round 1 had
// drawn the line at 0.25, between the 4K/256^3 and the 1080p/512^3 frames it was fitted on; re-measured in round 2 on the
// three bench frames, batches win or tie on all of them (1080p/64^3, f = 0.3: 24.8 -> 20.9 ms history-free; 1080p/512^3:
// 158 -> 155 ms history-free, 125.7 -> 126.9 replay), and no frame in the repository reaches f = 1.
__global__ __launch_bounds__(512) void order_plan_kernel(const uint32_t *__restrict__ hist, uint32_t g, uint32_t lanes, float max_share,
                                                         uint32_t *__restrict__ plan, int smooth, uint32_t *__restrict__ next_set, uint32_t *__restrict__ prefix) {
  // the counters of the NEXT order pass (it alternates between two sets): zeroed here, one launch instead of a memset per frame
  next_set[threadIdx.x] = 0u; next_set[512 + threadIdx.x] = 0u;
  __shared__ float s_sum[512];
  __shared__ uint32_t s_cnt[512];
  const uint32_t bins = 32u << g, k = threadIdx.x;
  float rep = 0.f; uint32_t cnt = 0;
  if (k < bins - 1u) {                                // the last bin holds the never-run slots (cost 0)
    const uint32_t idx = (bins - 1u) - k, e = idx >> g, frac = idx & ((1u << g) - 1u);
    rep = __builtin_ldexpf(1.0f + (float)frac / (float)(1u << g), (int)e);
    cnt = hist[k];
  }
  s_sum[k] = rep * (float)cnt; s_cnt[k] = cnt;
  __syncthreads();
  // inclusive prefix sums over the bins (counts) and the total cost, 9 doubling steps
  for (uint32_t o = 1; o < 512u; o <<= 1) {
    const uint32_t c = k >= o ? s_cnt[k - o] : 0u;
    const float t = k >= o ? s_sum[k - o] : 0.f;
    __syncthreads();
    s_cnt[k] += c; s_sum[k] += t;
    __syncthreads();
  }
  prefix[k] = s_cnt[k] - cnt;                         // where bin k starts in the output: every block of the scatter reads it instead of summing the bins before k
  if (k == 0) plan[0] = 0u;                           // batches, unless the test below says otherwise (tile-sum order: always batches)
  if (smooth) return;
  __syncthreads();
  const uint32_t n = s_cnt[511];
  const float total = s_sum[511];
  const uint32_t want = n / 1000u + 1u;
  // the first bin (in descending cost order) at which the running count reaches the 99.9th percentile
  if (s_cnt[k] >= want && (k == 0u || s_cnt[k - 1u] < want)) {
    const uint32_t idx = (bins - 1u) - k, e = idx >> g, frac = idx & ((1u << g) - 1u);
    const float c_hi = __builtin_ldexpf(1.0f + (float)frac / (float)(1u << g), (int)e);
    plan[0] = (total > 0.f && c_hi * (float)lanes > max_share * total) ? 1u : 0u;
  }
}

// Presentation: the RGBA8 frame the reference's quad pass (quad.frag:10, main.rs:582-600) leaves in a back
// buffer of the texture's size — clamp to [0,1] (NaN -> 0), x 255, round half to even (pinned on llvmpipe,
// tests/golden/present_*.npz) — so that a frame leaves the GPU as 4 instead of 16 bytes per pixel.  HBM-streaming.
__global__ __launch_bounds__(256) void present_kernel(const float4 *__restrict__ image, uint32_t *__restrict__ out, int w, int h, int top_down) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63);
  const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= w || y >= h) return;
  const float4 c = image[(size_t)y * w + x];
  auto u8 = [](float v) -> uint32_t {
    const float t = !(v > 0.0f) ? 0.0f : (v > 1.0f ? 1.0f : v);
    return (uint32_t)__builtin_rintf(t * 255.0f);
  };
  out[(size_t)(top_down ? h - 1 - y : y) * w + x] = u8(c.x) | (u8(c.y) << 8) | (u8(c.z) << 16) | (u8(c.w) << 24);
}

// De-interleave gathered per-rank tile buffers ([rank][k][32][32] RGBA) into a W x H image.
__global__ __launch_bounds__(256) void assemble_kernel(const float4 *__restrict__ tiles, float4 *__restrict__ image,
                                                       int image_width, int cover_w, int cover_h, int tiles_x,
                                                       int world, int tiles_per_rank) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63);
  const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= cover_w || y >= cover_h) return;
  const int t = (y >> 5) * tiles_x + (x >> 5);
  const int rank = t % world, k = t / world;
  const size_t src = ((size_t)rank * tiles_per_rank + k) * 1024 + (size_t)((y & 31) * 32 + (x & 31));
  image[(size_t)y * image_width + x] = tiles[src];
}

}  // namespace tdt

// ============================================================================ host ABI =====
#include "tdt_internal.hpp"

namespace {
thread_local std::string g_create_err;
}

namespace tdt {
int fail(tdt_ctx *ctx, int code, const std::string &msg) {
  if (ctx) ctx->err = msg; else g_create_err = msg;
  return code;
}
int hip_fail(tdt_ctx *ctx, hipError_t e, const char *what) {
  return fail(ctx, TDT_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
}
}  // namespace tdt
using tdt::fail; using tdt::hip_fail; using tdt::erase_from; using tdt::Cover; using tdt::Tiles; using tdt::cover_of; using tdt::tiles_of;

// ComputeShader::dispatch_compute's group arithmetic (compute_shader.rs:30-32) and what it covers
Cover tdt::cover_of(const tdt_compute *c, int width, int height) {
  Cover k;
  k.groups_x = width / 32 < 1 ? 1 : width / 32;
  k.groups_y = height / 32 < 1 ? 1 : height / 32;
  long cw = (long)k.groups_x * 32, ch = (long)k.groups_y * 32;
  k.cover_w = (int)(cw < c->image_width ? cw : c->image_width);
  k.cover_h = (int)(ch < c->image_height ? ch : c->image_height);
  if (k.cover_w < 0) k.cover_w = 0;
  if (k.cover_h < 0) k.cover_h = 0;
  return k;
}
// 32x32 work-groups of the covered image and the ones this rank owns (t % world == rank)
Tiles tdt::tiles_of(const tdt_compute *c, const Cover &k) {
  Tiles t;
  t.tiles_x = (k.cover_w + 31) / 32; t.tiles_y = (k.cover_h + 31) / 32;
  t.total = t.tiles_x * t.tiles_y;
  t.owned = t.total > c->part_rank ? (t.total - c->part_rank + c->part_world - 1) / c->part_world : 0;
  return t;
}
namespace {
int64_t owned_pixels(const tdt_compute *c, const Cover &k) {
  Tiles t = tiles_of(c, k);
  int64_t n = 0;
  for (int i = 0; i < t.owned; i++) {
    int id = c->part_rank + i * c->part_world;
    int gx = id % t.tiles_x, gy = id / t.tiles_x;
    int w = k.cover_w - gx * 32, h = k.cover_h - gy * 32;
    n += (int64_t)(w > 32 ? 32 : w) * (h > 32 ? 32 : h);
  }
  return n;
}

// camera, octree parameters, buffer identities and versions, partition, covered size of the dispatch about to be traced
void make_sig(const tdt_ctx *ctx, const tdt_compute *c, const Cover &k, const tdt_image *img, CostSig *sig) {
  std::memset(sig, 0, sizeof *sig);                   // padding too: signatures are compared with memcmp
  std::memcpy(sig->cam_i, &c->image_width, sizeof sig->cam_i);
  std::memcpy(sig->cam_f, c->horizontal, sizeof sig->cam_f);
  sig->part[0] = c->part_rank; sig->part[1] = c->part_world;
  std::memcpy(sig->octree_f, ctx->ssbo[TDT_SLOT_OCTREE_FLOATS]->shadow, sizeof sig->octree_f);
  std::memcpy(sig->octree_i, ctx->ssbo[TDT_SLOT_OCTREE_INTS]->shadow, sizeof sig->octree_i);
  sig->cover[0] = k.cover_w; sig->cover[1] = k.cover_h; sig->image[0] = img->w; sig->image[1] = img->h;
  for (int sl = 0; sl < kNumSlots; sl++)
    if (ctx->ssbo[sl]) { sig->slot[sl].buffer = ctx->ssbo[sl]; sig->slot[sl].version = ctx->ssbo[sl]->version; }
}

// The scene-specialised builds of the trace kernel (all SAFEV, COUNT = false), looked up by what the dispatch found out about the
// scene.  One row per instantiation; a build that is not listed does not exist, and launch() falls back to the general kernel.
using TraceFn = void (*)(const TraceParams);
struct TraceVariant { int form, depth; bool resident, full, brick, unit; TraceFn fn; };
#define TDT_V1(FORM, D, R, F, B, U) {tdt::FORM, D, R, F, B, U, tdt::trace_kernel<false, tdt::FORM, D, R, true, F, U, B>}
#define TDT_V(FORM, D, R, F, B) TDT_V1(FORM, D, R, F, B, false), TDT_V1(FORM, D, R, F, B, true)
const TraceVariant kTraceVariants[] = {
  // trees outside the LDS table, depth 6-10: the 32-bit level-5 table with per-position bands (+ bricks for depths 6-9) ...
  TDT_V(FORM_POW2, 6, false, false, true), TDT_V(FORM_POW2, 7, false, false, true), TDT_V(FORM_POW2, 8, false, false, true),
  TDT_V(FORM_POW2, 9, false, false, true), TDT_V(FORM_POW2, 10, false, false, true),
  // ... or, when that table cannot be built for the tree (or is switched off), the 16-bit 5-level table built per block and the memo walk
  TDT_V(FORM_POW2, 6, false, false, false), TDT_V(FORM_POW2, 7, false, false, false), TDT_V(FORM_POW2, 8, false, false, false),
  TDT_V(FORM_POW2, 9, false, false, false), TDT_V(FORM_POW2, 10, false, false, false),
  // small trees inside the LDS table: the whole-depth table
  TDT_V(FORM_POW2, 5, true, true, false), TDT_V(FORM_POW2, 6, true, true, false),
  // trees inside the LDS table: 4-level jump table (depth >= 4) + whole-cell LDS reads
  TDT_V(FORM_POW2, 3, true, false, false), TDT_V(FORM_POW2, 4, true, false, false), TDT_V(FORM_POW2, 5, true, false, false),
  TDT_V(FORM_POW2, 6, true, false, false), TDT_V(FORM_POW2, 7, true, false, false), TDT_V(FORM_POW2, 8, true, false, false),
  TDT_V(FORM_POW2, 9, true, false, false), TDT_V(FORM_POW2, 10, true, false, false),
  // the same for a cell_count that is not a power of two (per-cell thresholds, the band of the jump table computed from them)
  TDT_V(FORM_TABLE, 3, true, false, false), TDT_V(FORM_TABLE, 4, true, false, false), TDT_V(FORM_TABLE, 5, true, false, false),
  TDT_V(FORM_TABLE, 6, true, false, false), TDT_V(FORM_TABLE, 7, true, false, false), TDT_V(FORM_TABLE, 8, true, false, false),
  TDT_V(FORM_TABLE, 9, true, false, false), TDT_V(FORM_TABLE, 10, true, false, false),
  // ... and for such a count with the tree outside the LDS table: exact y / z digits, a 4-level jump table with the band its top cells'
  // thresholds give, node memo; the x index of a walked level by the formula itself
  TDT_V(FORM_TABLE, 6, false, false, false), TDT_V(FORM_TABLE, 7, false, false, false), TDT_V(FORM_TABLE, 8, false, false, false),
  TDT_V(FORM_TABLE, 9, false, false, false), TDT_V(FORM_TABLE, 10, false, false, false),
};
#undef TDT_V
#undef TDT_V1
TraceFn find_variant(int form, int depth, bool resident, bool full, bool brick, bool unit) {
  for (const TraceVariant &v : kTraceVariants)
    if (v.form == form && v.depth == depth && v.resident == resident && v.full == full && v.brick == brick && v.unit == unit) return v.fn;
  return nullptr;
}

int launch(tdt_compute *c, int width, int height, int depth, int mode, int spp_begin, int spp_count, void *carry,
           int total_spp, unsigned long long *counts_out) {
  tdt_ctx *ctx = c->ctx;
  (void)depth;   // raytracer.comp is a 2-D dispatch: groups_z = max(depth / 1, 1) layers all write the same pixels
  static const int required[] = {TDT_SLOT_CELLS, TDT_SLOT_MATERIALS, TDT_SLOT_ALBEDOS, TDT_SLOT_METAL,
                                 TDT_SLOT_DIELECTRIC, TDT_SLOT_OCTREE_FLOATS, TDT_SLOT_OCTREE_INTS};
  for (int s : required)
    if (!ctx->ssbo[s]) return fail(ctx, TDT_ERR_INCOMPLETE, "no buffer bound to shader-storage slot " + std::to_string(s));
  if (!ctx->image0) return fail(ctx, TDT_ERR_INCOMPLETE, "no image bound to unit 0");
  if (ctx->ssbo[TDT_SLOT_OCTREE_FLOATS]->bytes < 28 || ctx->ssbo[TDT_SLOT_OCTREE_INTS]->bytes < 12)
    return fail(ctx, TDT_ERR_INVALID_VALUE, "octree uniform buffers are too small (need 28 / 12 bytes)");
  tdt_image *img = ctx->image0;

  TraceParams P;
  std::memset(&P, 0, sizeof P);
  P.image_width = c->image_width; P.image_height = c->image_height;
  for (int i = 0; i < 3; i++) {
    P.hor[i] = c->horizontal[i]; P.ver[i] = c->vertical[i]; P.llc[i] = c->lower_left_corner[i]; P.org[i] = c->origin[i];
  }
  P.samples_per_pixel = c->samples_per_pixel; P.max_bounce = c->max_bounce;
  float of[7]; int32_t oi[3];
  std::memcpy(of, ctx->ssbo[TDT_SLOT_OCTREE_FLOATS]->shadow, sizeof of);
  std::memcpy(oi, ctx->ssbo[TDT_SLOT_OCTREE_INTS]->shadow, sizeof oi);
  P.min_x = of[0]; P.min_y = of[1]; P.min_z = of[2]; P.scale = of[4]; P.inv_scale = of[5]; P.inv_cell_count = of[6];
  P.max_depth = oi[0]; P.max_iter = oi[1]; P.cell_count = oi[2];
  auto dwords = [](const tdt_buffer *b) { size_t d = b->bytes >> 2; return (uint32_t)(d > 0xFFFFFFFFull ? 0xFFFFFFFFull : d); };
  P.cells = (const uint32_t *)ctx->ssbo[TDT_SLOT_CELLS]->dev; P.cells_dwords = dwords(ctx->ssbo[TDT_SLOT_CELLS]);
  P.materials = (const uint32_t *)ctx->ssbo[TDT_SLOT_MATERIALS]->dev; P.materials_dwords = dwords(ctx->ssbo[TDT_SLOT_MATERIALS]);
  P.albedos = (const uint32_t *)ctx->ssbo[TDT_SLOT_ALBEDOS]->dev; P.albedos_dwords = dwords(ctx->ssbo[TDT_SLOT_ALBEDOS]);
  P.metal = (const uint32_t *)ctx->ssbo[TDT_SLOT_METAL]->dev; P.metal_dwords = dwords(ctx->ssbo[TDT_SLOT_METAL]);
  P.dielectric = (const uint32_t *)ctx->ssbo[TDT_SLOT_DIELECTRIC]->dev; P.dielectric_dwords = dwords(ctx->ssbo[TDT_SLOT_DIELECTRIC]);
  P.image = img->dev; P.carry = (float *)carry; P.carry_final = ctx->carry_final ? 1 : 0;
  if (mode == 2) P.carry_final = ctx->use_done ? 1 : 0;      // resolve: skip the pixels the frame's miss pre-pass finished (only then: tdt_dispatch_resolve on its own resolves every pixel)
  Cover k = cover_of(c, width, height);
  Tiles t = tiles_of(c, k);
  P.cover_w = k.cover_w; P.cover_h = k.cover_h;
  P.tiles_x = t.tiles_x > 0 ? t.tiles_x : 1; P.owned_tiles = t.owned;
  // t / tiles_x by multiplication: with m = floor(2^32 / d) + 1, mulhi(t, m) = floor(t / d) whenever t * d < 2^32
  P.tiles_x_magic = ((unsigned long long)(t.total > 0 ? t.total : 1) * (unsigned long long)P.tiles_x < (1ull << 32) && P.tiles_x > 1)
                        ? (uint32_t)((1ull << 32) / (unsigned long long)P.tiles_x) + 1u : 0u;
  P.part_rank = c->part_rank; P.part_world = c->part_world;
  if (img->w == c->image_width && img->h == c->image_height) P.compact = 0;
  else if (img->w == 32 && img->h >= 32 * t.owned && (img->h % 32) == 0) P.compact = 1;   // tile buffer [k][32][32]
  else return fail(ctx, TDT_ERR_INVALID_OPERATION,
                   "bound image is neither camera.image_width x image_height nor a 32 x 32*tiles tile buffer");
  P.spp_begin = spp_begin; P.spp_count = spp_count; P.mode = mode; P.total_spp = total_spp;
  P.event_threshold = ctx->event_threshold;   // 0: adaptive (see trace_kernel)
  P.event_clamp = (float)ctx->event_clamp;
  {  // r of the adaptive event threshold (see trace_kernel): 0.10 up to 1.4 MiB of cells, 0.35 from 5 MiB on (refitted twice in round 2:
     // the traversal step lost a fifth of its instructions, which moves the optimum towards fewer, fuller event passes)
    const float mib = (float)ctx->ssbo[TDT_SLOT_CELLS]->bytes / 1048576.0f;
    const float r = 0.07f * mib;
    P.event_k = ctx->event_k > 0.0f ? ctx->event_k : (r < 0.10f ? 0.10f : (r > 0.35f ? 0.35f : r));
    { static const char *ks = getenv("TDT_EVENT_K_SCALE"); if (ks && atof(ks) > 0.0) P.event_k *= (float)atof(ks); }      // (diagnostics: refitting r)
  }

#ifdef TDT_STATS
  P.stats = (mode == 0 || mode == 1) && !counts_out && !(ctx->probe_launch && getenv("TDT_STATS_SKIP_PROBE")) ? ctx->stats : nullptr;   // (SKIP_PROBE: the main launch of a two-phase frame alone)
#endif
  TDT_HIP(ctx, hipSetDevice(ctx->device));
  if (ctx->ssbo[TDT_SLOT_CELLS]->bytes > 0xFFFFFFF8ull)
    return fail(ctx, TDT_ERR_INVALID_VALUE, "cells buffer larger than 4 GiB is not addressable by the shader's 32-bit offsets");
  if (counts_out) {
    if (!ctx->counters) TDT_HIP(ctx, hipMalloc((void **)&ctx->counters, (32 + 16384 + 256) * sizeof(unsigned long long)));
    TDT_HIP(ctx, hipMemsetAsync(ctx->counters, 0, 32 * sizeof(unsigned long long), ctx->stream));
    TDT_HIP(ctx, hipMemsetAsync(ctx->counters + 18, 0xFF, sizeof(unsigned long long), ctx->stream));   // running minima
    TDT_HIP(ctx, hipMemsetAsync(ctx->counters + 22, 0xFF, sizeof(unsigned long long), ctx->stream));
    TDT_HIP(ctx, hipMemsetAsync(ctx->counters + 32 + 16384, 0, 256 * sizeof(unsigned long long), ctx->stream));
    P.counters = ctx->counters;
    if (getenv("TDT_PIXEL_LOG")) {
      const size_t need = (size_t)t.owned * 1024 * 8;
      if (ctx->pixel_log_u32 < need) { if (ctx->pixel_log) (void)hipFree(ctx->pixel_log); TDT_HIP(ctx, hipMalloc((void **)&ctx->pixel_log, need * 4)); ctx->pixel_log_u32 = need; }
      TDT_HIP(ctx, hipMemsetAsync(ctx->pixel_log, 0, need * 4, ctx->stream));
      P.pixel_log = ctx->pixel_log;
    }
  }
#ifdef TDT_STATS
  if (P.stats && !counts_out && getenv("TDT_PIXEL_LOG")) {
    const size_t need = (size_t)t.owned * 1024 * 8;
    if (ctx->pixel_log_u32 < need) { if (ctx->pixel_log) (void)hipFree(ctx->pixel_log); TDT_HIP(ctx, hipMalloc((void **)&ctx->pixel_log, need * 4)); ctx->pixel_log_u32 = need; }
    TDT_HIP(ctx, hipMemsetAsync(ctx->pixel_log, 0, need * 4, ctx->stream));
    P.pixel_log = ctx->pixel_log;
  }
#endif
  const uint32_t buf_nodes = P.cells_dwords >> 1;
  if (mode == 3) {                                    // the miss pre-pass of a frame (see miss_prepass_kernel): done flags for every queue slot
    if (t.owned <= 0) return TDT_OK;
    if (ctx->done_capacity < (uint32_t)t.owned) {
      if (ctx->slot_done) (void)hipFree(ctx->slot_done);
      if (ctx->slot_live) (void)hipFree(ctx->slot_live);
      if (ctx->filter_counts) (void)hipFree(ctx->filter_counts);
      ctx->slot_done = nullptr; ctx->slot_live = nullptr; ctx->filter_counts = nullptr; ctx->done_capacity = 0;
      const size_t n = (size_t)t.owned * 1024;
      TDT_HIP(ctx, hipMalloc((void **)&ctx->slot_done, n));
      TDT_HIP(ctx, hipMalloc((void **)&ctx->slot_live, n * sizeof(uint32_t)));
      TDT_HIP(ctx, hipMalloc((void **)&ctx->filter_counts, ((n + tdt::kFilterChunk - 1) / tdt::kFilterChunk + 1) * sizeof(uint32_t)));
      ctx->done_capacity = (uint32_t)t.owned;
    }
    hipLaunchKernelGGL(tdt::miss_prepass_kernel, dim3((unsigned)t.owned * 4u), dim3(256), 0, ctx->stream, P, ctx->slot_done);
    TDT_HIP(ctx, hipGetLastError());
    return TDT_OK;
  }
  if (t.owned > 0 && mode != 2) {
    // LDS-table image of the bound cells buffer (rebuilt only when the buffer or its contents changed)
    const tdt_buffer *cb = ctx->ssbo[TDT_SLOT_CELLS];
    if (!ctx->packed) TDT_HIP(ctx, hipMalloc((void **)&ctx->packed, (size_t)tdt::kLdsCells * 8 * sizeof(uint16_t)));
    if (!ctx->queue) {
      TDT_HIP(ctx, hipMalloc((void **)&ctx->queue, 2 * sizeof(unsigned int)));
      const hipError_t e = hipMemsetAsync(ctx->queue, 0, 2 * sizeof(unsigned int), ctx->stream);
      if (e != hipSuccess) { (void)hipFree(ctx->queue); ctx->queue = nullptr; return hip_fail(ctx, e, "pixel-queue heads"); }      // (never a queue with undefined heads)
      ctx->queue_parity = 0;
    }
    P.lds_nodes = buf_nodes < tdt::kLdsCells * 8u ? (buf_nodes & ~7u) : tdt::kLdsCells * 8u;
    if (ctx->packed_of != cb || ctx->packed_version != cb->version) {
      if (!ctx->scan) TDT_HIP(ctx, hipMalloc((void **)&ctx->scan, 4 * sizeof(uint32_t)));
      TDT_HIP(ctx, hipMemsetAsync(ctx->scan, 0, 4 * sizeof(uint32_t), ctx->stream));
      if (P.lds_nodes)
        hipLaunchKernelGGL(tdt::pack_cells_kernel, dim3((P.lds_nodes + 255) / 256), dim3(256), 0, ctx->stream,
                           P.cells, P.cells_dwords, ctx->packed, P.lds_nodes);
      if (buf_nodes) {
        unsigned nb = (buf_nodes + 255) / 256; if (nb > 4096) nb = 4096;
        hipLaunchKernelGGL(tdt::scan_cells_kernel, dim3(nb), dim3(256), 0, ctx->stream, P.cells, buf_nodes, ctx->scan);
      }
      TDT_HIP(ctx, hipGetLastError());
      uint32_t res[3] = {0, 0, 0};
      TDT_HIP(ctx, hipMemcpyAsync(res, ctx->scan, sizeof res, hipMemcpyDeviceToHost, ctx->stream));
      TDT_HIP(ctx, hipStreamSynchronize(ctx->stream));    // once per cells buffer (version), not per frame
      ctx->max_parent_value = res[0]; ctx->max_any_value = res[1]; ctx->live_nodes = res[2];
      ctx->packed_of = cb; ctx->packed_version = cb->version;
    }
    P.packed = ctx->packed; P.queue = ctx->queue + ctx->queue_parity; P.queue_next = ctx->queue + (ctx->queue_parity ^ 1u);      // (the parity flips once the launch is in the stream)
    // cost-feedback hand-out order (see order_scatter_kernel); TDT_NO_COST_ORDER=1: image order
    if (!ctx->no_cost_order) {
      if (ctx->tile_capacity < (uint32_t)t.owned) {
        if (ctx->slot_cost) (void)hipFree(ctx->slot_cost);
        if (ctx->slot_order) (void)hipFree(ctx->slot_order);
        if (ctx->slot_acc) (void)hipFree(ctx->slot_acc);
        ctx->slot_cost = ctx->slot_order = ctx->slot_acc = nullptr; ctx->tile_capacity = 0; ctx->cost_tiles = 0; ctx->order_exact = false;
        TDT_HIP(ctx, hipMalloc((void **)&ctx->slot_cost, (size_t)t.owned * 1024 * sizeof(uint32_t)));
        TDT_HIP(ctx, hipMalloc((void **)&ctx->slot_order, (size_t)t.owned * 1024 * sizeof(uint32_t)));
        TDT_HIP(ctx, hipMalloc((void **)&ctx->slot_acc, (size_t)t.owned * 1024 * sizeof(uint32_t)));
        if (!ctx->order_hist) { TDT_HIP(ctx, hipMalloc((void **)&ctx->order_hist, (2048 + 4 + 512) * sizeof(uint32_t))); TDT_HIP(ctx, hipMemsetAsync(ctx->order_hist, 0, (2048 + 4 + 512) * sizeof(uint32_t), ctx->stream)); ctx->order_parity = 0; }
        ctx->tile_capacity = (uint32_t)t.owned;
      }
      // what this dispatch traces: if it equals what the recorded costs were measured on (a still camera: progressive
      // passes, repeated frames) every pixel will cost exactly what it did, and pixels are sorted one by one; otherwise
      // (the camera moved, the scene was edited) only the low-frequency part of the cost image is still true, and 8x8
      // tiles are sorted by their summed cost — measured on a 60 fps walk, exact-order-of-a-stale-frame is no better
      // than image order (256^3: 3 % worse), the tile form keeps about half of the gain
      CostSig sig;
      make_sig(ctx, c, k, img, &sig);
      P.slot_cost = ctx->slot_cost;
      // The same launch again (same inputs, same sample range) as the one whose costs the current order was built from, those costs
      // themselves recorded by such a launch: every pixel costs what it did, the sort would reproduce the order it produced last time
      // (the sums only double) — so a still camera's frames, from the third on, skip the three sort kernels and the cost stores
      // (1280x720 / 4 spp: 0.46 -> 0.41 ms; the order, a schedule, changes no pixel)
      const bool same_launch = ctx->cost_tiles == (uint32_t)t.owned && std::memcmp(&sig, &ctx->cost_sig, sizeof sig) == 0 &&
                               ctx->cost_range[0] == spp_begin && ctx->cost_range[1] == spp_count && ctx->force_smooth < 0;
      if (same_launch && ctx->order_exact && !ctx->no_order_reuse) {
        P.slot_order = ctx->slot_order; P.plan = ctx->order_hist + 2048; P.slot_cost = nullptr;
      } else
      if (ctx->cost_tiles == (uint32_t)t.owned) {
        if (ctx->order_exact) {
          // leaving the reuse state (the camera moved after standing still, another sample range): the launches that reused the order
          // recorded no costs and the last sort zeroed cost[], but acc[] still holds the sums that order was sorted from — they are the
          // prior for this launch (the tile sums of a moved camera's frame; a sort from all-zero costs would be image order)
          TDT_HIP(ctx, hipMemcpyAsync(ctx->slot_cost, ctx->slot_acc, (size_t)t.owned * 1024 * sizeof(uint32_t), hipMemcpyDeviceToDevice, ctx->stream));
          ctx->order_exact = false;
        }
        const int smooth = ctx->force_smooth >= 0 ? ctx->force_smooth : (std::memcmp(&sig, &ctx->cost_sig, sizeof sig) != 0 ? 1 : 0);
        // same inputs again (progressive passes, repeated frames): keep adding to the costs — every pass sharpens the
        // estimate of what a pixel costs; otherwise start over
        const bool keep_costs = !smooth && !ctx->no_cost_accum && ctx->cost_dispatches < 256;   // (restart before the sums can saturate)
        ctx->cost_dispatches = keep_costs ? ctx->cost_dispatches + 1 : 0;
        // samples per pixel behind the estimate this order is built from; a thin one (the probe of a two-phase frame) is
        // shrunk towards the 8x8-tile mean (blended_cost)
        ctx->acc_samples = (keep_costs ? ctx->acc_samples : 0u) + ctx->last_launch_samples;
        const float blend = (!smooth && ctx->acc_samples < 16u) ? ctx->order_blend : 0.0f;
        if (smooth) ctx->acc_samples = 0;               // (tile-sum mode drops the sums after use: order_scatter_kernel)
        const uint32_t n_slots = (uint32_t)t.owned * 1024u, n_chunks = (n_slots + tdt::kOrderChunk - 1) / tdt::kOrderChunk;
        const uint32_t og = tdt::kOrderBits;
        // two sets of sort counters alternate: the plan kernel of this pass zeroes the set of the next one
        uint32_t *hist = ctx->order_hist + 1024u * ctx->order_parity, *hist_next = ctx->order_hist + 1024u * (ctx->order_parity ^ 1u), *plan = ctx->order_hist + 2048;
        hipLaunchKernelGGL(tdt::order_hist_kernel, dim3(n_chunks), dim3(1024), 0, ctx->stream, ctx->slot_cost, ctx->slot_acc, n_slots, hist, og, smooth, keep_costs ? 1 : 0, blend);
        {
          const float max_share = ctx->max_share;
          const uint32_t lanes = (uint32_t)ctx->num_cus * TDT_BLOCKS_PER_CU * TDT_BLOCK;
          hipLaunchKernelGGL(tdt::order_plan_kernel, dim3(1), dim3(512), 0, ctx->stream, hist, og, lanes, max_share, plan, smooth, hist_next, plan + 4);
          P.plan = plan;
        }
        hipLaunchKernelGGL(tdt::order_scatter_kernel, dim3(n_chunks), dim3(1024), 0, ctx->stream, ctx->slot_cost, ctx->slot_acc, n_slots,
                           plan + 4, hist + 512, ctx->slot_order, og, smooth, blend);
        {
          const hipError_t e = hipGetLastError();
          if (e != hipSuccess) {
            // a launch of the sequence failed: the histogram may be half filled and the other counter set half zeroed — clear both and
            // drop the history, so that the next frame starts from image order instead of scattering through a dirty prefix
            (void)hipMemsetAsync(ctx->order_hist, 0, (2048 + 4 + 512) * sizeof(uint32_t), ctx->stream);
            ctx->cost_tiles = 0; ctx->order_exact = false; ctx->order_parity = 0;
            return hip_fail(ctx, e, "cost-order sort");
          }
        }
        ctx->order_parity ^= 1u;                        // (only now: the plan kernel of this pass zeroed the other set)
        P.slot_order = ctx->slot_order;
        ctx->order_exact = same_launch && !smooth && blend == 0.0f;      // built from what this very launch cost last time
      } else {                                        // no usable history: image order, fresh cost array
        TDT_HIP(ctx, hipMemsetAsync(ctx->slot_cost, 0, (size_t)t.owned * 1024 * sizeof(uint32_t), ctx->stream));
        TDT_HIP(ctx, hipMemsetAsync(ctx->slot_acc, 0, (size_t)t.owned * 1024 * sizeof(uint32_t), ctx->stream));
        ctx->cost_dispatches = 0; ctx->acc_samples = 0; ctx->order_exact = false;
      }
      if (P.slot_cost) {                              // the kernel launched below records this dispatch's costs
        ctx->last_launch_samples = (uint32_t)(spp_count > 0 ? spp_count : 0);
        ctx->cost_sig = sig; ctx->cost_range[0] = spp_begin; ctx->cost_range[1] = spp_count;
        ctx->cost_tiles = (uint32_t)t.owned;
      }
    }
  }
  if (t.owned > 0 && mode != 2 && ctx->use_done && !counts_out) {
    // the frame's pre-pass finished some pixels: hand out the others only (the order just built, or image order, minus the done slots)
    const uint32_t n_slots = (uint32_t)t.owned * 1024u, n_chunks = (n_slots + tdt::kFilterChunk - 1) / tdt::kFilterChunk;
    hipLaunchKernelGGL(tdt::filter_count_kernel, dim3(n_chunks), dim3(1024), 0, ctx->stream, P.slot_order, ctx->slot_done, n_slots, ctx->filter_counts);
    hipLaunchKernelGGL(tdt::filter_write_kernel, dim3(n_chunks), dim3(1024), 0, ctx->stream, P.slot_order, ctx->slot_done, n_slots, ctx->filter_counts, ctx->slot_live);
    TDT_HIP(ctx, hipGetLastError());
    P.slot_order = ctx->slot_live;
  }
  if (t.owned > 0) {
    // trace: one persistent block per CU (fewer when there is less work than lanes); resolve: a thread per pixel
    unsigned nblk = (unsigned)(((long)t.owned * 1024 + 1023) / 1024);
    if (nblk > (unsigned)ctx->num_cus * TDT_BLOCKS_PER_CU) nblk = (unsigned)ctx->num_cus * TDT_BLOCKS_PER_CU;
    dim3 grid(nblk, 1, 1), block(TDT_BLOCK, 1, 1), grid4((unsigned)t.owned * 4u, 1, 1), block4(256, 1, 1);
    // the exact-comparison form of treeLookup needs cell_count = 2^k <= 2^22 and inv_cell_count = 2^-k
    // bit-for-bit (true for every scene Octree::init_global_buffers builds from such a count,
    // octree.rs:49); any other count (the reference's own 100000, main.rs:459) takes the per-cell threshold form when the tree
    // sits in the LDS table (FORM_TABLE, below), else the literal float form
    const uint32_t cc = (uint32_t)P.cell_count;
    const bool depth_ok = P.max_depth >= 0 && P.max_depth <= 30;
    const bool pow2 = !ctx->force_generic && P.cell_count > 0 && (cc & (cc - 1)) == 0 && cc <= (1u << 22) &&
                      P.inv_cell_count == 1.0f / (float)cc && depth_ok;
    // scene-property specialisations of the hot kernel (see tree_lookup_pow2); every variant is bit-identical
    const bool safev = ctx->max_parent_value < (1u << 22);
    // resident: every node that is not all zeros sits in the LDS table (a pre-allocated buffer's tail of zero nodes reads as the
    // table's all-EMPTY sentinel cell does, and as nodes past the end of the buffer do) and fits its 16-bit entries
    const bool resident = buf_nodes > 0 && ctx->live_nodes <= P.lds_nodes && ctx->max_any_value <= tdt::kPackedMaxValue;
    // ... and then only the live cells are staged: the reference's demo scene is 19 cells in a buffer of 6259 (every block would copy
    // 82 KB of zeros at launch), and whatever lies past them reads as the sentinel cell
    if (resident && !ctx->no_specialise) P.lds_nodes = (ctx->live_nodes + 7u) & ~7u;
    // FORM_TABLE: thresholds for every cell of the LDS table (trees outside it: for the top cells, whose decisions the jump table's
    // band is computed from — their walk evaluates the formula), built once per (cell_count, inv_cell_count, cells)
    bool table_form = false;
    if (mode != 2 && !counts_out && !pow2 && !ctx->force_generic && !ctx->no_specialise && !ctx->no_table_form && depth_ok && P.cell_count > 0 && P.lds_nodes > 0) {
      const uint32_t lds_cells = (P.lds_nodes + 7u) >> 3;
      const uint32_t n_thr = resident ? lds_cells : (lds_cells < tdt::kThrTopCells ? lds_cells : tdt::kThrTopCells);
      uint32_t ic_bits; std::memcpy(&ic_bits, &P.inv_cell_count, 4);
      if (ctx->thr_cc != P.cell_count || ctx->thr_ic_bits != ic_bits || ctx->thr_n != n_thr || !ctx->thr) {
        if (!ctx->thr) TDT_HIP(ctx, hipMalloc((void **)&ctx->thr, ((size_t)tdt::kLdsCells + 1) * 2 * sizeof(float)));
        uint32_t *aux = reinterpret_cast<uint32_t *>(ctx->thr + (size_t)tdt::kLdsCells * 2);      // {shape flag, bits of F0max}
        TDT_HIP(ctx, hipMemsetAsync(aux, 0, 2 * sizeof(uint32_t), ctx->stream));
        hipLaunchKernelGGL(tdt::build_thresholds_kernel, dim3((n_thr + 255u) / 256u), dim3(256), 0, ctx->stream, P.inv_cell_count, P.cell_count, n_thr,
                           reinterpret_cast<float2 *>(ctx->thr), aux, (float4 *)nullptr);
        TDT_HIP(ctx, hipGetLastError());
        uint32_t res[2] = {1, 0};
        TDT_HIP(ctx, hipMemcpyAsync(res, aux, sizeof res, hipMemcpyDeviceToHost, ctx->stream));
        TDT_HIP(ctx, hipStreamSynchronize(ctx->stream));    // once per octree-uniform change, not per frame
        ctx->thr_cc = P.cell_count; ctx->thr_ic_bits = ic_bits; ctx->thr_n = n_thr; ctx->thr_ok = res[0] == 0;
        std::memcpy(&ctx->thr_f0max, &res[1], 4);
      }
      table_form = ctx->thr_ok && (!resident || ctx->max_parent_value < n_thr);      // resident: every cell index an x decision can meet has its thresholds
      P.thr = ctx->thr; P.thr_cells = n_thr; P.thr_f0max = ctx->thr_f0max;
    }
    bool launched = false;
    P.accumulate = mode == 1 ? 1 : 0;
    // small resident trees: the whole-depth lookup table (tree_lookup_pow2 FULL), built once per cells buffer
    bool full = false;
    if (mode != 2 && !counts_out && pow2 && safev && resident && !ctx->no_specialise && !ctx->no_full && (P.max_depth == 5 || P.max_depth == 6)) {
      const tdt_buffer *cb = ctx->ssbo[TDT_SLOT_CELLS];
      if (ctx->full_of != cb || ctx->full_version != cb->version || ctx->full_depth != P.max_depth) {
        const size_t entries = (size_t)1 << (3 * P.max_depth);
        if (!ctx->full_grid) TDT_HIP(ctx, hipMalloc((void **)&ctx->full_grid, ((size_t)1 << 18) * sizeof(uint16_t) + sizeof(uint32_t)));
        uint32_t *bad = reinterpret_cast<uint32_t *>(ctx->full_grid + ((size_t)1 << 18));
        TDT_HIP(ctx, hipMemsetAsync(bad, 0, sizeof(uint32_t), ctx->stream));
        hipLaunchKernelGGL(tdt::build_full_grid_kernel, dim3((unsigned)((entries + 255) / 256)), dim3(256), 0, ctx->stream, P.cells, P.cells_dwords, P.max_depth, ctx->full_grid, bad);
        TDT_HIP(ctx, hipGetLastError());
        uint32_t flag = 1;
        TDT_HIP(ctx, hipMemcpyAsync(&flag, bad, sizeof flag, hipMemcpyDeviceToHost, ctx->stream));
        TDT_HIP(ctx, hipStreamSynchronize(ctx->stream));    // once per cells buffer (version), not per frame
        ctx->full_of = cb; ctx->full_version = cb->version; ctx->full_depth = P.max_depth; ctx->full_ok = flag == 0;
      }
      full = ctx->full_ok;
      P.full_grid = ctx->full_grid;
    }
    // trees of depth 6-10 that are not LDS-resident: the 32-bit level-5 table with per-position bands, and for depth 8 / 9 the bricks
    // (tree_lookup_pow2 BRICK), built once per cells buffer
    bool brick = false;
    if (mode != 2 && !counts_out && pow2 && safev && !resident && !ctx->no_specialise && !ctx->no_bricks && P.max_depth >= 6 && P.max_depth <= 10) {
      const tdt_buffer *cb = ctx->ssbo[TDT_SLOT_CELLS];
      if (ctx->brick_of != cb || ctx->brick_version != cb->version || ctx->brick_depth != P.max_depth) {
        const bool has_bricks = tdt::brick_levels(P.max_depth) != 0u;     // (depth 10: the table alone, levels 6.. walked)
        const size_t need = has_bricks ? ((size_t)1 << 15) * tdt::brick_entries(P.max_depth) * sizeof(uint16_t) : 0;
        if (!ctx->brick_grid) TDT_HIP(ctx, hipMalloc((void **)&ctx->brick_grid, ((size_t)1 << 15) * sizeof(uint32_t) + sizeof(uint32_t)));
        if (ctx->bricks_bytes < need) {
          if (ctx->bricks) (void)hipFree(ctx->bricks);
          ctx->bricks = nullptr; ctx->bricks_bytes = 0; ctx->brick_of = nullptr;
          if (hipMalloc(&ctx->bricks, need) != hipSuccess) {      // (a device that cannot spare the address space: the levels are walked)
            (void)hipGetLastError();
            ctx->bricks = nullptr; ctx->no_bricks = true;
          } else ctx->bricks_bytes = need;
        }
      }
      if (!ctx->no_bricks && (ctx->brick_of != cb || ctx->brick_version != cb->version || ctx->brick_depth != P.max_depth)) {
        uint32_t *bad = ctx->brick_grid + ((size_t)1 << 15);
        TDT_HIP(ctx, hipMemsetAsync(bad, 0, sizeof(uint32_t), ctx->stream));
        switch (tdt::brick_levels(P.max_depth)) {
#define TDT_BUILD_BRICKS(B) case B: hipLaunchKernelGGL(tdt::build_bricks_kernel<B>, dim3(1u << 15), dim3(256), 0, ctx->stream, P.cells, P.cells_dwords, ctx->brick_grid, static_cast<uint16_t *>(ctx->bricks), bad); break
          TDT_BUILD_BRICKS(1); TDT_BUILD_BRICKS(2); TDT_BUILD_BRICKS(3); TDT_BUILD_BRICKS(4);
          default: TDT_BUILD_BRICKS(0);
#undef TDT_BUILD_BRICKS
        }
        TDT_HIP(ctx, hipGetLastError());
        uint32_t flag = 1;
        TDT_HIP(ctx, hipMemcpyAsync(&flag, bad, sizeof flag, hipMemcpyDeviceToHost, ctx->stream));
        TDT_HIP(ctx, hipStreamSynchronize(ctx->stream));    // once per cells buffer (version), not per frame
        ctx->brick_of = cb; ctx->brick_version = cb->version; ctx->brick_depth = P.max_depth; ctx->brick_ok = flag == 0;
      }
      brick = !ctx->no_bricks && ctx->brick_ok;
      if (brick) {
        P.brick_grid = ctx->brick_grid; P.bricks = ctx->bricks;
        if (P.lds_nodes > tdt::kBrickLdsCells * 8u) P.lds_nodes = tdt::kBrickLdsCells * 8u;      // (the BRICK builds' LDS node table)
      }
    }
    if (mode != 2 && !counts_out && (pow2 || table_form) && safev && !ctx->no_specialise) {
      // UNIT builds do not multiply by a scale of exactly 1.0f (x * 1.0f is x).  The probe launch of a two-phase frame runs the build
      // that multiplies — the same bits, and the short launch then has a row of its own in profiler statistics instead of halving the
      // average of the launches that do the work
      const bool unit = P.scale == 1.0f && P.inv_scale == 1.0f && P.min_x != 0.0f && P.min_y != 0.0f && P.min_z != 0.0f && !ctx->probe_launch;      // (a zero corner component: see the in-octree test)
      const int form = pow2 ? tdt::FORM_POW2 : tdt::FORM_TABLE;
      TraceFn fn = nullptr;
      if (brick) fn = find_variant(form, P.max_depth, false, false, true, unit);
      if (!fn && full) fn = find_variant(form, P.max_depth, true, true, false, unit);
      if (!fn) fn = find_variant(form, P.max_depth, resident, false, false, unit);
      if (fn) {
        hipLaunchKernelGGL(fn, grid, block, 0, ctx->stream, P); launched = true;
        const int v[6] = {form, P.max_depth, (brick || !resident) ? 0 : 1, (!brick && full) ? 1 : 0, brick ? 1 : 0, unit ? 1 : 0};
        std::memcpy(ctx->last_variant, v, sizeof v);
      }
    }
    if (!launched && mode != 2) { const int v[6] = {pow2 ? tdt::FORM_POW2 : tdt::FORM_LITERAL, 0, 0, 0, 0, 0}; std::memcpy(ctx->last_variant, v, sizeof v); }
#define TDT_LAUNCH(C) do { if (pow2) hipLaunchKernelGGL((tdt::trace_kernel<C, tdt::FORM_POW2>), grid, block, 0, ctx->stream, P); \
                           else hipLaunchKernelGGL((tdt::trace_kernel<C, tdt::FORM_LITERAL>), grid, block, 0, ctx->stream, P); } while (0)
    if (!launched && counts_out && getenv("TDT_COUNT_SPECIALISED") && pow2 && safev && resident && P.max_depth == 6) {
      hipLaunchKernelGGL((tdt::trace_kernel<true, tdt::FORM_POW2, 6, true, true>), grid, block, 0, ctx->stream, P); launched = true;   // diagnostics: region timers of the specialised form
    }
    if (launched) {}
    else if (mode != 2 && !counts_out) TDT_LAUNCH(false);
    else if (mode != 2) TDT_LAUNCH(true);
    else hipLaunchKernelGGL(tdt::resolve_kernel, grid4, block4, 0, ctx->stream, P);
#undef TDT_LAUNCH
    TDT_HIP(ctx, hipGetLastError());
    if (mode != 2) ctx->queue_parity ^= 1u;           // this launch zeroes the other head; a launch that failed leaves this one at zero
  }
  if (counts_out) {
    TDT_HIP(ctx, hipMemcpyAsync(counts_out, ctx->counters, 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
    TDT_HIP(ctx, hipStreamSynchronize(ctx->stream));
  }
  return TDT_OK;
}

}  // namespace

static unsigned long long next_buffer_version() {
  static std::atomic<unsigned long long> next_version{1};   // contexts may live on different threads
  return next_version.fetch_add(1);
}

int tdt::adopt_device_buffer(tdt_ctx *ctx, void *dev, size_t bytes, tdt_buffer **out) {
  tdt_buffer *b = new (std::nothrow) tdt_buffer();
  if (!b) return fail(ctx, TDT_ERR_HIP, "out of host memory");
  b->ctx = ctx; b->bytes = bytes; b->dev = dev; b->version = next_buffer_version();
  const size_t head = bytes < sizeof b->shadow ? bytes : sizeof b->shadow;
  hipError_t e = hipSetDevice(ctx->device);
  if (e == hipSuccess && head) e = hipMemcpyAsync(b->shadow, dev, head, hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  if (e != hipSuccess) { delete b; return hip_fail(ctx, e, "adopt_device_buffer"); }
  ctx->buffers.push_back(b);
  *out = b;
  return TDT_OK;
}

extern "C" {

int tdt_ctx_create(int device_id, void *stream, tdt_ctx **out) {
  if (!out) return fail(nullptr, TDT_ERR_INVALID_VALUE, "null out pointer");
  *out = nullptr;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0)
    return fail(nullptr, TDT_ERR_NO_DEVICE, std::string("no HIP device: ") + (e != hipSuccess ? hipGetErrorString(e) : "device count is 0"));
  if (device_id < 0 || device_id >= n) return fail(nullptr, TDT_ERR_INVALID_VALUE, "device id out of range");
  e = hipSetDevice(device_id);
  if (e != hipSuccess) return fail(nullptr, TDT_ERR_NO_DEVICE, std::string("hipSetDevice: ") + hipGetErrorString(e));
  tdt_ctx *ctx = new (std::nothrow) tdt_ctx();
  if (!ctx) return fail(nullptr, TDT_ERR_HIP, "out of host memory");
  ctx->device = device_id;
  for (auto &s : ctx->ssbo) s = nullptr;
  ctx->atomic0 = nullptr; ctx->image0 = nullptr; ctx->counters = nullptr; ctx->queue = nullptr; ctx->packed = nullptr; ctx->packed_of = nullptr; ctx->packed_version = 0; ctx->present = nullptr; ctx->present_bytes = 0; ctx->frame_carry = nullptr; ctx->frame_carry_bytes = 0; ctx->probe_launch = false; ctx->phase_timing = false; ctx->phase_n = 0; ctx->pixel_log = nullptr; ctx->pixel_log_u32 = 0; ctx->stats = nullptr; ctx->slot_cost = ctx->slot_acc = ctx->slot_order = ctx->order_hist = nullptr; ctx->tile_capacity = ctx->cost_tiles = 0; ctx->cost_dispatches = 0;
  { const char *nc = getenv("TDT_NO_COST_ORDER"); ctx->no_cost_order = nc && nc[0] == '1';
    const char *fs = getenv("TDT_ORDER_SMOOTH"); ctx->force_smooth = fs ? atoi(fs) : -1;
    ctx->no_cost_accum = getenv("TDT_NO_COST_ACCUM") != nullptr;
    ctx->no_two_phase = getenv("TDT_NO_TWO_PHASE") != nullptr;
    ctx->no_prepass = getenv("TDT_NO_PREPASS") != nullptr;
    ctx->no_order_reuse = getenv("TDT_NO_ORDER_REUSE") != nullptr;
    ctx->no_full = getenv("TDT_NO_FULL_GRID") != nullptr;
    ctx->no_table_form = getenv("TDT_NO_TABLE_FORM") != nullptr;
    ctx->no_bricks = getenv("TDT_NO_BRICKS") != nullptr;
    const char *ms = getenv("TDT_MAX_SHARE"); ctx->max_share = ms ? (float)atof(ms) : 1.0f;
    const char *ob = getenv("TDT_ORDER_BLEND"); ctx->order_blend = ob ? (float)atof(ob) : 0.5f;
    const char *tp = getenv("TDT_TWO_PHASE_MIN_SPP"); ctx->two_phase_min_spp = tp && atoi(tp) >= 2 ? atoi(tp) : 16;
    const char *pd = getenv("TDT_PROBE_DIV"); ctx->probe_div = pd && atoi(pd) >= 2 && atoi(pd) <= 64 ? atoi(pd) : 16; }
  ctx->scan = nullptr; ctx->max_parent_value = ctx->max_any_value = ctx->live_nodes = 0xFFFFFFFFu;
  ctx->thr = nullptr; ctx->thr_cc = 0; ctx->thr_ic_bits = 0; ctx->thr_n = 0; ctx->thr_ok = false;
  { hipDeviceProp_t prop; ctx->num_cus = (hipGetDeviceProperties(&prop, device_id) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256; }
  { const char *fg = getenv("TDT_FORCE_GENERIC"); ctx->force_generic = fg && fg[0] == '1';
    const char *ns_ = getenv("TDT_NO_SPECIALISE"); ctx->no_specialise = ns_ && ns_[0] == '1';
    const char *et = getenv("TDT_EVENT_THRESHOLD"); ctx->event_threshold = et ? atoi(et) : 0;
    const char *ek = getenv("TDT_EVENT_K"); ctx->event_k = ek ? (float)atof(ek) : 0.0f;
    const char *ec = getenv("TDT_EVENT_CLAMP"); ctx->event_clamp = ec && atoi(ec) >= 2 && atoi(ec) <= 64 ? atoi(ec) : 40; }
  if (stream) { ctx->stream = (hipStream_t)stream; ctx->own_stream = false; }
  else {
    e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete ctx; return fail(nullptr, TDT_ERR_NO_DEVICE, std::string("hipStreamCreate: ") + hipGetErrorString(e)); }
    ctx->own_stream = true;
  }
  *out = ctx;
  return TDT_OK;
}

void tdt_ctx_destroy(tdt_ctx *ctx) {
  if (!ctx) return;
  if (ctx->multi) { tdt::multi_destroy(ctx); return; }
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  for (tdt_compute *c : ctx->computes) delete c;
  for (tdt_buffer *b : ctx->buffers) { (void)hipFree(b->dev); delete b; }
  for (tdt_image *i : ctx->images) { if (i->owned) (void)hipFree(i->dev); delete i; }
  if (ctx->counters) (void)hipFree(ctx->counters);
  if (ctx->queue) (void)hipFree(ctx->queue);
  if (ctx->packed) (void)hipFree(ctx->packed);
  if (ctx->scan) (void)hipFree(ctx->scan);
  if (ctx->slot_cost) (void)hipFree(ctx->slot_cost);
  if (ctx->slot_order) (void)hipFree(ctx->slot_order);
  if (ctx->slot_acc) (void)hipFree(ctx->slot_acc);
  if (ctx->order_hist) (void)hipFree(ctx->order_hist);
  if (ctx->pixel_log) (void)hipFree(ctx->pixel_log);
  if (ctx->stats) (void)hipFree(ctx->stats);
  if (ctx->present) (void)hipFree(ctx->present);
  if (ctx->frame_carry) (void)hipFree(ctx->frame_carry);
  if (ctx->slot_done) (void)hipFree(ctx->slot_done);
  if (ctx->slot_live) (void)hipFree(ctx->slot_live);
  if (ctx->filter_counts) (void)hipFree(ctx->filter_counts);
  if (ctx->full_grid) (void)hipFree(ctx->full_grid);
  if (ctx->thr) (void)hipFree(ctx->thr);
  if (ctx->brick_grid) (void)hipFree(ctx->brick_grid);
  if (ctx->bricks) (void)hipFree(ctx->bricks);
  if (ctx->phase_timing) for (auto &e : ctx->phase_ev) (void)hipEventDestroy(e);
  tdt::edit_scratch_destroy(ctx);
  if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

int tdt_finish(tdt_ctx *ctx) {
  if (!ctx) return TDT_ERR_INVALID_VALUE;
  if (ctx->multi) return tdt::multi_finish(ctx);
  TDT_HIP(ctx, hipSetDevice(ctx->device));
  TDT_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return TDT_OK;
}

const char *tdt_last_error(const tdt_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_err.c_str(); }

const char *tdt_strerror(int code) {
  switch (code) {
    case TDT_OK: return "ok";
    case TDT_ERR_NO_DEVICE: return "no usable HIP device";
    case TDT_ERR_HIP: return "HIP runtime error";
    case TDT_ERR_INVALID_ENUM: return "gl error: invalid enum";            // mod.rs:47
    case TDT_ERR_INVALID_VALUE: return "gl error: invalid value";          // mod.rs:48
    case TDT_ERR_INVALID_OPERATION: return "gl error: invalid operation";  // mod.rs:49
    case TDT_ERR_VARIABLE_NOT_FOUND: return "failed to locate uniform";    // mod.rs:53-54
    case TDT_ERR_INCOMPLETE: return "dispatch with a missing binding";
    default: return "unknown error code";
  }
}

int tdt_compute_create(tdt_ctx *ctx, int kind, tdt_compute **out) {
  if (!ctx || !out) return fail(ctx, TDT_ERR_INVALID_VALUE, "null argument");
  *out = nullptr;
  if (kind != TDT_PROGRAM_RAYTRACER && kind != TDT_PROGRAM_OCTREE_UPDATE)
    return fail(ctx, TDT_ERR_INVALID_ENUM, "unknown program kind");
  if (ctx->multi) return tdt::multi_compute_create(ctx, kind, out);
  tdt_compute *c = new (std::nothrow) tdt_compute();
  if (!c) return fail(ctx, TDT_ERR_HIP, "out of host memory");
  c->ctx = ctx; c->kind = kind; c->part_rank = 0; c->part_world = 1;   // (uniforms start at 0, as GL's do: value-initialised)
  ctx->computes.push_back(c);
  *out = c;
  return TDT_OK;
}

void tdt_compute_destroy(tdt_compute *c) {
  if (!c) return;
  if (c->ctx->multi) { tdt::multi_compute_destroy(c); return; }
  erase_from(c->ctx->computes, c);
  delete c;
}

int tdt_compute_group_size(const tdt_compute *c, int out[3]) {
  if (!c || !out) return TDT_ERR_INVALID_VALUE;
  if (c->kind == TDT_PROGRAM_OCTREE_UPDATE) { out[0] = 1; out[1] = 1; out[2] = 1; return TDT_OK; }   // octree_update.comp:3
  out[0] = 32; out[1] = 32; out[2] = 1;   // layout(local_size_x = 32, local_size_y = 32) raytracer.comp:3
  return TDT_OK;
}

static int not_found(tdt_compute *c, const char *name, const char *type) {
  // InitializeErr::TypedVariableNotFound's Display, mod.rs:54
  return fail(c->ctx, TDT_ERR_VARIABLE_NOT_FOUND, std::string("failed to locate uniform ") + (name ? name : "(null)") + " with type " + type);
}

int tdt_set_i32(tdt_compute *c, const char *name, int32_t v) {
  if (!c) return TDT_ERR_INVALID_VALUE;
  if (c->ctx->multi) return tdt::multi_set_i32(c, name, v);
  if (name && c->kind == TDT_PROGRAM_RAYTRACER) {
    if (!std::strcmp(name, "camera.image_width")) { c->image_width = v; return TDT_OK; }
    if (!std::strcmp(name, "camera.image_height")) { c->image_height = v; return TDT_OK; }
    if (!std::strcmp(name, "camera.samples_per_pixel")) { c->samples_per_pixel = v; return TDT_OK; }
    if (!std::strcmp(name, "camera.max_bounce")) { c->max_bounce = v; return TDT_OK; }
  }
  return not_found(c, name, "i32");
}

int tdt_set_f32(tdt_compute *c, const char *name, float) {
  if (!c) return TDT_ERR_INVALID_VALUE;
  return not_found(c, name, "f32");   // raytracer.comp has no float uniform (set_f32 is unused: program.rs:61)
}

int tdt_set_vec3f(tdt_compute *c, const char *name, float x, float y, float z) {
  if (!c) return TDT_ERR_INVALID_VALUE;
  if (c->ctx->multi) return tdt::multi_set_vec3f(c, name, x, y, z);
  float *dst = nullptr;
  if (name && c->kind == TDT_PROGRAM_RAYTRACER) {
    if (!std::strcmp(name, "camera.horizontal")) dst = c->horizontal;
    else if (!std::strcmp(name, "camera.vertical")) dst = c->vertical;
    else if (!std::strcmp(name, "camera.lower_left_corner")) dst = c->lower_left_corner;
    else if (!std::strcmp(name, "camera.origin")) dst = c->origin;
  }
  if (!dst) return not_found(c, name, "vec3 f32");
  dst[0] = x; dst[1] = y; dst[2] = z;
  return TDT_OK;
}

int tdt_set_vec3i(tdt_compute *c, const char *name, int32_t, int32_t, int32_t) {
  if (!c) return TDT_ERR_INVALID_VALUE;
  return not_found(c, name, "vec3 i32");   // no ivec3 uniform in raytracer.comp (set_vector3_i32 is unused: program.rs:48)
}

int tdt_buffer_create(tdt_ctx *ctx, const void *data, size_t bytes, tdt_buffer **out) {
  if (!ctx || !out) return fail(ctx, TDT_ERR_INVALID_VALUE, "null argument");
  *out = nullptr;
  if (bytes && !data) return fail(ctx, TDT_ERR_INVALID_VALUE, "null data with non-zero size");
  if (ctx->multi) return tdt::multi_buffer_create(ctx, data, bytes, out);
  tdt_buffer *b = new (std::nothrow) tdt_buffer();
  if (!b) return fail(ctx, TDT_ERR_HIP, "out of host memory");
  b->ctx = ctx; b->bytes = bytes; b->dev = nullptr;
  b->version = next_buffer_version();
  std::memset(b->shadow, 0, sizeof b->shadow);
  TDT_HIP(ctx, hipSetDevice(ctx->device));
  // 16 bytes of zero slack so that the widest load at the last valid dword stays inside the allocation
  hipError_t e = hipMalloc(&b->dev, bytes + 16);
  if (e != hipSuccess) { delete b; return hip_fail(ctx, e, "hipMalloc"); }
  e = hipMemsetAsync((char *)b->dev + bytes, 0, 16, ctx->stream);
  if (e == hipSuccess && bytes) e = hipMemcpyAsync(b->dev, data, bytes, hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);   // glBufferData semantics: `data` may be freed on return
  if (e != hipSuccess) { (void)hipFree(b->dev); delete b; return hip_fail(ctx, e, "upload"); }
  std::memcpy(b->shadow, data, bytes < sizeof b->shadow ? bytes : sizeof b->shadow);
  ctx->buffers.push_back(b);
  *out = b;
  return TDT_OK;
}

void tdt_buffer_destroy(tdt_buffer *b) {
  if (!b) return;
  if (b->ctx->multi) { tdt::multi_buffer_destroy(b); return; }
  tdt_ctx *ctx = b->ctx;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  for (auto &s : ctx->ssbo) if (s == b) s = nullptr;
  if (ctx->atomic0 == b) ctx->atomic0 = nullptr;
  erase_from(ctx->buffers, b);
  (void)hipFree(b->dev);
  delete b;
}

int tdt_bind_buffer_base(tdt_ctx *ctx, int target, unsigned slot, tdt_buffer *b) {
  if (!ctx) return TDT_ERR_INVALID_VALUE;
  if (b && b->ctx != ctx) return fail(ctx, TDT_ERR_INVALID_OPERATION, "buffer belongs to another context");
  if (ctx->multi) return tdt::multi_bind_buffer_base(ctx, target, slot, b);
  if (target == TDT_SHADER_STORAGE_BUFFER) {
    if (slot >= (unsigned)kNumSlots) return fail(ctx, TDT_ERR_INVALID_VALUE, "shader-storage slot out of range (0..7)");
    ctx->ssbo[slot] = b;
    return TDT_OK;
  }
  if (target == TDT_ATOMIC_COUNTER_BUFFER) {
    if (slot != 0) return fail(ctx, TDT_ERR_INVALID_VALUE, "atomic-counter slot out of range (0)");
    ctx->atomic0 = b;
    return TDT_OK;
  }
  return fail(ctx, TDT_ERR_INVALID_ENUM, "unknown buffer target");
}

int tdt_buffer_read(tdt_buffer *b, size_t offset, size_t bytes, void *dst) {
  if (!b) return TDT_ERR_INVALID_VALUE;
  if (b->ctx->multi) return b->replicas.empty() ? TDT_ERR_INVALID_VALUE : tdt_buffer_read(b->replicas[0], offset, bytes, dst);
  tdt_ctx *ctx = b->ctx;
  if (offset > b->bytes || bytes > b->bytes - offset) return fail(ctx, TDT_ERR_INVALID_VALUE, "read range outside the buffer");
  if (!bytes) return TDT_OK;
  if (!dst) return fail(ctx, TDT_ERR_INVALID_VALUE, "null destination");
  TDT_HIP(ctx, hipSetDevice(ctx->device));
  TDT_HIP(ctx, hipMemcpyAsync(dst, (const char *)b->dev + offset, bytes, hipMemcpyDeviceToHost, ctx->stream));
  TDT_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return TDT_OK;
}

int tdt_buffer_sub_data(tdt_buffer *b, size_t offset, size_t bytes, const void *data) {
  if (!b) return TDT_ERR_INVALID_VALUE;
  if (b->ctx->multi) return tdt::multi_buffer_sub_data(b, offset, bytes, data);
  tdt_ctx *ctx = b->ctx;
  if (offset > b->bytes || bytes > b->bytes - offset) return fail(ctx, TDT_ERR_INVALID_VALUE, "sub-data range outside the buffer");
  if (!bytes) return TDT_OK;
  if (!data) return fail(ctx, TDT_ERR_INVALID_VALUE, "null data");
  TDT_HIP(ctx, hipSetDevice(ctx->device));
  TDT_HIP(ctx, hipMemcpyAsync((char *)b->dev + offset, data, bytes, hipMemcpyHostToDevice, ctx->stream));
  TDT_HIP(ctx, hipStreamSynchronize(ctx->stream));
  b->version += 0x100000000ull;
  if (offset < sizeof b->shadow) {
    size_t n = sizeof b->shadow - offset; if (n > bytes) n = bytes;
    std::memcpy(b->shadow + offset, data, n);
  }
  return TDT_OK;
}

int tdt_image_create_rgba32f(tdt_ctx *ctx, int width, int height, tdt_image **out) {
  if (!ctx || !out) return fail(ctx, TDT_ERR_INVALID_VALUE, "null argument");
  *out = nullptr;
  if (width <= 0 || height <= 0) return fail(ctx, TDT_ERR_INVALID_VALUE, "image size must be positive");
  if (ctx->multi) return tdt::multi_image_create(ctx, nullptr, width, height, out);
  tdt_image *img = new (std::nothrow) tdt_image();
  if (!img) return fail(ctx, TDT_ERR_HIP, "out of host memory");
  img->ctx = ctx; img->w = width; img->h = height; img->owned = true; img->dev = nullptr;
  size_t bytes = (size_t)width * height * 16;
  TDT_HIP(ctx, hipSetDevice(ctx->device));
  hipError_t e = hipMalloc((void **)&img->dev, bytes);
  if (e == hipSuccess) e = hipMemsetAsync(img->dev, 0, bytes, ctx->stream);
  if (e != hipSuccess) { if (img->dev) (void)hipFree(img->dev); delete img; return hip_fail(ctx, e, "image allocation"); }
  ctx->images.push_back(img);
  *out = img;
  return TDT_OK;
}

int tdt_image_wrap_device(tdt_ctx *ctx, void *device_ptr, int width, int height, tdt_image **out) {
  if (!ctx || !out) return fail(ctx, TDT_ERR_INVALID_VALUE, "null argument");
  *out = nullptr;
  if (!device_ptr || width <= 0 || height <= 0) return fail(ctx, TDT_ERR_INVALID_VALUE, "bad device pointer or size");
  if (((uintptr_t)device_ptr & 15) != 0) return fail(ctx, TDT_ERR_INVALID_VALUE, "image memory must be 16-byte aligned");
  if (ctx->multi) return tdt::multi_image_create(ctx, device_ptr, width, height, out);
  tdt_image *img = new (std::nothrow) tdt_image();
  if (!img) return fail(ctx, TDT_ERR_HIP, "out of host memory");
  img->ctx = ctx; img->w = width; img->h = height; img->owned = false; img->dev = (float *)device_ptr;
  ctx->images.push_back(img);
  *out = img;
  return TDT_OK;
}

void tdt_image_destroy(tdt_image *img) {
  if (!img) return;
  if (img->ctx->multi) { tdt::multi_image_destroy(img); return; }
  tdt_ctx *ctx = img->ctx;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  if (ctx->image0 == img) ctx->image0 = nullptr;
  erase_from(ctx->images, img);
  if (img->owned) (void)hipFree(img->dev);
  delete img;
}

int tdt_bind_image(tdt_ctx *ctx, unsigned unit, tdt_image *img) {
  if (!ctx) return TDT_ERR_INVALID_VALUE;
  if (unit != 0) return fail(ctx, TDT_ERR_INVALID_VALUE, "only image unit 0 exists (raytracer.comp:4)");
  if (img && img->ctx != ctx) return fail(ctx, TDT_ERR_INVALID_OPERATION, "image belongs to another context");
  ctx->image0 = img;
  return TDT_OK;
}

int tdt_image_width(const tdt_image *img) { return img ? img->w : 0; }
int tdt_image_height(const tdt_image *img) { return img ? img->h : 0; }
void *tdt_image_device_ptr(const tdt_image *img) { return img ? img->dev : nullptr; }

int tdt_image_read(tdt_image *img, float *dst) {
  if (!img || !dst) return TDT_ERR_INVALID_VALUE;
  if (img->ctx->multi) return tdt_image_read(img->full, dst);
  tdt_ctx *ctx = img->ctx;
  TDT_HIP(ctx, hipSetDevice(ctx->device));
  TDT_HIP(ctx, hipMemcpyAsync(dst, img->dev, (size_t)img->w * img->h * 16, hipMemcpyDeviceToHost, ctx->stream));
  TDT_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return TDT_OK;
}

int tdt_image_read_rgba8(tdt_image *img, int top_down, uint8_t *dst) {
  if (!img || !dst) return TDT_ERR_INVALID_VALUE;
  if (img->ctx->multi) return tdt_image_read_rgba8(img->full, top_down, dst);
  tdt_ctx *ctx = img->ctx;
  TDT_HIP(ctx, hipSetDevice(ctx->device));
  const size_t bytes = (size_t)img->w * img->h * 4;
  if (ctx->present_bytes < bytes) {
    if (ctx->present) (void)hipFree(ctx->present);
    ctx->present = nullptr; ctx->present_bytes = 0;
    TDT_HIP(ctx, hipMalloc((void **)&ctx->present, bytes));
    ctx->present_bytes = bytes;
  }
  hipLaunchKernelGGL(tdt::present_kernel, dim3((unsigned)(img->w + 63) / 64, (unsigned)(img->h + 3) / 4), dim3(256), 0, ctx->stream,
                     (const float4 *)img->dev, ctx->present, img->w, img->h, top_down ? 1 : 0);
  TDT_HIP(ctx, hipGetLastError());
  TDT_HIP(ctx, hipMemcpyAsync(dst, ctx->present, bytes, hipMemcpyDeviceToHost, ctx->stream));
  TDT_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return TDT_OK;
}

// events around the launches of a frame (only when tdt_debug_phase_timing switched them on)
static void phase_mark(tdt_ctx *ctx, int i) {
  if (!ctx->phase_timing) return;
  if (hipEventRecord(ctx->phase_ev[i], ctx->stream) == hipSuccess) ctx->phase_n = i + 1;
}

static int dispatch_frame(tdt_compute *c, int width, int height, int depth);

int tdt_dispatch_compute(tdt_compute *c, int width, int height, int depth) {
  if (!c) return TDT_ERR_INVALID_VALUE;
  if (c->ctx->multi) return tdt::multi_dispatch_compute(c, width, height, depth);
  if (c->kind == TDT_PROGRAM_OCTREE_UPDATE) return tdt::launch_update(c, width, height, depth);
  tdt_ctx *ctx = c->ctx;
  // a camera outside the octree: the pixels whose rays all miss it are finished by the miss pre-pass (miss_prepass_kernel) and
  // taken out of the hand-out order of this frame's launches.  (Inside the octree every ray starts in the root cube: nothing to find.)
  ctx->use_done = false;
  const tdt_buffer *of = ctx->ssbo[TDT_SLOT_OCTREE_FLOATS];
  if (!ctx->no_prepass && of && of->bytes >= 28 && c->samples_per_pixel >= 1 && c->max_bounce >= 1) {
    float f[7];
    std::memcpy(f, of->shadow, sizeof f);
    bool outside = false;
    for (int a = 0; a < 3; a++) outside = outside || c->origin[a] < f[a] || c->origin[a] > f[a] + f[4];
    if (outside) {
      const int rc = launch(c, width, height, depth, 3, 0, 0, nullptr, 0, nullptr);
      if (rc != TDT_OK) return rc;
      ctx->use_done = true;
    }
  }
  const int rc = dispatch_frame(c, width, height, depth);
  ctx->use_done = false;
  return rc;
}

static int dispatch_frame(tdt_compute *c, int width, int height, int depth) {
  // A frame whose inputs differ from what the recorded pixel costs were measured on (first frame, moved camera, edited
  // scene) is traced in two phases: spp/16 probe samples per pixel in image (or tile-sum) order, then — the launch below
  // sees identical inputs and fresh costs — the rest in the per-pixel cost order of THIS frame's probe, and a resolve.  Running
  // sums and the hit-record carry go through HBM between the phases exactly as in progressive rendering, so the frame is the
  // same bits as one pass (tests/test_gpu_fullsize.py).  Frames that repeat their inputs take one pass in the exact order.
  tdt_ctx *ctx = c->ctx;
  const int spp = c->samples_per_pixel;
  bool ready = ctx->image0 != nullptr && !ctx->no_cost_order && !ctx->no_two_phase && spp >= ctx->two_phase_min_spp && spp >= 2;
  for (int sl : {TDT_SLOT_CELLS, TDT_SLOT_MATERIALS, TDT_SLOT_ALBEDOS, TDT_SLOT_METAL, TDT_SLOT_DIELECTRIC, TDT_SLOT_OCTREE_FLOATS, TDT_SLOT_OCTREE_INTS})
    ready = ready && ctx->ssbo[sl] != nullptr;
  if (ready && ctx->ssbo[TDT_SLOT_OCTREE_FLOATS]->bytes >= 28 && ctx->ssbo[TDT_SLOT_OCTREE_INTS]->bytes >= 12) {
    const Cover k = cover_of(c, width, height);
    const Tiles t = tiles_of(c, k);
    CostSig sig;
    make_sig(ctx, c, k, ctx->image0, &sig);
    const bool replay = ctx->cost_tiles == (uint32_t)t.owned && std::memcmp(&sig, &ctx->cost_sig, sizeof sig) == 0;
    if (!replay && t.owned > 0) {
      TDT_HIP(ctx, hipSetDevice(ctx->device));
      const size_t px = (size_t)ctx->image0->w * (size_t)ctx->image0->h, slots = (size_t)t.owned * 1024;
      const size_t need = (px > slots ? px : slots) * 16 * sizeof(float);
      if (ctx->frame_carry_bytes < need) {
        if (ctx->frame_carry) (void)hipFree(ctx->frame_carry);
        ctx->frame_carry = nullptr; ctx->frame_carry_bytes = 0; ctx->probe_launch = false;
        TDT_HIP(ctx, hipMalloc(&ctx->frame_carry, need));
        ctx->frame_carry_bytes = need;
      }
      const int probe = spp / ctx->probe_div >= 1 ? spp / ctx->probe_div : 1;        // spp/16; measured: 1/8 and 1/32 are 0-3 % slower, 1/64 5 % (TDT_PROBE_DIV)
      ctx->probe_launch = true;
      phase_mark(ctx, 0);
      int rc = launch(c, width, height, depth, 1, 0, probe, ctx->frame_carry, 0, nullptr);
      ctx->probe_launch = false;
      phase_mark(ctx, 1);
      ctx->carry_final = true;                       // 64 B per pixel that the resolve does not read: not written
      if (rc == TDT_OK) rc = launch(c, width, height, depth, 1, probe, spp - probe, ctx->frame_carry, 0, nullptr);
      ctx->carry_final = false;
      phase_mark(ctx, 2);
      if (rc == TDT_OK) rc = launch(c, width, height, depth, 2, 0, 0, nullptr, spp, nullptr);
      phase_mark(ctx, 3);
      return rc;
    }
  }
  phase_mark(ctx, 0);
  const int rc = launch(c, width, height, depth, 0, 0, spp, nullptr, spp, nullptr);
  phase_mark(ctx, 1);
  return rc;
}

int tdt_set_partition(tdt_compute *c, int rank, int world) {
  if (!c) return TDT_ERR_INVALID_VALUE;
  if (c->ctx->multi) return fail(c->ctx, TDT_ERR_INVALID_OPERATION, "a multi-device context partitions the image over its devices itself");
  if (world < 1 || rank < 0 || rank >= world) return fail(c->ctx, TDT_ERR_INVALID_VALUE, "need 0 <= rank < world");
  c->part_rank = rank; c->part_world = world;
  return TDT_OK;
}

int tdt_dispatch_accumulate(tdt_compute *c, int width, int height, int depth, int spp_begin, int spp_count, void *carry) {
  if (!c) return TDT_ERR_INVALID_VALUE;
  if (c->kind != TDT_PROGRAM_RAYTRACER) return fail(c->ctx, TDT_ERR_INVALID_OPERATION, "not the raytracer program");
  if (spp_begin < 0 || spp_count < 0) return fail(c->ctx, TDT_ERR_INVALID_VALUE, "negative sample range");
  if (c->ctx->multi) return tdt::multi_dispatch_accumulate(c, width, height, depth, spp_begin, spp_count, carry);
  if (carry && ((uintptr_t)carry & 15) != 0) return fail(c->ctx, TDT_ERR_INVALID_VALUE, "carry memory must be 16-byte aligned");
  return launch(c, width, height, depth, 1, spp_begin, spp_count, carry, 0, nullptr);
}

int tdt_dispatch_resolve(tdt_compute *c, int width, int height, int depth, int total_spp) {
  if (!c) return TDT_ERR_INVALID_VALUE;
  if (c->kind != TDT_PROGRAM_RAYTRACER) return fail(c->ctx, TDT_ERR_INVALID_OPERATION, "not the raytracer program");
  if (c->ctx->multi) return tdt::multi_dispatch_resolve(c, width, height, depth, total_spp);
  return launch(c, width, height, depth, 2, 0, 0, nullptr, total_spp, nullptr);
}

int64_t tdt_covered_pixels(const tdt_compute *c, int width, int height, int depth) {
  (void)depth;
  if (!c) return 0;
  return owned_pixels(c, cover_of(c, width, height));
}

int tdt_owned_tiles(const tdt_compute *c, int width, int height, int depth, int *tiles_x, int *tiles_total) {
  (void)depth;
  if (!c) return 0;
  Cover k = cover_of(c, width, height);
  Tiles t = tiles_of(c, k);
  if (tiles_x) *tiles_x = t.tiles_x;
  if (tiles_total) *tiles_total = t.total;
  return t.owned;
}

int tdt_dispatch_counted(tdt_compute *c, int width, int height, int depth, uint64_t counts[8]) {
  if (!c || !counts) return TDT_ERR_INVALID_VALUE;
  if (c->ctx->multi) return tdt::multi_dispatch_counted(c, width, height, depth, counts);
  if (c->kind != TDT_PROGRAM_RAYTRACER) return fail(c->ctx, TDT_ERR_INVALID_OPERATION, "not the raytracer program");
  static_assert(sizeof(unsigned long long) == sizeof(uint64_t), "");
  return launch(c, width, height, depth, 0, 0, c->samples_per_pixel, nullptr, c->samples_per_pixel,
                reinterpret_cast<unsigned long long *>(counts));
}

int tdt_dispatch_counted_range(tdt_compute *c, int width, int height, int depth, int spp_begin, int spp_count, void *carry,
                               uint64_t counts[8]) {
  if (!c || !counts) return TDT_ERR_INVALID_VALUE;
  if (c->ctx->multi) return fail(c->ctx, TDT_ERR_INVALID_OPERATION, "progressive passes are a single-device feature (per-device carry memory)");
  if (c->kind != TDT_PROGRAM_RAYTRACER) return fail(c->ctx, TDT_ERR_INVALID_OPERATION, "not the raytracer program");
  if (spp_begin < 0 || spp_count < 0) return fail(c->ctx, TDT_ERR_INVALID_VALUE, "negative sample range");
  if (carry && ((uintptr_t)carry & 15) != 0) return fail(c->ctx, TDT_ERR_INVALID_VALUE, "carry memory must be 16-byte aligned");
  return launch(c, width, height, depth, 1, spp_begin, spp_count, carry, 0, reinterpret_cast<unsigned long long *>(counts));
}

int tdt_forget_costs(tdt_ctx *ctx) {
  if (!ctx) return TDT_ERR_INVALID_VALUE;
  if (ctx->multi) return tdt::multi_forget_costs(ctx);
  ctx->cost_tiles = 0; ctx->cost_dispatches = 0; ctx->order_exact = false;      // launch(): "no usable history" -> image order, fresh cost arrays
  return TDT_OK;
}

int tdt_debug_phase_timing(tdt_ctx *ctx, int enable, float ms[3]) {
  if (!ctx) return TDT_ERR_INVALID_VALUE;
  TDT_HIP(ctx, hipSetDevice(ctx->device));
  if (ms) {
    ms[0] = ms[1] = ms[2] = 0.f;
    if (ctx->phase_timing && ctx->phase_n >= 2) {
      TDT_HIP(ctx, hipEventSynchronize(ctx->phase_ev[ctx->phase_n - 1]));
      for (int i = 0; i + 1 < ctx->phase_n; i++) TDT_HIP(ctx, hipEventElapsedTime(&ms[ctx->phase_n == 2 ? 1 : i], ctx->phase_ev[i], ctx->phase_ev[i + 1]));
    }
  }
  if (enable && !ctx->phase_timing) {
    for (auto &e : ctx->phase_ev) TDT_HIP(ctx, hipEventCreate(&e));
    ctx->phase_timing = true; ctx->phase_n = 0;
  } else if (!enable && ctx->phase_timing) {
    for (auto &e : ctx->phase_ev) (void)hipEventDestroy(e);
    ctx->phase_timing = false; ctx->phase_n = 0;
  }
  return TDT_OK;
}

/* which build of the trace kernel the last trace launch of the context ran: {form (0 literal, 1 power-of-two, 2 thresholds), compile-time
 * depth (0: the general kernel), resident, full, brick, unit} — so that tests can tell a scene that silently fell back to the general
 * kernel from one that runs its specialised build (same pixels either way) */
int tdt_debug_last_variant(const tdt_ctx *ctx, int out[6]) {
  if (!ctx || !out) return TDT_ERR_INVALID_VALUE;
  if (ctx->multi) ctx = tdt::multi_first_member(const_cast<tdt_ctx *>(ctx));
  std::memcpy(out, ctx->last_variant, 6 * sizeof(int));
  return TDT_OK;
}

/* -DTDT_STATS builds of the library only (tools/loss_budget.py): pass / lane statistics of the product trace kernels launched on this
 * context since the last reset (the STAT_* rows of tdt_rt.hip); the first call switches the collection on.  The product library
 * answers TDT_ERR_INVALID_OPERATION. */
int tdt_debug_stats(tdt_ctx *ctx, uint64_t *out, int n_words, int reset) {
  if (!ctx) return TDT_ERR_INVALID_VALUE;
  if (ctx->multi) ctx = tdt::multi_first_member(ctx);
#ifdef TDT_STATS
  TDT_HIP(ctx, hipSetDevice(ctx->device));
  const bool fresh = ctx->stats == nullptr;
  constexpr size_t kStatWords = tdt::STAT_COUNT + 1 + 8192;      // totals, the time the queue ran dry, per-wave end times
  if (fresh) TDT_HIP(ctx, hipMalloc((void **)&ctx->stats, kStatWords * sizeof(unsigned long long)));
  TDT_HIP(ctx, hipStreamSynchronize(ctx->stream));
  const size_t n_out = n_words > 0 ? ((size_t)n_words < kStatWords ? (size_t)n_words : kStatWords) : (size_t)tdt::STAT_COUNT;
  if (out && !fresh) TDT_HIP(ctx, hipMemcpy(out, ctx->stats, n_out * sizeof(uint64_t), hipMemcpyDeviceToHost));
  else if (out) std::memset(out, 0, n_out * sizeof(uint64_t));
  if (reset || fresh) {
    std::vector<unsigned long long> init(kStatWords, 0ull);
    init[tdt::STAT_T_FIRST] = ~0ull; init[tdt::STAT_COUNT] = ~0ull;
    TDT_HIP(ctx, hipMemcpy(ctx->stats, init.data(), kStatWords * sizeof(unsigned long long), hipMemcpyHostToDevice));
  }
  return TDT_OK;
#else
  (void)out; (void)n_words; (void)reset;
  return fail(ctx, TDT_ERR_INVALID_OPERATION, "pass statistics need a -DTDT_STATS build of the library (tools/loss_budget.py)");
#endif
}

/* lane-utilisation diagnostics of the last tdt_dispatch_counted on this context (see Counters) */
int tdt_debug_counters(tdt_ctx *ctx, uint64_t out[32]) {
  if (!ctx || !out || !ctx->counters) return TDT_ERR_INVALID_VALUE;
  TDT_HIP(ctx, hipSetDevice(ctx->device));
  TDT_HIP(ctx, hipMemcpy(out, ctx->counters, 32 * sizeof(uint64_t), hipMemcpyDeviceToHost));
  return TDT_OK;
}
int tdt_debug_pixel_log(tdt_ctx *ctx, uint32_t *out, size_t n_u32) {
  if (!ctx || !out || !ctx->pixel_log || n_u32 > ctx->pixel_log_u32) return TDT_ERR_INVALID_VALUE;
  TDT_HIP(ctx, hipSetDevice(ctx->device));
  TDT_HIP(ctx, hipMemcpy(out, ctx->pixel_log, n_u32 * 4, hipMemcpyDeviceToHost));
  return TDT_OK;
}
/* per-wave end times (100 MHz ticks) of the last instrumented dispatch: n <= 16384 entries */
int tdt_debug_wave_ends(tdt_ctx *ctx, uint64_t *out, int n) {
  if (!ctx || !out || !ctx->counters || n < 0 || n > 16384 + 256) return TDT_ERR_INVALID_VALUE;
  TDT_HIP(ctx, hipSetDevice(ctx->device));
  TDT_HIP(ctx, hipMemcpy(out, ctx->counters + 32, (size_t)n * sizeof(uint64_t), hipMemcpyDeviceToHost));
  return TDT_OK;
}

/* Exhaustive self-test of the kernels' short correctly-rounded rcp / sqrt / rsq forms against the
 * IEEE expressions on all 2^32 inputs; *mismatches must come back 0 (which: 0 rcp, 1 sqrt, 2 rsq). */
int tdt_selftest(tdt_ctx *ctx, int which, uint64_t *mismatches) {
  if (!ctx || !mismatches || which < 0 || which > 16) return TDT_ERR_INVALID_VALUE;
  TDT_HIP(ctx, hipSetDevice(ctx->device));
  if (!ctx->counters) TDT_HIP(ctx, hipMalloc((void **)&ctx->counters, (32 + 16384 + 256) * sizeof(unsigned long long)));
  TDT_HIP(ctx, hipMemsetAsync(ctx->counters, 0, sizeof(unsigned long long), ctx->stream));
  hipLaunchKernelGGL(tdt::selftest_kernel, dim3(4096), dim3(256), 0, ctx->stream, which, ctx->counters);
  TDT_HIP(ctx, hipGetLastError());
  TDT_HIP(ctx, hipMemcpyAsync(mismatches, ctx->counters, sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
  TDT_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return TDT_OK;
}

/* Exhaustive check of the per-cell x-index thresholds (FORM_TABLE builds, x_thresholds in trace_device.hpp) for one cell_count:
 * every f in [0,1) x every cell index below n_cells against the literal formula.  *shape_ok = 0 when some cell's index is not of the
 * two-threshold shape (such a scene runs the literal kernel); shift != 0 moves the thresholds by that many ulps (the harness:
 * must report mismatches). */
int tdt_selftest_index(tdt_ctx *ctx, int32_t cell_count, float inv_cell_count, uint32_t n_cells, int shift, uint64_t *mismatches, int *shape_ok) {
  if (!ctx || !mismatches || !shape_ok || n_cells == 0 || n_cells > 65535u) return TDT_ERR_INVALID_VALUE;
  if (ctx->multi) ctx = tdt::multi_first_member(ctx);
  TDT_HIP(ctx, hipSetDevice(ctx->device));
  float4 *thr = nullptr;
  TDT_HIP(ctx, hipMalloc((void **)&thr, (size_t)n_cells * sizeof(float4) + 16));
  unsigned long long *cnt = reinterpret_cast<unsigned long long *>(thr + n_cells);
  uint32_t *aux = reinterpret_cast<uint32_t *>(cnt + 1);
  hipError_t e = hipMemsetAsync(cnt, 0, 16, ctx->stream);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(tdt::build_thresholds_kernel, dim3((n_cells + 255u) / 256u), dim3(256), 0, ctx->stream, inv_cell_count, cell_count, n_cells, (float2 *)nullptr, aux, thr);
    hipLaunchKernelGGL(tdt::selftest_index_kernel, dim3(512, n_cells), dim3(256), 0, ctx->stream, inv_cell_count, cell_count, thr, shift, cnt);
    e = hipGetLastError();
  }
  unsigned long long host[2] = {0, 0};
  if (e == hipSuccess) e = hipMemcpyAsync(host, cnt, 16, hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  (void)hipFree(thr);
  if (e != hipSuccess) return hip_fail(ctx, e, "tdt_selftest_index");
  *mismatches = host[0]; *shape_ok = (uint32_t)host[1] == 0u ? 1 : 0;
  return TDT_OK;
}

int tdt_assemble_tiles(tdt_compute *c, const void *gathered, int world, int tiles_per_rank, tdt_image *dst,
                       int width, int height, int depth) {
  (void)depth;
  if (!c || !gathered || !dst) return TDT_ERR_INVALID_VALUE;
  tdt_ctx *ctx = c->ctx;
  if (world < 1 || tiles_per_rank < 0) return fail(ctx, TDT_ERR_INVALID_VALUE, "bad world / tiles_per_rank");
  if (dst->w != c->image_width || dst->h != c->image_height)
    return fail(ctx, TDT_ERR_INVALID_OPERATION, "destination image must be camera.image_width x image_height");
  Cover k = cover_of(c, width, height);
  if (k.cover_w <= 0 || k.cover_h <= 0) return TDT_OK;
  const int tiles_x = (k.cover_w + 31) / 32, total = tiles_x * ((k.cover_h + 31) / 32);
  if ((int64_t)tiles_per_rank * world < total) return fail(ctx, TDT_ERR_INVALID_VALUE, "gathered buffer holds fewer tiles than the image has");
  TDT_HIP(ctx, hipSetDevice(ctx->device));
  dim3 grid((unsigned)((k.cover_w + 63) / 64), (unsigned)((k.cover_h + 3) / 4), 1), block(256, 1, 1);
  hipLaunchKernelGGL(tdt::assemble_kernel, grid, block, 0, ctx->stream, (const float4 *)gathered, (float4 *)dst->dev,
                     c->image_width, k.cover_w, k.cover_h, tiles_x, world, tiles_per_rank);
  TDT_HIP(ctx, hipGetLastError());
  return TDT_OK;
}

}  // extern "C"
