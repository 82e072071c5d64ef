// The wavefront form of the trace kernel (round 2, experimental: TDT_WAVEFRONT=1) for small trees — the ones whose whole-depth
// lookup table (tree_lookup_pow2 FULL) leaves the CU's LDS free.  Included by tdt_rt.hip inside namespace tdt.
//
// trace_kernel keeps a pixel's whole state in the registers of ONE lane, so a lane that waits for the (long, material-divergent)
// event code idles through traversal passes, an event pass serves 25-30 lanes of 64, and its three material branches run one
// after the other with a third of those each: 53 % of the lanes are live per issued instruction.  Here a pixel's path state is
// a CONTEXT in LDS (1280 per 1024-lane block), and lanes are workers:
//   traversal   a lane takes a context from the TRAV queue, loads its ray, and steps it through the octree (the same step
//               as trace_kernel's) until it hits a leaf or leaves; then it writes what it found back and queues the context
//               for the code it needs next — by MATERIAL for a hit, END for a path that is over — and takes the next ray;
//   service     when a queue holds a wave's worth of contexts, a wave takes 64 of them and runs that one piece of event code
//               with every lane live: one material's scatter + the next ray's root test, or sample end + next primary ray
//               (+ pixel end + next pixel).
// Every lane still performs exactly the reference's operation sequence for the pixel it works on (same device functions, same
// order; a pixel's samples are summed in order because a context is in one place at a time), so the image is the same bits.
// The two hit records CubeHit's call sites carry from call to call (Carry) live per context in global memory (L2): they are
// written once per ray / hit and read only by the rare ray that needs an old one.
#pragma once

constexpr int kWfContexts = 1280;
enum : int { WQ_TRAV = 0, WQ_MAT0 = 1, WQ_MAT1 = 2, WQ_MAT2 = 3, WQ_END = 4, WQ_COUNT = 5 };
constexpr int kWfPcWords = 16;     // root record (7) + root t, leaf record (7), pad: per context, in global memory

struct WfShared {
  float ox[kWfContexts], oy[kWfContexts], oz[kWfContexts], dx[kWfContexts], dy[kWfContexts], dz[kWfContexts];
  float ts[kWfContexts], tmax[kWfContexts];                 // OctreeHit's prologue: where the walk starts, where the octree ends
  float ar[kWfContexts], ag[kWfContexts], ab[kWfContexts];  // accumulative_attenuation rc:267
  float sr[kWfContexts], sg[kWfContexts], sb[kWfContexts];  // color rc:237
  uint32_t slot[kWfContexts];                               // queue slot of the pixel
  uint32_t sl[kWfContexts];                                 // sample index | loop_count << 16
  uint32_t work[kWfContexts];                               // the pixel's cost so far (hand-out order of the next dispatch)
  uint32_t hit[kWfContexts];                                // hit_index << 2 | use_leaf << 1 | leaf_rec
  float hbx[kWfContexts], hby[kWfContexts], hbz[kWfContexts], hsz[kWfContexts], ht[kWfContexts];   // the leaf's cube and entry t
  uint32_t ring[WQ_COUNT][kWfContexts];                     // ticket queues of context ids: valid bit | ticket (15 bits) << 16 | id; 0 = never written
  uint32_t head[8], tail[8];                                // monotonic tickets per queue
  uint32_t live;                                            // contexts that hold a pixel
#ifdef TDT_WF_DEBUG
  uint32_t own[kWfContexts];                                // 0 free, 1 queued, 2 held by a worker
#endif
  uint8_t mtype[1024];                                      // materials[i].type for i < 1024 (3 = anything else)
};

TDT_DEV uint32_t wf_lane() { return threadIdx.x & 63u; }
// the per-context hit records in global memory are written by one wave and read by another of the same block: loads that
// bypass the CU's vector L1 (sc1, served by L2), stores as they come (write-through)
TDT_DEV float wf_gld(const float *p) { return __uint_as_float(__hip_atomic_load(reinterpret_cast<const uint32_t *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)); }

// the lanes with `want` append their context to queue q (one LDS atomic per wave)
TDT_DEV void wf_push(WfShared &S, int q, bool want, uint32_t id) {
  const unsigned long long m = __ballot(want);
  if (m == 0ull) return;
  const uint32_t first = (uint32_t)__builtin_ctzll(m);
  uint32_t base = 0;
  if (wf_lane() == first) base = atomicAdd(&S.tail[q], (uint32_t)__popcll(m));
  base = (uint32_t)__shfl((int)base, (int)first, 64);
  if (want) {
#ifdef TDT_WF_DEBUG
    { const uint32_t old = atomicExch(&S.own[id], 1u); if (old != 2u) printf("WF: push of context %u to queue %d while its state is %u (block %u)\n", id, q, old, blockIdx.x); }
#endif
    const uint32_t t = base + (uint32_t)__popcll(m & ((1ull << wf_lane()) - 1ull));
    // the context's fields, then its id: a release store (LDS to LDS, work-group scope)
    __hip_atomic_store(&S.ring[q][t % (uint32_t)kWfContexts], 0x80000000u | ((t & 0x7FFFu) << 16) | id, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
}
// this wave takes up to n entries of queue q: returns how many, and the first ticket (wave-uniform)
TDT_DEV uint32_t wf_claim(WfShared &S, int q, uint32_t n, uint32_t &base) {
  uint32_t got = 0, b = 0;
  if (wf_lane() == 0u) {
    uint32_t h = __hip_atomic_load(&S.head[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    for (int tries = 0; tries < 8; tries++) {
      const uint32_t avail = __hip_atomic_load(&S.tail[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) - h;
      const uint32_t k = avail < n ? avail : n;
      if (k == 0u || k > (uint32_t)kWfContexts) break;
      const uint32_t old = atomicCAS(&S.head[q], h, h + k);
      if (old == h) { got = k; b = h; break; }
      h = old;
    }
  }
  base = (uint32_t)__builtin_amdgcn_readfirstlane((int)b);
  return (uint32_t)__builtin_amdgcn_readfirstlane((int)got);
}
TDT_DEV uint32_t wf_take(WfShared &S, int q, uint32_t ticket) {
  const uint32_t want = 0x8000u | (ticket & 0x7FFFu);
  uint32_t e;
  do { e = __hip_atomic_load(&S.ring[q][ticket % (uint32_t)kWfContexts], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP); } while ((e >> 16) != want);   // (its producer is mid-write in another wave)
#ifdef TDT_WF_DEBUG
  { const uint32_t old = atomicExch(&S.own[e & 0xFFFFu], 2u); if (old != 1u) printf("WF: take of context %u from queue %d while its state is %u (block %u ticket %u)\n", e & 0xFFFFu, q, old, blockIdx.x, ticket); }
#endif
  return e & 0xFFFFu;
}
TDT_DEV uint32_t wf_avail(WfShared &S, int q) {
  return __hip_atomic_load(&S.tail[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) - __hip_atomic_load(&S.head[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

template <int DEPTH, bool UNIT, bool PROBE>
__global__ __launch_bounds__(TDT_BLOCK) void trace_wf_kernel(const TraceParams P) {
  __shared__ WfShared S;
  __shared__ __attribute__((aligned(16))) uint16_t s_nodes[16];       // no LDS node table: the few steps that walk read nodes through L2
  if (threadIdx.x < 16) s_nodes[threadIdx.x] = (uint16_t)kPackedEscape;
  for (uint32_t i = threadIdx.x; i < 1024u; i += (uint32_t)TDT_BLOCK)
    S.mtype[i] = (uint8_t)((3u * i + 2u < P.materials_dwords && P.materials[3u * i] <= 2u) ? P.materials[3u * i] : 3u);
  if (threadIdx.x < 8) { S.head[threadIdx.x] = 0u; S.tail[threadIdx.x] = 0u; }
  for (uint32_t i = threadIdx.x; i < (uint32_t)(WQ_COUNT * kWfContexts); i += (uint32_t)TDT_BLOCK) (&S.ring[0][0])[i] = 0u;   // (LDS keeps the last launch's entries, tickets and all)
  if (threadIdx.x == 0) S.live = 0u;
#ifdef TDT_WF_ZERO
  for (uint32_t i = threadIdx.x + (uint32_t)(TDT_WF_ZERO * kWfContexts); i < (uint32_t)(TDT_WF_ZEND * kWfContexts); i += (uint32_t)TDT_BLOCK) reinterpret_cast<uint32_t *>(&S.ox[0])[i] = 0u;
#endif
#ifdef TDT_WF_DEBUG
  for (uint32_t i = threadIdx.x; i < (uint32_t)kWfContexts; i += (uint32_t)TDT_BLOCK) S.own[i] = 2u;     // the start-up code holds them
#endif
  __syncthreads();
  NodeSource ns;
  ns.lds = s_nodes; ns.lds_nodes = 0u; ns.lds_cells = 0u;
  ns.grid = nullptr; ns.grid_ok = false; ns.grid_band = Grid<5>::kBand;
  ns.full = P.full_grid;
  ns.cells = __builtin_amdgcn_make_buffer_rsrc((void *)P.cells, 0, (int)((P.cells_dwords >> 1) << 3), 0x00020000);
  const uint32_t total_slots = (uint32_t)P.owned_tiles * 1024u;
  const float inf = __builtin_inff();
  const int s_end = P.spp_begin + P.spp_count;
  float *gpc = P.wf_pc + (size_t)blockIdx.x * kWfPcWords * kWfContexts;     // [word][context] of this block
  Counters cnt = {};
  NodeMemo<kMemoLevels> memo;
#pragma unroll
  for (int l = 0; l < kMemoLevels; l++) { memo.key[l] = 0x3FFFFFFFu; memo.val[l] = 0u; }

  // ---- OctreeHit's prologue for a new ray of context c (rc:399-408), and the context is ready to be traversed
  auto new_ray = [&](uint32_t c, const Ray &r) {
    float ix, iy, iz;
    q_rcp3(r.dx, r.dy, r.dz, ix, iy, iz);
    const float lx = (P.min_x + -r.ox) * ix, ly = (P.min_y + -r.oy) * iy, lz = (P.min_z + -r.oz) * iz;
    const float ux = ((P.min_x + P.scale) + -r.ox) * ix, uy = ((P.min_y + P.scale) + -r.oy) * iy, uz = ((P.min_z + P.scale) + -r.oz) * iz;
    const float mnx = hw_min(lx, ux), mny = hw_min(ly, uy), mnz = hw_min(lz, uz);
    const float mxx = hw_max(lx, ux), mxy = hw_max(ly, uy), mxz = hw_max(lz, uz);
    const float t_enter = hw_max(hw_max(hw_max(mnx, 0.0003f), mny), mnz);
    const float t_exit = hw_min(hw_min(hw_min(mxx, inf), mxy), mxz);
    float tmax = inf, ts;
    if (t_exit >= t_enter) {
      HitTmp h;
      cube_hit_record(r, t_enter, P.min_x, P.min_y, P.min_z, P.scale, h);
      gpc[0 * kWfContexts + c] = h.nx; gpc[1 * kWfContexts + c] = h.ny; gpc[2 * kWfContexts + c] = h.nz;
      gpc[3 * kWfContexts + c] = h.px; gpc[4 * kWfContexts + c] = h.py; gpc[5 * kWfContexts + c] = h.pz;
      gpc[6 * kWfContexts + c] = h.ff ? 1.f : 0.f; gpc[7 * kWfContexts + c] = t_enter;
      tmax = t_exit; ts = t_enter;
    } else {
      ts = wf_gld(&gpc[7 * kWfContexts + c]);                  // the root call site's old t (rc: uninitialised out parameter keeps its value)
    }
    S.ox[c] = r.ox; S.oy[c] = r.oy; S.oz[c] = r.oz; S.dx[c] = r.dx; S.dy[c] = r.dy; S.dz[c] = r.dz;
    S.ts[c] = ts; S.tmax[c] = tmax;
  };

  // ---- sample end / pixel end / next pixel / next primary ray for up to 64 contexts (all lanes of the wave may be live)
  // fresh: the contexts hold nothing yet (kernel start)
  auto service_end = [&](bool have, uint32_t c, bool fresh) {
    bool need_pixel = have && fresh, alive = have && !fresh;
    uint32_t slot = 0, sl = 0;
    float sr = 0.f, sg = 0.f, sb = 0.f;
    if (alive) {
      slot = S.slot[c]; sl = S.sl[c];
      const uint32_t loop_count = sl >> 16;
      float cr, cg, cb;
      if (loop_count > 0u) { cr = S.ar[c]; cg = S.ag[c]; cb = S.ab[c]; }      // rc:297-301
      else {
        const float yp = S.dy[c] + 1.0f;
        const float w = 1.0f + -(0.5f * yp);
        cr = w + 0.25f * yp; cg = w + 0.35f * yp; cb = 1.0f;
      }
      sr = S.sr[c] + cr; sg = S.sg[c] + cg; sb = S.sb[c] + cb;
      sl = (sl & 0xFFFFu) + 1u;                                                 // s++, loop_count = 0
      if ((int)sl >= s_end) {                                                   // the pixel is done: main()'s last lines rc:249-251
        int x, y; size_t pix; bool inside;
        decode_pixel(P, (int)(slot >> 10), slot & 1023u, x, y, pix, inside);
        float4 *dst = reinterpret_cast<float4 *>(P.image) + pix;
        if (P.accumulate) {
          *dst = make_float4(sr, sg, sb, 0.f);
          if (P.carry && !P.carry_final) {
            float4 *cc = reinterpret_cast<float4 *>(P.carry) + pix * 4;
            cc[0] = make_float4(wf_gld(&gpc[0 * kWfContexts + c]), wf_gld(&gpc[1 * kWfContexts + c]), wf_gld(&gpc[2 * kWfContexts + c]), wf_gld(&gpc[3 * kWfContexts + c]));
            cc[1] = make_float4(wf_gld(&gpc[4 * kWfContexts + c]), wf_gld(&gpc[5 * kWfContexts + c]), wf_gld(&gpc[6 * kWfContexts + c]), wf_gld(&gpc[7 * kWfContexts + c]));
            cc[2] = make_float4(wf_gld(&gpc[8 * kWfContexts + c]), wf_gld(&gpc[9 * kWfContexts + c]), wf_gld(&gpc[10 * kWfContexts + c]), wf_gld(&gpc[11 * kWfContexts + c]));
            cc[3] = make_float4(wf_gld(&gpc[12 * kWfContexts + c]), wf_gld(&gpc[13 * kWfContexts + c]), wf_gld(&gpc[14 * kWfContexts + c]), 0.f);
          }
        } else {
          const float n = (float)P.samples_per_pixel;
          float4 o;
          o.x = f_min(f_max(__builtin_sqrtf(sr / n), 0.f), 1.f);
          o.y = f_min(f_max(__builtin_sqrtf(sg / n), 0.f), 1.f);
          o.z = f_min(f_max(__builtin_sqrtf(sb / n), 0.f), 1.f);
          o.w = 1.0f;
          *dst = o;
        }
        if (P.slot_cost) P.slot_cost[slot] = (S.work[c] + kCostEvent) | 1u;
        need_pixel = true; alive = false;
      }
    }
    // next pixel from the global queue (as trace_kernel draws them: one atomic per wave and refill)
    unsigned long long m = __ballot(need_pixel);
    bool retired = false, got_new = false;
    while (m != 0ull) {
      const uint32_t chunk = (uint32_t)__popcll(m);   // exactly as many slots as lanes ask for (a context is not tied to a lane: no batches to keep)
      uint32_t base = 0;
      if (wf_lane() == 0u) base = atomicAdd(P.queue, chunk);
      base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
      const uint32_t bq = base + wf_lane();
      const uint32_t mine = (wf_lane() < chunk && bq < total_slots) ? (P.slot_order ? P.slot_order[bq] : bq) : 0xFFFFFFFFu;
      const uint32_t rank = (uint32_t)__popcll(m & ((1ull << wf_lane()) - 1ull));
      const uint32_t q = (uint32_t)__shfl((int)mine, (int)(rank & 63u), 64);
      if (need_pixel && rank < chunk) {
        if (q == 0xFFFFFFFFu) { need_pixel = false; retired = true; }
        else {
          int x, y; size_t pix; bool inside;
          decode_pixel(P, (int)(q >> 10), q & 1023u, x, y, pix, inside);
          if (inside) {                               // outside the covered image: ask again
            slot = q; sl = (uint32_t)P.spp_begin; sr = 0.f; sg = 0.f; sb = 0.f;
            for (int k = 0; k < 15; k++) gpc[k * kWfContexts + c] = 0.f;
            if (P.accumulate && P.spp_begin != 0) {   // (a range that starts at sample 0 starts from nothing)
              const float4 acc = *(reinterpret_cast<const float4 *>(P.image) + pix);
              sr = acc.x; sg = acc.y; sb = acc.z;
              if (P.carry) {
                const float4 *cc = reinterpret_cast<const float4 *>(P.carry) + pix * 4;
                const float4 c0 = cc[0], c1 = cc[1], c2 = cc[2], c3 = cc[3];
                gpc[0 * kWfContexts + c] = c0.x; gpc[1 * kWfContexts + c] = c0.y; gpc[2 * kWfContexts + c] = c0.z; gpc[3 * kWfContexts + c] = c0.w;
                gpc[4 * kWfContexts + c] = c1.x; gpc[5 * kWfContexts + c] = c1.y; gpc[6 * kWfContexts + c] = c1.z != 0.f ? 1.f : 0.f; gpc[7 * kWfContexts + c] = c1.w;
                gpc[8 * kWfContexts + c] = c2.x; gpc[9 * kWfContexts + c] = c2.y; gpc[10 * kWfContexts + c] = c2.z; gpc[11 * kWfContexts + c] = c2.w;
                gpc[12 * kWfContexts + c] = c3.x; gpc[13 * kWfContexts + c] = c3.y; gpc[14 * kWfContexts + c] = c3.z != 0.f ? 1.f : 0.f;
              }
            }
            S.work[c] = 0u;
            if ((int)sl < s_end) { need_pixel = false; alive = true; got_new = true; }
            else if (!P.accumulate) {                 // zero samples: main() still stores sqrt(0/0) clamped; then the next pixel
              const float n = (float)P.samples_per_pixel;
              const float v0 = f_min(f_max(__builtin_sqrtf(0.f / n), 0.f), 1.f);
              *(reinterpret_cast<float4 *>(P.image) + pix) = make_float4(v0, v0, v0, 1.0f);
            }
          }
        }
      }
      m = __ballot(need_pixel);
    }
    // contexts that hold a pixel: fresh ones count in, retired ones (queue dry) count out
    {
      const unsigned long long m_in = __ballot(got_new && fresh), m_out = __ballot(retired && !fresh);
      if (wf_lane() == 0u) {
        if (m_in) atomicAdd(&S.live, (uint32_t)__popcll(m_in));
        if (m_out) atomicSub(&S.live, (uint32_t)__popcll(m_out));
      }
    }
    // next primary ray rc:240-245 + the bounce loop's first condition rc:271
    bool to_trav = false, to_end = false;
    if (alive) {
      int x, y; size_t pix; bool inside;
      decode_pixel(P, (int)(slot >> 10), slot & 1023u, x, y, pix, inside);
      const Ray r = primary_ray(P, x, y, (int)(sl & 0xFFFFu));
      S.slot[c] = slot; S.sl[c] = sl & 0xFFFFu; S.sr[c] = sr; S.sg[c] = sg; S.sb[c] = sb;
      S.ar[c] = 1.f; S.ag[c] = 1.f; S.ab[c] = 1.f;
      if (0 < P.max_bounce) { new_ray(c, r); to_trav = true; }
      else { S.dx[c] = r.dx; S.dy[c] = r.dy; S.dz[c] = r.dz; to_end = true; }      // no bounce allowed: the path ends on the sky colour
    }
    wf_push(S, WQ_TRAV, to_trav, c);
    wf_push(S, WQ_END, to_end, c);
  };

  // ---- RayColor's loop body rc:272-295 for up to 64 hits (of ONE material, as the queues are filled), then the next ray
  auto service_hit = [&](bool have, uint32_t c) {
    bool to_trav = false, to_end = false;
    if (have) {
      const uint32_t hw = S.hit[c], hit_index = hw >> 2;
      const bool use_leaf = (hw & 2u) != 0u, leaf_rec = (hw & 1u) != 0u;
      const MatRef mat = material_fetch(P, hit_index);
      Ray r = {S.ox[c], S.oy[c], S.oz[c], S.dx[c], S.dy[c], S.dz[c]};
      HitTmp src;
      if (leaf_rec) {
        cube_hit_record(r, S.ht[c], S.hbx[c], S.hby[c], S.hbz[c], S.hsz[c], src);
        gpc[8 * kWfContexts + c] = src.nx; gpc[9 * kWfContexts + c] = src.ny; gpc[10 * kWfContexts + c] = src.nz;
        gpc[11 * kWfContexts + c] = src.px; gpc[12 * kWfContexts + c] = src.py; gpc[13 * kWfContexts + c] = src.pz;
        gpc[14 * kWfContexts + c] = src.ff ? 1.f : 0.f;
      } else {
        const int o = use_leaf ? 8 : 0;               // an old record: the leaf call site's, or (a leaf at the first step) the root's
        src.nx = wf_gld(&gpc[(o + 0) * kWfContexts + c]); src.ny = wf_gld(&gpc[(o + 1) * kWfContexts + c]); src.nz = wf_gld(&gpc[(o + 2) * kWfContexts + c]);
        src.px = wf_gld(&gpc[(o + 3) * kWfContexts + c]); src.py = wf_gld(&gpc[(o + 4) * kWfContexts + c]); src.pz = wf_gld(&gpc[(o + 5) * kWfContexts + c]);
        src.ff = wf_gld(&gpc[(o + 6) * kWfContexts + c]) != 0.f;
      }
      uint32_t sl = S.sl[c] + 0x10000u;               // loop_count += 1
      Hit h;
      h.px = src.px; h.py = src.py; h.pz = src.pz; h.nx = src.nx; h.ny = src.ny; h.nz = src.nz; h.ff = src.ff;
      h.index = hit_index;
      Ray nr; float tr, tg, tb;
      S.work[c] += kCostEvent;
      if (scatter<false>(P, r, h, mat, nr, tr, tg, tb, cnt)) {
        S.ar[c] = S.ar[c] * tr; S.ag[c] = S.ag[c] * tg; S.ab[c] = S.ab[c] * tb;
        if ((int)(sl >> 16) < P.max_bounce) { new_ray(c, nr); to_trav = true; }
        else { S.dy[c] = nr.dy; to_end = true; }
      } else {
        to_end = true;
      }
      S.sl[c] = sl;
    }
    wf_push(S, WQ_TRAV, to_trav, c);
    wf_push(S, WQ_END, to_end, c);
  };

  // ---- start: every context asks for a pixel
  {
    const uint32_t wave = threadIdx.x >> 6;
    for (uint32_t b = wave * 64u; b < (uint32_t)kWfContexts; b += (uint32_t)TDT_BLOCK) {
      const uint32_t c = b + wf_lane();
      service_end(c < (uint32_t)kWfContexts, c, true);
    }
  }
  __syncthreads();

  if (P.event_threshold == 777 && threadIdx.x >= 64) return;      // DEBUG: one worker wave per block
  // ---- the worker loop
  int cid = -1;                                       // the context this lane is traversing
  int fin = 0;                                        // 1: its walk hit a leaf, 2: it left the octree / ran out of steps
  Ray r = {0.f, 0.f, 0.f, 0.f, 0.f, 1.f};
  float ix = 0.f, iy = 0.f, iz = 0.f, t_stride = 0.f, t_octree_max = 0.f, inv_pow_depth = 0.5f;
  float leaf_box_x = 0.f, leaf_box_y = 0.f, leaf_box_z = 0.f;
  int it = 0; uint32_t lane_work = 0u, hit_word = 0u;
  uint32_t idle_spins = 0u;
  for (;;) {
    // -------------------------------------------------------- one traversal step rc:410-447 (as trace_kernel's)
    {
      const bool trav = cid >= 0 && fin == 0;
      const bool go = trav && (it < P.max_iter) && (t_stride < t_octree_max);
      const float adv = f_max(0.0001f * (inv_pow_depth + 0.1f), 0.000001f);
      const float tt = t_stride + adv;
      const float wx = tt * r.dx + r.ox, wy = tt * r.dy + r.oy, wz = tt * r.dz + r.oz;
      const float lx = UNIT ? (wx + -P.min_x) : (wx + -P.min_x) * P.inv_scale, ly = UNIT ? (wy + -P.min_y) : (wy + -P.min_y) * P.inv_scale,
                  lz = UNIT ? (wz + -P.min_z) : (wz + -P.min_z) * P.inv_scale;
      const uint32_t ux = __float_as_uint(lx + 0.0f), uy = __float_as_uint(ly + 0.0f), uz = __float_as_uint(lz + 0.0f);
      const uint32_t um = ux > uy ? ux : uy;
      const bool in_box = (um > uz ? um : uz) < 0x3F800000u;
      const bool inside = go && in_box;
      if (inside) {
        float ugx, ugy, ugz; uint32_t value;
        const bool leaf = tree_lookup_pow2<false, kMemoLevels, DEPTH, false, true, true>(P, ns, lx, ly, lz, inv_pow_depth, ugx, ugy, ugz, value, memo, cnt);
        lane_work += kCostStep + (127u - (__float_as_uint(inv_pow_depth) >> 23));
        const float bx = (UNIT ? ugx : ugx * P.scale) + P.min_x, by = (UNIT ? ugy : ugy * P.scale) + P.min_y, bz = (UNIT ? ugz : ugz * P.scale) + P.min_z;
        const float cs0 = UNIT ? inv_pow_depth : P.scale * inv_pow_depth;
        const float pad = leaf ? -0.0f : -0.00001f;
        const float cx = bx + pad, cy = by + pad, cz = bz + pad;
        const float cs = leaf ? cs0 : cs0 + 0.00002f;
        float t_enter, t_exit;
        cube_slabs(r, ix, iy, iz, cx, cy, cz, cs, t_stride, t_octree_max, t_enter, t_exit);
        const bool cube_ok = !(t_exit < t_enter);
        if (leaf) {
          leaf_box_x = cx; leaf_box_y = cy; leaf_box_z = cz; inv_pow_depth = cs; t_stride = t_enter;
          hit_word = (value << 2) | (it > 0 ? 2u : 0u) | ((it > 0 && cube_ok) ? 1u : 0u);
          fin = 1;
        } else {
          t_stride = cube_ok ? t_exit : t_octree_max;
          it++;
        }
      }
      if (trav && !inside) fin = 2;
    }
    // -------------------------------------------------------- hand finished walks on, take new ones
    const unsigned long long m_fin = __ballot(fin != 0), m_act = __ballot(cid >= 0 && fin == 0);
    const int n_fin = __popcll(m_fin), n_act = __popcll(m_act);
    if (n_fin >= 8 || (n_fin > 0 && n_act == 0)) {
      const uint32_t c = (uint32_t)cid;
      uint32_t type = 3u;
      if (fin == 1) {
        S.hit[c] = hit_word; S.hbx[c] = leaf_box_x; S.hby[c] = leaf_box_y; S.hbz[c] = leaf_box_z; S.hsz[c] = inv_pow_depth; S.ht[c] = t_stride;
        const uint32_t hi = hit_word >> 2;
        type = hi < 1024u ? (uint32_t)S.mtype[hi] : 3u;
      }
      if (fin != 0) S.work[c] += lane_work;
      wf_push(S, WQ_MAT0, fin == 1 && (type == 0u || type == 3u), c);      // (an unknown material type ends the path in the hit code)
      wf_push(S, WQ_MAT1, fin == 1 && type == 1u, c);
      wf_push(S, WQ_MAT2, fin == 1 && type == 2u, c);
      wf_push(S, WQ_END, fin == 2, c);
      if (fin != 0) { cid = -1; fin = 0; }
    }
    {
      const unsigned long long m_idle = __ballot(cid < 0);
      const int n_idle = __popcll(m_idle);
      if (n_idle >= 8 && wf_avail(S, WQ_TRAV) != 0u) {
        uint32_t base;
        const uint32_t got = wf_claim(S, WQ_TRAV, (uint32_t)n_idle, base);
        const uint32_t rank = (uint32_t)__popcll(m_idle & ((1ull << wf_lane()) - 1ull));
        if (cid < 0 && rank < got) {
          const uint32_t c = wf_take(S, WQ_TRAV, base + rank);
          cid = (int)c; fin = 0;
          r = {S.ox[c], S.oy[c], S.oz[c], S.dx[c], S.dy[c], S.dz[c]};
          q_rcp3(r.dx, r.dy, r.dz, ix, iy, iz);
          t_stride = S.ts[c]; t_octree_max = S.tmax[c]; inv_pow_depth = 0.5f; it = 0; lane_work = 0u;
        }
      }
    }
    // -------------------------------------------------------- service: a wave's worth of one kind of event code
    {
      const int n_busy = __popcll(__ballot(cid >= 0));
      int q = -1;
      for (int k = WQ_MAT0; k <= WQ_END; k++) {
        const uint32_t a = wf_avail(S, k);
        if (a >= 64u || (a != 0u && n_busy == 0)) { q = k; if (a >= 64u) break; }
      }
      q = __builtin_amdgcn_readfirstlane(q);
      if (q >= 0) {
        uint32_t base;
        const uint32_t got = wf_claim(S, q, 64u, base);
        if (got != 0u) {
          const bool have = wf_lane() < got;
          const uint32_t c = have ? wf_take(S, q, base + wf_lane()) : 0u;
          if (q == WQ_END) service_end(have, c, false);
          else service_hit(have, c);
          idle_spins = 0u;
        }
      } else if (n_busy == 0) {
        // nothing to walk, nothing to serve: done when no context of the block holds a pixel any more
        if (__hip_atomic_load(&S.live, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0u && wf_avail(S, WQ_TRAV) == 0u) break;
        __builtin_amdgcn_s_sleep(8);
        idle_spins++;
      }
    }
  }
  (void)idle_spins;
}
