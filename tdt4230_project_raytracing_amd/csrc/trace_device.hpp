// Device-side arithmetic of the voxel path trace for gfx950.
//
// Every function states the reference lines it implements (rc:N = /root/reference/
// assets/shaders/raytracer.comp line N).  The integrand is chaotic (each hash feeds on the
// bits of the previous hit point), so the arithmetic keeps the exact fp32 operation order of
// the reference as it is compiled for its only runnable target (Mesa/llvmpipe): unfused
// mul+add (build with -ffp-contract=off), IEEE divide / sqrt, z-y-x dot accumulation, and the
// polynomial sin/cos/pow of that implementation (the only places that fuse).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "trace_params.h"

namespace tdt {

#define TDT_DEV __device__ __forceinline__
// -DTDT_MARKERS (tools/isa_regions.py; never the product library): assembler comments at the boundaries of the trace kernel's code regions,
// so that the compiler's output can be cut into traversal step / gate / scatter branches / end of path / fetch / primary ray / new ray
#ifdef TDT_MARKERS
#define TDT_MARK(name) asm volatile("; TDT_MARK " #name)
#else
#define TDT_MARK(name) do {} while (0)
#endif

// how a kernel build computes treeLookup's x index (rc:376-378): the float formula as written; the exact-comparison form for
// cell_count = 2^k (tree_lookup_pow2); the same walk with per-cell thresholds for any other cell_count (XThreshold)
enum : int { FORM_LITERAL = 0, FORM_POW2 = 1, FORM_TABLE = 2 };

TDT_DEV float f_fract(float x) { return x - __builtin_floorf(x); }
// for x >= 0 the v_fract_f32 instruction returns exactly x - floor(x) (checked on all inputs by
// tdt_selftest mode 4); negative arguments keep the two-instruction form (they differ: fract(-tiny) = 1.0)
TDT_DEV float f_fract_nonneg(float x) { return __builtin_amdgcn_fractf(x); }
TDT_DEV float f_rcp(float x) { return 1.0f / x; }                          // IEEE-rounded
TDT_DEV float f_rsq(float x) { return 1.0f / __builtin_sqrtf(x); }          // two roundings, as the reference

// ---- short correctly-rounded forms -------------------------------------------------------------
// hipcc's IEEE divide / sqrt expansions carry operand scaling for the extreme exponents
// (v_div_scale / v_div_fmas / v_div_fixup, denormal rescue): ~11 and ~15 instructions.  Inside
// a safe exponent window a hardware seed + fused Newton steps give the SAME correctly rounded
// result in 3 / 5 instructions.  "Same" is not argued but checked exhaustively: tdt_selftest()
// compares these with the IEEE expressions on all 2^32 inputs (tests/test_gpu_api.py); inputs
// outside the window take the IEEE expression (wave-uniform branch).
TDT_DEV bool exp_in_window(float x) {      // biased exponent in [27, 228): |x| in [2^-100, 2^101)
  return ((__float_as_uint(x) & 0x7FFFFFFFu) - 0x0D800000u) < 0x65000000u;
}
TDT_DEV bool pos_in_window(float x) {      // x > 0 and in the window, in one unsigned compare (negative floats wrap past it)
  return (__float_as_uint(x) - 0x0D800000u) < 0x65000000u;
}
TDT_DEV float rcp_core(float y) {          // RN(1/y) for y in the window
  const float r = __builtin_amdgcn_rcpf(y);
  const float e = __builtin_fmaf(-y, r, 1.0f);
  return __builtin_fmaf(e, r, r);
}
TDT_DEV float sqrt_core(float x) {         // RN(sqrt(x)) for positive x in the window
  const float s = __builtin_amdgcn_sqrtf(x), h = 0.5f * __builtin_amdgcn_rsqf(x);
  return __builtin_fmaf(__builtin_fmaf(-s, s, x), h, s);     // seed + one step on the exact residual
}
TDT_DEV float q_rcp(float y) {
  if (__builtin_expect(__ballot(!exp_in_window(y)) != 0ull, 0)) return 1.0f / y;
  return rcp_core(y);
}
TDT_DEV void q_rcp3(float a, float b, float c, float &ra, float &rb, float &rc) {   // one guard for a vec3
  if (__builtin_expect(__ballot(!(exp_in_window(a) && exp_in_window(b) && exp_in_window(c))) != 0ull, 0)) {
    ra = 1.0f / a; rb = 1.0f / b; rc = 1.0f / c;
  } else { ra = rcp_core(a); rb = rcp_core(b); rc = rcp_core(c); }
}
TDT_DEV float q_sqrt(float x) {
  if (__builtin_expect(__ballot(!pos_in_window(x)) != 0ull, 0)) return __builtin_sqrtf(x);
  return sqrt_core(x);
}
TDT_DEV float q_rsq(float x) {             // RN(1 / RN(sqrt(x))): sqrt of a window value stays in the window
  if (__builtin_expect(__ballot(!pos_in_window(x)) != 0ull, 0)) return 1.0f / __builtin_sqrtf(x);
  return rcp_core(sqrt_core(x));
}
// a / b through the correctly rounded reciprocal and one residual step.  NOT claimed for all operand pairs: used at two call sites
// whose operands are functions of ONE float, and checked there for every bit pattern of it (tdt_selftest mode 11).
TDT_DEV float div_core(float a, float b) {
  const float r = rcp_core(b), q0 = a * r;
  return __builtin_fmaf(__builtin_fmaf(-b, q0, a), r, q0);
}
// (m - 1) / (m + 1) for a mantissa m in [1, 2): pow_poly's log2 argument (operands always in the window)
TDT_DEV float pow_t(float m) {
#ifdef TDT_LITERAL_NORMAL
  return (m - 1.0f) / (m + 1.0f);
#else
  return div_core(m - 1.0f, m + 1.0f);
#endif
}
// (1 - x) / (1 + x): reflectance's r0 before squaring (rc:495), x = the refraction ratio
TDT_DEV float schlick_q(float x) {
  const float a = 1.0f + -x, b = 1.0f + x;
#ifndef TDT_LITERAL_NORMAL
  if (__builtin_expect(__ballot(!(exp_in_window(a) && exp_in_window(b))) == 0ull, 1)) return div_core(a, b);
#endif
  return a / b;
}
// min/max where a NaN operand yields the other operand (and ties return b)
TDT_DEV float f_min(float a, float b) { return (b != b) ? a : (a < b ? a : b); }
TDT_DEV float f_max(float a, float b) { return (b != b) ? a : (a > b ? a : b); }
// v_min_f32 / v_max_f32 (IEEE minNum / maxNum): identical to f_min / f_max except for which
// zero a (+0,-0) tie returns — which no consumer below can observe (the slab chains only
// compare these values, or use them when they are strictly positive)
TDT_DEV float hw_min(float a, float b) { return __builtin_fminf(a, b); }
TDT_DEV float hw_max(float a, float b) { return __builtin_fmaxf(a, b); }
TDT_DEV float b2f(bool b) { return b ? 1.0f : 0.0f; }
TDT_DEV int32_t f2i(float f) {   // truncating convert; out of range / NaN -> INT_MIN
  return (f > -2147483904.0f && f < 2147483648.0f) ? (int32_t)f : (int32_t)0x80000000;
}

// sin/cos: octant reduction + two short polynomials with fused steps (the form the reference
// target evaluates; needed bit-for-bit because fract(sin(x)*43758.5453) is the RNG, rc:53)
TDT_DEV void sincos_poly(float a, float &s_out, float &c_out) {
  float x = __builtin_fabsf(a);
  int32_t j = f2i(x * 1.27323954473516f);
  j = (j + 1) & ~1;
  float y = (float)j;
  x = __builtin_fmaf(y, -0.78515625f, x);
  x = __builtin_fmaf(y, -2.4187564849853515625e-4f, x);
  x = __builtin_fmaf(y, -3.77489497744594108e-8f, x);
  float z = x * x;
  float c = 2.443315711809948e-5f;
  c = __builtin_fmaf(c, z, -1.388731625493765e-3f);
  c = __builtin_fmaf(c, z, 4.166664568298827e-2f);
  c = c * z;
  c = c * z;
  c = __builtin_fmaf(-0.5f, z, c);
  c = c + 1.0f;
  float s = -1.9515295891e-4f;
  s = __builtin_fmaf(s, z, 8.3321608736e-3f);
  s = __builtin_fmaf(s, z, -1.6666654611e-1f);
  s = s * z;
  s = __builtin_fmaf(s, x, x);
  float rs = ((j & 2) == 0) ? s : c;
  bool neg_s = (((j & 4) != 0) != ((__float_as_uint(a) >> 31) != 0));
  s_out = neg_s ? -rs : rs;
  int32_t k = j - 2;
  float rc = ((k & 2) == 0) ? s : c;
  c_out = (((~k) & 4) != 0) ? -rc : rc;
}
TDT_DEV float sin_poly(float a) { float s, c; sincos_poly(a, s, c); return s; }

// pow(x,y) = exp2(log2(x)*y) with the reference target's polynomials (only use: rc:496)
TDT_DEV float pow_poly(float x, float yy) {
  uint32_t u = __float_as_uint(x);
  float e = (float)((int32_t)((u >> 23) & 0xff) - 127);
  float m = __uint_as_float((u & 0x007fffffu) | 0x3f800000u);
  float t = pow_t(m);
  float z = t * t;
  float z2 = z * z;
  float even = __builtin_fmaf(z2, __builtin_fmaf(z2, 0.406718052498846252698f, 0.577440339438736392009f), 2.88539009343309178325f);
  float odd = __builtin_fmaf(z2, 0.403343858251329912514f, 0.961791550404184197881f);
  float p = __builtin_fmaf(odd, z, even);
  float l2 = __builtin_fmaf(t, p, e);
  float w = l2 * yy;
  w = f_min(f_max(w, -126.99999f), 128.0f);
  float i = __builtin_floorf(w);
  float f = w - i;
  float f2 = f * f;
  float ev = __builtin_fmaf(f2, __builtin_fmaf(f2, 0.00898934009049466391101f, 0.240153617044375388211f), 1.0f);
  float od = __builtin_fmaf(f2, __builtin_fmaf(f2, 0.00187757667519147912699f, 0.0558263180532956664775f), 0.693153073200168932794f);
  float q = __builtin_fmaf(od, f, ev);
  int32_t ii = f2i(i);
  return __uint_as_float((uint32_t)(ii + 127) << 23) * q;
}

// SSBO read with robust-access semantics: dword index outside the buffer reads 0
TDT_DEV uint32_t ld_dw(const uint32_t *buf, uint32_t dwords, uint32_t byte_off) {
  uint32_t i = byte_off >> 2;
  return (i < dwords) ? buf[i] : 0u;
}

// ---- octree node fetch ----------------------------------------------------------------------
// The first kLdsCells cells of the breadth-first cell array (= the top levels of the tree, or the
// whole tree for a 64^3 scene) are staged in LDS by every block, 16 bits per node
// (value << 2 | code; code 0 EMPTY, 1 PARENT / any other type, 2 LEAF; 0xFFFF = value does not
// fit in 14 bits, read the original).  Everything else comes from the linearised octree in HBM / L2 through
// a raw buffer descriptor whose range check IS the reference's robust-access rule (reads past the
// end return 0), so there is no bounds branch.
#ifndef TDT_LDS_CELLS
#define TDT_LDS_CELLS 5120
#endif
constexpr uint32_t kLdsCells = TDT_LDS_CELLS;        // 5120 cells * 8 nodes * 2 B = 81,920 B of the 160 KiB LDS
constexpr uint32_t kPackedEscape = 0xFFFFu;
constexpr uint32_t kPackedMaxValue = 0x3FFEu;        // largest value an LDS entry can hold

struct NodeSource {
  const uint16_t *lds;                                // LDS table
  uint32_t lds_nodes;                                 // valid entries
  uint32_t lds_cells;                                 // cells they make up (the last may be partial): index of the sentinel cell
  const void *grid;                                   // top-level jump table (Grid<GL>::Entry[], see build_top_grid), or unusable when !grid_ok
  bool grid_ok; float grid_band;                      // grid_band = kGridBand, or 2 when the table is unusable
  const uint16_t *full;                               // FULL builds: the whole-depth table in global memory (see tree_lookup_pow2)
  const uint32_t *grid32; const void *bricks;         // BRICK builds: the 5-level table with brick headers (LDS) and the bricks (global memory, 16-bit entries)
  const float2 *thr; float thr_f0max;                 // FORM_TABLE builds: (F1, F2) per cell (LDS) and the scene-wide bound on F0, see x_thresholds
  __amdgpu_buffer_rsrc_t cells;                       // raw buffer over the cells payload (8-byte granules)
};

// returns the node's value; code = 0 EMPTY, 2 LEAF, 1 otherwise (rc:384-386 only distinguishes these)
TDT_DEV uint32_t fetch_node(const NodeSource &ns, uint32_t idx, uint32_t &code) {
  idx &= 0x1FFFFFFFu;                                 // byte offset idx << 3 wraps at 32 bits (rc:184 on a 32-bit offset)
  {
    const uint32_t n = ns.lds[idx < ns.lds_nodes ? idx : ns.lds_nodes];     // sentinel slot: see tree_lookup_pow2
    if (n != kPackedEscape) { code = n & 3u; return n >> 2; }
  }
  const auto n2 = __builtin_amdgcn_raw_buffer_load_b64(ns.cells, (int)(idx << 3), 0, 0);
  const uint32_t node_type = (uint32_t)n2[1];
  code = (node_type == 0u) ? 0u : (node_type == 2u ? 2u : 1u);
  return (uint32_t)n2[0];
}

// event counts of an instrumented launch (defines the algorithmic bytes)
struct Counters {
  uint32_t octree_hit_calls, iterations, node_loads, lambertian, metal, dielectric, unknown;
  // lane-utilisation diagnostics (instrumented builds only): *_slots counts 64 per wave-level
  // execution of a code region, *_active the lanes that were live in it
  uint32_t trav_slots, trav_active, level_slots, level_active, event_slots, event_active,
           scatter_slots, scatter_active, memo_miss, leaf_records;
};
// 64 for the first active lane of the wave, 0 for the others
TDT_DEV uint32_t slot64() {
  const unsigned long long m = __ballot(1);
  return ((threadIdx.x & 63) == (uint32_t)__builtin_ctzll(m)) ? 64u : 0u;
}

struct Ray { float ox, oy, oz, dx, dy, dz; };
// what one CubeHit call site last produced (rc:336-354); the root and leaf call sites keep
// theirs across calls, because on a miss the reference's out-parameter copy hands back the
// previous contents (see oracle/pathtrace_oracle.c)
struct HitTmp { float nx, ny, nz, px, py, pz; bool ff; };
struct Carry { HitTmp root; float root_t; HitTmp leaf; };
struct Hit { float px, py, pz, nx, ny, nz; bool ff; uint32_t index; };

// The normal and the front-face flag of CubeHit's record (rc:341-353) from q = p - centre, as the shader computes them: mask all
// but the dominant axis, normalise, orient against the ray, normalise again.
TDT_DEV void cube_normal_literal(const Ray &r, float nx, float ny, float nz, HitTmp &h) {
  float ax = __builtin_fabsf(nx), ay = __builtin_fabsf(ny), az = __builtin_fabsf(nz);
  nx = nx * b2f(ax >= f_max(ay, az));
  ny = ny * b2f(f_max(ax, az) < ay);
  nz = nz * b2f(f_max(ax, ay) < az);
  float rs = q_rsq((nz * nz + ny * ny) + nx * nx);
  nx = nx * rs; ny = ny * rs; nz = nz * rs;
  bool ff = (r.dz * nz + r.dy * ny) < -(r.dx * nx);
  float flip = -2.0f * b2f(!ff) + 1.0f;
  nx = nx * flip; ny = ny * flip; nz = nz * flip;
  rs = q_rsq((nz * nz + ny * ny) + nx * nx);
  h.nx = nx * rs; h.ny = ny * rs; h.nz = nz * rs;
  h.ff = ff;
}
// What that sequence yields when exactly one axis is selected and its |v| is far from the ends of the exponent range (the
// others are finite): sqrt(RN(v^2)) is |v| in binary floating point, g(x) = RN(x * RN(1/x)) is 1 or 1 - 2^-24, and g(g(x)) = 1
// — so the selected component ends as +-1.0 and the masked ones as zeros that keep q's sign, all flipped for a back face;
// the front-face test only looks at the sign of d_k * n_k, which +-1.0 has in common with the once-normalised value.  Two
// reciprocal square roots and 14 multiplications less, per record (two records per bounce).  Checked against the literal
// form for every bit pattern of the selected component x a grid of the other operands (tdt_selftest mode 9); anything else
// (ties between axes — an edge hit selects no axis and yields NaNs —, NaN / inf / out-of-window operands) takes the literal form.
TDT_DEV bool cube_normal_fast_ok(float nx, float ny, float nz, bool &sx, bool &sy, bool &sz) {
  const float ax = __builtin_fabsf(nx), ay = __builtin_fabsf(ny), az = __builtin_fabsf(nz);
  sx = ax >= hw_max(ay, az); sy = hw_max(ax, az) < ay; sz = hw_max(ax, ay) < az;     // (NaN operands fail the window test below)
  const float sum = (ax + ay) + az;                    // max <= sum <= 3 max; NaN and inf propagate
  return (sx || sy || sz) && (__float_as_uint(sum) - 0x27000000u) < (0x58000000u - 0x27000000u);      // 2^-49 <= sum < 2^49
}
TDT_DEV void cube_normal_fast(const Ray &r, float nx, float ny, float nz, bool sx, bool sy, bool sz, HitTmp &h) {
  const uint32_t ex = (__float_as_uint(nx) & 0x80000000u) | (sx ? 0x3F800000u : 0u);
  const uint32_t ey = (__float_as_uint(ny) & 0x80000000u) | (sy ? 0x3F800000u : 0u);
  const uint32_t ez = (__float_as_uint(nz) & 0x80000000u) | (sz ? 0x3F800000u : 0u);
  const bool ff = (r.dz * __uint_as_float(ez) + r.dy * __uint_as_float(ey)) < -(r.dx * __uint_as_float(ex));
  const uint32_t fl = ff ? 0u : 0x80000000u;
  h.nx = __uint_as_float(ex ^ fl); h.ny = __uint_as_float(ey ^ fl); h.nz = __uint_as_float(ez ^ fl);
  h.ff = ff;
}
// CubeHit's record for entry parameter t (rc:336-354)
TDT_DEV void cube_hit_record(const Ray &r, float t, float cx, float cy, float cz, float size, HitTmp &h) {
  float px = t * r.dx + r.ox, py = t * r.dy + r.oy, pz = t * r.dz + r.oz;
  float radius = size * 0.5f;
  float nx = px + -(cx + radius), ny = py + -(cy + radius), nz = pz + -(cz + radius);
  h.px = px; h.py = py; h.pz = pz;
#ifdef TDT_LITERAL_NORMAL                               // (A/B builds)
  cube_normal_literal(r, nx, ny, nz, h);
#else
  bool sx, sy, sz;
  const bool ok = cube_normal_fast_ok(nx, ny, nz, sx, sy, sz);
  if (__builtin_expect(__ballot(!ok) != 0ull, 0)) cube_normal_literal(r, nx, ny, nz, h);
  else cube_normal_fast(r, nx, ny, nz, sx, sy, sz, h);
#endif
}

// slab test rc:317-331 (t_min / t_max in the first operand slot of the min/max chain)
TDT_DEV void cube_slabs(const Ray &r, float ix, float iy, float iz, float cx, float cy, float cz, float size,
                        float t_min, float t_max, float &t_enter, float &t_exit) {
  float lx = (cx + -r.ox) * ix, ly = (cy + -r.oy) * iy, lz = (cz + -r.oz) * iz;
  float ux = ((cx + size) + -r.ox) * ix, uy = ((cy + size) + -r.oy) * iy, uz = ((cz + size) + -r.oz) * iz;
  float mnx = hw_min(lx, ux), mny = hw_min(ly, uy), mnz = hw_min(lz, uz);
  float mxx = hw_max(lx, ux), mxy = hw_max(ly, uy), mxz = hw_max(lz, uz);
  // the chain through the instructions themselves: __builtin_fmaxf canonicalises operands the compiler cannot prove quiet (t_min, t_max
  // are loop-carried: a v_max_f32 x, x each, half-rate instructions); every operand here comes out of fp32 arithmetic, which never yields a
  // signalling NaN, and v_max / v_min return the other operand for a quiet NaN exactly as fmaxf / fminf do
  float e1, e3, x1, x3;
  asm("v_max_f32 %0, %1, %2" : "=v"(e1) : "v"(t_min), "v"(mnx));
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(e3) : "v"(e1), "v"(mny), "v"(mnz));
  asm("v_min_f32 %0, %1, %2" : "=v"(x1) : "v"(t_max), "v"(mxx));
  asm("v_min3_f32 %0, %1, %2, %3" : "=v"(x3) : "v"(x1), "v"(mxy), "v"(mxz));
  t_enter = e3; t_exit = x3;      // = hw_max(hw_max(hw_max(t_min, mnx), mny), mnz), hw_min(hw_min(hw_min(t_max, mxx), mxy), mxz)
}

// treeLookup rc:359-394: one dependent 8-byte Node load per level
template <bool COUNT>
TDT_DEV bool tree_lookup(const TraceParams &P, const NodeSource &ns, float cx, float cy, float cz, float &inv_pow_depth,
                         float &gx, float &gy, float &gz, uint32_t &value, Counters &cnt) {
  float ipd = 1.0f, ux = 0.0f, uy = 0.0f, uz = 0.0f;
  uint32_t node_value = 0;
  bool is_leaf = false;
  const float two_cc = (float)(int32_t)((uint32_t)P.cell_count << 1);
  const float fdepth = (float)P.max_depth;
  for (float i = 0.0f; i < fdepth; i = i + 1.0f) {
    ipd = ipd * 0.5f;
    float fx = f_fract(cx), fy = f_fract(cy), fz = f_fract(cz);
    float rx = __builtin_rintf((((float)node_value + fx) * P.inv_cell_count) * two_cc + -0.5f);
    float ry = __builtin_rintf(fy * 2.0f + -0.5f);
    float rz = __builtin_rintf(fz * 2.0f + -0.5f);
    int32_t ix = f2i(rx), iy = f2i(ry), iz = f2i(rz);
    float tx = __builtin_truncf(rx), ty = __builtin_truncf(ry), tz = __builtin_truncf(rz);
    float bx = tx + -(2.0f * __builtin_floorf(tx / 2.0f));
    float by = ty + -(2.0f * __builtin_floorf(ty / 2.0f));
    float bz = tz + -(2.0f * __builtin_floorf(tz / 2.0f));
    ux = ux + bx * ipd; uy = uy + by * ipd; uz = uz + bz * ipd;
    uint32_t idx = (((uint32_t)ix << 1) + (uint32_t)iy);
    idx = (idx << 1) + (uint32_t)iz;
    uint32_t node_type;
    node_value = fetch_node(ns, idx, node_type);
    if (COUNT) cnt.node_loads++;
    if (node_type == 0u || node_type == 2u) { is_leaf = (node_type == 2u); break; }
    cx = cx * 2.0f; cy = cy * 2.0f; cz = cz * 2.0f;
  }
  inv_pow_depth = ipd; gx = ux; gy = uy; gz = uz; value = node_value;
  return is_leaf;
}

// ---- treeLookup's x index for ANY cell_count (FORM_TABLE) ------------------------------------------------------------------
// The x index of a level, as compiled (rc:376-378):
//     ix(v, f) = int(round_even(((float(v) + f) * inv_cell_count) * float(2 * cell_count) + -0.5))
// is, for a fixed cell index v, a composition of monotone roundings of f, hence a non-decreasing step function of the level's
// coordinate f in [0, 1).  With a power-of-two cell_count the products are exact and the steps are tree_lookup_pow2's q > 0.5 /
// q == 1 tests.  With the reference's own 100000 (main.rs:459: fl(1e-5) * 200000 = 2 (1 - 2.5e-8)) they sit within an ulp or two
// of 1/2 and 1, differently for every v — and for many v (7, 11, 14, 15, ... : a third of all cells) the products fall short at the
// bottom of the cell as well: ix(v, f) = 2v - 1, the upper half of the PREVIOUS cell, for f below half an ulp of v.  The
// reference really reads that node there, so this kernel must too.  Wherever ix(v, 0) is 2v - 1 or 2v and ix(v, 1 - 2^-24) is
// 2v + 1 or 2v + 2 (checked per cell when the table is built, never assumed) the function is fully described by three thresholds
//     ix(v, f) = 2v - 1 + (f >= F0(v)) + (f >= F1(v)) + (f >= F2(v)),    F0 <= F1 <= F2, F0 = 0: never 2v - 1, F2 = 2: never 2v + 2
// which build_thresholds_kernel finds by bisection over the bit patterns of f, evaluating the literal formula.  No closed form,
// so a table: (F1, F2), 8 bytes per cell, read beside the cell's nodes — two compares on f itself, not even the addition the
// power-of-two form does — and ONE scene-wide bound F0max = max F0(v) (about half an ulp of the largest cell index: 1e-7 ... 1e-4)
// below which a lane evaluates the literal formula for that level (a wave-uniform, rare branch, like the 2v + 2 case).  The band
// a jump table needs around the integers of 2^L c is exact as well: the decision of level l on cell v differs from the
// coordinate's binary digit only for f between 1/2 and F1(v), is 2v + 2 only for f >= F2(v) and 2v - 1 only for f < F0(v), i.e.
//     need_l(v) = 2^(L - l + 1) max(|F1(v) - 1/2|, 1 - F2(v), F0max)                      (the differences are exact in fp32)
// and a lane is on the table's side whenever |2^L c - rint(2^L c)| > the largest need along its descent (build_top_grid computes
// the largest over the whole table: one band per scene).  The claim "thresholds == literal formula" is checked exhaustively —
// every f in [0,1) x every cell — by tdt_selftest_index (tests/test_gpu_table_form.py).
TDT_DEV int32_t x_index_literal(uint32_t v, float f, float inv_cell_count, float two_cc) {
  return f2i(__builtin_rintf((((float)v + f) * inv_cell_count) * two_cc + -0.5f));
}
TDT_DEV float x_threshold_need(float2 F, float f0max) {  // how far from 1/2, 1 and 0 the decisions of this cell leave the binary digit
  const float d1 = __builtin_fabsf(F.x - 0.5f), d2 = F.y <= 1.0f ? 1.0f - F.y : 0.0f;
  const float d = d1 > d2 ? d1 : d2;
  return d > f0max ? d : f0max;
}
// one cell's thresholds (x = F1, y = F2, z = F0); *bad is raised when ix(v, .) is not of the three-threshold shape
TDT_DEV float4 x_thresholds(uint32_t v, float inv_cell_count, int32_t cell_count, uint32_t *bad) {
  const float two_cc = (float)(int32_t)((uint32_t)cell_count << 1);
  const int32_t base = (int32_t)(2u * v);
  const uint32_t kOne = 0x3F800000u;                   // f runs over the bit patterns [0, kOne): +0 ... 1 - 2^-24, in value order
  const int32_t g0 = x_index_literal(v, 0.0f, inv_cell_count, two_cc), g1 = x_index_literal(v, __uint_as_float(kOne - 1u), inv_cell_count, two_cc);
  if (v >= (1u << 22) || cell_count <= 0 || g0 < base - 1 || g0 > base || g1 < base + 1 || g1 > base + 2 || (v == 0u && g0 != 0)) {
    atomicOr(bad, 1u);
    return make_float4(2.0f, 2.0f, 0.0f, 0.0f);
  }
  auto first = [&](int32_t target) {                  // smallest f with ix(v, f) >= target
    uint32_t lo = 0u, hi = kOne;
    while (lo < hi) {
      const uint32_t mid = (lo + hi) >> 1;
      if (x_index_literal(v, __uint_as_float(mid), inv_cell_count, two_cc) >= target) hi = mid; else lo = mid + 1u;
    }
    return lo == kOne ? 2.0f : __uint_as_float(lo);
  };
  return make_float4(first(base + 1), first(base + 2), first(base), 0.0f);
}
// Per-lane memo of the last node fetched from HBM/L2 at each of CL levels (levels kMemoFirst+1 ..
// kMemoFirst+CL; shallower levels always sit in the LDS table).  The scene is read-only, so
// "node idx at level l" fetched for the previous traversal step is still the node: consecutive
// steps of a ray (and consecutive rays of a pixel) share most of their root path, which turns the
// reference's restart-from-root into ~1 dependent load per step without changing a decision.
// key = node index (29 bits) | code << 30.
template <int CL>
struct NodeMemo { uint32_t key[CL]; uint32_t val[CL]; };
// Top-level jump table (kGridLevels = 4 levels).  The y and z child digits of treeLookup are exact integer digits
// (tree_lookup_pow2); the x decision of a level is  a = fl(v + f) - v > 0.5  (and b = ... == 1), which differs from the
// plain binary digit floor(2 f) only when f lies within ulp(v + f) above 1/2 or below 1: <= 2^-17 for a cell index
// v < 128, <= 2^-14 for v < 1024.  So whenever 16 c_x is farther than 2^-12 from an integer, the first four levels are a
// pure function of the top four digits of (x, y, z):  grid[x4 << 8 | y4 << 4 | z4] = value << 5 | levels << 2 | code of the
// node that descent reaches (it may stop at an EMPTY / LEAF node earlier).  One LDS read then replaces four dependent
// ones; lanes inside a band (about 0.05 % of the steps) send their wave through the level-by-level form, so the result
// is the same bits either way.  Built per block from the LDS node table; unusable (grid_ok = false) when the tree is
// shallower than 4 levels, when a top node is not LDS-resident, or when a PARENT of levels 1-2 points at a cell index
// >= 128 or one of level 3 at >= 1024 (a tree edited so that its top cells were re-allocated — octree_update.comp can do
// that).  Used by all scene-specialised kernels of depth >= 4 (4K/256^3: 321 -> 297 ms, 1080p/512^3: 177 -> 165 ms,
// 64^3: 33.9 -> 33.3 ms; a 3-level table was 4 % SLOWER on the LDS-resident 64^3 tree, whose levels are a ds_read_b128 and
// 20 instructions each).
// Trees that do not fit the LDS table: a node read at level >= kNoProbeFrom skips the table and goes to the memo / HBM path
// directly.  The table holds the first 5120 cells of the breadth-first array, i.e. depths 0-4 and the start of depth 5, so a
// probe for a node of a depth >= 5 cell almost never hits; it is only a cache, so skipping it is always correct
// (4K/256^3: 292 -> 280 ms, 1080p/512^3: 166 -> 153 ms; from level 5 on: 286 / 154 ms).
constexpr int kNoProbeFrom = 6;
// Levels per table: FOUR (4096 32-bit entries, 16 KB) for trees that are wholly LDS-resident — their levels are one ds_read_b128
// and ~20 instructions each, and a fifth level in the table was 1 % slower than walking it (a 3-level table had been 4 % slower
// than 4) — and FIVE (round 2: 32768 entries of 16 bits, 64 KB beside the 82 KB node table: 146 of the CU's 160 KB) for trees
// that are not, whose level 5 otherwise costs an LDS probe, a memo compare and sometimes an L2 round trip (4K/256^3: -1.1 %,
// 1080p/512^3: -5.8 %).  The cell index the level-5 decision uses can reach 8191 (levels 1-4 hold at most 585 cells, level 5
// up to 4096 more), where ulp(v + f) = 2^-11: the bands around the integers of 32 c are 2^-11 wide, 2^-10 of all coordinates
// fall into them (a wave with a lane inside one — about 5 % of the steps — walks the levels one by one), and the exhaustive
// check (tdt_selftest 5 / 7) covers every coordinate x every cell index below the bounds x every level of the table.
TDT_DEV uint32_t grid_v_bound(int level) { return level < 3 ? 128u : (level == 3 ? 1024u : 8192u); }   // bound on the cell index a level-`level` PARENT may hold
template <int GL> struct Grid;
template <> struct Grid<4> {
  static constexpr int kLevels = 4;
  static constexpr uint32_t kEntries = 1u << 12;
  static constexpr float kBand = 0x1.0p-12f;
  typedef uint32_t Entry;                             // value << 5 | levels << 2 | code
  static TDT_DEV bool encode(uint32_t v, uint32_t m, uint32_t code, Entry &e) { e = (v << 5) | (m << 2) | code; return true; }
  static TDT_DEV void decode(uint32_t g, uint32_t &v, uint32_t &m, uint32_t &code) { code = g & 3u; m = (g >> 2) & 7u; v = g >> 5; }
};
template <> struct Grid<5> {
  static constexpr int kLevels = 5;
  static constexpr uint32_t kEntries = 1u << 15;
  static constexpr float kBand = 0x1.0p-11f;
  typedef uint16_t Entry;     // code 1 (PARENT, always after all five levels): value << 2 | 1;  code 0 / 2 (EMPTY / LEAF after `levels` levels): value << 5 | levels << 2 | code, value < 2048
  static TDT_DEV bool encode(uint32_t v, uint32_t m, uint32_t code, Entry &e) {
    if (code == 1u) { e = (Entry)((v << 2) | 1u); return v <= kPackedMaxValue && m == 5u; }
    e = (Entry)((v << 5) | (m << 2) | code);
    return v < 2048u;
  }
  static TDT_DEV void decode(uint32_t g, uint32_t &v, uint32_t &m, uint32_t &code) {
    code = g & 3u;
    const bool parent = code == 1u;
    m = parent ? 5u : (g >> 2) & 7u;
    v = parent ? g >> 2 : g >> 5;
  }
};
template <int GL, bool TABLE = false>
TDT_DEV void build_top_grid(const uint16_t *lds, uint32_t lds_nodes, int depth, typename Grid<GL>::Entry *grid, int *grid_ok,
                            const float2 *thr = nullptr, uint32_t thr_cells = 0u, float f0max = 0.0f, uint32_t *band_bits = nullptr) {
  constexpr int kGridLevels = GL;
  if (threadIdx.x == 0) { *grid_ok = depth >= kGridLevels ? 1 : 0; if (TABLE) *band_bits = 0u; }
  __syncthreads();
  if (depth >= kGridLevels) {
    for (uint32_t e = threadIdx.x; e < Grid<GL>::kEntries; e += blockDim.x) {
      const uint32_t xg = e >> (2 * kGridLevels), yg = (e >> kGridLevels) & ((1u << kGridLevels) - 1u), zg = e & ((1u << kGridLevels) - 1u);
      uint32_t v = 0, code = 1u, m = 0;
      bool ok = true;
      float need = 0.0f;
      for (int l = 1; l <= kGridLevels && code == 1u; l++) {
        const int sh = kGridLevels - l;
        if (TABLE) {                                  // this level's x decision is made on cell v: see x_thresholds
          if (v >= thr_cells) { ok = false; break; }
          const float n = x_threshold_need(thr[v], f0max) * (float)(2 << sh);
          need = n > need ? n : need;
        }
        const uint32_t idx = ((2u * v + ((xg >> sh) & 1u)) << 2) + (((yg >> sh) & 1u) << 1) + ((zg >> sh) & 1u);
        const uint32_t n = lds[idx < lds_nodes ? idx : lds_nodes];
        if (n == kPackedEscape) { ok = false; break; }
        v = n >> 2; code = n & 3u; m = (uint32_t)l;
        if (!TABLE && code == 1u && l < kGridLevels && v >= grid_v_bound(l)) ok = false;     // this v feeds the next level's x decision
      }
      if (code == 1u && m < (uint32_t)kGridLevels) ok = false;     // (the descent was cut short by a table that is unusable anyway)
      if (!Grid<GL>::encode(v, m, code, grid[e])) ok = false;
      if (!ok) atomicAnd(grid_ok, 0);
      if (TABLE) atomicMax(band_bits, __float_as_uint(need));     // (non-negative floats order as their bit patterns)
    }
  }
  __syncthreads();
}

// a pixel's cost for the hand-out order of the next dispatch (tdt_rt.hip, "Cost-feedback scheduling"):
// tree levels visited + kCostStep per traversal step + kCostEvent per path event (measured plateau 24..128)
constexpr uint32_t kCostStep = 3, kCostEvent = 64, kCostRayStep = 7;      // (kCostRayStep: per step when the levels are not counted — 3 + the ~4 levels a step used to visit)
constexpr uint32_t kEventWindow = 1024;   // rays after which the adaptive event threshold's running counts are halved
constexpr int kMemoLevels = 9;
constexpr int kBrickMemoLevels = 3;   // BRICK builds (see trace_kernel): >= 2, the levels their jump covers beyond kMemoFirst
// levels of the LDS jump table: 4 for trees inside the LDS table, 5 for the others (see Grid<GL>) — except FORM_TABLE trees outside
// it, which keep 4: the thresholds the table's band is computed from are staged for the first kThrTopCells cells only (levels 1-4
// of a breadth-first tree: at most 585 cells), beside the 82 KB node table
__host__ __device__ constexpr int top_grid_levels(bool resident, bool table) { return (resident || table) ? 4 : 5; }
constexpr uint32_t kThrTopCells = 1024u;
constexpr int kMemoFirst = 3;       // levels 1..3 = cells 0..72 at most: always inside the LDS table

// treeLookup rc:359-394 specialised for cell_count = 2^k <= 2^22 with inv_cell_count = 2^-k and
// max_depth <= 30 (checked on the host), where the float index arithmetic collapses to exact
// integer / comparison forms:
//   y,z: round_even(f*2 - 0.5) = (f > 0.5) for f = fract(c * 2^(l-1)) in [0,1).  Over all levels
//        these are the binary digits of floor(c * 2^D), except that an exact tie (f == 0.5: c * 2^D
//        an integer whose lowest set bit is this level's) picks the LOWER child and all-zero
//        digits below it: clear the lowest set bit when c * 2^D is an integer.  One integer per
//        axis, computed once per lookup; a level is one bit-field extract.
//   x  : ((v + fx) * 2^-k) * 2^(k+1) - 0.5 = 2*s - 0.5 exactly, s = fl(float(v) + fx); with
//        q = s - float(v) (exact) round_even gives 2v + (q > 0.5) + (q == 1): the second term is
//        the reference's own rounding artefact (fx rounded up to 1.0 at large v lands in the
//        NEXT cell), kept.  Needs v < 2^22 so that 2s - 0.5 is exact; larger v takes the
//        literal formula.  fx needs the float recurrence fract(2 f) (exact).
// Coordinates: c in [0,1) on entry (OctreeHit's outside test), so f starts as c itself.
// Scene-property specialisations chosen by the host (all bit-identical to the general form):
//   DEPTH    > 0: max_depth is this compile-time constant (digit shifts become immediates)
//   RESIDENT    : the whole cells buffer sits in the LDS table and no node needed the escape code
//   SAFEV       : every PARENT value in the buffer is < 2^22 (scanned once per buffer), so the
//                 literal-formula branch for huge cell indices cannot be taken
// FULL (round 2; small trees: max_depth 5 or 6, wholly LDS-resident): the jump table idea taken to the last level — one 16-bit
// entry per finest-level voxel position (8^depth of them: 64 KB / 512 KB, built once per cells buffer by build_full_grid_kernel,
// read through L2) holds what the whole descent ends on, so a traversal step does ONE load instead of a table read plus two
// more levels; the bands are those of the 5-level table (cell indices of a resident tree stay below 8192: 2^-11 around the
// integers of 2^depth c), and a wave with a lane inside one walks all levels from the LDS node table.
// BRICK (round 2; depth-8 and depth-9 trees that are NOT LDS-resident, every PARENT value < 2^22): the levels below 5 in ONE load
// as well.  Below level 5 cell indices are large and the x decision  a = fl(v + f) - v > 0.5, b = ... == 1  is far from the
// coordinate's binary digit (v ~ 2^20: ulp(v + f) = 2^-3), so a table indexed by position would be wrong for a tenth of all
// coordinates.  But for an integer v < 2^22 the sum v + f is rounded on a grid of 2^(e-23), e = floor(log2 v), that v itself
// lies on, with ties that do not depend on v: fl(v + f) - v is a function of e and f alone (checked for every f in [0,1) x every
// e x the ends and the middle of its binade: tdt_selftest 13).  And the cells of one level below one level-5 position are
// neighbours in the breadth-first array, so they share e.  Hence a "brick" per level-5 position, indexed not by position but
// by DECISION sequence — (a + b) of levels 6, 7, 8 (and 9: 27 or 81 combinations) x the y and z digits of those levels (64 or 256)
// — holds exactly what the reference's walk ends on, 2v + 2 jumps into the neighbour cell included, because
// build_bricks_kernel fills it by doing that walk; the level-5 table entry carries the exponents of the cell indices the
// decisions of levels 6.. add the coordinate to (31: the cells of a level below this position do not share one — the wave
// walks).  A traversal step is then one LDS read and one 2-byte load instead of a table read, three or four memo compares and
// as many dependent L2 round trips.  Table entries (32 bits): PARENT 1 | e6 << 2 | e7 << 7 | e8 << 12 | e9 << 17 | k << 29; EMPTY /
// LEAF code | (depth - levels) << 2 | value << 6 | k << 29 (depth 10, no bricks: levels << 2).  Brick entries (16 bits): (depth - levels) << 2 | code, and a LEAF's value << 6 (a material
// index >= 1024 below a position: it walks).  k: the position's own band — the x decisions of its five levels are the
// coordinate's digits unless 32 c is within 2^-(11 + k) of an integer, where 2^-(11 + k) >= 2^(5 - l) ulp(v_l + f) for the
// cell index v_l each level l really uses (the table-wide 2^-11 of the 16-bit tables assumes the largest index the bounds allow,
// 8191 at level 5; a sparse tree's top cells have indices in the hundreds, and every wave that walks costs a handful of
// dependent L2 round trips: 4K/256^3 -20 %).  Checked for every coordinate x level x cell index below the bounds, each with its
// own band (tdt_selftest 15).
__host__ __device__ constexpr uint32_t brick_levels(int depth) { return depth >= 6 && depth <= 9 ? (uint32_t)(depth - 5) : 0u; }   // levels a brick covers: all below the table (depth 10: none, its levels are walked)
__host__ __device__ constexpr uint32_t brick_entries(int depth) { uint32_t n = 1u; for (uint32_t j = 0; j < brick_levels(depth); j++) n *= 12u; return n; }   // (3 decisions x 2 y x 2 z) per level   // per level-5 position, 2 bytes each; 32768 positions: 113 MB / 1.36 GB of address space, touched where the tree is
constexpr uint32_t kBrickLdsCells = 1024u;            // BRICK builds keep a small node table (the walk of waves with a lane in a band starts in it)
// exponent B of the band (in units of 2^L c, L = 5) a level-l decision with cell index v needs: 2^(5 - l) ulp(v + f), ulp = 2^(floor(log2 v) - 23)
TDT_DEV int brick_band_exp(int l, uint32_t v) { return (v == 0u ? -40 : (31 - (int)__builtin_clz(v)) + 5 - l - 23); }
// fl(v + f) - v for any integer v in [2^e, 2^(e+1)), e <= 21
TDT_DEV float brick_q(uint32_t e, float f) { const float V = __uint_as_float((127u + e) << 23); return (V + f) - V; }
template <bool COUNT, int CL, int DEPTH, bool RESIDENT, bool SAFEV, bool FULL = false, bool BRICK = false, bool TABLE = false>
TDT_DEV bool tree_lookup_pow2(const TraceParams &P, const NodeSource &ns, float fx, float fy, float fz, float &inv_pow_depth,
                              float &gx, float &gy, float &gz, uint32_t &value, NodeMemo<CL> &memo, Counters &cnt) {
  static_assert(!TABLE || (SAFEV && !BRICK && !FULL), "per-cell thresholds: the resident walk, or (trees outside the LDS table) the jump table's bands only");
  const int depth = DEPTH > 0 ? DEPTH : P.max_depth;
  const float scale_d = __uint_as_float((uint32_t)(127 + depth) << 23);     // 2^depth
  const float Yf = fy * scale_d, Zf = fz * scale_d;                        // exact
  const float Yfl = __builtin_floorf(Yf), Zfl = __builtin_floorf(Zf);
  uint32_t Yi = (uint32_t)Yfl, Zi = (uint32_t)Zfl;
  if (__builtin_expect(__ballot((Yf == Yfl) | (Zf == Zfl)) != 0ull, 0)) {   // a coordinate exactly on a finest-level boundary: the tie rule
    Yi = (Yf == Yfl) ? (Yi & (Yi - 1u)) : Yi;
    Zi = (Zf == Zfl) ? (Zi & (Zi - 1u)) : Zi;
  }
  uint32_t qx = 1u, v = 0, code = 1u;                 // qx: x digits below a sentinel bit that counts the levels visited
  const float fx0 = fx;
  bool jumped = false;
  constexpr int kTableLevels = top_grid_levels(RESIDENT, TABLE);      // (see Grid<GL>)
  constexpr int kGridLevels = FULL ? DEPTH : kTableLevels;            // levels the jump covers (BRICK builds: never continue after their jump)
  if constexpr (FULL && !COUNT) {
    const float tg = fx0 * (float)(1 << DEPTH);       // exact
    const bool safe = __builtin_fabsf(tg - __builtin_rintf(tg)) > ns.grid_band;
    if (__builtin_expect(__ballot(!safe) == 0ull, 1)) {
      // the whole lookup: levels visited, the cell's digits and what it holds (see build_full_grid_kernel)
      const uint32_t xg = (uint32_t)tg;
      const uint32_t g = ns.full[(xg << (2 * DEPTH)) | (Yi << DEPTH) | Zi];
      const uint32_t sh = (g >> 2) & 7u;                                      // DEPTH - levels (build_full_grid_kernel stores it that way: one subtraction less per step)
      const float ipd = __uint_as_float(((127u - (uint32_t)DEPTH) << 23) + (sh << 23));      // 2^-levels
      gx = (float)(xg >> sh) * ipd; gy = (float)(Yi >> sh) * ipd; gz = (float)(Zi >> sh) * ipd;
      inv_pow_depth = ipd;
      value = g >> 5;
      return (g & 3u) == 2u;
    }
  } else if constexpr (BRICK && !COUNT) {
    static_assert(!BRICK || (DEPTH >= 6 && DEPTH <= 10 && !RESIDENT && SAFEV), "the 32-bit table: trees of depth 6-10 outside the LDS table");
    constexpr int BL = (int)brick_levels(DEPTH);      // levels a brick covers: all of them below the table (depths 6-9), or no bricks (depth 10)
    const float tg = fx0 * 32.0f;                     // exact
    const uint32_t xg = (uint32_t)tg;
    const uint32_t e = (xg << 10) | ((Yi >> (DEPTH - 5)) << 5) | (Zi >> (DEPTH - 5));
    const uint32_t g = ns.grid32[e];
    // the band of THIS position: 2^-(11 + k), k from the cell indices its five levels really add the coordinate to (brick_band_exp)
    const float band = __uint_as_float((116u - (g >> 29)) << 23);
    if constexpr (BL == 0) {                          // the table alone: levels 6.. are walked (PARENT entry: 1 | v << 2 | k << 29)
      if (__builtin_expect(__ballot(!(__builtin_fabsf(tg - __builtin_rintf(tg)) > band)) == 0ull, 1)) {
        code = g & 3u;
        const bool parent = code == 1u;
        const uint32_t mg = parent ? 5u : (g >> 2) & 15u;
        v = parent ? (g >> 2) & 0x3FFFFFu : (g >> 6) & 0x7FFFFFu;
        qx = (1u << mg) | (xg >> (5u - mg));
        fx = f_fract_nonneg(tg);                      // fract(c * 2^5): level 6's coordinate (only used when code == 1)
        jumped = true;
      }
    } else
    // (two ballots: one of the AND-ed condition goes through a 0 / 1 register and a second compare)
    if (__builtin_expect((__ballot(!(__builtin_fabsf(tg - __builtin_rintf(tg)) > band)) | __ballot((g & 0x7Fu) == 0x7Du)) == 0ull, 1)) {      // 0x7D: PARENT, first exponent 31
      uint32_t ent = g & 0x1FFFFFFFu, xd = xg << BL;  // what the descent ends on (a non-PARENT table entry IS a brick entry below its k); its x digits (the top `levels` count)
      if ((g & 3u) == 1u) {
        uint32_t ci = 0u, xlow = 0u;
#pragma unroll
        for (int j = 0; j < BL; j++) {                // level 6 + j: its coordinate, the decision for ANY cell index of the exponent the entry names
          const float f = j == 0 ? f_fract_nonneg(tg) : f_fract_nonneg(fx0 * (float)(32 << j));
          const float q = brick_q((g >> (2 + 5 * j)) & 31u, f);
          const uint32_t a = q > 0.5f ? 1u : 0u, b = q == 1.0f ? 1u : 0u;
          ci = ci * 3u + a + b;
          xlow = (xlow << 1) | (a & ~b);
        }
        constexpr uint32_t kCombos = BL == 1 ? 3u : (BL == 2 ? 9u : (BL == 3 ? 27u : 81u)), kMask = (1u << BL) - 1u;
        const uint32_t bi = ((__umul24(e, kCombos) + ci) << (2 * BL)) | ((Yi & kMask) << BL) | (Zi & kMask);
        ent = static_cast<const uint16_t *>(ns.bricks)[bi];
        xd |= xlow;
      }
      const uint32_t sh = (ent >> 2) & 15u;                                   // DEPTH - levels (as build_bricks_kernel stores it)
      const float ipd = __uint_as_float(((127u - (uint32_t)DEPTH) << 23) + (sh << 23));      // 2^-levels
      gx = (float)(xd >> sh) * ipd; gy = (float)(Yi >> sh) * ipd; gz = (float)(Zi >> sh) * ipd;
      inv_pow_depth = ipd;
      value = ent >> 6;
      return (ent & 3u) == 2u;
    }
  } else if constexpr (!COUNT && DEPTH >= kTableLevels) {        // the top levels in one step (see build_top_grid)
    const float tg = fx0 * (float)(1 << kGridLevels);  // exact
    const bool safe = __builtin_fabsf(tg - __builtin_rintf(tg)) > ns.grid_band;   // (an unusable table has band 2: never safe)
    if (__builtin_expect(__ballot(!safe) == 0ull, 1)) {
      const uint32_t xg = (uint32_t)tg;               // floor: tg in [0, 2^kGridLevels)
      const uint32_t g = static_cast<const typename Grid<kTableLevels>::Entry *>(ns.grid)[(xg << (2 * kGridLevels)) | ((Yi >> (depth - kGridLevels)) << kGridLevels) | (Zi >> (depth - kGridLevels))];
      uint32_t mg;
      Grid<kTableLevels>::decode(g, v, mg, code);
      qx = (1u << mg) | (xg >> ((uint32_t)kGridLevels - mg));
      fx = f_fract_nonneg(tg);                        // fract(c * 2^kGridLevels): the next level's coordinate (only used when code == 1)
      jumped = true;
    }
  }
  auto level = [&](int l, uint32_t *mkey, uint32_t *mval) {    // l = 1-based level
    const int sh = depth - l;
    const float fv = (float)v;
    if (COUNT) { cnt.level_slots += slot64(); cnt.level_active++; cnt.node_loads++; }
    if (RESIDENT && SAFEV) {
      // Whole tree in the LDS table: the 8 children of cell v are 16 consecutive bytes, so one ds_read_b128 issued
      // as soon as v is known overlaps the x decision below, and the child is then picked in registers.  Cells
      // past the table read the all-EMPTY sentinel cell (a read past the end of the buffer IS 0: robust access).
      const uint32_t cell = v < ns.lds_cells ? v : ns.lds_cells;
      const uint4 c = *reinterpret_cast<const uint4 *>(ns.lds + (cell << 3));
      bool a, b, low = false;
      if constexpr (TABLE) { const float2 F = ns.thr[cell]; a = fx >= F.x; b = fx >= F.y; low = fx < ns.thr_f0max; }       // (x_thresholds)
      else { const float q = (fv + fx) - fv; a = q > 0.5f; b = (q == 1.0f); }
      uint32_t bitx = (a && !b) ? 1u : 0u;
      const uint32_t yb = (Yi >> sh) & 1u, zb = (Zi >> sh) & 1u;
      const uint32_t lo = a ? c.z : c.x, hi = a ? c.w : c.y;
      uint32_t n = ((yb ? hi : lo) >> (zb << 4)) & 0xFFFFu;
      if (__builtin_expect(__ballot(b || low) != 0ull, 0)) {    // q == 1: x index 2v + 2, the first half of the NEXT cell
        if (b || low) {
          uint32_t ix = 2u * v + 2u;
          if constexpr (TABLE) if (low) {                // f below the largest F0 of the scene: the formula itself (2v - 1 for f < F0(v))
            ix = (uint32_t)x_index_literal(v, fx, P.inv_cell_count, (float)(int32_t)((uint32_t)P.cell_count << 1));
            bitx = ix & 1u;
          }
          const uint32_t idx = ((ix << 2) + (yb << 1) + zb) & 0x1FFFFFFFu;
          n = ns.lds[idx < ns.lds_nodes ? idx : ns.lds_nodes];
        }
      }
      qx = qx + qx + bitx;
      v = n >> 2; code = n & 3u;
      fx = f_fract_nonneg(fx0 * __uint_as_float((uint32_t)(127 + l) << 23));
      return;
    }
    uint32_t ix; uint32_t bitx;
    if constexpr (TABLE) {
      // a cell_count that is not a power of two, tree outside the LDS table: the x index by the formula itself (rc:376-378) — the
      // thresholds of x_thresholds would cost a dependent load per level here; they serve the jump table's band
      const float two_cc = (float)(int32_t)((uint32_t)P.cell_count << 1);
      const float rx = __builtin_rintf(((fv + fx) * P.inv_cell_count) * two_cc + -0.5f);
      ix = (uint32_t)f2i(rx);
      const float tx = __builtin_truncf(rx);
      bitx = ((tx + -(2.0f * __builtin_floorf(tx / 2.0f))) != 0.0f) ? 1u : 0u;
    } else
    if (SAFEV || __builtin_expect(__ballot(v >= (1u << 22)) == 0ull, 1)) {
      const float q = (fv + fx) - fv;
      const bool a = q > 0.5f, b = (q == 1.0f);
      ix = (v + v + (uint32_t)a) + (uint32_t)b;
      bitx = (a && !b) ? 1u : 0u;
    } else {
      const float two_cc = (float)(int32_t)((uint32_t)P.cell_count << 1);
      const float rx = __builtin_rintf(((fv + fx) * P.inv_cell_count) * two_cc + -0.5f);
      ix = (uint32_t)f2i(rx);
      const float tx = __builtin_truncf(rx);
      bitx = ((tx + -(2.0f * __builtin_floorf(tx / 2.0f))) != 0.0f) ? 1u : 0u;
      if (v < (1u << 22)) {                           // lanes that did not need the literal form
        const float q = (fv + fx) - fv;
        const uint32_t a = q > 0.5f ? 1u : 0u, b = (q == 1.0f) ? 1u : 0u;
        ix = 2u * v + a + b; bitx = a & ~b;
      }
    }
    qx = qx + qx + bitx;
    const uint32_t idx = ((ix << 2) + (((Yi >> sh) & 1u) << 1) + ((Zi >> sh) & 1u)) & 0x1FFFFFFFu;
    // LDS table first: an unconditional read of min(idx, lds_nodes).  The slot just past the table holds
    // a sentinel: EMPTY (0) when the whole buffer is resident — a read past the end of the buffer IS 0
    // (robust access) — and the escape code otherwise, which sends the lane to the range-checked HBM path.
    bool resident = false;
    if (RESIDENT || l < kNoProbeFrom) {
      const uint32_t li = idx < ns.lds_nodes ? idx : ns.lds_nodes;
      const uint32_t n = ns.lds[li];
      resident = RESIDENT || (n != kPackedEscape);
      v = n >> 2; code = n & 3u;
    }
    if (!resident) {
      if (mkey && (*mkey & 0x1FFFFFFFu) == idx) {
        v = *mval; code = *mkey >> 30;
      } else {
        if (COUNT) cnt.memo_miss++;
        const auto n2 = __builtin_amdgcn_raw_buffer_load_b64(ns.cells, (int)(idx << 3), 0, 0);
        const uint32_t node_type = (uint32_t)n2[1];
        code = (node_type == 0u) ? 0u : (node_type == 2u ? 2u : 1u);
        v = (uint32_t)n2[0];
        if (mkey) { *mkey = idx | (code << 30); *mval = v; }
      }
    }
    fx = f_fract_nonneg(fx0 * __uint_as_float((uint32_t)(127 + l) << 23));   // fract(c * 2^l): next level's coordinate
  };
  constexpr int kFirstAfterJump = (kGridLevels > kMemoFirst ? kGridLevels : kMemoFirst) + 1;
  TDT_MARK(walk);
  if (!jumped) {                                      // (wave-uniform) the levels a jump would have covered
#pragma unroll
    for (int l = 1; l <= kMemoFirst; l++) {
      if (code == 1u && l <= depth) level(l, nullptr, nullptr);
    }
#pragma unroll
    for (int l = kMemoFirst + 1; l < kFirstAfterJump; l++) {
      if (code == 1u && l <= depth) {
        if (l - kMemoFirst - 1 < CL) level(l, &memo.key[l - kMemoFirst - 1 < CL ? l - kMemoFirst - 1 : 0], &memo.val[l - kMemoFirst - 1 < CL ? l - kMemoFirst - 1 : 0]);
        else level(l, nullptr, nullptr);               // (a build with fewer memo levels than the jump covers)
      }
    }
  }
#pragma unroll
  for (int l = kFirstAfterJump; l <= kMemoFirst + CL; l++) {
    if (code == 1u && l <= depth) level(l, &memo.key[l - kMemoFirst - 1], &memo.val[l - kMemoFirst - 1]);
  }
  for (int l = (kMemoFirst + CL + 1 > kFirstAfterJump ? kMemoFirst + CL + 1 : kFirstAfterJump); code == 1u && l <= depth; l++) level(l, nullptr, nullptr);      // (levels past the memo; never one the jump covered)
  const int m = 31 - __builtin_clz(qx);               // levels visited
  qx ^= 1u << m;
  const float ipd = __uint_as_float((uint32_t)(127 - m) << 23);             // 2^-m = inv_pow_depth after m halvings
  const int sh = depth - m;
  gx = (float)qx * ipd; gy = (float)(Yi >> sh) * ipd; gz = (float)(Zi >> sh) * ipd;
  if (DEPTH > 0) {
    inv_pow_depth = ipd;                              // a compile-time depth >= 1: level 1 always runs, m >= 1
  } else {
    inv_pow_depth = (m == 0) ? 1.0f : ipd;            // max_depth 0: the loop body never ran
    if (m == 0) { gx = 0.f; gy = 0.f; gz = 0.f; }
  }
  value = v;
  return code == 2u;
}

TDT_DEV float rand2(float cx, float cy) {   // Rand(vec2) rc:53
  return f_fract(sin_poly(cy * 78.233f + cx * 12.9898f) * 43758.5453f);
}

// The material tables (rc:189-222) behind raw buffer descriptors: the hardware's range check returns 0 for a dword past the end,
// which IS the reference's robust-access rule, and — unlike a compare-and-branch per load — leaves the loads free to overlap:
// one round trip for materials[hit.index]'s three dwords, one more for everything they point at (albedo, fuzz, index of
// refraction: all issued, whatever the type turns out to be; a table the material does not use is read and ignored).
struct MatSource { __amdgpu_buffer_rsrc_t materials, albedos, metal, dielectric; };
TDT_DEV __amdgpu_buffer_rsrc_t table_rsrc(const uint32_t *p, uint32_t dwords) {      // (a 32-bit byte offset cannot reach past 2^30 dwords)
  return __builtin_amdgcn_make_buffer_rsrc((void *)p, 0, (int)(dwords < (1u << 30) ? dwords << 2 : 0xFFFFFFFFu), 0x00020000);
}
TDT_DEV MatSource material_source(const TraceParams &P) {
  MatSource ms;
  ms.materials = table_rsrc(P.materials, P.materials_dwords); ms.albedos = table_rsrc(P.albedos, P.albedos_dwords);
  ms.metal = table_rsrc(P.metal, P.metal_dwords); ms.dielectric = table_rsrc(P.dielectric, P.dielectric_dwords);
  return ms;
}
TDT_DEV uint32_t ld_buf(__amdgpu_buffer_rsrc_t rs, uint32_t byte_off) { return (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rs, (int)byte_off, 0, 0); }
// materials[hit.index] (rc:189-196): fetched as early as the hit is known, so that the round trip overlaps the hit record's
// arithmetic instead of standing in front of the material switch
struct MatRef { uint32_t type, attribute, albedo; };
TDT_DEV MatRef material_fetch(const MatSource &ms, uint32_t index) {
  const uint32_t mo = index * 12u;
  MatRef m;
  m.type = ld_buf(ms.materials, mo);
  m.attribute = ld_buf(ms.materials, mo + 4u);
  m.albedo = ld_buf(ms.materials, mo + 8u);
  return m;
}

// normalize(n) of a hit record's normal (rc:471, rc:485).  CubeHit's normals are +-1 on one axis and zeros on the others
// (cube_normal_fast), so the squared length is exactly 1.0f, 1 / sqrt(1.0f) is 1.0f and n * 1.0f is n: nothing to do.  Whatever
// else a record may hold (the zeros of a pixel's first, never-written record; NaNs) takes the arithmetic.
TDT_DEV void unit_or_normalised(float nx, float ny, float nz, float &mx, float &my, float &mz) {
  const float nn = (nz * nz + ny * ny) + nx * nx;
#ifndef TDT_LITERAL_NORMAL
  if (__builtin_expect(__ballot(nn != 1.0f) == 0ull, 1)) { mx = nx; my = ny; mz = nz; return; }
#endif
  const float rs = q_rsq(nn);
  mx = nx * rs; my = ny * rs; mz = nz * rs;
}

// switch (materials[hit.index].type) rc:278-291; returns false when the path ends
// SHARED_RAND: Rand(hit.point.xy) is the first draw of BOTH ScatterMetal (rc:488 via RandVec3) and ScatterDielectric (rc:513) — one evaluation
// for the lanes of either kind (bench frame -1.2 %; the brick builds, at the edge of their register budget, lose 0.5 % to it and keep two)
template <bool COUNT, bool SHARED_RAND = false, bool SHARED_NORM = false>
TDT_DEV bool scatter(const MatSource &ms, const Ray &r, const Hit &h, const MatRef &mat, Ray &out, float &ar, float &ag, float &ab, Counters &cnt) {
  const int32_t type = (int32_t)mat.type;
  // the second-level reads, all of them, before the switch (see MatSource); used at the end of the branches
  const uint32_t ai = mat.albedo * 12u, at = mat.attribute << 2;
  const float alb_r = __uint_as_float(ld_buf(ms.albedos, ai)), alb_g = __uint_as_float(ld_buf(ms.albedos, ai + 4u)), alb_b = __uint_as_float(ld_buf(ms.albedos, ai + 8u));
  const float m_fuzz = __uint_as_float(ld_buf(ms.metal, at)), ir = __uint_as_float(ld_buf(ms.dielectric, at));
  float dx = r.dx, dy = r.dy, dz = r.dz;
  float nx = h.nx, ny = h.ny, nz = h.nz;
  out.ox = h.px; out.oy = h.py; out.oz = h.pz;
  if (COUNT) { cnt.lambertian += (type == 0); cnt.metal += (type == 1); cnt.dielectric += (type == 2); cnt.unknown += ((uint32_t)type > 2u); }
  float rnd_md = 0.0f;
  if (SHARED_RAND && (type == 1 || type == 2)) rnd_md = rand2(h.px, h.py);
  float ux = 1.0f, uy = 0.0f, uz = 0.0f;              // SHARED_NORM: the scattered direction before normalize()
  if (type == 0) {   // ScatterLambertian rc:470-482, constructFrisvad rc:453-468, SampleGGXVNDF rc:27-49
    TDT_MARK(lambert);
    float mx, my, mz; float rs;
    unit_or_normalised(nx, ny, nz, mx, my, mz);
    bool sing = nz < -0.9999f;
    float a = q_rcp(1.0f + nz);
    float b = -((nx * ny) * a);
    float r0x = 1.0f + -((nx * nx) * a);
    float r2y = 1.0f + -((ny * ny) * a);
    float e_b = sing ? -1.0f : b;
    float e_r2y = sing ? 0.0f : r2y;
    float e_r2z = sing ? 0.0f : -ny;
    float e_r0x = sing ? 0.0f : r0x;
    float e_r0z = sing ? 0.0f : -nx;
    float vx = (dz * e_r0z + dy * e_b) + dx * e_r0x;
    float vy = (dz * mz + dy * my) + dx * mx;
    float vz = (dz * e_r2z + dy * e_r2y) + dx * e_b;
    // hash23(point * 100 + 0) rc:85-90,230-232 ("*100" and the hash scale folded in fp32)
    float p3x = f_fract((100.0f * .1031f) * h.px), p3y = f_fract((100.0f * .1030f) * h.py),
          p3z = f_fract((100.0f * .0973f) * h.pz);
    float d = (p3z * (p3x + 33.33f) + p3y * (p3z + 33.33f)) + p3x * (p3y + 33.33f);
    p3x = p3x + d; p3y = p3y + d; p3z = p3z + d;
    float U0 = f_fract((p3x + p3y) * p3z), U1 = f_fract((p3x + p3z) * p3y);
    float sx = vx * 0.85f, sy = vy * 0.85f;
    rs = q_rsq((vz * vz + sy * sy) + sx * sx);
    float va = sx * rs, vb = sy * rs, vc = vz * rs;      // Vh = (-va,-vb,-vc)
    float vhz = -vc;
    float lensq = va * va + vb * vb;
    bool nzl = 0.0f < lensq;
    float rl = q_rsq(lensq);
    float t1x = nzl ? vb * rl : 1.0f;
    float t1y = nzl ? -(va * rl) : 0.0f;
    float rr = q_sqrt(U0);
    float phi = (2.0f * 3.14159265358f) * U1;
    float sn, cs; sincos_poly(phi, sn, cs);
    float t1 = rr * cs, t2 = rr * sn;
    float s = 0.5f * (1.0f + vhz);
    float om = 1.0f + -(t1 * t1);
    t2 = (1.0f + -s) * q_sqrt(om) + s * t2;
    float T2x = vc * t1y;
    float T2yn = vc * t1x;
    float T2z = -(va * t1y) + vb * t1x;
    float nhx = t1 * t1x + t2 * T2x;
    float nhy = t1 * t1y + -(T2yn * t2);
    float nhz = t2 * T2z;
    float sq = q_sqrt(f_max(om + -(t2 * t2), 0.0f));
    nhx = nhx + -(va * sq); nhy = nhy + -(vb * sq); nhz = nhz + -(vc * sq);
    float ex = 0.85f * nhx, ey = 0.85f * nhy, ez = f_max(nhz, 0.0f);
    rs = q_rsq((ez * ez + ey * ey) + ex * ex);
    ex = ex * rs; ey = ey * rs; ez = ez * rs;
    float dt = ((ez * dz + ey * dy) + ex * dx) * 2.0f;
    float qx = nx + (dx + -(dt * ex)), qy = ny + (dy + -(dt * ey)), qz = nz + (dz + -(dt * ez));
    ar = alb_r; ag = alb_g; ab = alb_b;
    if (!SHARED_NORM) { rs = q_rsq((qz * qz + qy * qy) + qx * qx); out.dx = qx * rs; out.dy = qy * rs; out.dz = qz * rs; return true; }
    ux = qx; uy = qy; uz = qz;
  }
  if (type == 1) {   // ScatterMetal rc:484-491, RandInHemisphere rc:106-115 (one cube sample, as compiled)
    TDT_MARK(metal);
    float mx, my, mz; float rs;
    unit_or_normalised(nx, ny, nz, mx, my, mz);
    float dt = ((mz * dz + my * dy) + mx * dx) * 2.0f;
    float rx = dx + -(dt * mx), ry = dy + -(dt * my), rz = dz + -(dt * mz);
    const float fuzz = m_fuzz;
    float hx = -1.0f + 2.0f * (SHARED_RAND ? rnd_md : rand2(h.px, h.py));
    float hy = -1.0f + 2.0f * rand2(h.px + hx, h.py + hx);
    float hz = -1.0f + 2.0f * rand2(h.px + hy, h.py + hy);
    bool same = -(hz * nz + hy * ny) < hx * nx;
    if (!same) { hx = -hx; hy = -hy; hz = -hz; }
    float qx = rx + fuzz * hx, qy = ry + fuzz * hy, qz = rz + fuzz * hz;
    ar = alb_r; ag = alb_g; ab = alb_b;
    if (!SHARED_NORM) {
      rs = q_rsq((qz * qz + qy * qy) + qx * qx);
      qx = qx * rs; qy = qy * rs; qz = qz * rs;
      out.dx = qx; out.dy = qy; out.dz = qz;
      return -(qz * nz + qy * ny) < qx * nx;
    }
    ux = qx; uy = qy; uz = qz;
  }
  if (type == 2) {   // ScatterDielectric rc:499-522, reflectance rc:494-497
    TDT_MARK(dielectric);
    float ratio = h.ff ? q_rcp(ir) : ir;
    float pz_ = dz * nz, py_ = dy * ny, px_ = dx * nx;
    float cos_t = f_min((-pz_ + -py_) + -px_, 1.0f);
    float sin_t = q_sqrt(1.0f + -(cos_t * cos_t));
    bool cannot = 1.0f < ratio * sin_t;
    float q = schlick_q(ratio);
    float r0 = q * q;
    float rnd = SHARED_RAND ? rnd_md : rand2(h.px, h.py);
    float refl = pow_poly(1.0f + -cos_t, 5.0f) * (1.0f + -r0) + r0;
    float dn = (pz_ + py_) + px_;
    float ox_, oy_, oz_;
    if (cannot || (rnd < refl)) {
      float dt = dn * 2.0f;
      ox_ = dx + -(dt * nx); oy_ = dy + -(dt * ny); oz_ = dz + -(dt * nz);
    } else {
      float k = 1.0f + -(ratio * (ratio * (1.0f + -(dn * dn))));
      if (!(k < 0.0f)) {
        float m = ratio * dn + q_sqrt(k);
        ox_ = ratio * dx + -(m * nx); oy_ = ratio * dy + -(m * ny); oz_ = ratio * dz + -(m * nz);
      } else { ox_ = 0.0f; oy_ = 0.0f; oz_ = 0.0f; }
    }
    ar = 1.0f; ag = 1.0f; ab = 1.0f;
    if (!SHARED_NORM) { float rs = q_rsq((oz_ * oz_ + oy_ * oy_) + ox_ * ox_); out.dx = ox_ * rs; out.dy = oy_ * rs; out.dz = oz_ * rs; return true; }
    ux = ox_; uy = oy_; uz = oz_;
  }
  if (!SHARED_NORM || (uint32_t)type > 2u) return false;
  // normalize(...) — the last operation of all three scatter functions (rc:481, rc:489, rc:521), in the same association — once, for
  // the lanes of every material, behind the branches: three transcendentals + 16 other instructions that each branch used to issue for
  // its own few lanes.  ScatterMetal's test of the normalised direction against the normal (rc:490) follows it.
  const float rs = q_rsq((uz * uz + uy * uy) + ux * ux);
  ux = ux * rs; uy = uy * rs; uz = uz * rs;
  out.dx = ux; out.dy = uy; out.dz = uz;
  return type != 1 || -(uz * nz + uy * ny) < ux * nx;
}

// primary ray of sample s at pixel (px,py): rc:240-245, CameraGetRay rc:304-307
TDT_DEV Ray primary_ray(const TraceParams &P, int px, int py, int s) {
  float x = (float)px, y = (float)py, fs = (float)s;
  const float K = 0.2f * .1031f;
  // pixel coordinates and sample indices are non-negative, so every fract argument here is too
  float a = f_fract_nonneg(K * (x + fs)), b = f_fract_nonneg(K * y);
  float d = (a + 33.33f) * (a + b) + a * (b + 33.33f);
  float h1 = f_fract_nonneg(((a + d) + (b + d)) * (a + d));
  float a2 = f_fract_nonneg(K * x), b2 = f_fract_nonneg(K * (y + fs));
  float d2 = (a2 + 33.33f) * (a2 + b2) + a2 * (b2 + 33.33f);
  float h2 = f_fract_nonneg(((a2 + d2) + (b2 + d2)) * (a2 + d2));
  float u = (x + h1) / (float)(P.image_width - 1);
  float v = (y + h2) / (float)(P.image_height - 1);
  float rx = (P.hor[0] * u + P.llc[0]) + (v * P.ver[0] + -P.org[0]);
  float ry = (P.hor[1] * u + P.llc[1]) + (v * P.ver[1] + -P.org[1]);
  float rz = (P.hor[2] * u + P.llc[2]) + (v * P.ver[2] + -P.org[2]);
  float rs = q_rsq((rz * rz + ry * ry) + rx * rx);
  Ray r = { P.org[0], P.org[1], P.org[2], rx * rs, ry * rs, rz * rs };
  return r;
}

}  // namespace tdt
