/*
 * pathtrace_oracle.c — TEST INFRASTRUCTURE ONLY (see pathtrace_oracle.h).
 *
 * Scalar fp32 restatement of /root/reference/assets/shaders/raytracer.comp following the
 * operation order the shader has after Mesa's GLSL->NIR compile (ST_DEBUG=nir dump of the
 * unmodified file), because the integrand is chaotic: every hash feeds on the bits of the
 * previous hit point, so only a bit-for-bit restatement reproduces the reference image.
 * Build with -ffp-contract=off (a*b+c below is two roundings, as in the compiled shader:
 * GLSL fma() is lowered to fmul+fadd); fmaf() is used only inside the sin/cos/pow
 * polynomials where llvmpipe itself fuses.
 *
 * Reference lines are cited as rc:N = raytracer.comp line N.
 */
#include "pathtrace_oracle.h"

#include <limits.h>
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ---------------------------------------------------------------- scalar helpers -------- */
static inline float f_fract(float x) { return x - floorf(x); }          /* GLSL fract, unclamped */
static inline float f_rcp(float x) { return 1.0f / x; }                 /* llvmpipe: IEEE divide */
static inline float f_rsq(float x) { return 1.0f / sqrtf(x); }          /* llvmpipe: rcp(sqrt)   */
/* llvmpipe min/max: a NaN operand yields the other one; otherwise SSE minps/maxps(a,b) */
static inline float f_min(float a, float b) { if (b != b) return a; return a < b ? a : b; }
static inline float f_max(float a, float b) { if (b != b) return a; return a > b ? a : b; }
static inline float b2f(int b) { return b ? 1.0f : 0.0f; }
static inline uint32_t f_bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float bits_f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
/* float -> int as cvttps2dq does it: out of range / NaN -> INT_MIN */
static inline int32_t f2i(float f) {
  if (!(f > -2147483904.0f && f < 2147483648.0f)) return INT32_MIN;
  return (int32_t)f;
}

/* gallivm's sin/cos: Cephes-style range reduction by pi/4 octants, two short polynomials,
 * fused multiply-adds inside (llvm.fmuladd on an FMA host). */
static void sincos_poly(float a, float *s_out, float *c_out) {
  float x = fabsf(a);
  int32_t j = f2i(x * 1.27323954473516f);
  j = (j + 1) & ~1;
  float y = (float)j;
  x = fmaf(y, -0.78515625f, x);
  x = fmaf(y, -2.4187564849853515625e-4f, x);
  x = fmaf(y, -3.77489497744594108e-8f, x);
  float z = x * x;
  float c = 2.443315711809948e-5f;
  c = fmaf(c, z, -1.388731625493765e-3f);
  c = fmaf(c, z, 4.166664568298827e-2f);
  c = c * z;
  c = c * z;
  c = fmaf(-0.5f, z, c);
  c = c + 1.0f;
  float s = -1.9515295891e-4f;
  s = fmaf(s, z, 8.3321608736e-3f);
  s = fmaf(s, z, -1.6666654611e-1f);
  s = s * z;
  s = fmaf(s, x, x);
  /* sin */
  {
    float r = ((j & 2) == 0) ? s : c;
    int neg = (((j & 4) != 0) ? 1 : 0) ^ ((f_bits(a) >> 31) & 1);
    *s_out = neg ? -r : r;
  }
  /* cos */
  {
    int32_t k = j - 2;
    float r = ((k & 2) == 0) ? s : c;
    int neg = ((~k) & 4) != 0;
    *c_out = neg ? -r : r;
  }
}
float oracle_sin(float a) { float s, c; sincos_poly(a, &s, &c); return s; }
float oracle_cos(float a) { float s, c; sincos_poly(a, &s, &c); return c; }

/* gallivm's pow(x,y) = exp2(log2(x) * y), polynomial log2 / exp2 */
static float log2_poly(float x) {
  uint32_t u = f_bits(x);
  float e = (float)((int32_t)((u >> 23) & 0xff) - 127);
  float m = bits_f((u & 0x007fffffu) | 0x3f800000u);
  float t = (m - 1.0f) / (m + 1.0f);
  float z = t * t;
  const float L0 = 2.88539009343309178325f, L1 = 0.961791550404184197881f, L2 = 0.577440339438736392009f,
              L3 = 0.403343858251329912514f, L4 = 0.406718052498846252698f;
  float z2 = z * z;
  float even = fmaf(z2, fmaf(z2, L4, L2), L0);
  float odd = fmaf(z2, L3, L1);
  float p = fmaf(odd, z, even);
  return fmaf(t, p, e);
}
static float exp2_poly(float u) {
  u = f_min(f_max(u, -126.99999f), 128.0f);
  float i = floorf(u);
  float f = u - i;
  float f2 = f * f;
  const float E0 = 1.0f, E1 = 0.693153073200168932794f, E2 = 0.240153617044375388211f,
              E3 = 0.0558263180532956664775f, E4 = 0.00898934009049466391101f, E5 = 0.00187757667519147912699f;
  float even = fmaf(f2, fmaf(f2, E4, E2), E0);
  float odd = fmaf(f2, fmaf(f2, E5, E3), E1);
  float q = fmaf(odd, f, even);
  int32_t ii = f2i(i);
  return bits_f((uint32_t)(ii + 127) << 23) * q;
}
float oracle_pow(float x, float y) { return exp2_poly(log2_poly(x) * y); }

/* ---------------------------------------------------------------- scene access ---------- */
typedef struct {
  const oracle_scene *sc;
  oracle_camera cam;
  /* OctreeFloats / OctreeInts (rc:150-166), read once: they are uniform data */
  float min_x, min_y, min_z, scale, inv_scale, inv_cell_count;
  int32_t max_depth, max_iter, cell_count;
} tracer;

static inline uint32_t ld_u32(const void *buf, size_t bytes, uint32_t byte_off) {
  /* llvmpipe SSBO load: dword index compared against size in dwords, 0 when outside */
  if ((size_t)(byte_off >> 2) >= (bytes >> 2)) return 0;
  uint32_t v; memcpy(&v, (const char *)buf + (byte_off & ~3u), 4); return v;
}
static inline float ld_f32(const void *buf, size_t bytes, uint32_t byte_off) {
  return bits_f(ld_u32(buf, bytes, byte_off));
}

/* Out-parameter temporaries that survive between calls (uninitialised `out HitRecord` copies
 * in rc:186,316 become loop-carried values after inlining). */
typedef struct {
  float nx, ny, nz; int ff; float px, py, pz;
} hit_tmp;
typedef struct {
  hit_tmp root; float root_t;   /* CubeHit call at rc:407 */
  hit_tmp leaf;                 /* CubeHit call at rc:432 */
} pixel_carry;

typedef struct { float ox, oy, oz, dx, dy, dz; } ray;
typedef struct { float px, py, pz, nx, ny, nz; int ff; uint32_t index; } hit_record;

/* CubeHit's hit-record part (rc:336-354) for entry parameter t */
static inline void cube_hit_record(const ray *r, float t, float cx, float cy, float cz, float size, hit_tmp *h) {
  float px = t * r->dx + r->ox, py = t * r->dy + r->oy, pz = t * r->dz + r->oz;   /* RayAt rc:258-261 */
  float radius = size * 0.5f;
  float nx = px + -(cx + radius), ny = py + -(cy + radius), nz = pz + -(cz + radius);
  float ax = fabsf(nx), ay = fabsf(ny), az = fabsf(nz);
  nx = nx * b2f(ax >= f_max(ay, az));        /* rc:346-348, ties favour x */
  ny = ny * b2f(f_max(ax, az) < ay);
  nz = nz * b2f(f_max(ax, ay) < az);
  float rs = f_rsq(nz * nz + ny * ny + nx * nx);
  nx = nx * rs; ny = ny * rs; nz = nz * rs;
  int ff = (r->dz * nz + r->dy * ny) < -(r->dx * nx);      /* dot(d,n) < 0, rc:350-351 */
  float flip = -2.0f * b2f(!ff) + 1.0f;                       /* rc:352 */
  nx = nx * flip; ny = ny * flip; nz = nz * flip;
  rs = f_rsq(nz * nz + ny * ny + nx * nx);
  h->nx = nx * rs; h->ny = ny * rs; h->nz = nz * rs;
  h->ff = ff; h->px = px; h->py = py; h->pz = pz;
}

/* slab test (rc:317-334): returns entry/exit parameters */
static inline void cube_slabs(const ray *r, float ix, float iy, float iz, float cx, float cy, float cz, float size,
                              float t_min, float t_max, float *t_enter, float *t_exit) {
  float lx = (cx + -r->ox) * ix, ly = (cy + -r->oy) * iy, lz = (cz + -r->oz) * iz;
  float ux = ((cx + size) + -r->ox) * ix, uy = ((cy + size) + -r->oy) * iy, uz = ((cz + size) + -r->oz) * iz;
  float mnx = f_min(lx, ux), mny = f_min(ly, uy), mnz = f_min(lz, uz);
  float mxx = f_max(lx, ux), mxy = f_max(ly, uy), mxz = f_max(lz, uz);
  /* t_min / t_max sit in the FIRST operand slot of the compiled min/max chain; for the root
   * call they are constants and Mesa's code generator swaps them into the second slot. */
  *t_enter = f_max(f_max(f_max(t_min, mnx), mny), mnz);
  *t_exit = f_min(f_min(f_min(t_max, mxx), mxy), mxz);
}

/* treeLookup rc:359-394 */
static inline int tree_lookup(const tracer *T, float cx, float cy, float cz,
                              float *inv_pow_depth, float *gx, float *gy, float *gz, uint32_t *value,
                              oracle_stats *st) {
  const oracle_scene *sc = T->sc;
  float ipd = 1.0f, ux = 0.0f, uy = 0.0f, uz = 0.0f;
  uint32_t node_value = 0;
  int is_leaf = 0;
  float two_cc = (float)(int32_t)((uint32_t)T->cell_count << 1);
  for (float i = 0.0f; i < (float)T->max_depth; i = i + 1.0f) {   /* float loop counter, rc:372 */
    ipd = ipd * 0.5f;
    float fx = f_fract(cx), fy = f_fract(cy), fz = f_fract(cz);
    float rx = rintf((((float)node_value + fx) * T->inv_cell_count) * two_cc + -0.5f);   /* round-half-even */
    float ry = rintf(fy * 2.0f + -0.5f);
    float rz = rintf(fz * 2.0f + -0.5f);
    int32_t ix = f2i(rx), iy = f2i(ry), iz = f2i(rz);
    float tx = truncf(rx), ty = truncf(ry), tz = truncf(rz);
    float bx = tx + -(2.0f * floorf(tx / 2.0f));     /* mod(point, 2) rc:378 */
    float by = ty + -(2.0f * floorf(ty / 2.0f));
    float bz = tz + -(2.0f * floorf(tz / 2.0f));
    ux = ux + bx * ipd; uy = uy + by * ipd; uz = uz + bz * ipd;
    uint32_t idx = (((uint32_t)ix << 1) + (uint32_t)iy);
    idx = (idx << 1) + (uint32_t)iz;                  /* AccessIndirectCell rc:184 */
    uint32_t off = idx << 3;
    node_value = ld_u32(sc->cells, sc->cells_bytes, off);
    uint32_t node_type = ld_u32(sc->cells, sc->cells_bytes, off + 4);
    if (st) st->node_loads++;
    if (node_type == 0u || node_type == 2u) { is_leaf = (node_type == 2u); goto done; }
    cx = cx * 2.0f; cy = cy * 2.0f; cz = cz * 2.0f;
  }
  is_leaf = 0;
done:
  *inv_pow_depth = ipd; *gx = ux; *gy = uy; *gz = uz; *value = node_value;
  return is_leaf;
}

/* OctreeHit rc:397-450 with t_min = 0.0003, t_max = +inf (rc:271) */
static int octree_hit(const tracer *T, const ray *r, pixel_carry *pc, hit_record *hit, oracle_stats *st) {
  if (st) st->octree_hit_calls++;
  float ix = f_rcp(r->dx), iy = f_rcp(r->dy), iz = f_rcp(r->dz);      /* rc:319, ray-invariant */
  float t_enter, t_exit;
  {
    /* constants are swapped into the second operand slot here (see cube_slabs) */
    float lx = (T->min_x + -r->ox) * ix, ly = (T->min_y + -r->oy) * iy, lz = (T->min_z + -r->oz) * iz;
    float ux = ((T->min_x + T->scale) + -r->ox) * ix, uy = ((T->min_y + T->scale) + -r->oy) * iy,
          uz = ((T->min_z + T->scale) + -r->oz) * iz;
    float mnx = f_min(lx, ux), mny = f_min(ly, uy), mnz = f_min(lz, uz);
    float mxx = f_max(lx, ux), mxy = f_max(ly, uy), mxz = f_max(lz, uz);
    t_enter = f_max(f_max(f_max(mnx, 0.0003f), mny), mnz);
    t_exit = f_min(f_min(f_min(mxx, INFINITY), mxy), mxz);
  }
  float t_octree_max = INFINITY;
  if (t_exit >= t_enter) {        /* compiled form of !(t_cube_min > t_cube_max): false on NaN */
    cube_hit_record(r, t_enter, T->min_x, T->min_y, T->min_z, T->scale, &pc->root);
    pc->root_t = t_enter;
    t_octree_max = t_exit;
  }
  float t_stride = pc->root_t;    /* rc:408: stale value of the out-temp when the root test missed */
  float inv_pow_depth = 0.5f;     /* rc:403 */

  for (int32_t i = 0; i < T->max_iter && t_stride < t_octree_max; i++) {
    float adv = f_max(0.0001f * (inv_pow_depth + 0.1f), 0.000001f);       /* rc:412 */
    float t = t_stride + adv;
    float wx = t * r->dx + r->ox, wy = t * r->dy + r->oy, wz = t * r->dz + r->oz;
    float lx = (wx + -T->min_x) * T->inv_scale, ly = (wy + -T->min_y) * T->inv_scale, lz = (wz + -T->min_z) * T->inv_scale;
    {   /* rc:417 "fract(p) - p != vec3(0)" as compiled */
      float ex = f_fract(lx) + -lx, ey = f_fract(ly) + -ly, ez = f_fract(lz) + -lz;
      if ((fabsf(ez) + fabsf(ey)) != -fabsf(ex)) return 0;
    }
    float gx, gy, gz; uint32_t value;
    if (st) st->iterations++;
    int leaf = tree_lookup(T, lx, ly, lz, &inv_pow_depth, &gx, &gy, &gz, &value, st);
    if (leaf) {
      const hit_tmp *src = &pc->root;
      if (i > 0) {                                                             /* rc:426-433 */
        float cx = gx * T->scale + T->min_x, cy = gy * T->scale + T->min_y, cz = gz * T->scale + T->min_z;
        float cs = T->scale * inv_pow_depth;
        cube_slabs(r, ix, iy, iz, cx, cy, cz, cs, t_stride, t_octree_max, &t_enter, &t_exit);
        if (!(t_exit < t_enter))
          cube_hit_record(r, t_enter, cx, cy, cz, cs, &pc->leaf);
        src = &pc->leaf;             /* on a miss: whatever this call site produced last time */
      }
      hit->px = src->px; hit->py = src->py; hit->pz = src->pz;
      hit->nx = src->nx; hit->ny = src->ny; hit->nz = src->nz;
      hit->ff = src->ff; hit->index = value;
      return 1;
    }
    /* empty cell: padded cube, only its exit parameter matters (rc:441-446) */
    float cx = (gx * T->scale + T->min_x) + -0.00001f, cy = (gy * T->scale + T->min_y) + -0.00001f,
          cz = (gz * T->scale + T->min_z) + -0.00001f;
    float cs = T->scale * inv_pow_depth + 0.00002f;
    cube_slabs(r, ix, iy, iz, cx, cy, cz, cs, t_stride, t_octree_max, &t_enter, &t_exit);
    t_stride = (!(t_exit < t_enter)) ? t_exit : t_octree_max;
  }
  return 0;
}

/* Rand(vec2) rc:53 */
static inline float rand2(float cx, float cy) {
  return f_fract(oracle_sin(cy * 78.233f + cx * 12.9898f) * 43758.5453f);
}

/* returns 1 when the path continues; writes the scattered ray and attenuation */
static int scatter(const tracer *T, const ray *r, const hit_record *h, ray *out, float atten[3], oracle_stats *st) {
  const oracle_scene *sc = T->sc;
  uint32_t mo = h->index * 12u;
  int32_t type = (int32_t)ld_u32(sc->materials, sc->materials_bytes, mo);            /* rc:278 */
  float dx = r->dx, dy = r->dy, dz = r->dz;
  float nx = h->nx, ny = h->ny, nz = h->nz;
  if (type == 0) {            /* ScatterLambertian rc:470-482 */
    if (st) st->lambertian++;
    float rs = f_rsq(nz * nz + ny * ny + nx * nx);
    float mx = nx * rs, my = ny * rs, mz = nz * rs;          /* constructFrisvad ret[1] rc:455 */
    int sing = nz < -0.9999f;                                /* rc:457 */
    float a = f_rcp(1.0f + nz);
    float b = -((nx * ny) * a);
    float r0x = 1.0f + -((nx * nx) * a);
    float r2y = 1.0f + -((ny * ny) * a);
    float e_b = sing ? -1.0f : b;                            /* ret[0].y and ret[2].x */
    float e_r2y = sing ? 0.0f : r2y;
    float e_r2z = sing ? 0.0f : -ny;
    float e_r0x = sing ? 0.0f : r0x;
    float e_r0z = sing ? 0.0f : -nx;
    float vx = (dz * e_r0z + dy * e_b) + dx * e_r0x;         /* dot(d, ret[0]) rc:472 */
    float vy = (dz * mz + dy * my) + dx * mx;
    float vz = (dz * e_r2z + dy * e_r2y) + dx * e_b;
    /* hash23(RngSample(hit.point)) rc:85-90,230-232: sample_i is the file-scope 0 */
    float p3x = f_fract((100.0f * .1031f) * h->px), p3y = f_fract((100.0f * .1030f) * h->py),
          p3z = f_fract((100.0f * .0973f) * h->pz);   /* "*100" rc:231 and rc:87 folded in fp32 */
    float d = (p3z * (p3x + 33.33f) + p3y * (p3z + 33.33f)) + p3x * (p3y + 33.33f);
    p3x = p3x + d; p3y = p3y + d; p3z = p3z + d;
    float U0 = f_fract((p3x + p3y) * p3z), U1 = f_fract((p3x + p3z) * p3y);
    /* SampleGGXVNDF(-view, (0.85,0.85), U) rc:27-49; signs folded as in the compiled code */
    float sx = vx * 0.85f, sy = vy * 0.85f;
    rs = f_rsq((vz * vz + sy * sy) + sx * sx);
    float va = sx * rs, vb = sy * rs, vc = vz * rs;         /* Vh = (-va,-vb,-vc) */
    float vhz = -vc;
    float lensq = va * va + vb * vb;
    int nz_len = 0.0f < lensq;
    float rl = f_rsq(lensq);
    float t1x = nz_len ? vb * rl : 1.0f;                     /* T1 rc:33 */
    float t1y = nz_len ? -(va * rl) : 0.0f;
    float rr = sqrtf(U0);
    float phi = (2.0f * 3.14159265358f) * U1;               /* rc:14,38 folded in fp32 */
    float sn, cs; sincos_poly(phi, &sn, &cs);
    float t1 = rr * cs, t2 = rr * sn;
    float s = 0.5f * (1.0f + vhz);
    float one_m_t1sq = 1.0f + -(t1 * t1);
    t2 = (1.0f + -s) * sqrtf(one_m_t1sq) + s * t2;
    float T2x = vc * t1y;                                    /* cross(Vh, T1), T1.z = 0 */
    float T2y_neg = vc * t1x;
    float T2z = -(va * t1y) + vb * t1x;
    float nhx = t1 * t1x + t2 * T2x;
    float nhy = t1 * t1y + -(T2y_neg * t2);
    float nhz = t2 * T2z;
    float sq = sqrtf(f_max(one_m_t1sq + -(t2 * t2), 0.0f));
    nhx = nhx + -(va * sq); nhy = nhy + -(vb * sq); nhz = nhz + -(vc * sq);
    float ex = 0.85f * nhx, ey = 0.85f * nhy, ez = f_max(nhz, 0.0f);
    rs = f_rsq((ez * ez + ey * ey) + ex * ex);
    ex = ex * rs; ey = ey * rs; ez = ez * rs;
    float dt = ((ez * dz + ey * dy) + ex * dx) * 2.0f;       /* reflect(d, Ne) rc:474 */
    float sdx = dx + -(dt * ex), sdy = dy + -(dt * ey), sdz = dz + -(dt * ez);
    float qx = nx + sdx, qy = ny + sdy, qz = nz + sdz;       /* rc:475 */
    rs = f_rsq((qz * qz + qy * qy) + qx * qx);
    out->dx = qx * rs; out->dy = qy * rs; out->dz = qz * rs;
    out->ox = h->px; out->oy = h->py; out->oz = h->pz;
    uint32_t ai = ld_u32(sc->materials, sc->materials_bytes, mo + 8u) * 12u;          /* AlbedoColor rc:309-313 */
    atten[0] = ld_f32(sc->albedos, sc->albedos_bytes, ai);
    atten[1] = ld_f32(sc->albedos, sc->albedos_bytes, ai + 4u);
    atten[2] = ld_f32(sc->albedos, sc->albedos_bytes, ai + 8u);
    return 1;
  }
  if (type == 1) {            /* ScatterMetal rc:484-491 */
    if (st) st->metal++;
    float rs = f_rsq((nz * nz + ny * ny) + nx * nx);
    float mx = nx * rs, my = ny * rs, mz = nz * rs;
    float dt = ((mz * dz + my * dy) + mx * dx) * 2.0f;
    float rx = dx + -(dt * mx), ry = dy + -(dt * my), rz = dz + -(dt * mz);
    uint32_t at = ld_u32(sc->materials, sc->materials_bytes, mo + 4u);
    float fuzz = ld_f32(sc->metal, sc->metal_bytes, at << 2);
    /* RandInHemisphere(hit.point.xy, n) as compiled: ONE cube sample in [-1,1]^3, rc:106-115 */
    float hx = -1.0f + 2.0f * rand2(h->px, h->py);
    float hy = -1.0f + 2.0f * rand2(h->px + hx, h->py + hx);
    float hz = -1.0f + 2.0f * rand2(h->px + hy, h->py + hy);
    int same = -(hz * nz + hy * ny) < hx * nx;
    if (!same) { hx = -hx; hy = -hy; hz = -hz; }
    float qx = rx + fuzz * hx, qy = ry + fuzz * hy, qz = rz + fuzz * hz;
    rs = f_rsq((qz * qz + qy * qy) + qx * qx);
    qx = qx * rs; qy = qy * rs; qz = qz * rs;
    out->dx = qx; out->dy = qy; out->dz = qz;
    out->ox = h->px; out->oy = h->py; out->oz = h->pz;
    uint32_t ai = ld_u32(sc->materials, sc->materials_bytes, mo + 8u) * 12u;
    atten[0] = ld_f32(sc->albedos, sc->albedos_bytes, ai);
    atten[1] = ld_f32(sc->albedos, sc->albedos_bytes, ai + 4u);
    atten[2] = ld_f32(sc->albedos, sc->albedos_bytes, ai + 8u);
    return -(qz * nz + qy * ny) < qx * nx;                  /* dot(scattered.dir, n) > 0 rc:490 */
  }
  if (type == 2) {            /* ScatterDielectric rc:499-522 */
    if (st) st->dielectric++;
    uint32_t at = ld_u32(sc->materials, sc->materials_bytes, mo + 4u);
    float ir = ld_f32(sc->dielectric, sc->dielectric_bytes, at << 2);
    float ratio = h->ff ? f_rcp(ir) : ir;
    float pz_ = dz * nz, py_ = dy * ny, px_ = dx * nx;
    float cos_t = f_min((-pz_ + -py_) + -px_, 1.0f);
    float sin_t = sqrtf(1.0f + -(cos_t * cos_t));
    int cannot = 1.0f < ratio * sin_t;
    float q = (1.0f + -ratio) / (1.0f + ratio);
    float r0 = q * q;                                        /* pow(.,2) rc:495 */
    float rnd = rand2(h->px, h->py);
    float refl = oracle_pow(1.0f + -cos_t, 5.0f) * (1.0f + -r0) + r0;   /* rc:496 */
    float ox_, oy_, oz_;
    float dn = (pz_ + py_) + px_;
    if (cannot || (rnd < refl)) {                            /* reflect rc:515 */
      float dt = dn * 2.0f;
      ox_ = dx + -(dt * nx); oy_ = dy + -(dt * ny); oz_ = dz + -(dt * nz);
    } else {                                                 /* refract rc:517 (Mesa lowering) */
      float k = 1.0f + -(ratio * (ratio * (1.0f + -(dn * dn))));
      if (!(k < 0.0f)) {
        float m = ratio * dn + sqrtf(k);
        ox_ = ratio * dx + -(m * nx); oy_ = ratio * dy + -(m * ny); oz_ = ratio * dz + -(m * nz);
      } else { ox_ = 0.0f; oy_ = 0.0f; oz_ = 0.0f; }
    }
    float rs = f_rsq((oz_ * oz_ + oy_ * oy_) + ox_ * ox_);
    out->dx = ox_ * rs; out->dy = oy_ * rs; out->dz = oz_ * rs;
    out->ox = h->px; out->oy = h->py; out->oz = h->pz;
    atten[0] = 1.0f; atten[1] = 1.0f; atten[2] = 1.0f;
    return 1;
  }
  if (st) st->unknown_material++;
  return 0;                    /* default: rc:288-291 */
}

/* RayColor rc:264-302 */
static void ray_color(const tracer *T, ray r, pixel_carry *pc, float rgb[3], oracle_stats *st) {
  float ar = 1.0f, ag = 1.0f, ab = 1.0f;
  int32_t loop_count = 0;
  hit_record h;
  while (loop_count < T->cam.max_bounce && octree_hit(T, &r, pc, &h, st)) {
    loop_count += 1;
    ray nr; float at[3];
    if (!scatter(T, &r, &h, &nr, at, st)) break;
    ar = ar * at[0]; ag = ag * at[1]; ab = ab * at[2];
    r = nr;
  }
  if (loop_count > 0) { rgb[0] = ar; rgb[1] = ag; rgb[2] = ab; return; }
  float yp = r.dy + 1.0f;                                   /* sky rc:299-300 as compiled */
  float w = 1.0f + -(0.5f * yp);
  rgb[0] = w + 0.25f * yp; rgb[1] = w + 0.35f * yp; rgb[2] = 1.0f;
}

/* one sample of main()'s loop body rc:240-246 */
static void sample_pixel(const tracer *T, int px, int py, int s, pixel_carry *pc, float sum[3], oracle_stats *st) {
  const oracle_camera *c = &T->cam;
  float x = (float)px, y = (float)py, fs = (float)s;
  const float K = 0.2f * .1031f;        /* "* 0.2" rc:243 and "* .1031" rc:72 folded in fp32 */
  float a = f_fract(K * (x + fs)), b = f_fract(K * y);
  float d = (a + 33.33f) * (a + b) + a * (b + 33.33f);
  float h1 = f_fract(((a + d) + (b + d)) * (a + d));
  float a2 = f_fract(K * x), b2 = f_fract(K * (y + fs));
  float d2 = (a2 + 33.33f) * (a2 + b2) + a2 * (b2 + 33.33f);
  float h2 = f_fract(((a2 + d2) + (b2 + d2)) * (a2 + d2));
  float u = (x + h1) / (float)(c->image_width - 1);
  float v = (y + h2) / (float)(c->image_height - 1);
  /* CameraGetRay rc:304-307 */
  float rx = (c->horizontal[0] * u + c->lower_left_corner[0]) + (v * c->vertical[0] + -c->origin[0]);
  float ry = (c->horizontal[1] * u + c->lower_left_corner[1]) + (v * c->vertical[1] + -c->origin[1]);
  float rz = (c->horizontal[2] * u + c->lower_left_corner[2]) + (v * c->vertical[2] + -c->origin[2]);
  float rs = f_rsq((rz * rz + ry * ry) + rx * rx);
  ray r = { c->origin[0], c->origin[1], c->origin[2], rx * rs, ry * rs, rz * rs };
  float rgb[3];
  ray_color(T, r, pc, rgb, st);
  sum[0] = sum[0] + rgb[0]; sum[1] = sum[1] + rgb[1]; sum[2] = sum[2] + rgb[2];
}

/* ---------------------------------------------------------------- driver ---------------- */
typedef struct {
  tracer T;
  int x_end, y_begin, y_end;        /* pixel ranges actually covered */
  int spp_begin, spp_count;
  int mode;                          /* 0 = full render, 1 = accumulate */
  float *out;                        /* image or accum */
  float *carry;                      /* optional W*H*16 floats of per-pixel carry (accumulate) */
  int next_row;                      /* dynamic row scheduler */
  pthread_mutex_t mu;
  oracle_stats st; int want_stats;
} job;

static void run_rows(job *J) {
  oracle_stats local; memset(&local, 0, sizeof local);
  oracle_stats *st = J->want_stats ? &local : NULL;
  const int W = J->T.cam.image_width;
  for (;;) {
    pthread_mutex_lock(&J->mu);
    int y = J->next_row++;
    pthread_mutex_unlock(&J->mu);
    if (y >= J->y_end) break;
    for (int x = 0; x < J->x_end; x++) {
      float *px = J->out + ((size_t)y * W + x) * 4;
      pixel_carry pc; memset(&pc, 0, sizeof pc);
      float sum[3] = { 0.0f, 0.0f, 0.0f };
      if (J->mode == 1) {
        sum[0] = px[0]; sum[1] = px[1]; sum[2] = px[2];
        if (J->carry) memcpy(&pc, J->carry + ((size_t)y * W + x) * 16, sizeof pc);
      }
      for (int s = J->spp_begin; s < J->spp_begin + J->spp_count; s++) sample_pixel(&J->T, x, y, s, &pc, sum, st);
      if (J->mode == 1) {
        px[0] = sum[0]; px[1] = sum[1]; px[2] = sum[2];
        if (J->carry) memcpy(J->carry + ((size_t)y * W + x) * 16, &pc, sizeof pc);
      } else {
        float n = (float)J->T.cam.samples_per_pixel;       /* rc:249-251 */
        px[0] = f_min(f_max(sqrtf(sum[0] / n), 0.0f), 1.0f);
        px[1] = f_min(f_max(sqrtf(sum[1] / n), 0.0f), 1.0f);
        px[2] = f_min(f_max(sqrtf(sum[2] / n), 0.0f), 1.0f);
        px[3] = 1.0f;
      }
      if (st) { st->pixels++; st->samples += (uint64_t)J->spp_count; }
    }
  }
  if (J->want_stats) {
    pthread_mutex_lock(&J->mu);
    uint64_t *dst = (uint64_t *)&J->st; const uint64_t *src = (const uint64_t *)&local;
    for (size_t i = 0; i < sizeof(oracle_stats) / sizeof(uint64_t); i++) dst[i] += src[i];
    pthread_mutex_unlock(&J->mu);
  }
}
static void *thread_main(void *p) { run_rows((job *)p); return NULL; }

static void covered(const oracle_camera *cam, int dispatch_w, int dispatch_h, int *x_end, int *y_end) {
  /* ComputeShader::dispatch_compute: groups = max(dim / 32, 1) (compute_shader.rs:30-32);
   * stores outside the image are dropped */
  int gx = dispatch_w / 32; if (gx < 1) gx = 1;
  int gy = dispatch_h / 32; if (gy < 1) gy = 1;
  *x_end = gx * 32 < cam->image_width ? gx * 32 : cam->image_width;
  *y_end = gy * 32 < cam->image_height ? gy * 32 : cam->image_height;
}

static int setup(job *J, const oracle_scene *scene, const oracle_camera *cam, int dispatch_w, int dispatch_h,
                 int row_begin, int row_end) {
  memset(J, 0, sizeof *J);
  J->T.sc = scene; J->T.cam = *cam;
  const void *of = scene->octree_floats; size_t ofb = scene->octree_floats_bytes;
  J->T.min_x = ld_f32(of, ofb, 0); J->T.min_y = ld_f32(of, ofb, 4); J->T.min_z = ld_f32(of, ofb, 8);
  J->T.scale = ld_f32(of, ofb, 16); J->T.inv_scale = ld_f32(of, ofb, 20); J->T.inv_cell_count = ld_f32(of, ofb, 24);
  const void *oi = scene->octree_ints; size_t oib = scene->octree_ints_bytes;
  J->T.max_depth = (int32_t)ld_u32(oi, oib, 0); J->T.max_iter = (int32_t)ld_u32(oi, oib, 4);
  J->T.cell_count = (int32_t)ld_u32(oi, oib, 8);
  int ye; covered(cam, dispatch_w, dispatch_h, &J->x_end, &ye);
  J->y_begin = row_begin < 0 ? 0 : row_begin;
  J->y_end = row_end < ye ? row_end : ye;
  J->next_row = J->y_begin;
  pthread_mutex_init(&J->mu, NULL);
  return 0;
}

static void launch(job *J, int nthreads) {
  if (nthreads <= 1) { run_rows(J); return; }
  if (nthreads > 256) nthreads = 256;
  pthread_t th[256];
  for (int i = 0; i < nthreads; i++) pthread_create(&th[i], NULL, thread_main, J);
  for (int i = 0; i < nthreads; i++) pthread_join(th[i], NULL);
}

int oracle_render(const oracle_scene *scene, const oracle_camera *cam, int dispatch_w, int dispatch_h,
                  int row_begin, int row_end, float *image, int nthreads, oracle_stats *stats) {
  job J; setup(&J, scene, cam, dispatch_w, dispatch_h, row_begin, row_end);
  J.mode = 0; J.out = image; J.spp_begin = 0; J.spp_count = cam->samples_per_pixel; J.want_stats = stats != NULL;
  launch(&J, nthreads);
  if (stats) *stats = J.st;
  pthread_mutex_destroy(&J.mu);
  return 0;
}

int oracle_accumulate_carry(const oracle_scene *scene, const oracle_camera *cam, int dispatch_w, int dispatch_h,
                            int row_begin, int row_end, int spp_begin, int spp_count, float *accum, float *carry,
                            int nthreads, oracle_stats *stats) {
  job J; setup(&J, scene, cam, dispatch_w, dispatch_h, row_begin, row_end);
  J.mode = 1; J.out = accum; J.carry = carry; J.spp_begin = spp_begin; J.spp_count = spp_count; J.want_stats = stats != NULL;
  launch(&J, nthreads);
  if (stats) *stats = J.st;
  pthread_mutex_destroy(&J.mu);
  return 0;
}

int oracle_accumulate(const oracle_scene *scene, const oracle_camera *cam, int dispatch_w, int dispatch_h,
                      int row_begin, int row_end, int spp_begin, int spp_count, float *accum, int nthreads,
                      oracle_stats *stats) {
  return oracle_accumulate_carry(scene, cam, dispatch_w, dispatch_h, row_begin, row_end, spp_begin, spp_count, accum,
                                 NULL, nthreads, stats);
}

int oracle_resolve(const oracle_camera *cam, int dispatch_w, int dispatch_h, int row_begin, int row_end,
                   int total_spp, const float *accum, float *image) {
  int xe, ye; covered(cam, dispatch_w, dispatch_h, &xe, &ye);
  if (row_begin < 0) row_begin = 0;
  if (row_end > ye) row_end = ye;
  float n = (float)total_spp;
  for (int y = row_begin; y < row_end; y++)
    for (int x = 0; x < xe; x++) {
      size_t o = ((size_t)y * cam->image_width + x) * 4;
      for (int c = 0; c < 3; c++) image[o + c] = f_min(f_max(sqrtf(accum[o + c] / n), 0.0f), 1.0f);
      image[o + 3] = 1.0f;
    }
  return 0;
}

/* ---------------------------------------------------------------- octree_update.comp ---- */
/* uc:N = assets/shaders/octree_update.comp line N.  float -> uint as llvmpipe's f2u32 (values used by
 * the reference host are small non-negative integers, main.rs:566-567). */
static inline uint32_t f2u(float f) {
  if (!(f > -1.0f)) return 0u;
  if (f >= 4294967296.0f) return 0xFFFFFFFFu;
  return (uint32_t)f;
}
static inline void st_u32(void *buf, size_t bytes, uint32_t byte_off, uint32_t v) {
  if ((size_t)(byte_off >> 2) >= (bytes >> 2)) return;            /* robust access: dropped */
  memcpy((char *)buf + (byte_off & ~3u), &v, 4);
}

static void update_invocation(void *cells, size_t cb, const void *delta, size_t db, float inv_cell_count,
                              int32_t max_depth, int32_t cell_count, uint32_t *counter, uint32_t delta_index) {
  const uint32_t dof = delta_index << 5;                                   /* DeltaNode stride 32 */
  float cx = ld_f32(delta, db, dof), cy = ld_f32(delta, db, dof + 4), cz = ld_f32(delta, db, dof + 8);
  const float d_type = ld_f32(delta, db, dof + 12), d_value = ld_f32(delta, db, dof + 16);
  const float two_cc = (float)(int32_t)((uint32_t)cell_count << 1);
  uint32_t node_value = 0, index = 0;                                      /* index: undefined when the loop is empty */
  for (float i = 0.0f; i < (float)(max_depth - 1); i = i + 1.0f) {        /* treeLookupLeaf uc:57-80 */
    float fx = f_fract(cx), fy = f_fract(cy), fz = f_fract(cz);
    float rx = rintf((((float)node_value + fx) * inv_cell_count) * two_cc + -0.5f);
    float ry = rintf(fy * 2.0f + -0.5f), rz = rintf(fz * 2.0f + -0.5f);
    index = ((((uint32_t)f2i(rx) << 1) + (uint32_t)f2i(ry)) << 1) + (uint32_t)f2i(rz);
    const uint32_t off = index << 3;
    /* atomicCompSwap(type, EMPTY, PARENT) == EMPTY -> allocate a cell (uc:72-74) */
    {   /* an out-of-range atomic returns 0 and writes nothing (llvmpipe), so it still takes a counter value */
      const uint32_t old = ld_u32(cells, cb, off + 4);
      if (old == 0u) {
        st_u32(cells, cb, off + 4, 1u);
        st_u32(cells, cb, off, (*counter)++);
      }
    }
    node_value = ld_u32(cells, cb, off);                                   /* node = indirect_cells[index] uc:76 */
    cx = cx * 2.0f; cy = cy * 2.0f; cz = cz * 2.0f;
  }
  st_u32(cells, cb, index << 3, f2u(d_value));                             /* uc:101 */
  st_u32(cells, cb, (index << 3) + 4, f2u(d_type));
}

int oracle_octree_update(void *cells, size_t cells_bytes, const void *delta, size_t delta_bytes,
                         const void *octree_floats, size_t octree_floats_bytes,
                         const void *octree_ints, size_t octree_ints_bytes, uint32_t *counter,
                         int dispatch_w, int dispatch_h, int dispatch_d) {
  const float inv_cell_count = ld_f32(octree_floats, octree_floats_bytes, 24);
  const int32_t max_depth = (int32_t)ld_u32(octree_ints, octree_ints_bytes, 0);
  const int32_t cell_count = (int32_t)ld_u32(octree_ints, octree_ints_bytes, 8);
  const int gx = dispatch_w < 1 ? 1 : dispatch_w, gy = dispatch_h < 1 ? 1 : dispatch_h, gz = dispatch_d < 1 ? 1 : dispatch_d;
  for (int z = 0; z < gz; z++)
    for (int y = 0; y < gy; y++)
      for (int x = 0; x < gx; x++)
        update_invocation(cells, cells_bytes, delta, delta_bytes, inv_cell_count, max_depth, cell_count, counter,
                          (uint32_t)x + (uint32_t)y + (uint32_t)z);            /* uc:99 */
  return 0;
}
