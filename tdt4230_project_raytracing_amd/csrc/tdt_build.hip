// Scene ingest on the GPU — SURVEY §8f-1: the step the reference never wrote (its call is commented out, main.rs:218-224):
// the voxel list of ply_point_loader::from_resources (ply_point_loader.rs:102-319) -> the indirect-cell octree that
// raytracer.comp reads (Node{value,type}, 8 nodes per cell, node index = cell*8 + x*4 + y*2 + z; rc:172-184,375-376).
// Byte-identical to the host builder of libtdthost.so (csrc/host_scene.cpp build_octree / tdt_scene_from_ply), which
// works on a dense grid + pyramid + FIFO queue; here nothing is dense and nothing is sequential:
//
//   keys      one lane per voxel: grid coordinates -> Morton key whose 3-bit digits, most significant level first, are the
//             child index x*4 + y*2 + z of each level (so key order inside a level IS the host's breadth-first order)
//   sort      stable LSD radix sort, 8-bit digits, 4 passes (keys <= 30 bits + the all-ones "dropped" sentinel): per-tile
//             LDS histograms -> one prefix sum over [digit][tile] -> scatter with wave-ballot ranking (stable: of duplicate
//             voxels the LAST in file order must win, as Grid::set overwrites)
//   unique    flag the last of every run of equal keys, prefix sum, compact  -> the leaf level D
//   levels    for l = D-1 .. 1: flag segment heads (key >> 3 changes), prefix sum = parent index, one lane per head reduces
//             its <= 8 children: all eight present, same material, none MIXED -> uniform (one LEAF at this level) else MIXED
//   number    ONE prefix sum over the MIXED flags of levels 1..D-1 laid end to end = breadth-first cell numbers - 1
//   emit      one lane per node of every level whose parent is MIXED writes its (value, type) into its parent's cell
//
// All of it is HBM-streaming integer work (a few 4-byte reads and writes per voxel and pass); the 156 942-voxel
// monument of the reference builds in well under a millisecond of kernel time.  Two host synchronisations: one to size the
// grid from the voxel extent (from_points only), one to size the cells buffer from the cell count.
#include <cstring>
#include <new>
#include <vector>

#include "device_scan.hpp"
#include "tdt_internal.hpp"

namespace tdt {

constexpr uint32_t kDropped = 0xFFFFFFFFu;   // key of a voxel outside the grid: sorts behind everything
constexpr uint32_t kMixed = 0xFFu;           // host_scene.cpp MIXED
constexpr uint32_t kSortTile = 2048;         // items per 256-thread block and pass

struct KeyArgs {
  const int32_t *vox; uint32_t n;
  int ax, ay, az;                            // which file axis feeds octree x, y, z
  int32_t sub[3], add[3];                    // grid = file - sub + add, per octree axis
  int depth;
  const uint32_t *pal_keys; uint32_t n_pal; const uint32_t *pal_rank;   // null: voxel[3] is material + 1 already
};

__device__ __forceinline__ uint32_t spread3(uint32_t v) {      // 10 bits -> every third bit
  v = (v | (v << 16)) & 0x030000FFu;
  v = (v | (v << 8)) & 0x0300F00Fu;
  v = (v | (v << 4)) & 0x030C30C3u;
  v = (v | (v << 2)) & 0x09249249u;
  return v;
}
__device__ __forceinline__ int pal_find(const uint32_t *keys, uint32_t n, uint32_t k) {
  uint32_t lo = 0, hi = n;
  while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (keys[mid] < k) lo = mid + 1; else hi = mid; }
  return (lo < n && keys[lo] == k) ? (int)lo : -1;
}

// per-axis maxima of the voxel list (tdt_scene_from_ply's `mx`)
__global__ __launch_bounds__(256) void build_extent_kernel(const int32_t *vox, uint32_t n, int32_t *mx) {
  int32_t m[3] = {INT32_MIN, INT32_MIN, INT32_MIN};
  for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u)
    for (int a = 0; a < 3; a++) { const int32_t v = vox[4 * (size_t)i + a]; m[a] = v > m[a] ? v : m[a]; }
  for (int a = 0; a < 3; a++) {
    for (int o = 32; o > 0; o >>= 1) { const int32_t t = __shfl_xor(m[a], o, 64); m[a] = t > m[a] ? t : m[a]; }
    if ((threadIdx.x & 63) == 0) atomicMax(&mx[a], m[a]);
  }
}

// which palette entries a voxel really uses (the host builds its materials from those, not from the whole palette)
__global__ __launch_bounds__(256) void build_palette_mark_kernel(const int32_t *vox, uint32_t n, const uint32_t *pal_keys, uint32_t n_pal,
                                                                uint32_t *used, uint32_t *err) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  const int p = pal_find(pal_keys, n_pal, (uint32_t)vox[4 * (size_t)i + 3]);
  if (p < 0) atomicOr(err, 1u); else used[p] = 1u;
}

__global__ __launch_bounds__(256) void build_keys_kernel(const KeyArgs A, uint32_t *keys, uint32_t *vals) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= A.n) return;
  const int32_t *v = A.vox + 4 * (size_t)i;
  const long long x = (long long)v[A.ax] - A.sub[0] + A.add[0], y = (long long)v[A.ay] - A.sub[1] + A.add[1],
                  z = (long long)v[A.az] - A.sub[2] + A.add[2];
  const long long N = 1ll << A.depth;
  uint32_t m;
  if (A.pal_keys) { const int p = pal_find(A.pal_keys, A.n_pal, (uint32_t)v[3]); m = p < 0 ? 0u : 1u + A.pal_rank[p]; }
  else m = (uint32_t)v[3];
  const bool ok = x >= 0 && y >= 0 && z >= 0 && x < N && y < N && z < N && m >= 1u && m < kMixed;
  keys[i] = ok ? ((spread3((uint32_t)x) << 2) | (spread3((uint32_t)y) << 1) | spread3((uint32_t)z)) : kDropped;
  vals[i] = m;
}

// ---- stable LSD radix sort, one 8-bit digit per pass -------------------------------------------------------------------
__global__ __launch_bounds__(256) void sort_hist_kernel(const uint32_t *keys, uint32_t n, int shift, uint32_t tiles, uint32_t *hist) {
  __shared__ uint32_t s_bin[256];
  s_bin[threadIdx.x] = 0;
  __syncthreads();
  const uint32_t lo = blockIdx.x * kSortTile;
  for (uint32_t r = 0; r < kSortTile; r += 256u) {
    const uint32_t i = lo + r + threadIdx.x;
    if (i < n) atomicAdd(&s_bin[(keys[i] >> shift) & 255u], 1u);
  }
  __syncthreads();
  hist[threadIdx.x * tiles + blockIdx.x] = s_bin[threadIdx.x];       // [digit][tile]: one prefix sum orders digits, then tiles
}

__global__ __launch_bounds__(256) void sort_scatter_kernel(const uint32_t *keys, const uint32_t *vals, uint32_t n, int shift, uint32_t tiles,
                                                          const uint32_t *offsets, uint32_t *keys_out, uint32_t *vals_out) {
  __shared__ uint32_t s_off[256], s_wave[4][256];
  s_off[threadIdx.x] = offsets[threadIdx.x * tiles + blockIdx.x];
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const uint32_t lo = blockIdx.x * kSortTile;
  for (uint32_t r = 0; r < kSortTile; r += 256u) {                   // rounds in index order: ranks below keep file order
    for (int w = 0; w < 4; w++) s_wave[w][threadIdx.x] = 0;
    __syncthreads();
    const uint32_t i = lo + r + threadIdx.x;
    const bool valid = i < n;
    const uint32_t k = valid ? keys[i] : 0u, v = valid ? vals[i] : 0u, d = (k >> shift) & 255u;
    unsigned long long peers = __ballot(valid);                      // lanes of this wave with the same digit
#pragma unroll
    for (int b = 0; b < 8; b++) {
      const bool bit = (d >> b) & 1u;
      const unsigned long long m = __ballot(valid && bit);
      peers &= bit ? m : ~m;
    }
    const uint32_t rank = (uint32_t)__popcll(peers & ((1ull << lane) - 1ull));
    if (valid && rank == 0) s_wave[wave][d] = (uint32_t)__popcll(peers);
    __syncthreads();
    if (valid) {
      uint32_t pos = s_off[d] + rank;
      for (uint32_t w = 0; w < wave; w++) pos += s_wave[w][d];
      keys_out[pos] = k; vals_out[pos] = v;
    }
    __syncthreads();
    s_off[threadIdx.x] += s_wave[0][threadIdx.x] + s_wave[1][threadIdx.x] + s_wave[2][threadIdx.x] + s_wave[3][threadIdx.x];
    __syncthreads();
  }
}

// ---- leaf level: of every run of equal keys keep the LAST (file order: Grid::set overwrites) ---------------------------
__global__ __launch_bounds__(256) void build_last_flags_kernel(const uint32_t *keys, uint32_t n, uint32_t *flag /* n + 1 */) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i > n) return;
  flag[i] = (i < n && keys[i] != kDropped && (i + 1u == n || keys[i + 1u] != keys[i])) ? 1u : 0u;
}
__global__ __launch_bounds__(256) void build_compact_kernel(const uint32_t *keys, const uint32_t *vals, uint32_t n, const uint32_t *excl /* n + 1 */,
                                                           uint32_t *lk, uint32_t *lv, uint32_t *count) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i == 0) *count = excl[n];
  if (i >= n) return;
  if (excl[i + 1u] != excl[i]) { lk[excl[i]] = keys[i]; lv[excl[i]] = vals[i]; }
}

// ---- one level up: children (keys ck, values cv, *c_count of them) -> parents ------------------------------------------
__global__ __launch_bounds__(256) void build_head_flags_kernel(const uint32_t *ck, const uint32_t *c_count, uint32_t cap, uint32_t *flag /* cap + 1 */) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i > cap) return;
  flag[i] = (i < *c_count && (i == 0 || (ck[i] >> 3) != (ck[i - 1u] >> 3))) ? 1u : 0u;
}
__global__ __launch_bounds__(256) void build_reduce_kernel(const uint32_t *ck, const uint32_t *cv, const uint32_t *c_count, uint32_t cap,
                                                          const uint32_t *excl /* cap + 1 */, uint32_t *c_parent, uint32_t *pk, uint32_t *pv, uint32_t *p_count) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i == 0) *p_count = excl[cap];
  const uint32_t n = *c_count;
  if (i >= n || excl[i + 1u] == excl[i]) return;      // not a segment head
  const uint32_t p = excl[i], key = ck[i] >> 3, first = cv[i];
  uint32_t c = 0; bool same = true;
  for (uint32_t j = i; j < n && j < i + 8u && (ck[j] >> 3) == key; j++) { same = same && cv[j] == first; c_parent[j] = p; c++; }
  pk[p] = key;
  pv[p] = (c == 8u && same && first != kMixed) ? first : kMixed;        // eight equal children: one LEAF at this level
}

// MIXED flags of level l into the end-to-end array (capacity offsets; the tail of a level stays 0)
__global__ __launch_bounds__(256) void build_mixed_flags_kernel(const uint32_t *lv, const uint32_t *count, uint32_t cap, uint32_t *flag) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= cap) return;
  flag[i] = (i < *count && lv[i] == kMixed) ? 1u : 0u;
}

// nodes of level l (1-based; its cells are numbered by the parents of level l-1) -> (value, type) in their parent's cell
__global__ __launch_bounds__(256) void build_emit_kernel(const uint32_t *lk, const uint32_t *lv, const uint32_t *count, const uint32_t *parent /* null: level 1 */,
                                                        const uint32_t *pv, const uint32_t *p_rank /* cell number - 1 of the parents */,
                                                        const uint32_t *my_rank /* null: the leaf level */, uint32_t *cells, uint32_t n_cells) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= *count) return;
  uint32_t cell = 0;
  if (parent) {
    const uint32_t p = parent[i];
    if (pv[p] != kMixed) return;                      // inside a uniform block: the parent is a LEAF
    cell = 1u + p_rank[p];
  }
  if (cell >= n_cells) return;
  const uint32_t v = lv[i];
  uint32_t *node = cells + (size_t)cell * 16u + (lk[i] & 7u) * 2u;
  if (v == kMixed) { node[0] = 1u + my_rank[i]; node[1] = 1u; }          // PARENT of the cell this node's children fill
  else { node[0] = v - 1u; node[1] = 2u; }                             // LEAF: the material index
}

// materials {LAMBERTIAN, 0, m} and albedos rgb / 255 for the palette entries in use, in key order (tdt_scene_from_ply)
__global__ __launch_bounds__(256) void build_materials_kernel(const uint32_t *used, const uint32_t *rank, const uint8_t *rgb, uint32_t n_pal,
                                                             uint32_t *materials, float *albedos) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n_pal || !used[i]) return;
  const uint32_t m = rank[i];
  materials[3u * m] = 0u; materials[3u * m + 1u] = 0u; materials[3u * m + 2u] = m;
  for (int c = 0; c < 3; c++) albedos[3u * m + c] = (float)rgb[3u * i + c] / 255.0f;
}

namespace {

struct DeviceArena {          // temporaries of one build, freed together
  tdt_ctx *ctx; std::vector<void *> ptrs;
  explicit DeviceArena(tdt_ctx *c) : ctx(c) {}
  ~DeviceArena() { for (void *p : ptrs) (void)hipFree(p); }
  template <class T> T *get(size_t n) {
    void *p = nullptr;
    if (hipMalloc(&p, (n ? n : 1) * sizeof(T)) != hipSuccess) return nullptr;
    ptrs.push_back(p);
    return (T *)p;
  }
};
#define TDT_ALLOC(var, T, n) T *var = arena.get<T>(n); if (!var) return fail(ctx, TDT_ERR_HIP, "out of device memory in the octree builder")
inline unsigned blocks(size_t n) { return (unsigned)((n + 255) / 256); }

// voxels already on the device -> a cells buffer; pal_* null: vox[3] is material + 1
int build_cells_device(tdt_ctx *ctx, const int32_t *d_vox, uint32_t n, const KeyArgs &key_args, tdt_buffer **out, uint32_t *n_cells_out) {
  hipStream_t st = ctx->stream;
  const int D = key_args.depth;
  DeviceArena arena(ctx);
  TDT_ALLOC(k0, uint32_t, n); TDT_ALLOC(k1, uint32_t, n); TDT_ALLOC(v0, uint32_t, n); TDT_ALLOC(v1, uint32_t, n);
  KeyArgs A = key_args; A.vox = d_vox; A.n = n;
  hipLaunchKernelGGL(build_keys_kernel, dim3(blocks(n)), dim3(256), 0, st, A, k0, v0);
  // sort
  const uint32_t tiles = (n + kSortTile - 1) / kSortTile;
  TDT_ALLOC(hist, uint32_t, (size_t)256 * tiles);
  TDT_ALLOC(hscr, uint32_t, scan_scratch_words((size_t)256 * tiles));
  for (int pass = 0; pass < 4; pass++) {
    hipLaunchKernelGGL(sort_hist_kernel, dim3(tiles), dim3(256), 0, st, (const uint32_t *)k0, n, pass * 8, tiles, hist);
    TDT_HIP(ctx, exclusive_scan_u32(st, hist, hist, 256u * tiles, hscr));
    hipLaunchKernelGGL(sort_scatter_kernel, dim3(tiles), dim3(256), 0, st, (const uint32_t *)k0, (const uint32_t *)v0, n, pass * 8, tiles,
                       (const uint32_t *)hist, k1, v1);
    std::swap(k0, k1); std::swap(v0, v1);
  }
  // level arrays: capacity of level l = min(n, 8^l)
  std::vector<uint32_t> cap(D + 1);
  for (int l = 1; l <= D; l++) { const unsigned long long c = 1ull << (3 * l); cap[l] = c < n ? (uint32_t)c : n; }
  std::vector<uint32_t *> lk(D + 1, nullptr), lv(D + 1, nullptr), par(D + 1, nullptr);
  for (int l = 1; l <= D; l++) {
    lk[l] = arena.get<uint32_t>(cap[l]); lv[l] = arena.get<uint32_t>(cap[l]); par[l] = arena.get<uint32_t>(cap[l]);
    if (!lk[l] || !lv[l] || !par[l]) return fail(ctx, TDT_ERR_HIP, "out of device memory in the octree builder");
  }
  TDT_ALLOC(count, uint32_t, D + 2);
  TDT_ALLOC(flag, uint32_t, (size_t)n + 1); TDT_ALLOC(fscr, uint32_t, scan_scratch_words((size_t)n + 1));
  hipLaunchKernelGGL(build_last_flags_kernel, dim3(blocks((size_t)n + 1)), dim3(256), 0, st, (const uint32_t *)k0, n, flag);
  TDT_HIP(ctx, exclusive_scan_u32(st, flag, flag, n + 1u, fscr));
  hipLaunchKernelGGL(build_compact_kernel, dim3(blocks(n)), dim3(256), 0, st, (const uint32_t *)k0, (const uint32_t *)v0, n, (const uint32_t *)flag,
                     lk[D], lv[D], count + D);
  for (int l = D - 1; l >= 1; l--) {
    const uint32_t cc = cap[l + 1];
    hipLaunchKernelGGL(build_head_flags_kernel, dim3(blocks((size_t)cc + 1)), dim3(256), 0, st, (const uint32_t *)lk[l + 1], (const uint32_t *)(count + l + 1), cc, flag);
    TDT_HIP(ctx, exclusive_scan_u32(st, flag, flag, cc + 1u, fscr));
    hipLaunchKernelGGL(build_reduce_kernel, dim3(blocks(cc)), dim3(256), 0, st, (const uint32_t *)lk[l + 1], (const uint32_t *)lv[l + 1], (const uint32_t *)(count + l + 1), cc,
                       (const uint32_t *)flag, par[l + 1], lk[l], lv[l], count + l);
  }
  // breadth-first numbers: one prefix sum over the MIXED flags of levels 1..D-1 laid end to end
  std::vector<size_t> off(D + 1, 0);
  size_t total_cap = 0;
  for (int l = 1; l < D; l++) { off[l] = total_cap; total_cap += cap[l]; }
  TDT_ALLOC(mixed, uint32_t, total_cap + 1); TDT_ALLOC(mscr, uint32_t, scan_scratch_words(total_cap + 1));
  TDT_HIP(ctx, hipMemsetAsync(mixed + total_cap, 0, sizeof(uint32_t), st));
  for (int l = 1; l < D; l++)
    hipLaunchKernelGGL(build_mixed_flags_kernel, dim3(blocks(cap[l])), dim3(256), 0, st, (const uint32_t *)lv[l], (const uint32_t *)(count + l), cap[l], mixed + off[l]);
  TDT_HIP(ctx, exclusive_scan_u32(st, mixed, mixed, (uint32_t)(total_cap + 1), mscr));
  TDT_HIP(ctx, hipGetLastError());
  uint32_t n_mixed = 0;
  TDT_HIP(ctx, hipMemcpyAsync(&n_mixed, mixed + total_cap, sizeof n_mixed, hipMemcpyDeviceToHost, st));
  TDT_HIP(ctx, hipStreamSynchronize(st));             // the one thing the host must know: how large the cells buffer is
  const uint32_t n_cells = 1u + n_mixed;
  const size_t bytes = (size_t)n_cells * 64;
  void *d_cells = nullptr;
  TDT_HIP(ctx, hipMalloc(&d_cells, bytes + 16));       // + the zero slack every buffer of this library carries
  hipError_t e = hipMemsetAsync(d_cells, 0, bytes + 16, st);
  if (e != hipSuccess) { (void)hipFree(d_cells); return hip_fail(ctx, e, "hipMemsetAsync"); }
  for (int l = 1; l <= D; l++)
    hipLaunchKernelGGL(build_emit_kernel, dim3(blocks(cap[l])), dim3(256), 0, st, (const uint32_t *)lk[l], (const uint32_t *)lv[l], (const uint32_t *)(count + l),
                       (const uint32_t *)(l > 1 ? par[l] : nullptr), (const uint32_t *)(l > 1 ? lv[l - 1] : nullptr),
                       (const uint32_t *)(l > 1 ? mixed + off[l - 1] : nullptr), (const uint32_t *)(l < D ? mixed + off[l] : nullptr),
                       (uint32_t *)d_cells, n_cells);
  e = hipGetLastError();
  if (e == hipSuccess) e = hipStreamSynchronize(st);   // the temporaries are freed when this function returns
  if (e != hipSuccess) { (void)hipFree(d_cells); return hip_fail(ctx, e, "octree builder"); }
  const int rc = adopt_device_buffer(ctx, d_cells, bytes, out);
  if (rc != TDT_OK) { (void)hipFree(d_cells); return rc; }
  if (n_cells_out) *n_cells_out = n_cells;
  return TDT_OK;
}

int upload_voxels(tdt_ctx *ctx, DeviceArena &arena, const int32_t *host, size_t n, int32_t **dev) {
  *dev = arena.get<int32_t>(n * 4);
  if (!*dev) return fail(ctx, TDT_ERR_HIP, "out of device memory in the octree builder");
  TDT_HIP(ctx, hipMemcpyAsync(*dev, host, n * 4 * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
  return TDT_OK;
}

// a multi-device context builds on its first device and replicates the result like any other upload
int replicate_to_front(tdt_ctx *front, tdt_ctx *m0, tdt_buffer *built, tdt_buffer **out) {
  std::vector<unsigned char> host(built->bytes);
  int rc = tdt_buffer_read(built, 0, built->bytes, host.data());
  if (rc == TDT_OK) rc = tdt_buffer_create(front, host.data(), host.size(), out);
  else fail(front, rc, tdt_last_error(m0));
  tdt_buffer_destroy(built);
  return rc;
}

}  // namespace
}  // namespace tdt

extern "C" {

int tdt_octree_build_cells(tdt_ctx *ctx, const int32_t *voxels_xyzm, size_t n_voxels, int depth, tdt_buffer **cells, uint32_t *n_cells) {
  using namespace tdt;
  if (!ctx || !cells) return fail(ctx, TDT_ERR_INVALID_VALUE, "null argument");
  *cells = nullptr;
  if (depth < 1 || depth > 10) return fail(ctx, TDT_ERR_INVALID_VALUE, "depth must be 1..10");
  if (n_voxels && !voxels_xyzm) return fail(ctx, TDT_ERR_INVALID_VALUE, "null voxel list");
  if (n_voxels >= (1ull << 31)) return fail(ctx, TDT_ERR_INVALID_VALUE, "more than 2^31 voxels");
  if (ctx->multi) {
    tdt_ctx *m0 = multi_first_member(ctx);
    tdt_buffer *b = nullptr;
    const int rc = tdt_octree_build_cells(m0, voxels_xyzm, n_voxels, depth, &b, n_cells);
    if (rc != TDT_OK) return fail(ctx, rc, tdt_last_error(m0));
    return replicate_to_front(ctx, m0, b, cells);
  }
  TDT_HIP(ctx, hipSetDevice(ctx->device));
  if (n_voxels == 0) {                                  // an empty grid is the root cell with eight EMPTY nodes
    const uint32_t zero[16] = {0};
    if (n_cells) *n_cells = 1;
    return tdt_buffer_create(ctx, zero, sizeof zero, cells);
  }
  DeviceArena arena(ctx);
  int32_t *d_vox = nullptr;
  int rc = upload_voxels(ctx, arena, voxels_xyzm, n_voxels, &d_vox);
  if (rc != TDT_OK) return rc;
  KeyArgs A;
  std::memset(&A, 0, sizeof A);
  A.ax = 0; A.ay = 1; A.az = 2; A.depth = depth;
  rc = build_cells_device(ctx, d_vox, (uint32_t)n_voxels, A, cells, n_cells);
  (void)hipStreamSynchronize(ctx->stream);              // the upload borrowed the caller's memory
  return rc;
}

int tdt_octree_build_from_points(tdt_ctx *ctx, const int32_t *voxels_xyzk, size_t n_voxels, const int32_t min_point[3],
                                 const uint32_t *palette_keys, const uint8_t *palette_rgb, size_t n_palette, int z_up,
                                 int max_iter, tdt_buffer *out_slots[8], int32_t *max_depth, int32_t *cell_count) {
  using namespace tdt;
  if (!ctx || !out_slots) return fail(ctx, TDT_ERR_INVALID_VALUE, "null argument");
  for (int s = 0; s < 8; s++) out_slots[s] = nullptr;
  if (!voxels_xyzk || n_voxels == 0) return fail(ctx, TDT_ERR_INVALID_VALUE, "the PLY holds no voxels");
  if (!min_point || !palette_keys || !palette_rgb || n_palette == 0) return fail(ctx, TDT_ERR_INVALID_VALUE, "null min_point / palette");
  if (n_voxels >= (1ull << 31) || n_palette >= (1ull << 24)) return fail(ctx, TDT_ERR_INVALID_VALUE, "voxel list or palette too large");
  for (size_t i = 1; i < n_palette; i++)
    if (palette_keys[i] <= palette_keys[i - 1]) return fail(ctx, TDT_ERR_INVALID_VALUE, "palette keys must be strictly ascending");
  tdt_ctx *dev_ctx = ctx->multi ? multi_first_member(ctx) : ctx;          // kernels run here; buffers are created on `ctx`
  TDT_HIP(ctx, hipSetDevice(dev_ctx->device));
  hipStream_t st = dev_ctx->stream;
  DeviceArena arena(dev_ctx);
  const uint32_t n = (uint32_t)n_voxels, np = (uint32_t)n_palette;
  int32_t *d_vox = nullptr;
  int rc = upload_voxels(dev_ctx, arena, voxels_xyzk, n_voxels, &d_vox);
  if (rc != TDT_OK) return fail(ctx, rc, tdt_last_error(dev_ctx));
  // extent -> depth (tdt_scene_from_ply: ext = max - min_point + 1; the smallest depth >= 1 whose grid holds it)
  TDT_ALLOC(d_mx, int32_t, 3);
  const int32_t lowest[3] = {INT32_MIN, INT32_MIN, INT32_MIN};
  TDT_HIP(ctx, hipMemcpyAsync(d_mx, lowest, sizeof lowest, hipMemcpyHostToDevice, st));
  unsigned nb = blocks(n); if (nb > 1024) nb = 1024;
  hipLaunchKernelGGL(build_extent_kernel, dim3(nb), dim3(256), 0, st, (const int32_t *)d_vox, n, d_mx);
  // palette entries in use -> material index = rank among them (ascending key, as the host's std::map iterates)
  TDT_ALLOC(d_pk, uint32_t, np); TDT_ALLOC(d_rgb, uint8_t, (size_t)np * 3);
  TDT_ALLOC(d_used, uint32_t, (size_t)np + 1); TDT_ALLOC(d_rank, uint32_t, (size_t)np + 1);
  TDT_ALLOC(d_pscr, uint32_t, scan_scratch_words((size_t)np + 1)); TDT_ALLOC(d_err, uint32_t, 1);
  TDT_HIP(ctx, hipMemcpyAsync(d_pk, palette_keys, np * sizeof(uint32_t), hipMemcpyHostToDevice, st));
  TDT_HIP(ctx, hipMemcpyAsync(d_rgb, palette_rgb, (size_t)np * 3, hipMemcpyHostToDevice, st));
  TDT_HIP(ctx, hipMemsetAsync(d_used, 0, ((size_t)np + 1) * sizeof(uint32_t), st));
  TDT_HIP(ctx, hipMemsetAsync(d_err, 0, sizeof(uint32_t), st));
  hipLaunchKernelGGL(build_palette_mark_kernel, dim3(blocks(n)), dim3(256), 0, st, (const int32_t *)d_vox, n, (const uint32_t *)d_pk, np, d_used, d_err);
  TDT_HIP(ctx, exclusive_scan_u32(st, d_used, d_rank, np + 1u, d_pscr));
  int32_t mx[3]; uint32_t n_colours = 0, err = 0;
  TDT_HIP(ctx, hipMemcpyAsync(mx, d_mx, sizeof mx, hipMemcpyDeviceToHost, st));
  TDT_HIP(ctx, hipMemcpyAsync(&n_colours, d_rank + np, sizeof n_colours, hipMemcpyDeviceToHost, st));
  TDT_HIP(ctx, hipMemcpyAsync(&err, d_err, sizeof err, hipMemcpyDeviceToHost, st));
  TDT_HIP(ctx, hipStreamSynchronize(st));
  if (err) return fail(ctx, TDT_ERR_INVALID_VALUE, "a voxel's colour key is not in the palette");
  if (n_colours > 254) return fail(ctx, TDT_ERR_INVALID_VALUE, "more than 254 distinct colours");
  long long ext[3], emax = 0;
  for (int a = 0; a < 3; a++) { ext[a] = (long long)mx[a] - min_point[a] + 1; emax = ext[a] > emax ? ext[a] : emax; }
  int depth = 1;
  while ((1ll << depth) < emax) depth++;
  if (depth > 9) return fail(ctx, TDT_ERR_INVALID_VALUE, "model larger than 512 voxels on an edge");
  // file axes -> octree axes; centred in x, on the floor, at the far (z = 0) side
  KeyArgs A;
  std::memset(&A, 0, sizeof A);
  A.ax = 0; A.ay = z_up ? 2 : 1; A.az = z_up ? 1 : 2; A.depth = depth;
  A.sub[0] = min_point[A.ax]; A.sub[1] = min_point[A.ay]; A.sub[2] = min_point[A.az];
  A.add[0] = (int32_t)(((1ll << depth) - ext[A.ax]) / 2); A.add[1] = 0; A.add[2] = 0;
  A.pal_keys = d_pk; A.n_pal = np; A.pal_rank = d_rank;
  tdt_buffer *built = nullptr;
  uint32_t n_cells = 0;
  rc = build_cells_device(dev_ctx, d_vox, n, A, &built, &n_cells);
  if (rc != TDT_OK) return ctx == dev_ctx ? rc : fail(ctx, rc, tdt_last_error(dev_ctx));
  int cc = 1 << 10;
  while ((uint32_t)cc < n_cells && cc < (1 << 22)) cc <<= 1;
  if ((uint32_t)cc < n_cells) { tdt_buffer_destroy(built); return fail(ctx, TDT_ERR_INVALID_VALUE, "scene needs more than 2^22 cells"); }
  // material tables on the device, then the small constant payloads
  const size_t tab_bytes = (size_t)n_colours * 3 * 4;
  void *d_mat = nullptr, *d_alb = nullptr;
  hipError_t e = hipMalloc(&d_mat, tab_bytes + 16);
  if (e == hipSuccess) e = hipMalloc(&d_alb, tab_bytes + 16);
  if (e == hipSuccess) e = hipMemsetAsync(d_mat, 0, tab_bytes + 16, st);
  if (e == hipSuccess) e = hipMemsetAsync(d_alb, 0, tab_bytes + 16, st);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(build_materials_kernel, dim3(blocks(np)), dim3(256), 0, st, (const uint32_t *)d_used, (const uint32_t *)d_rank, (const uint8_t *)d_rgb, np,
                       (uint32_t *)d_mat, (float *)d_alb);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  if (e != hipSuccess) { if (d_mat) (void)hipFree(d_mat); if (d_alb) (void)hipFree(d_alb); tdt_buffer_destroy(built); return hip_fail(ctx, e, "material tables"); }
  tdt_buffer *b_mat = nullptr, *b_alb = nullptr;
  rc = adopt_device_buffer(dev_ctx, d_mat, tab_bytes, &b_mat);
  if (rc != TDT_OK) { (void)hipFree(d_mat); (void)hipFree(d_alb); tdt_buffer_destroy(built); return fail(ctx, rc, tdt_last_error(dev_ctx)); }
  rc = adopt_device_buffer(dev_ctx, d_alb, tab_bytes, &b_alb);
  if (rc != TDT_OK) { (void)hipFree(d_alb); tdt_buffer_destroy(b_mat); tdt_buffer_destroy(built); return fail(ctx, rc, tdt_last_error(dev_ctx)); }
  if (ctx != dev_ctx) {                                 // multi-device: replicate what the first device built
    rc = replicate_to_front(ctx, dev_ctx, built, &out_slots[TDT_SLOT_CELLS]);
    const int rc2 = replicate_to_front(ctx, dev_ctx, b_mat, &out_slots[TDT_SLOT_MATERIALS]);
    const int rc3 = replicate_to_front(ctx, dev_ctx, b_alb, &out_slots[TDT_SLOT_ALBEDOS]);
    rc = rc != TDT_OK ? rc : (rc2 != TDT_OK ? rc2 : rc3);
  } else { out_slots[TDT_SLOT_CELLS] = built; out_slots[TDT_SLOT_MATERIALS] = b_mat; out_slots[TDT_SLOT_ALBEDOS] = b_alb; }
  const float metal[4] = {0.1f, 0.3f, 0.4f, 0.8f}, dielectric[1] = {1.2f};     // the reference's tables (main.rs:418-441); unused by Lambertian voxels
  const float scale = 1.0f;
  const float of[7] = {-0.5f, -0.5f, -1.0f, 0.0f, scale, 1.0f / scale, 1.0f / (float)cc};   // main.rs:456-457 AABB; octree.rs:44-50
  const int32_t oi[3] = {depth, max_iter, cc};
  if (rc == TDT_OK) rc = tdt_buffer_create(ctx, metal, sizeof metal, &out_slots[TDT_SLOT_METAL]);
  if (rc == TDT_OK) rc = tdt_buffer_create(ctx, dielectric, sizeof dielectric, &out_slots[TDT_SLOT_DIELECTRIC]);
  if (rc == TDT_OK) rc = tdt_buffer_create(ctx, of, sizeof of, &out_slots[TDT_SLOT_OCTREE_FLOATS]);
  if (rc == TDT_OK) rc = tdt_buffer_create(ctx, oi, sizeof oi, &out_slots[TDT_SLOT_OCTREE_INTS]);
  if (rc != TDT_OK) { for (int s = 0; s < 8; s++) if (out_slots[s]) { tdt_buffer_destroy(out_slots[s]); out_slots[s] = nullptr; } return rc; }
  if (max_depth) *max_depth = depth;
  if (cell_count) *cell_count = cc;
  return TDT_OK;
}

}  // extern "C"
