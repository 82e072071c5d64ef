// Device-wide exclusive prefix sum over uint32 (HBM-streaming: 4 B read + 4 B written per item and level).
// Used by the GPU octree builder (tdt_build.hip) and the parallel voxel-edit planner (tdt_edit.hip).
//
// Tiles of 2048 items per 256-thread block (8 consecutive items per lane: two 16-byte loads), wave64 shuffles for the
// in-wave scan, LDS for the four wave totals; the per-tile totals are scanned by the same code one level up (2048^2 =
// 4 M items need two levels, 8 G three) and added back.  Scanning n + 1 items whose last one is 0 leaves the grand total
// in out[n] — that is how callers get counts without a separate reduction.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace tdt {

constexpr uint32_t kScanTile = 2048;

__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t t = (uint32_t)__shfl_up((int)v, o, 64);
    if ((int)(threadIdx.x & 63) >= o) v += t;
  }
  return v;
}

// out[i] = sum of in[tile_start .. i) for i in the tile; tile_sums[tile] = the tile's total (may be null for one tile).
// in == out is allowed (each lane reads its 8 items before it writes them).
static __global__ __launch_bounds__(256) void scan_tiles_kernel(const uint32_t *in, uint32_t *out, uint32_t n, uint32_t *tile_sums) {
  __shared__ uint32_t s_wave[4];
  const uint32_t base = blockIdx.x * kScanTile + threadIdx.x * 8u;
  uint32_t v[8];
  if (base + 8u <= n && ((reinterpret_cast<uintptr_t>(in) & 15u) == 0)) {
    const uint4 a = *reinterpret_cast<const uint4 *>(in + base), b = *reinterpret_cast<const uint4 *>(in + base + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
  } else {
#pragma unroll
    for (int k = 0; k < 8; k++) v[k] = base + k < n ? in[base + k] : 0u;
  }
  uint32_t mine = 0;
#pragma unroll
  for (int k = 0; k < 8; k++) { const uint32_t t = v[k]; v[k] = mine; mine += t; }   // v[k] = exclusive prefix inside the lane
  const uint32_t incl = wave_inclusive_scan(mine);
  if ((threadIdx.x & 63) == 63) s_wave[threadIdx.x >> 6] = incl;
  __syncthreads();
  uint32_t before = incl - mine;
  for (uint32_t w = 0; w < (threadIdx.x >> 6); w++) before += s_wave[w];
#pragma unroll
  for (int k = 0; k < 8; k++) if (base + k < n) out[base + k] = before + v[k];
  if (tile_sums && threadIdx.x == 255) tile_sums[blockIdx.x] = before + mine;
}

static __global__ __launch_bounds__(256) void scan_add_kernel(uint32_t *out, uint32_t n, const uint32_t *tile_offsets) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i < n) out[i] += tile_offsets[i / kScanTile];
}

// words of scratch exclusive_scan_u32 needs for n items
inline size_t scan_scratch_words(size_t n) {
  size_t w = 0;
  while (n > kScanTile) { n = (n + kScanTile - 1) / kScanTile; w += n; }
  return w + 1;
}

// out[0..n) = exclusive prefix sums of in[0..n) (in == out allowed); asynchronous on `stream`
inline hipError_t exclusive_scan_u32(hipStream_t stream, const uint32_t *in, uint32_t *out, uint32_t n, uint32_t *scratch) {
  if (n == 0) return hipSuccess;
  const uint32_t tiles = (n + kScanTile - 1) / kScanTile;
  hipLaunchKernelGGL(scan_tiles_kernel, dim3(tiles), dim3(256), 0, stream, in, out, n, tiles > 1 ? scratch : nullptr);
  if (tiles > 1) {
    const hipError_t e = exclusive_scan_u32(stream, scratch, scratch, tiles, scratch + tiles);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(scan_add_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, out, n, scratch);
  }
  return hipGetLastError();
}

}  // namespace tdt
