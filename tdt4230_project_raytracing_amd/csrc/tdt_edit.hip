// Voxel edits — SURVEY §8f-2: assets/shaders/octree_update.comp (uc:N = its line N) behind tdt_dispatch_compute of a
// TDT_PROGRAM_OCTREE_UPDATE program (Octree::update_vbo, octree.rs:170-183).
//
// One voxel edit per invocation: walk max_depth-1 levels with treeLookup's index arithmetic, turning EMPTY nodes on the way
// into PARENTs of freshly counted cells (atomicCompSwap + atomic counter, uc:72-74), then overwrite the last node visited
// with the delta (uc:101).  The reference's invocations race when their paths collide (its own comment, uc:70-71); the
// only implementation of it that can be run (llvmpipe) executes work-groups one after the other, x fastest, and THAT
// order defines the result here (pinned by tests/golden/edit_*.npz).
//
// Two executions of that definition:
//   ordered walk  octree_update_kernel: one lane walks the dispatch in order.  Always right; one invocation (the
//                 reference's click path, main.rs:561-568) never needs more.
//   parallel      one lane per invocation, chosen ON THE DEVICE when it provably gives the same bytes:
//                 1. edit_plan_kernel  walks every invocation's path read-only on the tree as it is, finds the first
//                    EMPTY node (from there on the serial walk would allocate one cell per remaining level) and claims
//                    the node it will write with an atomicCAS on a mark word per node;
//                 2. a prefix sum over the invocations' cell counts = the counter values the serial order would hand
//                    each of them (a bump allocator without a lock and without an order dependence);
//                 3. edit_check_kernel re-walks: no node it READS may carry another invocation's mark, and the nodes
//                    it will touch inside its fresh cells must be inside the buffer, EMPTY, and unclaimed;
//                 4. edit_apply_kernel: atomicCAS(type, EMPTY, PARENT) + value stores along the path, the delta store at
//                    the end; edit_finish_kernel bumps the counter by the total.
//                 Any doubt (two invocations on one node, a walk that leaves the buffer, a non-empty node in the free
//                 pool, the same delta twice: delta_index = x+y+z, uc:99) raises a device flag: apply does nothing and the
//                 ordered walk, launched behind it, runs instead.  No host round trip either way.
#include <cstring>

#include "device_scan.hpp"
#include "tdt_internal.hpp"
#include "trace_device.hpp"

namespace tdt {

struct EditArgs {
  uint32_t *cells; uint32_t cells_dwords;
  const uint32_t *delta; uint32_t delta_dwords;
  float inv_cell_count; int max_depth; int cell_count;
  uint32_t *counter;
  int gx, gy, gz;
};

struct EditScratch {
  uint32_t *mark = nullptr; size_t mark_words = 0;    // one word per node of the cells buffer: id + 1 of the invocation that writes it
  uint32_t *plan = nullptr; size_t plan_words = 0;    // per invocation: cells it allocates -> (scanned in place) its first counter offset; [n] = total
  uint32_t *scan = nullptr; size_t scan_words = 0;
  uint32_t *flags = nullptr;                          // [0] doubt raised, [1] path the last dispatch took (1 ordered, 2 parallel)
  int mode = 0;                                       // tdt_debug_edit_mode
  int last_direct = 0;                                // the last dispatch went straight to the ordered walk (one invocation / mode 1)
};

// treeLookupLeaf's index arithmetic for one level (uc:65-69): as compiled, fma is two roundings and round() is
// round-half-even; returns the node's dword offset in the cells buffer ((index << 3) >> 2 on 32 bits, as uc:76 addresses it)
TDT_DEV uint32_t edit_level_dword(uint32_t node_value, float cx, float cy, float cz, float inv_cell_count, float two_cc, uint32_t &index) {
  const float fx = f_fract(cx), fy = f_fract(cy), fz = f_fract(cz);
  const float rx = __builtin_rintf((((float)node_value + fx) * inv_cell_count) * two_cc + -0.5f);
  const float ry = __builtin_rintf(fy * 2.0f + -0.5f), rz = __builtin_rintf(fz * 2.0f + -0.5f);
  index = ((((uint32_t)f2i(rx) << 1) + (uint32_t)f2i(ry)) << 1) + (uint32_t)f2i(rz);
  return (index << 3) >> 2;
}
TDT_DEV uint32_t f2u(float v) {     // uint(float) as compiled: negative / NaN -> 0, >= 2^32 -> 0xFFFFFFFF
  return !(v > -1.0f) ? 0u : (v >= 4294967296.0f ? 0xFFFFFFFFu : (uint32_t)v);
}
struct EditDelta { float cx, cy, cz, type, value; };
TDT_DEV EditDelta edit_load_delta(const EditArgs &A, uint32_t x, uint32_t y, uint32_t z) {
  const uint32_t dof = (x + y + z) << 5;              // delta_index uc:99, DeltaNode stride 32
  EditDelta d;
  d.cx = __uint_as_float(ld_dw(A.delta, A.delta_dwords, dof)); d.cy = __uint_as_float(ld_dw(A.delta, A.delta_dwords, dof + 4));
  d.cz = __uint_as_float(ld_dw(A.delta, A.delta_dwords, dof + 8));
  d.type = __uint_as_float(ld_dw(A.delta, A.delta_dwords, dof + 12)); d.value = __uint_as_float(ld_dw(A.delta, A.delta_dwords, dof + 16));
  return d;
}

// ---- the ordered walk ------------------------------------------------------------------------------------------------
// flags == null: unconditional; otherwise only when the parallel plan raised a doubt (flags[0] != 0)
__global__ __launch_bounds__(64) void octree_update_kernel(const EditArgs A, uint32_t *flags) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  if (flags) { if (flags[0] == 0u) return; flags[1] = 1u; }
  const float two_cc = (float)(int32_t)((uint32_t)A.cell_count << 1);
  for (int z = 0; z < A.gz; z++) for (int y = 0; y < A.gy; y++) for (int x = 0; x < A.gx; x++) {
    const EditDelta d = edit_load_delta(A, (uint32_t)x, (uint32_t)y, (uint32_t)z);
    float cx = d.cx, cy = d.cy, cz = d.cz;
    uint32_t node_value = 0, index = 0;
    for (float i = 0.0f; i < (float)(A.max_depth - 1); i = i + 1.0f) {          // treeLookupLeaf uc:57-80
      const uint32_t dw = edit_level_dword(node_value, cx, cy, cz, A.inv_cell_count, two_cc, index);
      const uint32_t old = (dw + 1u < A.cells_dwords) ? A.cells[dw + 1u] : 0u;    // an out-of-range atomic returns 0, writes nothing
      if (old == 0u) {
        const uint32_t fresh = (*A.counter)++;
        if (dw + 1u < A.cells_dwords) A.cells[dw + 1u] = 1u;
        if (dw < A.cells_dwords) A.cells[dw] = fresh;
      }
      node_value = (dw < A.cells_dwords) ? A.cells[dw] : 0u;                      // node = indirect_cells[index] uc:76
      cx = cx * 2.0f; cy = cy * 2.0f; cz = cz * 2.0f;
    }
    const uint32_t dw = (index << 3) >> 2;
    if (dw < A.cells_dwords) A.cells[dw] = f2u(d.value);                          // uc:101
    if (dw + 1u < A.cells_dwords) A.cells[dw + 1u] = f2u(d.type);
  }
}

// ---- the parallel form -----------------------------------------------------------------------------------------------
TDT_DEV bool edit_invocation(const EditArgs &A, uint32_t &x, uint32_t &y, uint32_t &z, uint32_t &id) {
  const unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  const unsigned long long n = (unsigned long long)A.gx * (unsigned long long)A.gy * (unsigned long long)A.gz;
  if (i >= n) return false;
  id = (uint32_t)i;                                   // x fastest: the ordered walk's position of this invocation
  x = (uint32_t)(i % (unsigned)A.gx); y = (uint32_t)((i / (unsigned)A.gx) % (unsigned)A.gy); z = (uint32_t)(i / ((unsigned long long)A.gx * (unsigned)A.gy));
  return true;
}
// claim a node for writing; false when another invocation holds it
TDT_DEV bool edit_claim(uint32_t *mark, uint32_t node, uint32_t id) {
  const uint32_t old = atomicCAS(&mark[node], 0u, id + 1u);
  return old == 0u || old == id + 1u;
}

__global__ __launch_bounds__(256) void edit_plan_kernel(const EditArgs A, uint32_t *mark, uint32_t *plan, uint32_t *flags) {
  uint32_t x, y, z, id;
  if (!edit_invocation(A, x, y, z, id)) return;
  const float two_cc = (float)(int32_t)((uint32_t)A.cell_count << 1);
  const EditDelta d = edit_load_delta(A, x, y, z);
  float cx = d.cx, cy = d.cy, cz = d.cz;
  const int L = A.max_depth - 1;
  uint32_t node_value = 0, index = 0, n_alloc = 0;
  bool doubt = false;
  uint32_t dw = 0;
  for (int l = 0; l < L; l++) {
    dw = edit_level_dword(node_value, cx, cy, cz, A.inv_cell_count, two_cc, index);
    if (!(dw + 1u < A.cells_dwords)) { doubt = true; break; }                      // leaves the buffer: the ordered walk knows what that means
    if (A.cells[dw + 1u] == 0u) { n_alloc = (uint32_t)(L - l); break; }            // EMPTY: this level and every deeper one allocate
    node_value = A.cells[dw];
    cx = cx * 2.0f; cy = cy * 2.0f; cz = cz * 2.0f;
  }
  plan[id] = n_alloc;
  // the node of the tree as it is that this invocation writes: its first EMPTY node, or (no allocation) the last node visited
  if (!doubt) {
    if (L <= 0) dw = 0u;
    if (!(dw + 1u < A.cells_dwords) || !edit_claim(mark, dw >> 1, id)) doubt = true;
  }
  if (doubt) atomicOr(&flags[0], 1u);
}

__global__ __launch_bounds__(256) void edit_check_kernel(const EditArgs A, uint32_t *mark, const uint32_t *plan, uint32_t *flags) {
  uint32_t x, y, z, id;
  if (!edit_invocation(A, x, y, z, id)) return;
  if (flags[0] != 0u) return;
  const float two_cc = (float)(int32_t)((uint32_t)A.cell_count << 1);
  const EditDelta d = edit_load_delta(A, x, y, z);
  float cx = d.cx, cy = d.cy, cz = d.cz;
  const int L = A.max_depth - 1;
  const uint32_t fresh0 = *A.counter + plan[id];      // (the counter is bumped by edit_finish_kernel, after everybody has read it)
  uint32_t node_value = 0, index = 0, k = 0;
  bool doubt = false, fresh = false;
  for (int l = 0; l < L && !doubt; l++) {
    const uint32_t dw = edit_level_dword(node_value, cx, cy, cz, A.inv_cell_count, two_cc, index);
    if (!(dw + 1u < A.cells_dwords)) { doubt = true; break; }
    const uint32_t node = dw >> 1;
    if (!fresh) {
      const uint32_t m = mark[node];
      if (m != 0u && m != id + 1u) doubt = true;      // a node this walk reads is written by another invocation
      if (A.cells[dw + 1u] == 0u) { fresh = true; node_value = fresh0; k = 1; }     // allocate: value = the counter
      else node_value = A.cells[dw];
    } else {                                          // inside a cell this invocation allocated: must be untouched free pool
      if (A.cells[dw + 1u] != 0u || !edit_claim(mark, node, id)) doubt = true;
      node_value = fresh0 + k; k++;
    }
    cx = cx * 2.0f; cy = cy * 2.0f; cz = cz * 2.0f;
  }
  if (doubt) atomicOr(&flags[0], 1u);
}

__global__ __launch_bounds__(256) void edit_apply_kernel(const EditArgs A, const uint32_t *plan, const uint32_t *flags) {
  uint32_t x, y, z, id;
  if (!edit_invocation(A, x, y, z, id)) return;
  if (flags[0] != 0u) return;
  const float two_cc = (float)(int32_t)((uint32_t)A.cell_count << 1);
  const EditDelta d = edit_load_delta(A, x, y, z);
  float cx = d.cx, cy = d.cy, cz = d.cz;
  const int L = A.max_depth - 1;
  uint32_t next = *A.counter + plan[id];
  uint32_t node_value = 0, index = 0;
  for (int l = 0; l < L; l++) {
    const uint32_t dw = edit_level_dword(node_value, cx, cy, cz, A.inv_cell_count, two_cc, index);
    if (atomicCAS(&A.cells[dw + 1u], 0u, 1u) == 0u) { A.cells[dw] = next; node_value = next; next++; }    // uc:72-74
    else node_value = A.cells[dw];
    cx = cx * 2.0f; cy = cy * 2.0f; cz = cz * 2.0f;
  }
  const uint32_t dw = (index << 3) >> 2;
  A.cells[dw] = f2u(d.value);                                                     // uc:101 (inside the buffer: checked by the plan)
  A.cells[dw + 1u] = f2u(d.type);
}

__global__ void edit_finish_kernel(uint32_t *counter, const uint32_t *plan_total, uint32_t *flags) {
  if (flags[0] == 0u) { *counter += *plan_total; flags[1] = 2u; }
}

static int ensure(tdt_ctx *ctx, uint32_t **p, size_t *have, size_t need) {
  if (*have >= need) return TDT_OK;
  if (*p) (void)hipFree(*p);
  *p = nullptr; *have = 0;
  TDT_HIP(ctx, hipMalloc((void **)p, need * sizeof(uint32_t)));
  *have = need;
  return TDT_OK;
}

int launch_update(tdt_compute *c, int width, int height, int depth) {
  tdt_ctx *ctx = c->ctx;
  static const int required[] = {TDT_SLOT_CELLS, TDT_SLOT_DELTA, TDT_SLOT_OCTREE_FLOATS, TDT_SLOT_OCTREE_INTS};
  for (int s : required)
    if (!ctx->ssbo[s]) return fail(ctx, TDT_ERR_INCOMPLETE, "no buffer bound to shader-storage slot " + std::to_string(s));
  if (!ctx->atomic0 || ctx->atomic0->bytes < 4) return fail(ctx, TDT_ERR_INCOMPLETE, "no atomic-counter buffer bound to slot 0");
  if (ctx->ssbo[TDT_SLOT_OCTREE_FLOATS]->bytes < 28 || ctx->ssbo[TDT_SLOT_OCTREE_INTS]->bytes < 12)
    return fail(ctx, TDT_ERR_INVALID_VALUE, "octree uniform buffers are too small (need 28 / 12 bytes)");
  float of[7]; int32_t oi[3];
  std::memcpy(of, ctx->ssbo[TDT_SLOT_OCTREE_FLOATS]->shadow, sizeof of);
  std::memcpy(oi, ctx->ssbo[TDT_SLOT_OCTREE_INTS]->shadow, sizeof oi);
  tdt_buffer *cells = ctx->ssbo[TDT_SLOT_CELLS], *delta = ctx->ssbo[TDT_SLOT_DELTA];
  auto dwords = [](const tdt_buffer *b) { size_t d = b->bytes >> 2; return (uint32_t)(d > 0xFFFFFFFFull ? 0xFFFFFFFFull : d); };
  EditArgs A;
  A.cells = (uint32_t *)cells->dev; A.cells_dwords = dwords(cells);
  A.delta = (const uint32_t *)delta->dev; A.delta_dwords = dwords(delta);
  A.inv_cell_count = of[6]; A.max_depth = oi[0]; A.cell_count = oi[2];
  A.counter = (uint32_t *)ctx->atomic0->dev;
  // ComputeShader::dispatch_compute with group_size {1,1,1}: groups = max(dim / 1, 1) (compute_shader.rs:30-32)
  A.gx = width < 1 ? 1 : width; A.gy = height < 1 ? 1 : height; A.gz = depth < 1 ? 1 : depth;
  TDT_HIP(ctx, hipSetDevice(ctx->device));
  if (!ctx->edit) ctx->edit = new (std::nothrow) EditScratch();
  if (!ctx->edit) return fail(ctx, TDT_ERR_HIP, "out of host memory");
  EditScratch &S = *ctx->edit;
  const unsigned long long n = (unsigned long long)A.gx * (unsigned long long)A.gy * (unsigned long long)A.gz;
  const bool direct = n == 1ull || S.mode == 1 || n > (1ull << 26) || A.max_depth > 31;
  S.last_direct = direct ? 1 : 0;
  if (direct) {
    hipLaunchKernelGGL(octree_update_kernel, dim3(1), dim3(64), 0, ctx->stream, A, (uint32_t *)nullptr);
  } else {
    const size_t nodes = (size_t)(A.cells_dwords >> 1) + 1;
    int rc = ensure(ctx, &S.mark, &S.mark_words, nodes);
    if (rc == TDT_OK) rc = ensure(ctx, &S.plan, &S.plan_words, (size_t)n + 1);
    if (rc == TDT_OK) rc = ensure(ctx, &S.scan, &S.scan_words, scan_scratch_words((size_t)n + 1));
    if (rc == TDT_OK && !S.flags) { size_t z = 0; rc = ensure(ctx, &S.flags, &z, 4); }
    if (rc != TDT_OK) return rc;
    TDT_HIP(ctx, hipMemsetAsync(S.mark, 0, nodes * sizeof(uint32_t), ctx->stream));
    TDT_HIP(ctx, hipMemsetAsync(S.flags, 0, 2 * sizeof(uint32_t), ctx->stream));
    TDT_HIP(ctx, hipMemsetAsync(S.plan + n, 0, sizeof(uint32_t), ctx->stream));
    const unsigned nb = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(edit_plan_kernel, dim3(nb), dim3(256), 0, ctx->stream, A, S.mark, S.plan, S.flags);
    TDT_HIP(ctx, exclusive_scan_u32(ctx->stream, S.plan, S.plan, (uint32_t)n + 1u, S.scan));   // plan[n] = cells allocated in total
    hipLaunchKernelGGL(edit_check_kernel, dim3(nb), dim3(256), 0, ctx->stream, A, S.mark, (const uint32_t *)S.plan, S.flags);
    hipLaunchKernelGGL(edit_apply_kernel, dim3(nb), dim3(256), 0, ctx->stream, A, (const uint32_t *)S.plan, (const uint32_t *)S.flags);
    hipLaunchKernelGGL(edit_finish_kernel, dim3(1), dim3(1), 0, ctx->stream, A.counter, (const uint32_t *)(S.plan + n), S.flags);
    hipLaunchKernelGGL(octree_update_kernel, dim3(1), dim3(64), 0, ctx->stream, A, S.flags);
  }
  TDT_HIP(ctx, hipGetLastError());
  cells->version += 0x100000000ull;      // the trace's LDS-table image / scan of this buffer are stale now
  return TDT_OK;
}

void edit_scratch_destroy(tdt_ctx *ctx) {
  if (!ctx->edit) return;
  EditScratch &S = *ctx->edit;
  for (uint32_t *p : {S.mark, S.plan, S.scan, S.flags}) if (p) (void)hipFree(p);
  delete ctx->edit;
  ctx->edit = nullptr;
}

}  // namespace tdt

extern "C" {

int tdt_debug_edit_mode(tdt_ctx *ctx, int mode) {
  if (!ctx || mode < 0 || mode > 1) return TDT_ERR_INVALID_VALUE;
  if (ctx->multi) return tdt::fail(ctx, TDT_ERR_INVALID_OPERATION, "set the edit mode on a single-device context");
  if (!ctx->edit) ctx->edit = new (std::nothrow) tdt::EditScratch();
  if (!ctx->edit) return tdt::fail(ctx, TDT_ERR_HIP, "out of host memory");
  ctx->edit->mode = mode;
  return TDT_OK;
}

int tdt_debug_last_edit_path(tdt_ctx *ctx) {
  if (!ctx || ctx->multi || !ctx->edit) return 0;
  if (ctx->edit->last_direct) return 1;
  if (!ctx->edit->flags) return 0;
  uint32_t f[2] = {0, 0};
  if (hipSetDevice(ctx->device) != hipSuccess) return 0;
  if (hipMemcpyAsync(f, ctx->edit->flags, sizeof f, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) return 0;
  if (hipStreamSynchronize(ctx->stream) != hipSuccess) return 0;
  return (int)f[1];
}

}  // extern "C"
