// The multi-device context — SURVEY §8b ("tdt_ctx_create(int n_devices, const int* device_ids) — replication + gather are
// internal") and §8e: one context, one owning thread (the reference's shape: main.rs:58-61,105-112), n GPUs.
//
// A multi-device tdt_ctx is a FRONT over n ordinary single-device contexts ("members", one per entry of device_ids, each
// with its own HIP stream).  Handles created from the front are fronts too: a buffer handle stands for n replicas, a
// program handle for n programs whose work-group partition is (i, n), an image handle for the assembled frame on the first
// device.  Every entry point of tdt_rt.hip forwards here when its handle belongs to a front, and the functions below are
// written against the public single-device API — a member is driven exactly as a host program would drive it.
//
// Frame (tdt_dispatch_compute of the raytracer):
//   for each member i: bind its tile buffer [k][32][32] RGBA, tdt_dispatch_compute -> its own stream        (no sync)
//   ONE gather of the tile buffers to member 0:
//       "rccl": ncclGroupStart; ncclGather(tile_i -> gathered, root 0, comm_i, stream_i) for every i; ncclGroupEnd
//               (librccl.so.1 is dlopen'ed on first use: a single-device user never needs it; one single-process
//                communicator over the devices, created by ncclCommInitAll)
//       "copy": devices repeat in device_ids (several shares on one GPU: how a one-GPU box tests this path), or
//               TDT_MULTI_TRANSPORT=copy — hipMemcpyPeerAsync on member 0's stream behind one event per member
//   assemble_kernel (tdt_assemble_tiles) on member 0's stream de-interleaves into the front's image.
// Nothing blocks the host; tdt_finish / tdt_image_read on the front wait for member 0's stream, which is ordered behind
// every other member's work by the gather.
#include <dlfcn.h>

#include <cstdlib>
#include <cstring>
#include <new>

#include "tdt_internal.hpp"

namespace tdt {

// the handful of RCCL entry points used, bound at run time (types as in <rccl/rccl.h>: ncclComm_t is an opaque pointer,
// ncclResult_t / ncclDataType_t are ints, ncclFloat32 = 7)
struct Rccl {
  void *lib = nullptr;
  int (*CommInitAll)(void **comms, int ndev, const int *devlist) = nullptr;
  int (*CommDestroy)(void *comm) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  int (*Gather)(const void *send, void *recv, size_t sendcount, int datatype, int root, void *comm, hipStream_t stream) = nullptr;
  const char *(*GetErrorString)(int) = nullptr;
};
constexpr int kNcclFloat32 = 7;

struct Multi {
  std::vector<tdt_ctx *> member;
  std::vector<tdt_image *> tile;          // per member: its tile buffer (image of 32 x 32*tiles_per_member)
  int tiles_per_member = 0;
  float *gathered = nullptr; size_t gathered_bytes = 0;   // on member 0: [n][tiles_per_member][32][32] RGBA
  std::vector<hipEvent_t> ev_begin, ev_traced;             // per member, on its stream
  hipEvent_t ev_gathered = nullptr, ev_assembled = nullptr;   // on member 0's stream
  bool timed = false;                                      // a frame has been dispatched since the events were made
  bool distinct = true;                                    // no device id repeats
  int transport = 0;                                       // 0 undecided, 1 rccl, 2 copy
  const char *last_transport = "";
  Rccl rccl;
  std::vector<void *> comm;
  int fail_member = -1;                                    // tdt_debug_multi_fail: the next raytracer dispatch fails at this member (tests)
  std::vector<void *> carry; size_t carry_bytes = 0;       // progressive passes: per member, the hit-record carry of its tile buffer
  int acc_tiles = 0;                                       // tiles per member of the running sums the tile buffers hold (0: none)
};

static int member_fail(tdt_ctx *front, tdt_ctx *m, int rc) {
  return fail(front, rc, std::string("device ") + std::to_string(m->device) + ": " + tdt_last_error(m));
}
#define TDT_MEMBER(front, m, call) do { const int rc_ = (call); if (rc_ != TDT_OK) return member_fail((front), (m), rc_); } while (0)

static int rccl_load(tdt_ctx *front) {
  Rccl &R = front->multi->rccl;
  if (R.lib) return TDT_OK;
  // a copy that is already in the process (a host that linked RCCL, or PyTorch's) wins: one RCCL per process
  for (const char *name : {"librccl.so.1", "librccl.so"}) { R.lib = dlopen(name, RTLD_NOW | RTLD_NOLOAD); if (R.lib) break; }
  if (!R.lib) for (const char *name : {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"}) { R.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL); if (R.lib) break; }
  if (!R.lib) return fail(front, TDT_ERR_HIP, std::string("cannot load librccl.so.1 for the multi-device gather: ") + dlerror());
  auto sym = [&](const char *n) { return dlsym(R.lib, n); };
  R.CommInitAll = (decltype(R.CommInitAll))sym("ncclCommInitAll");
  R.CommDestroy = (decltype(R.CommDestroy))sym("ncclCommDestroy");
  R.GroupStart = (decltype(R.GroupStart))sym("ncclGroupStart");
  R.GroupEnd = (decltype(R.GroupEnd))sym("ncclGroupEnd");
  R.Gather = (decltype(R.Gather))sym("ncclGather");
  R.GetErrorString = (decltype(R.GetErrorString))sym("ncclGetErrorString");
  if (!R.CommInitAll || !R.CommDestroy || !R.GroupStart || !R.GroupEnd || !R.Gather || !R.GetErrorString) {
    R.lib = nullptr;
    return fail(front, TDT_ERR_HIP, "librccl.so.1 lacks ncclCommInitAll / ncclGather / ncclGroupStart");
  }
  return TDT_OK;
}
static int rccl_fail(tdt_ctx *front, int rc, const char *what) {
  return fail(front, TDT_ERR_HIP, std::string(what) + ": " + front->multi->rccl.GetErrorString(rc));
}

// decide how tile buffers reach member 0 (once per context, at the first frame)
static int choose_transport(tdt_ctx *front) {
  Multi &M = *front->multi;
  if (M.transport) return TDT_OK;
  const char *env = getenv("TDT_MULTI_TRANSPORT");
  const bool want_copy = env && !std::strcmp(env, "copy"), want_rccl = env && !std::strcmp(env, "rccl");
  if (want_rccl && !M.distinct) return fail(front, TDT_ERR_INVALID_OPERATION, "TDT_MULTI_TRANSPORT=rccl, but a device id repeats: RCCL cannot form a communicator");
  if (want_copy || !M.distinct) { M.transport = 2; return TDT_OK; }
  const int rc = rccl_load(front);
  if (rc != TDT_OK) return rc;
  std::vector<int> devs;
  for (tdt_ctx *m : M.member) devs.push_back(m->device);
  M.comm.assign(M.member.size(), nullptr);
  const int nrc = M.rccl.CommInitAll(M.comm.data(), (int)devs.size(), devs.data());
  if (nrc != 0) { M.comm.clear(); return rccl_fail(front, nrc, "ncclCommInitAll"); }
  M.transport = 1;
  return TDT_OK;
}

void multi_destroy(tdt_ctx *front) {
  Multi *M = front->multi;
  for (tdt_ctx *m : M->member) { (void)hipSetDevice(m->device); (void)hipStreamSynchronize(m->stream); }
  for (void *c : M->comm) if (c) (void)M->rccl.CommDestroy(c);
  for (tdt_compute *c : front->computes) delete c;          // the replicas die with their member contexts
  for (tdt_buffer *b : front->buffers) delete b;
  for (tdt_image *i : front->images) delete i;
  if (!M->member.empty()) {
    (void)hipSetDevice(M->member[0]->device);
    if (front->counters) (void)hipFree(front->counters);      // tdt_selftest on the front
    if (M->gathered) (void)hipFree(M->gathered);
    for (size_t i = 0; i < M->carry.size(); i++) if (M->carry[i]) { (void)hipSetDevice(M->member[i]->device); (void)hipFree(M->carry[i]); }
    (void)hipSetDevice(M->member[0]->device);
    if (M->ev_gathered) (void)hipEventDestroy(M->ev_gathered);
    if (M->ev_assembled) (void)hipEventDestroy(M->ev_assembled);
  }
  for (size_t i = 0; i < M->member.size(); i++) {
    (void)hipSetDevice(M->member[i]->device);
    if (i < M->ev_begin.size()) { (void)hipEventDestroy(M->ev_begin[i]); (void)hipEventDestroy(M->ev_traced[i]); }
    tdt_ctx_destroy(M->member[i]);
  }
  delete M;
  delete front;
}

int multi_finish(tdt_ctx *front) {
  Multi &M = *front->multi;
  for (size_t i = M.member.size(); i-- > 0;) TDT_MEMBER(front, M.member[i], tdt_finish(M.member[i]));
  return TDT_OK;
}

int multi_compute_create(tdt_ctx *front, int kind, tdt_compute **out) {
  Multi &M = *front->multi;
  tdt_compute *c = new (std::nothrow) tdt_compute();
  if (!c) return fail(front, TDT_ERR_HIP, "out of host memory");
  c->ctx = front; c->kind = kind; c->part_rank = 0; c->part_world = 1;
  const int n = (int)M.member.size();
  for (int i = 0; i < n; i++) {
    tdt_compute *r = nullptr;
    int rc = tdt_compute_create(M.member[i], kind, &r);
    if (rc == TDT_OK && kind == TDT_PROGRAM_RAYTRACER) rc = tdt_set_partition(r, i, n);
    if (rc != TDT_OK) { delete c; return member_fail(front, M.member[i], rc); }
    c->replicas.push_back(r);
  }
  front->computes.push_back(c);
  *out = c;
  return TDT_OK;
}

void multi_compute_destroy(tdt_compute *c) {
  for (tdt_compute *r : c->replicas) tdt_compute_destroy(r);
  erase_from(c->ctx->computes, c);
  delete c;
}

int multi_set_i32(tdt_compute *c, const char *name, int32_t v) {
  for (tdt_compute *r : c->replicas) TDT_MEMBER(c->ctx, r->ctx, tdt_set_i32(r, name, v));
  if (!c->replicas.empty() && c->kind == TDT_PROGRAM_RAYTRACER) {   // the front keeps the camera too: it sizes the tile buffers
    const tdt_compute *r = c->replicas[0];
    c->image_width = r->image_width; c->image_height = r->image_height; c->samples_per_pixel = r->samples_per_pixel; c->max_bounce = r->max_bounce;
  }
  return TDT_OK;
}

int multi_set_vec3f(tdt_compute *c, const char *name, float x, float y, float z) {
  for (tdt_compute *r : c->replicas) TDT_MEMBER(c->ctx, r->ctx, tdt_set_vec3f(r, name, x, y, z));
  return TDT_OK;
}

int multi_buffer_create(tdt_ctx *front, const void *data, size_t bytes, tdt_buffer **out) {
  Multi &M = *front->multi;
  tdt_buffer *b = new (std::nothrow) tdt_buffer();
  if (!b) return fail(front, TDT_ERR_HIP, "out of host memory");
  b->ctx = front; b->dev = nullptr; b->bytes = bytes; b->version = 0;
  for (tdt_ctx *m : M.member) {
    tdt_buffer *r = nullptr;
    const int rc = tdt_buffer_create(m, data, bytes, &r);
    if (rc != TDT_OK) { for (tdt_buffer *q : b->replicas) tdt_buffer_destroy(q); delete b; return member_fail(front, m, rc); }
    b->replicas.push_back(r);
  }
  front->buffers.push_back(b);
  *out = b;
  return TDT_OK;
}

void multi_buffer_destroy(tdt_buffer *b) {
  tdt_ctx *front = b->ctx;
  for (tdt_buffer *r : b->replicas) tdt_buffer_destroy(r);      // (unbinds it from its member)
  for (auto &s : front->ssbo) if (s == b) s = nullptr;
  if (front->atomic0 == b) front->atomic0 = nullptr;
  erase_from(front->buffers, b);
  delete b;
}

int multi_bind_buffer_base(tdt_ctx *front, int target, unsigned slot, tdt_buffer *b) {
  Multi &M = *front->multi;
  for (size_t i = 0; i < M.member.size(); i++)
    TDT_MEMBER(front, M.member[i], tdt_bind_buffer_base(M.member[i], target, slot, b ? b->replicas[i] : nullptr));
  if (target == TDT_SHADER_STORAGE_BUFFER && slot < (unsigned)kNumSlots) front->ssbo[slot] = b;
  if (target == TDT_ATOMIC_COUNTER_BUFFER && slot == 0) front->atomic0 = b;
  return TDT_OK;
}

int multi_buffer_sub_data(tdt_buffer *b, size_t offset, size_t bytes, const void *data) {
  for (tdt_buffer *r : b->replicas) TDT_MEMBER(b->ctx, r->ctx, tdt_buffer_sub_data(r, offset, bytes, data));
  return TDT_OK;
}

int multi_image_create(tdt_ctx *front, void *device_ptr, int width, int height, tdt_image **out) {
  Multi &M = *front->multi;
  tdt_image *img = new (std::nothrow) tdt_image();
  if (!img) return fail(front, TDT_ERR_HIP, "out of host memory");
  tdt_ctx *m0 = M.member[0];
  const int rc = device_ptr ? tdt_image_wrap_device(m0, device_ptr, width, height, &img->full)
                            : tdt_image_create_rgba32f(m0, width, height, &img->full);
  if (rc != TDT_OK) { delete img; return member_fail(front, m0, rc); }
  img->ctx = front; img->w = width; img->h = height; img->owned = false; img->dev = img->full->dev;
  front->images.push_back(img);
  *out = img;
  return TDT_OK;
}

void multi_image_destroy(tdt_image *img) {
  tdt_ctx *front = img->ctx;
  if (front->image0 == img) front->image0 = nullptr;
  tdt_image_destroy(img->full);
  erase_from(front->images, img);
  delete img;
}

// per-member tile buffers, the gather target and the timing events for a frame of `tiles_per_member` work-groups per member
static int ensure_frame_buffers(tdt_ctx *front, int tiles_per_member) {
  Multi &M = *front->multi;
  const size_t n = M.member.size();
  if (M.tile.size() != n || M.tiles_per_member < tiles_per_member) {
    for (size_t i = 0; i < M.tile.size(); i++) if (M.tile[i]) tdt_image_destroy(M.tile[i]);
    M.tile.assign(n, nullptr);
    for (size_t i = 0; i < n; i++)
      TDT_MEMBER(front, M.member[i], tdt_image_create_rgba32f(M.member[i], 32, 32 * tiles_per_member, &M.tile[i]));
    M.tiles_per_member = tiles_per_member;
  }
  const size_t need = n * (size_t)M.tiles_per_member * 1024 * 16;
  TDT_HIP(front, hipSetDevice(M.member[0]->device));
  if (M.gathered_bytes < need) {
    if (M.gathered) (void)hipFree(M.gathered);
    M.gathered = nullptr; M.gathered_bytes = 0;
    TDT_HIP(front, hipMalloc((void **)&M.gathered, need));
    M.gathered_bytes = need;
  }
  if (M.ev_begin.empty()) {
    TDT_HIP(front, hipEventCreate(&M.ev_gathered));
    TDT_HIP(front, hipEventCreate(&M.ev_assembled));
    M.ev_begin.resize(n); M.ev_traced.resize(n);
    for (size_t i = 0; i < n; i++) {
      TDT_HIP(front, hipSetDevice(M.member[i]->device));
      TDT_HIP(front, hipEventCreate(&M.ev_begin[i]));
      TDT_HIP(front, hipEventCreate(&M.ev_traced[i]));
    }
  }
  return TDT_OK;
}

// A frame that fails half-way must not leave the front in a state the next frame trips over: members that were launched are
// drained (their streams may still be writing tile buffers a retry would rebind) and the "a frame has been dispatched" flag —
// which the next frame's hipStreamWaitEvent(ev_gathered) and tdt_debug_multi_timing rely on — is reset.
static void abandon_frame(tdt_ctx *front, int launched) {
  Multi &M = *front->multi;
  for (int j = 0; j < launched && j < (int)M.member.size(); j++) {
    (void)hipSetDevice(M.member[j]->device);
    (void)hipStreamSynchronize(M.member[j]->stream);
  }
  M.timed = false; M.acc_tiles = 0;
}

// the ONE gather of per-device tile buffers to the first device, then the de-interleave into the front's image
static int gather_and_assemble(tdt_ctx *front, tdt_compute *c, tdt_image *img, int width, int height, int depth) {
  Multi &M = *front->multi;
  const int n = (int)M.member.size();
  tdt_ctx *m0 = M.member[0];
  const size_t count = (size_t)M.tiles_per_member * 1024 * 4;        // floats per member
  if (M.transport == 1) {
    int nrc = M.rccl.GroupStart();
    if (nrc != 0) return rccl_fail(front, nrc, "ncclGroupStart");
    for (int i = 0; i < n && nrc == 0; i++)
      nrc = M.rccl.Gather(M.tile[i]->dev, M.gathered /* significant at the root only */, count, kNcclFloat32, 0, M.comm[i], M.member[i]->stream);
    const int erc = M.rccl.GroupEnd();
    if (nrc != 0) return rccl_fail(front, nrc, "ncclGather");
    if (erc != 0) return rccl_fail(front, erc, "ncclGroupEnd");
    M.last_transport = "rccl";
  } else {
    TDT_HIP(front, hipSetDevice(m0->device));
    for (int i = 0; i < n; i++) {
      if (i > 0) TDT_HIP(front, hipStreamWaitEvent(m0->stream, M.ev_traced[i], 0));
      TDT_HIP(front, hipMemcpyPeerAsync((char *)M.gathered + (size_t)i * count * 4, m0->device, M.tile[i]->dev, M.member[i]->device, count * 4, m0->stream));
    }
    M.last_transport = "copy";
  }
  TDT_HIP(front, hipSetDevice(m0->device));
  TDT_HIP(front, hipEventRecord(M.ev_gathered, m0->stream));
  TDT_MEMBER(front, m0, tdt_assemble_tiles(c->replicas[0], M.gathered, n, M.tiles_per_member, img->full, width, height, depth));
  TDT_HIP(front, hipEventRecord(M.ev_assembled, m0->stream));
  return TDT_OK;
}

// what: 0 = tdt_dispatch_compute (a whole frame), 1 = tdt_dispatch_accumulate (samples [spp_begin, spp_begin + spp_count) added to
// the running sums in the members' tile buffers; no gather), 2 = tdt_dispatch_resolve (resolve on every member, then gather + assemble)
static int member_launch(tdt_ctx *front, tdt_compute *c, int i, int what, int width, int height, int depth, int spp_begin, int spp_count, bool use_carry, int total_spp) {
  Multi &M = *front->multi;
  tdt_ctx *m = M.member[i];
  TDT_HIP(front, hipSetDevice(m->device));
  // copy transport: member 0's stream reads this member's tile buffer; the next frame must not overwrite it earlier
  // (with RCCL the gather runs on the member's own stream, which orders it)
  if (M.transport == 2 && M.timed && i > 0 && what != 2) TDT_HIP(front, hipStreamWaitEvent(m->stream, M.ev_gathered, 0));
  if (what != 2) TDT_HIP(front, hipEventRecord(M.ev_begin[i], m->stream));
  TDT_MEMBER(front, m, tdt_bind_image(m, 0, M.tile[i]));
  if (M.fail_member == i) { M.fail_member = -1; return fail(front, TDT_ERR_INVALID_OPERATION, "device " + std::to_string(m->device) + ": failure injected by tdt_debug_multi_fail"); }
  if (what == 0) TDT_MEMBER(front, m, tdt_dispatch_compute(c->replicas[i], width, height, depth));
  else if (what == 1) TDT_MEMBER(front, m, tdt_dispatch_accumulate(c->replicas[i], width, height, depth, spp_begin, spp_count, use_carry ? M.carry[i] : nullptr));
  else TDT_MEMBER(front, m, tdt_dispatch_resolve(c->replicas[i], width, height, depth, total_spp));
  TDT_HIP(front, hipEventRecord(M.ev_traced[i], m->stream));
  return TDT_OK;
}

static int multi_frame(tdt_compute *c, int what, int width, int height, int depth, int spp_begin, int spp_count, bool use_carry, int total_spp) {
  tdt_ctx *front = c->ctx;
  Multi &M = *front->multi;
  const int n = (int)M.member.size();
  if (!front->image0) return fail(front, TDT_ERR_INCOMPLETE, "no image bound to unit 0");
  tdt_image *img = front->image0;
  if (img->w != c->image_width || img->h != c->image_height)
    return fail(front, TDT_ERR_INVALID_OPERATION, "bound image is not camera.image_width x image_height");
  const Cover k = cover_of(c, width, height);
  const Tiles t = tiles_of(c, k);                        // the front's partition is (0, 1): totals
  if (t.total <= 0) return TDT_OK;
  const int tpm = (t.total + n - 1) / n;                 // what member 0 owns: the most
  if (what == 2 && M.acc_tiles != tpm) return fail(front, TDT_ERR_INVALID_OPERATION, "resolve without accumulated passes of the same dispatch size on this context");
  if (what == 1 && spp_begin != 0 && M.acc_tiles != tpm)
    return fail(front, TDT_ERR_INVALID_OPERATION, "a pass that continues running sums (spp_begin > 0) needs earlier passes of the same dispatch size on this context");
  int rc = ensure_frame_buffers(front, tpm);             // (a larger frame re-allocates the tile buffers: running sums start over, checked above)
  if (rc == TDT_OK) rc = choose_transport(front);
  if (rc == TDT_OK && what == 1 && use_carry) {
    const size_t need = (size_t)M.tiles_per_member * 1024 * 16 * sizeof(float);
    if (M.carry.size() != (size_t)n || M.carry_bytes < need) {
      for (size_t i = 0; i < M.carry.size(); i++) if (M.carry[i]) { (void)hipSetDevice(M.member[i]->device); (void)hipFree(M.carry[i]); }
      M.carry.assign(n, nullptr); M.carry_bytes = 0;
      for (int i = 0; i < n && rc == TDT_OK; i++) {
        if (hipSetDevice(M.member[i]->device) != hipSuccess || hipMalloc(&M.carry[i], need) != hipSuccess) rc = fail(front, TDT_ERR_HIP, "carry allocation on device " + std::to_string(M.member[i]->device));
      }
      if (rc == TDT_OK) M.carry_bytes = need;
    }
  }
  if (rc != TDT_OK) return rc;
  for (int i = 0; i < n; i++) {
    rc = member_launch(front, c, i, what, width, height, depth, spp_begin, spp_count, use_carry, total_spp);
    if (rc != TDT_OK) { abandon_frame(front, i + 1); return rc; }      // (member i may have kernels in flight already: its stream is drained too)
  }
  if (what == 1) { M.acc_tiles = tpm; return TDT_OK; }   // running sums stay in the tile buffers until the resolve
  rc = gather_and_assemble(front, c, img, width, height, depth);
  if (rc != TDT_OK) { abandon_frame(front, n); return rc; }
  M.timed = true; M.acc_tiles = 0;
  return TDT_OK;
}

int multi_dispatch_compute(tdt_compute *c, int width, int height, int depth) {
  tdt_ctx *front = c->ctx;
  Multi &M = *front->multi;
  if (c->kind == TDT_PROGRAM_OCTREE_UPDATE) {          // an edit changes every replica of the scene, identically
    for (size_t i = 0; i < M.member.size(); i++) TDT_MEMBER(front, M.member[i], tdt_dispatch_compute(c->replicas[i], width, height, depth));
    return TDT_OK;
  }
  return multi_frame(c, 0, width, height, depth, 0, 0, false, 0);
}

// Progressive passes on a node (BASELINE configs[4]'s 1024 spp as passes): every member keeps the running sums of ITS work-groups in
// its tile buffer and the hit-record carry beside it (allocated here: `carry` device memory of a single-device host would be on
// one GPU only — a non-null pointer just says "carry the records"); the resolve runs per member, then the one gather + assemble.
int multi_dispatch_accumulate(tdt_compute *c, int width, int height, int depth, int spp_begin, int spp_count, void *carry) {
  return multi_frame(c, 1, width, height, depth, spp_begin, spp_count, carry != nullptr, 0);
}
int multi_dispatch_resolve(tdt_compute *c, int width, int height, int depth, int total_spp) {
  return multi_frame(c, 2, width, height, depth, 0, 0, false, total_spp);
}

int multi_dispatch_counted(tdt_compute *c, int width, int height, int depth, uint64_t counts[8]) {
  tdt_ctx *front = c->ctx;
  Multi &M = *front->multi;
  if (c->kind != TDT_PROGRAM_RAYTRACER) return fail(front, TDT_ERR_INVALID_OPERATION, "not the raytracer program");
  const Cover k = cover_of(c, width, height);
  const Tiles t = tiles_of(c, k);
  const int n = (int)M.member.size();
  const int rc = ensure_frame_buffers(front, (t.total + n - 1) / n > 0 ? (t.total + n - 1) / n : 1);
  if (rc != TDT_OK) return rc;
  for (int j = 0; j < 8; j++) counts[j] = 0;
  for (int i = 0; i < n; i++) {
    uint64_t part[8];
    TDT_MEMBER(front, M.member[i], tdt_bind_image(M.member[i], 0, M.tile[i]));
    TDT_MEMBER(front, M.member[i], tdt_dispatch_counted(c->replicas[i], width, height, depth, part));
    for (int j = 0; j < 8; j++) counts[j] += part[j];
  }
  return TDT_OK;
}

tdt_ctx *multi_first_member(tdt_ctx *front) { return front->multi->member[0]; }

int multi_forget_costs(tdt_ctx *front) {
  for (tdt_ctx *m : front->multi->member) (void)tdt_forget_costs(m);
  return TDT_OK;
}

}  // namespace tdt

extern "C" {

int tdt_ctx_create_multi(int n_devices, const int *device_ids, tdt_ctx **out) {
  using namespace tdt;
  if (!out) return fail(nullptr, TDT_ERR_INVALID_VALUE, "null out pointer");
  *out = nullptr;
  if (n_devices < 1 || n_devices > 64 || !device_ids) return fail(nullptr, TDT_ERR_INVALID_VALUE, "need 1..64 device ids");
  tdt_ctx *front = new (std::nothrow) tdt_ctx();
  Multi *M = new (std::nothrow) Multi();
  if (!front || !M) { delete front; delete M; return fail(nullptr, TDT_ERR_HIP, "out of host memory"); }
  for (int i = 0; i < n_devices; i++) {
    tdt_ctx *m = nullptr;
    const int rc = tdt_ctx_create(device_ids[i], nullptr, &m);
    if (rc != TDT_OK) {                                  // (the message of a failed creation is already in place)
      for (tdt_ctx *q : M->member) tdt_ctx_destroy(q);
      delete M; delete front;
      return rc;
    }
    for (int j = 0; j < i; j++) if (device_ids[j] == device_ids[i]) M->distinct = false;
    M->member.push_back(m);
  }
  front->device = device_ids[0];
  front->stream = M->member[0]->stream; front->own_stream = false;
  front->num_cus = M->member[0]->num_cus;
  front->multi = M;
  *out = front;
  return TDT_OK;
}

int tdt_ctx_device_count(const tdt_ctx *ctx) { return !ctx ? 0 : (ctx->multi ? (int)ctx->multi->member.size() : 1); }

int tdt_debug_multi_timing(tdt_ctx *ctx, float *trace_ms, float *gather_ms, float *assemble_ms) {
  using namespace tdt;
  if (!ctx || !ctx->multi) return TDT_ERR_INVALID_VALUE;
  Multi &M = *ctx->multi;
  if (!M.timed) return fail(ctx, TDT_ERR_INVALID_OPERATION, "no frame has been dispatched yet");
  TDT_HIP(ctx, hipSetDevice(M.member[0]->device));
  TDT_HIP(ctx, hipEventSynchronize(M.ev_assembled));
  for (size_t i = 0; i < M.member.size(); i++) {
    TDT_HIP(ctx, hipSetDevice(M.member[i]->device));
    TDT_HIP(ctx, hipEventSynchronize(M.ev_traced[i]));
    float ms = 0.f;
    TDT_HIP(ctx, hipEventElapsedTime(&ms, M.ev_begin[i], M.ev_traced[i]));
    if (trace_ms) trace_ms[i] = ms;
  }
  TDT_HIP(ctx, hipSetDevice(M.member[0]->device));
  float g = 0.f, a = 0.f;
  TDT_HIP(ctx, hipEventElapsedTime(&g, M.ev_traced[0], M.ev_gathered));
  TDT_HIP(ctx, hipEventElapsedTime(&a, M.ev_gathered, M.ev_assembled));
  if (gather_ms) *gather_ms = g;
  if (assemble_ms) *assemble_ms = a;
  return TDT_OK;
}

int tdt_debug_multi_rccl_ranks(const tdt_ctx *ctx) {
  if (!ctx || !ctx->multi || ctx->multi->transport != 1) return 0;
  int n = 0;
  for (void *c : ctx->multi->comm) n += c ? 1 : 0;
  return n;
}

int tdt_debug_multi_fail(tdt_ctx *ctx, int member) {
  if (!ctx || !ctx->multi || member < -1 || member >= (int)ctx->multi->member.size()) return TDT_ERR_INVALID_VALUE;
  ctx->multi->fail_member = member;
  return TDT_OK;
}

const char *tdt_debug_multi_transport(const tdt_ctx *ctx) { return (ctx && ctx->multi) ? ctx->multi->last_transport : ""; }

}  // extern "C"
