/*
 * tdt_rt.h — C ABI of libtdtrt.so, the MI355X (gfx950) drop-in for the per-pixel voxel path
 * trace of Avokadoen/tdt4230_project_raytracing.
 *
 * The reference has no FFI layer: its boundary is the `src/renderer` GL-wrapper API that
 * main.rs / camera.rs / octree.rs call on the thread that owns the GL context.  Every entry
 * point below replaces one of those calls one-for-one (cited as file:line of the reference);
 * a Rust maintainer binds them with the `extern "C"` block shown in INTEGRATION.md.
 *
 * Conventions (the GL contract, kept):
 *   - context-affine, no internal locking: call from one thread per context;
 *   - host pointers are borrowed for the duration of the call only; uploads COPY (glBufferData);
 *   - every call returns an int: 0 = TDT_OK, otherwise a TDT_ERR_* code (never aborts); the
 *     message for the last error of a context is available from tdt_last_error();
 *   - handles are owned by their context and die with it;
 *   - all dispatches are asynchronous on the context's HIP stream; tdt_finish() is glFinish().
 * There is NO CPU fallback: without a usable HIP device tdt_ctx_create fails with
 * TDT_ERR_NO_DEVICE and nothing else can be called.
 */
#ifndef TDT_RT_H
#define TDT_RT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct tdt_ctx tdt_ctx;         /* the GL context            main.rs:58-61,108-112 */
typedef struct tdt_compute tdt_compute; /* renderer::ComputeShader   compute_shader.rs:10-13 */
typedef struct tdt_buffer tdt_buffer;   /* renderer::vbo::VertexBufferObject   vbo.rs:9-13 */
typedef struct tdt_image tdt_image;     /* renderer::texture::Texture          texture.rs:7-14 */

enum {
  TDT_OK = 0,
  TDT_ERR_NO_DEVICE = 1,          /* no HIP device / HIP runtime failure at context creation  */
  TDT_ERR_HIP = 2,                /* a HIP call failed (check_for_gl_error analogue, mod.rs:62-68) */
  TDT_ERR_INVALID_ENUM = 0x0500,  /* = GL_INVALID_ENUM,      mod.rs:47 */
  TDT_ERR_INVALID_VALUE = 0x0501, /* = GL_INVALID_VALUE,     mod.rs:48 */
  TDT_ERR_INVALID_OPERATION = 0x0502, /* = GL_INVALID_OPERATION, mod.rs:49 */
  TDT_ERR_VARIABLE_NOT_FOUND = 3, /* InitializeErr::VariableNotFound, program.rs:144-165, mod.rs:31 */
  TDT_ERR_INCOMPLETE = 4          /* dispatch with a required binding / uniform missing */
};

/* kinds for tdt_compute_create (the two compute programs the reference links) */
enum {
  TDT_PROGRAM_RAYTRACER = 0,      /* assets/shaders/raytracer.comp */
  TDT_PROGRAM_OCTREE_UPDATE = 1   /* assets/shaders/octree_update.comp (next row SURVEY §8f-2): one voxel edit per
                                     invocation; reads slots 0,5,6,7 + the atomic counter, group size {1,1,1} */
};
/* targets for tdt_bind_buffer_base (values are the GL enums the reference passes) */
enum {
  TDT_SHADER_STORAGE_BUFFER = 0x90D2, /* main.rs:352,383,408,430,448; octree.rs:67,98,144 */
  TDT_ATOMIC_COUNTER_BUFFER = 0x92C0  /* octree.rs:115 (accepted, unused by the trace) */
};
/* SSBO slots read by raytracer.comp (binding = N in the shader) */
enum {
  TDT_SLOT_CELLS = 0,        /* Node{uint value; uint type}[]          raytracer.comp:180-182 */
  TDT_SLOT_MATERIALS = 1,    /* {int type, attribute_index, albedo_index}[]   :189-196 */
  TDT_SLOT_ALBEDOS = 2,      /* {float x,y,z}[]                                :203-210 */
  TDT_SLOT_METAL = 3,        /* {float fuzz}[]                                 :214-219 */
  TDT_SLOT_DIELECTRIC = 4,   /* {float ir}[]                                   :222-227 */
  TDT_SLOT_DELTA = 5,        /* DeltaNode{vec3 pos; float type; float value}[] stride 32   octree_update.comp:41-48 */
  TDT_SLOT_OCTREE_FLOATS = 6,/* {vec4 min_point; float scale, inv_scale, inv_cell_count} :150-158 */
  TDT_SLOT_OCTREE_INTS = 7   /* {int max_depth, max_iter, cell_count}          :159-166 */
};

/* ---- context ------------------------------------------------------------------------- */
/* Replaces GL context creation + make_current (main.rs:58-61,108-112).  `stream` is a
 * hipStream_t to launch on (NULL: the context creates its own non-blocking stream). */
int tdt_ctx_create(int device_id, void *stream, tdt_ctx **out);
/* ONE context over n_devices GPUs, driven by the one thread that owns it — the reference's shape (a single
 * GL context on a single thread, main.rs:58-61,105-112), so a host bound per INTEGRATION.md drives a whole node without
 * knowing it.  Every call on the returned context and on handles created from it means what it means on a single-device
 * context: buffer uploads / sub-data / binds / uniforms / edit dispatches are replicated to every device (the scene is
 * read-only and small: <= 64 MB against 288 GB); tdt_dispatch_compute of the raytracer launches each device's share of the
 * 32x32 work-groups (t % n == i, as tdt_set_partition) on that device's own stream, brings the per-device tile buffers to
 * the first device with ONE RCCL gather (librccl.so.1, loaded on first use; a single-process communicator over the
 * devices) and de-interleaves them there into the bound image, so tdt_image_read / tdt_image_read_rgba8 /
 * tdt_image_device_ptr see the assembled frame on device_ids[0].  Device ids may repeat (several shares on one GPU — how
 * the path is tested on a one-GPU box); RCCL cannot form a communicator then and the gather becomes peer copies ordered
 * by events (TDT_MULTI_TRANSPORT=copy forces that, =rccl forbids it).  Progressive passes work on a node too:
 * tdt_dispatch_accumulate keeps every device's running sums (and, when carry_device_ptr is non-NULL, its hit-record carry, in
 * memory the context allocates per device — the pointer itself is not used) in that device's tile buffer, and
 * tdt_dispatch_resolve resolves per device, then gathers and assembles.  A dispatch that fails on one device drains the
 * devices already launched and leaves the context ready for the next frame.  Not available on a multi-device context
 * (TDT_ERR_INVALID_OPERATION): tdt_set_partition, tdt_dispatch_counted_range. */
int tdt_ctx_create_multi(int n_devices, const int *device_ids, tdt_ctx **out);
/* number of devices behind a context (1 for tdt_ctx_create) */
int tdt_ctx_device_count(const tdt_ctx *ctx);
void tdt_ctx_destroy(tdt_ctx *ctx);
/* glFinish: block until every dispatch of this context has completed. */
int tdt_finish(tdt_ctx *ctx);
/* message of the most recent error on this context ("" if none); ctx may be NULL for
 * errors of tdt_ctx_create itself */
const char *tdt_last_error(const tdt_ctx *ctx);
/* InitializeErr's Display (mod.rs:44-59) for a code */
const char *tdt_strerror(int code);

/* ---- programs / uniforms --------------------------------------------------------------- */
/* Shader::from_resources + Program::from_shaders + ComputeShader::new
 * (shader.rs:26, program.rs:101, compute_shader.rs:15-26; called main.rs:156-160). */
int tdt_compute_create(tdt_ctx *ctx, int kind, tdt_compute **out);
void tdt_compute_destroy(tdt_compute *c);          /* Program::drop, program.rs:169 */
/* the COMPUTE_WORK_GROUP_SIZE query of compute_shader.rs:18: {32,32,1} */
int tdt_compute_group_size(const tdt_compute *c, int out[3]);
/* Program::set_i32 / set_f32 / set_vector3_f32 / set_vector3_i32 (program.rs:35-83): uniform
 * addressed by its GLSL name, e.g. "camera.image_width"; unknown name or wrong type ->
 * TDT_ERR_VARIABLE_NOT_FOUND.  Takes effect for the next dispatch. */
int tdt_set_i32(tdt_compute *c, const char *name, int32_t value);
int tdt_set_f32(tdt_compute *c, const char *name, float value);
int tdt_set_vec3f(tdt_compute *c, const char *name, float x, float y, float z);
int tdt_set_vec3i(tdt_compute *c, const char *name, int32_t x, int32_t y, int32_t z);

/* ---- buffers ---------------------------------------------------------------------------- */
/* VertexBufferObject::new::<T>(Vec<T>, ..) = glGenBuffers + glBufferData (vbo.rs:32-55):
 * copies `bytes` bytes to device memory. */
int tdt_buffer_create(tdt_ctx *ctx, const void *data, size_t bytes, tdt_buffer **out);
void tdt_buffer_destroy(tdt_buffer *b);
/* gl::BindBufferBase(target, slot, id) as called by main.rs:352-448 and octree.rs:67-144 */
int tdt_bind_buffer_base(tdt_ctx *ctx, int target, unsigned slot, tdt_buffer *b);
/* gl::BufferSubData (octree.rs:174): how Octree::update_vbo hands the delta nodes to the edit program */
int tdt_buffer_sub_data(tdt_buffer *b, size_t offset, size_t bytes, const void *data);
/* NEW (the reference never reads a buffer back): finishes the stream and copies bytes to dst */
int tdt_buffer_read(tdt_buffer *b, size_t offset, size_t bytes, void *dst);

/* ---- image ------------------------------------------------------------------------------ */
/* Texture::new_2d(TEXTURE0, 0, RGBA32F, RGBA, w, h) (texture.rs:47-75, camera.rs:158-165):
 * W*H*4 floats in device memory, row 0 = bottom scan line, zero-filled (the reference leaves
 * it undefined). */
int tdt_image_create_rgba32f(tdt_ctx *ctx, int width, int height, tdt_image **out);
/* Same, but over device memory the caller owns (e.g. a torch tensor's data_ptr) — used by
 * the multi-GPU harness so the per-rank tile can be handed to RCCL without a copy. */
int tdt_image_wrap_device(tdt_ctx *ctx, void *device_ptr, int width, int height, tdt_image **out);
void tdt_image_destroy(tdt_image *img);
/* glBindImageTexture(unit, ..) of texture.rs:71: only unit 0 exists in raytracer.comp:4 */
int tdt_bind_image(tdt_ctx *ctx, unsigned unit, tdt_image *img);
int tdt_image_width(const tdt_image *img);
int tdt_image_height(const tdt_image *img);
void *tdt_image_device_ptr(const tdt_image *img);
/* NEW (the reference never reads back; quad.frag samples the texture): finishes the stream and
 * copies W*H*4 floats to dst */
int tdt_image_read(tdt_image *img, float *dst);
/* NEXT ROW SURVEY §8f-4, presentation: the RGBA8 frame the reference's quad pass (assets/shaders/quad.frag:10,
 * main.rs:582-600) leaves in a back buffer of the texture's size — per channel clamp to [0,1] (NaN -> 0), x 255, round half
 * to even (pinned on llvmpipe) — converted on the GPU, then W*H*4 bytes copied to dst after finishing the stream.
 * Texture row 0 is the bottom scan-line; top_down = 1 writes the top scan-line first (image-file order). */
int tdt_image_read_rgba8(tdt_image *img, int top_down, uint8_t *dst);

/* ---- dispatch --------------------------------------------------------------------------- */
/* ComputeShader::dispatch_compute(width, height, depth) (compute_shader.rs:28-38; called with
 * (W+1, H+1, 1) at main.rs:579, and by Octree::update_vbo octree.rs:179,181 for the edit program):
 * work-group counts are max(dim / group_size, 1) by integer floor division; for the raytracer the shader has no bounds check, image stores outside the image are dropped;
 * followed by the image-access barrier (= stream order here).  Asynchronous. */
int tdt_dispatch_compute(tdt_compute *c, int width, int height, int depth);

/* ---- extensions with no reference counterpart (documented in DESIGN.md) ------------------ */
/* Tile partition for one-process-per-GPU rendering: the covered image is cut into the
 * reference's own 32x32 work-groups, numbered row-major t = gy * ceil(cover_w/32) + gx; subsequent
 * dispatches of `c` trace only the groups with t % world == rank.  Default (0,1).
 * The bound image is then either the full W x H image (only owned groups are written) or this
 * rank's TILE BUFFER: an image of width 32 and height 32*n, n >= owned groups, i.e. the owned
 * groups packed as [k][32][32] RGBA in the order k = 0,1,.. <-> t = rank + k*world. */
int tdt_set_partition(tdt_compute *c, int rank, int world);
/* number of groups a dispatch of (width,height,depth) gives this rank; optionally the groups
 * per row and the total */
int tdt_owned_tiles(const tdt_compute *c, int width, int height, int depth, int *tiles_x, int *tiles_total);
/* number of pixels a dispatch of (width,height,depth) writes under the current partition */
int64_t tdt_covered_pixels(const tdt_compute *c, int width, int height, int depth);
/* De-interleave gathered tile buffers — `gathered` = device memory [world][tiles_per_rank][32][32]
 * RGBA as produced by `world` ranks — into the full image `dst` (the step after the RCCL gather). */
int tdt_assemble_tiles(tdt_compute *c, const void *gathered, int world, int tiles_per_rank, tdt_image *dst,
                       int width, int height, int depth);
/* Progressive form of the sample loop: adds samples [spp_begin, spp_begin+spp_count) of every
 * covered pixel, in sample order, to the bound image's rgb running sums (and keeps the
 * shader's loop-carried temporaries in `carry`, 16 floats per pixel of the bound image, in
 * device memory; NULL = start from / discard the initial state). */
int tdt_dispatch_accumulate(tdt_compute *c, int width, int height, int depth, int spp_begin, int spp_count,
                            void *carry_device_ptr);
/* image = clamp(sqrt(sum / total_spp), 0, 1), alpha = 1: raytracer.comp:249-251 — EVERY covered pixel of the bound image, whatever its
 * alpha (a pixel no pass wrote resolves from the zeros or whatever the caller put there).  (Inside tdt_dispatch_compute, and only for a
 * frame whose miss pre-pass ran, the library's own resolve leaves the pixels that pass finished — alpha 1 — alone.) */
int tdt_dispatch_resolve(tdt_compute *c, int width, int height, int depth, int total_spp);
/* Instrumented dispatch (measurement only, slower): same image as tdt_dispatch_compute, and
 * returns event totals: [0] pixels written, [1] OctreeHit calls, [2] traversal iterations,
 * [3] Node loads (tree levels visited), [4] Lambertian, [5] metal, [6] dielectric scatters,
 * [7] hits on unknown material types.  Synchronous.  These define the algorithmic bytes. */
int tdt_dispatch_counted(tdt_compute *c, int width, int height, int depth, uint64_t counts[8]);
/* the same for one launch of a progressive frame: tdt_dispatch_accumulate(.., spp_begin, spp_count, carry), instrumented */
int tdt_dispatch_counted_range(tdt_compute *c, int width, int height, int depth, int spp_begin, int spp_count,
                               void *carry_device_ptr, uint64_t counts[8]);
/* Drop the per-pixel cost history of the context: the next tdt_dispatch_compute is scheduled like the first frame of a
 * context (two-phase, probe in image order) whatever was traced before.  Only the schedule changes, never a pixel.
 * (Measurement: bench.py times frames "the scheduler has not seen" with it.) */
int tdt_forget_costs(tdt_ctx *ctx);
/* measurement aid: enable != 0 records HIP events around the launches of every following tdt_dispatch_compute; ms (may be
 * NULL) receives the times of the LAST frame: {probe launch, main launch (+ the sort that orders it), resolve} for a
 * two-phase frame, {0, the one launch, 0} otherwise.  Blocks until that frame has finished. */
int tdt_debug_phase_timing(tdt_ctx *ctx, int enable, float ms[3]);
/* multi-device contexts: times of the last raytracer dispatch — trace_ms[i] for each device (its own events), then on the
 * first device gather_ms (from the end of ITS trace to the end of the gather: includes waiting for the slowest device) and
 * assemble_ms.  Blocks until the frame has finished. */
int tdt_debug_multi_timing(tdt_ctx *ctx, float *trace_ms /* n_devices */, float *gather_ms, float *assemble_ms);
/* which transport the last gather of a multi-device context used: "rccl", "copy", or "" */
const char *tdt_debug_multi_transport(const tdt_ctx *ctx);
/* ranks of the RCCL communicator the context really created (0: none — single-device context, or the copy transport) */
int tdt_debug_multi_rccl_ranks(const tdt_ctx *ctx);
/* test hook: the next raytracer dispatch of a multi-device context fails at device index `member` after the devices before
 * it were launched (-1: cancel) — exercises the drain-and-reset path of a half-launched frame */
int tdt_debug_multi_fail(tdt_ctx *ctx, int member);

/* ---- scene ingest on the GPU ------------------------------------------------------------------------
 * The step the reference never wrote (its call is commented out, main.rs:218-224): turn the voxel list its PLY loader
 * yields (ply_point_loader.rs:102-319: PlyFileContent{voxels, albedos, min_point}) into the indirect-cell octree
 * raytracer.comp reads.  Everything between the upload of the voxel list and the finished buffers runs in HIP kernels:
 * Morton keys (x, y, z bit of a level = one child digit, most significant level first) -> stable LSD radix sort ->
 * last-duplicate-wins unique -> per level, bottom-up, segment heads + prefix sum + 8-child reduce (uniform subtrees merge
 * into one LEAF) -> one prefix sum over the MIXED flags of all levels = breadth-first cell numbers -> node emission.
 * Byte-identical to the host builder (tdt_scene_from_ply / tdt_scene_generate in libtdthost.so). */
/* core: n voxels {x, y, z, material index + 1 (1..254)} in grid coordinates [0, 2^depth)^3 (others are dropped; of
 * duplicates the last wins) -> a new cells buffer of *n_cells cells (64 B each) on the context. */
int tdt_octree_build_cells(tdt_ctx *ctx, const int32_t *voxels_xyzm, size_t n_voxels, int depth, tdt_buffer **cells,
                           uint32_t *n_cells);
/* the whole of tdt_scene_from_ply on the GPU: voxels {x, y, z, colour key}, the loader's min_point and palette
 * (key -> r,g,b; n_palette entries in ascending key order).  Creates the seven buffers raytracer.comp reads and returns
 * them in out_slots[0,1,2,3,4,6,7] (out_slots[5] = NULL) WITHOUT binding them; *max_depth / *cell_count are what
 * OctreeInts holds (cell_count = the smallest power of two >= 1024 that holds the tree). */
int tdt_octree_build_from_points(tdt_ctx *ctx, const int32_t *voxels_xyzk, size_t n_voxels, const int32_t min_point[3],
                                 const uint32_t *palette_keys, const uint8_t *palette_rgb, size_t n_palette, int z_up,
                                 int max_iter, tdt_buffer *out_slots[8], int32_t *max_depth, int32_t *cell_count);

/* ---- voxel edits --------------------------------------------------------------------------------------
 * tdt_dispatch_compute of a TDT_PROGRAM_OCTREE_UPDATE program runs octree_update.comp's invocations.  The reference's
 * invocations race when their paths collide (its own comment, octree_update.comp:70-71); the defined result here is the
 * one its only runnable implementation produces — invocations one after the other, x fastest.  A dispatch of more than one
 * invocation is executed in parallel when that provably gives the same bytes: a planning pass walks every invocation's path
 * read-only, marks the nodes it would write, allocates the cells it needs by a prefix sum over the invocations (so each gets
 * the counter values the serial order would hand it), and checks that no invocation reads or writes a node another one
 * writes and that the cells to be allocated are untouched; then one lane per invocation applies its edit.  Any doubt
 * (collision, a walk that leaves the buffer, a non-empty node in the free pool) and the ordered one-lane walk runs instead
 * — decided on the device, no host round trip.  mode: 0 = as described, 1 = always the ordered walk. */
int tdt_debug_edit_mode(tdt_ctx *ctx, int mode);
/* which path the last edit dispatch of the context took: 0 none yet, 1 ordered walk, 2 parallel.  Synchronises. */
int tdt_debug_last_edit_path(tdt_ctx *ctx);
/* which build of the trace kernel the context's last trace launch ran: out = {form: 0 the literal float index, 1 the exact form of a
 * power-of-two cell_count, 2 per-cell thresholds (any other count); compile-time depth (0 = the general kernel); tree inside the LDS
 * table; whole-depth table; bricks; the build that skips multiplications by a scale of 1.0f}.  Every build writes the same pixels; this
 * lets a test tell a scene that fell back to the general kernel from one that runs its specialised build. */
int tdt_debug_last_variant(const tdt_ctx *ctx, int out[6]);
/* pass / lane statistics of the PRODUCT trace kernels (how many traversal and event passes the waves ran, how many lanes were live in
 * each code region: the STAT_* rows of csrc/tdt_rt.hip) since the last reset — collected only by a -DTDT_STATS build of the library
 * (tools/loss_budget.py builds one beside the product library; the product library answers TDT_ERR_INVALID_OPERATION).  The first
 * call switches the collection on; reset != 0 clears the totals after reading them.  out: n_words entries (0: the 32 totals) —
 * 32 totals, then the time (100 MHz ticks) the first wave met the end of the pixel queue, then up to 8192 per-wave end times. */
int tdt_debug_stats(tdt_ctx *ctx, uint64_t *out, int n_words, int reset);
/* lane-utilisation diagnostics of the last tdt_dispatch_counted of this context (32 totals; layout in
 * csrc/trace_device.hpp `Counters`); development aid */
int tdt_debug_counters(tdt_ctx *ctx, uint64_t out[32]);
/* development aids, filled by tdt_dispatch_counted (tools/timeline.py, tools/pixel_log.py):
 * wave_ends: n <= 16384 + 256 entries — per-wave end times (100 MHz ticks; index = block * 16 + wave), then from
 * entry 16384 two 128-bin histograms (0.1 ms bins) of pixel durations: pixels finished while the queue still had
 * work / by waves that had seen its end.  pixel_log (only when TDT_PIXEL_LOG is set in the environment): 8 u32
 * per queue slot — traversal steps, path events, wave passes, start tick, end tick, wave, event passes, threshold. */
int tdt_debug_wave_ends(tdt_ctx *ctx, uint64_t *out, int n);
int tdt_debug_pixel_log(tdt_ctx *ctx, uint32_t *out, size_t n_u32);
/* exhaustive (all 2^32 inputs) check of the kernels' short correctly-rounded rcp (0) / sqrt (1) /
 * rsq (2) forms against the IEEE expressions, of v_fract_f32 against x - floor(x) for x >= 0 (4), and of the top-level
 * jump tables' claim (5: the 4-level table of LDS-resident trees, 7: the 5-level table of the others): for every coordinate
 * in [0,1) outside the table's bands and every cell index below its bounds (128 / 1024 / 8192 by level) the x decision of
 * treeLookup's first four / five levels is the coordinate's binary digit; and of the short form of CubeHit's normal (9: every
 * bit pattern of the dominant component x zeros / smaller values / ties / NaN on the other axes x ray directions with zero,
 * denormal, inf and NaN components: where its guard holds it returns the bits of the literal normalise-orient-normalise
 * sequence); and of the two one-parameter divisions taken through reciprocal + residual step (11: pow's (m - 1) / (m + 1) for
 * every mantissa, reflectance's (1 - x) / (1 + x) for every x); and of the claim the bricks of depth-8 trees rest on (13:
 * fl(v + f) - v depends on an integer v < 2^22 only through floor(log2 v), for every f in [0, 1); 15: the bands of their
 * level-5 table, one per cell index instead of one for all).  *mismatches must be 0.  Modes 3, 6, 8, 10, 12, 14 and 16 check the
 * harness: the raw reciprocal seed, the claims without the bands, the short normal without its guard, the divisions without
 * the residual step, the wrong binade, and half the band, must fail. */
int tdt_selftest(tdt_ctx *ctx, int which, uint64_t *mismatches);
/* A cell_count that is not a power of two (the reference's own: 100000, main.rs:459) takes treeLookup's x index
 * (raytracer.comp:376-378) through two per-cell thresholds on the level's coordinate instead of the float formula (csrc/
 * trace_device.hpp, x_thresholds).  This checks that claim exhaustively for one (cell_count, inv_cell_count) pair: every
 * coordinate f in [0,1) x every cell index below n_cells (<= 65535) against the literal formula; *mismatches must be 0.
 * *shape_ok = 0: some cell's index is not a two-step function of f — such a scene runs the literal kernel.  shift != 0 moves
 * every threshold by that many ulps first (the harness: mismatches must appear). */
int tdt_selftest_index(tdt_ctx *ctx, int32_t cell_count, float inv_cell_count, uint32_t n_cells, int shift, uint64_t *mismatches,
                       int *shape_ok);

#ifdef __cplusplus
}
#endif
#endif /* TDT_RT_H */
