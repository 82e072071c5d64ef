/*
 * pathtrace_oracle — TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C, scalar fp32) of the reference's per-pixel voxel path trace,
 * assets/shaders/raytracer.comp, *as compiled* by Mesa 23.2.1 for llvmpipe (the only
 * implementation of the reference that can run in this image; see oracle/README.md).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it; the
 * product (libtdtrt.so) never links or calls it.
 *
 * Pinned against: renders of the reference shader itself on llvmpipe (oracle/glref), stored
 * as fixtures under tests/golden/ by oracle/make_goldens.py.
 */
#ifndef PATHTRACE_ORACLE_H
#define PATHTRACE_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* `uniform Camera camera` — raytracer.comp:133-146; set by camera.rs:241-253. */
typedef struct {
  int32_t image_width, image_height;
  float horizontal[3], vertical[3], lower_left_corner[3], origin[3];
  int32_t samples_per_pixel, max_bounce;
} oracle_camera;

/* The seven SSBO payloads exactly as the reference host uploads them (main.rs:238-450,
 * octree.rs:44-100).  Sizes are in BYTES; reads past a buffer's end return 0 (GL robust
 * buffer access as implemented by llvmpipe). */
typedef struct {
  const void *cells;      size_t cells_bytes;       /* binding 0: Node{uint value; uint type}[]   */
  const void *materials;  size_t materials_bytes;   /* binding 1: {int type, attribute, albedo}[] */
  const void *albedos;    size_t albedos_bytes;     /* binding 2: {float x,y,z}[]                 */
  const void *metal;      size_t metal_bytes;       /* binding 3: {float fuzz}[]                  */
  const void *dielectric; size_t dielectric_bytes;  /* binding 4: {float ir}[]                    */
  const void *octree_floats; size_t octree_floats_bytes; /* binding 6: min.xyzw, scale, inv_scale, inv_cell_count */
  const void *octree_ints;   size_t octree_ints_bytes;   /* binding 7: max_depth, max_iter, cell_count */
} oracle_scene;

/* Event counts that define the ALGORITHMIC bytes of the path. */
typedef struct {
  uint64_t pixels;          /* pixels written                                            */
  uint64_t samples;         /* pixels * spp                                              */
  uint64_t octree_hit_calls;/* OctreeHit invocations (= rays traced)                     */
  uint64_t iterations;      /* OctreeHit loop iterations that reached treeLookup         */
  uint64_t node_loads;      /* tree levels visited = 8-byte Node loads (raytracer.comp:381) */
  uint64_t lambertian, metal, dielectric, unknown_material; /* scatter calls by type     */
} oracle_stats;

/* Renders what ComputeShader::dispatch_compute(dispatch_w, dispatch_h, 1) would write
 * (floor-div by 32, min 1 group; compute_shader.rs:28-38) into `image`
 * (image_width*image_height*4 floats, row 0 = bottom, pixels never written are left alone).
 * Rows [row_begin,row_end) only (clipped); pass 0,INT32_MAX for all.  nthreads<=1: scalar.
 * `stats` may be NULL. Returns 0. */
int oracle_render(const oracle_scene *scene, const oracle_camera *cam,
                  int dispatch_w, int dispatch_h, int row_begin, int row_end,
                  float *image, int nthreads, oracle_stats *stats);

/* Progressive form: adds spp_count samples starting at sample index spp_begin into
 * `accum` (W*H*4 floats; rgb running sums in sample order, a unused). */
int oracle_accumulate(const oracle_scene *scene, const oracle_camera *cam,
                      int dispatch_w, int dispatch_h, int row_begin, int row_end,
                      int spp_begin, int spp_count, float *accum, int nthreads, oracle_stats *stats);
/* resolve: image = clamp(sqrt(accum / total_spp), 0, 1), alpha = 1 for written pixels */
int oracle_resolve(const oracle_camera *cam, int dispatch_w, int dispatch_h, int row_begin, int row_end,
                   int total_spp, const float *accum, float *image);

/* ---- next row §8f-2: the voxel edit kernel, assets/shaders/octree_update.comp (as compiled) ----
 * Executes the dispatch ComputeShader::dispatch_compute(dispatch_w, dispatch_h, dispatch_d) of the
 * update program (work-group size 1: groups = max(dim, 1)), one invocation after the other in
 * x-fastest order (the reference's own outcome is racy when several invocations collide;
 * octree_update.comp:70-71).  `cells` is modified in place; `*counter` is the atomic counter
 * (binding 0).  delta = DeltaNode[] with the std430 layout the shader declares: pos at 0, type at
 * 12, value at 16, stride 32 (octree_update.comp:41-48).  Returns 0. */
int oracle_octree_update(void *cells, size_t cells_bytes, const void *delta, size_t delta_bytes,
                         const void *octree_floats, size_t octree_floats_bytes,
                         const void *octree_ints, size_t octree_ints_bytes, uint32_t *counter,
                         int dispatch_w, int dispatch_h, int dispatch_d);

/* llvmpipe's sin/cos/pow (gallivm polynomial forms) exposed for unit tests */
float oracle_sin(float a);
float oracle_cos(float a);
float oracle_pow(float x, float y);

#ifdef __cplusplus
}
#endif
#endif
