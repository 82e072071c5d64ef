/*
 * tdt_host.h — C ABI of libtdthost.so: the host-side inputs of the trace, restated from the
 * reference's Rust host code (which cannot be compiled here: no Rust toolchain).
 *
 *   - camera uniforms  : CameraBuilder::build / initial_uniforms   (src/renderer/camera.rs:135-196, 241-253)
 *   - octree payloads  : Octree::init_global_buffers                (src/renderer/octree.rs:40-100)
 *   - the demo scene   : the literal of src/main.rs:235-463 (as data)
 *   - synthetic scenes : deterministic generators for BASELINE.json's configs (the reference
 *                        has no octree builder and no other scene)
 *
 * Pure host code (no HIP).  A scene is nothing but the seven SSBO payloads, byte-for-byte in
 * the layout raytracer.comp reads, so the same blobs feed the reference shader, the oracle
 * and libtdtrt.so.
 */
#ifndef TDT_HOST_H
#define TDT_HOST_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* ---------------------------------------------------------------- camera ---------------- */
/* CameraBuilder (camera.rs:104-117); a has_* of 0 means "None" and build() applies the
 * reference's default. */
typedef struct {
  float vertical_fov;            /* CameraBuilder::new(vertical_fov, image_width) camera.rs:120 */
  int32_t image_width;
  int32_t has_aspect_ratio;    float aspect_ratio;      /* default 16/9  camera.rs:136 */
  int32_t has_viewport_height; float viewport_height;   /* default 2.0   camera.rs:140 */
  int32_t has_origin;          float origin[3];         /* default 0     camera.rs:143 */
  int32_t has_samples_per_pixel; int32_t samples_per_pixel; /* default 10 camera.rs:156 */
  int32_t has_max_bounce;      int32_t max_bounce;      /* default 3     camera.rs:157 */
} tdt_camera_builder;

/* what initial_uniforms() sends (camera.rs:241-253) = `uniform Camera camera` raytracer.comp:133-146 */
typedef struct {
  int32_t image_width, image_height;
  float horizontal[3], vertical[3], lower_left_corner[3], origin[3];
  int32_t samples_per_pixel, max_bounce;
} tdt_camera_uniforms;

int tdt_camera_build(const tdt_camera_builder *b, tdt_camera_uniforms *out);
/* the pose main.rs:165-168 builds: fov 90, origin (0,-0.1,-0.3), viewport_height 2.0,
 * aspect = width/height (f32 division) */
int tdt_camera_reference_pose(int width, int height, int spp, int max_bounce, tdt_camera_uniforms *out);

/* ---------------------------------------------------------------- scenes ---------------- */
typedef struct tdt_scene tdt_scene;

enum { TDT_SCENE_HASH_GRID = 0, TDT_SCENE_TERRAIN = 1, TDT_SCENE_SHELLS = 2 };

typedef struct {
  int32_t kind;          /* TDT_SCENE_*                                                     */
  int32_t max_depth;     /* log2 of the voxel grid edge (OctreeInts.max_depth)              */
  int32_t cell_count;    /* OctreeInts.cell_count: the divisor the shader uses; power of two */
  int32_t max_iter;      /* OctreeInts.max_iter                                             */
  uint64_t seed;
} tdt_scene_params;

/* main.rs:235-463: 19 cells (+ zero padding to 100144 u32), 13 materials, 7 albedos, 4 fuzz,
 * 1 ior; Octree::new(min (-.5,-.5,-1), scale 1, max_depth 10, cell_count 100000, max_iter 100) */
int tdt_scene_demo(tdt_scene **out);
int tdt_scene_generate(const tdt_scene_params *p, tdt_scene **out);
/* BASELINE.json configs 1..5 (config 4 uses the config-3 scene); see DESIGN.md */
int tdt_scene_config(int config, tdt_scene **out);
/* a scene from caller-provided payloads (copied) */
int tdt_scene_from_blobs(const void *const blobs[8], const size_t bytes[8], tdt_scene **out);
void tdt_scene_destroy(tdt_scene *s);
/* payload of SSBO binding `slot` (0,1,2,3,4,6,7); NULL/0 for other slots */
const void *tdt_scene_blob(const tdt_scene *s, int slot, size_t *bytes);
/* counts: [0]=cells in use, [1]=parent nodes, [2]=leaf nodes, [3]=empty nodes,
 * [4]=materials, [5]=occupied finest-level voxels (0 for the demo scene) */
int tdt_scene_counts(const tdt_scene *s, int64_t out[6]);
const char *tdt_host_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
