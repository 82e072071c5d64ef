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
  /* controller settings (only tdt_camera_init reads them; tdt_camera_build ignores them like build() does for the uniforms) */
  int32_t has_turn_rate;       float turn_rate;         /* default 0.025 camera.rs:167 */
  int32_t has_normal_speed;    float normal_speed;      /* default 1.0   camera.rs:168 */
  int32_t has_sprint_speed;    float sprint_speed;      /* default 2 * normal_speed camera.rs:169 */
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

/* ---- next row SURVEY §8f-3: the camera controller and its settings (camera.rs:8-16, 20-102) ----
 * The reference's vector / quaternion arithmetic is the cgmath crate's (Cargo.lock: cgmath 0.18.0; not vendored, no Rust
 * toolchain here): csrc/host_view.cpp restates its published formulas in f32.  PARITY UNPINNED (no reference fixture
 * covers these values). */
typedef struct {                 /* CameraSettings camera.rs:9-16 = assets/settings/camera.ron */
  int32_t samples_per_pixel, max_bounce;
  float turn_rate, normal_speed, sprint_speed;
} tdt_camera_settings;

typedef struct {                 /* Camera camera.rs:20-37, minus the GL texture */
  float horizontal[3], vertical[3];
  float viewport_width, viewport_height;
  float lower_left_corner[3], origin[3];
  float pitch[4], yaw[4];        /* quaternions in Quaternion::new(w, xi, yj, zk) order */
  int32_t image_width, image_height;
  tdt_camera_settings settings;
  float movement_speed;
} tdt_camera;

int tdt_camera_init(const tdt_camera_builder *b, tdt_camera *cam);                     /* CameraBuilder::build camera.rs:135-196 */
int tdt_camera_translate(tdt_camera *cam, const float by[3], double deltatime);        /* camera.rs:40-43; `by` e.g. Direction::into_vector3 utility/mod.rs:15-26 */
int tdt_camera_turn_pitch(tdt_camera *cam, float angle);                               /* camera.rs:46-53 */
int tdt_camera_turn_yaw(tdt_camera *cam, float angle);                                 /* camera.rs:56-62 */
void tdt_camera_set_speed_to_normal(tdt_camera *cam);                                  /* camera.rs:84-86 */
void tdt_camera_set_speed_to_sprint(tdt_camera *cam);                                  /* camera.rs:88-90 */
int tdt_camera_look_at_world_point(const tdt_camera *cam, float distance, float out[3]);   /* camera.rs:92-94 (main.rs:555) */
int tdt_camera_apply_settings(tdt_camera *cam, const tdt_camera_settings *s);          /* camera.rs:96-101 */
/* the uniforms propagate_changes / apply_settings / initial_uniforms have sent so far (camera.rs:78-81, 99-100, 241-253) */
int tdt_camera_get_uniforms(const tdt_camera *cam, tdt_camera_uniforms *out);
/* `ron::de::from_bytes::<CameraSettings>` (main.rs:171, 493) for the RON subset such a file uses: optional struct name,
 * `field: number` pairs in any order, comments, optional trailing comma; a missing field or a float where an i32 is
 * expected is an error (returns 1, message in tdt_host_last_error) */
int tdt_camera_settings_from_ron(const char *text, size_t n, tdt_camera_settings *out);

/* ---- next row SURVEY §8f-4: presentation ----
 * The reference shows the render texture with a full-window quad (assets/shaders/quad.vert, quad.frag:10; main.rs:113-153,
 * 582-600) and never writes a file.  tdt_present_rgba8 is what that pass leaves in an RGBA8 back buffer of the texture's
 * size (pinned on llvmpipe: tests/golden/present_*.npz): per channel clamp to [0,1] (NaN -> 0), x 255, round half to even.
 * Source rows are bottom-up (row 0 = bottom scan-line); top_down = 1 writes the top scan-line first (file order). */
int tdt_present_rgba8(const float *rgba, int w, int h, int top_down, uint8_t *dst);
/* 8-bit PNG (colour type 2, or 6 with_alpha) of a TOP-DOWN RGBA8 frame; *out is malloc'ed: release with tdt_host_free */
int tdt_png_encode(const uint8_t *rgba8, int w, int h, int with_alpha, uint8_t **out, size_t *len);
int tdt_png_write(const char *path, const uint8_t *rgba8, int w, int h, int with_alpha);
void tdt_host_free(void *p);

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

/* ---------------------------------------------------------------- scene ingest --- */
/* ply_point_loader::from_resources (src/utility/ply_point_loader.rs:102-319) restated over a byte
 * buffer: the MagicaVoxel ASCII point export ("float" x y z that are really integers, uchar r g b).
 * strict_crlf = 1 is the reference grammar to the byte ("ply\r\n", "format ascii 1.0\r\n", every
 * line CRLF: :121,:130,:158,:175,:207,:312-314); 0 also accepts LF-only files (as checked out on
 * Linux).  Quirks kept: a header word only has to match a PREFIX of its keyword (:138-152); the
 * albedo key is recomputed and inserted after EVERY property of a vertex (:300-307), so the palette
 * also holds the partial colours (0,0,0), (r,0,0), (r,g,0); the Cantor pairing runs in f64 and
 * saturates to u32 (:228-241); the vertex count of the header is not checked against the data.
 * One deviation: an unknown header keyword is an error here (the reference loops forever, :136-153).
 * PARITY UNPINNED: the Rust loader cannot be built or run in this image; tests pin the restatement
 * against hand-derived values for the reference's own 3x3x3 model. */
typedef struct tdt_ply tdt_ply;
int tdt_ply_parse(const void *data, size_t bytes, int strict_crlf, tdt_ply **out);
void tdt_ply_destroy(tdt_ply *p);
/* header.vertex, number of voxels read, min_point (:221), palette size */
int tdt_ply_info(const tdt_ply *p, int64_t *header_vertex, int64_t *n_voxels, int32_t min_point[3], int64_t *n_albedos);
/* 4 x i32 per voxel: x, y, z, albedo_key (as u32 bits) */
const int32_t *tdt_ply_voxels(const tdt_ply *p);
/* palette sorted by key (the reference's HashMap has no order): returns the number written */
int64_t tdt_ply_albedos(const tdt_ply *p, uint32_t *keys, uint8_t *rgb, int64_t capacity);
/* The step the reference never wrote (the loader's result is unused: main.rs:218-224): voxels ->
 * breadth-first indirect cells + one Lambertian material per distinct voxel colour (rgb / 255).
 * The grid edge is the next power of two >= the model's extent; z_up = 1 maps the file's z to the
 * octree's y (MagicaVoxel is z-up); the model is centred in x, stands on the floor and is pushed to
 * the far (z = 0) side of the octree so that the reference camera (main.rs:165-168) looks at it. */
int tdt_scene_from_ply(const tdt_ply *p, int max_iter, int z_up, tdt_scene **out);

#ifdef __cplusplus
}
#endif
#endif
