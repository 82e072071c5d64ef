// renderer.hpp — C++ mirror of the reference's `src/renderer` module over the C ABI (tdt_rt.h / tdt_host.h).
//
// The reference's host is Rust (no toolchain in this image), so this header plays the part of
// src/renderer/{mod,program,compute_shader,vbo,texture,vao,octree,camera}.rs for a C++ host: same type and
// method names, same argument meaning, same error behaviour — `Result<T, InitializeErr>` becomes "returns T or
// throws InitializeErr" (the reference `.unwrap()`s nearly every one of them: an uncaught throw is that panic).
// Header-only; link with -ltdtrt -ltdthost.  Every item cites the reference lines it stands for.
#pragma once

#include <array>
#include <cstdint>
#include <cstring>
#include <string>
#include <utility>
#include <vector>

#include "tdt_host.h"
#include "tdt_rt.h"

namespace renderer {

// renderer/mod.rs:20-24
enum class Material : uint32_t { Lambertian = 0, Metal, Dielectric };

// renderer/mod.rs:28-59 (InitializeErr + its Display)
struct InitializeErr {
  enum Kind { GL, VariableNotFound, TypedVariableNotFound, InvalidCStr } kind;
  unsigned code = 0;            // GL(code)
  std::string name, type_str;   // VariableNotFound(name) / TypedVariableNotFound(name, type)
  std::string detail;           // message of the C ABI for this failure

  InitializeErr var_into_typed(const std::string &t) const {   // mod.rs:36-41
    InitializeErr e = *this;
    if (e.kind == VariableNotFound) { e.kind = TypedVariableNotFound; e.type_str = t; }
    return e;
  }
  std::string to_string() const {                              // impl Display, mod.rs:44-59
    switch (kind) {
      case GL:
        if (code == 0x0500) return "gl error: invalid enum";
        if (code == 0x0501) return "gl error: invalid value";
        if (code == 0x0502) return "gl error: invalid operation";
        return "got gl error code: " + std::to_string(code);
      case VariableNotFound: return "failed to locate uniform " + name;
      case TypedVariableNotFound: return "failed to locate uniform " + name + " with type " + type_str;
      default: return detail;
    }
  }
};

using Vector3f = std::array<float, 3>;   // cgmath::Vector3<f32>
using Vector3i = std::array<int32_t, 3>;

// The GL context the render thread makes current (main.rs:105-112): one HIP device + stream.
class Context {
 public:
  explicit Context(int device = 0, void *stream = nullptr) {
    if (int rc = tdt_ctx_create(device, stream, &ctx_)) throw InitializeErr{InitializeErr::GL, (unsigned)rc, "", "", tdt_last_error(nullptr)};
  }
  ~Context() { tdt_ctx_destroy(ctx_); }
  Context(const Context &) = delete;
  Context &operator=(const Context &) = delete;
  tdt_ctx *raw() const { return ctx_; }
  // check_for_gl_error (mod.rs:62-68): GL's sticky error is the return code of every call here
  void check(int rc) const {
    if (rc != TDT_OK) throw InitializeErr{InitializeErr::GL, (unsigned)rc, "", "", tdt_last_error(ctx_)};
  }
  void finish() const { check(tdt_finish(ctx_)); }   // glFinish
 private:
  tdt_ctx *ctx_ = nullptr;
};

// renderer/vbo.rs:9-55.  `target` / `usage` are accepted for signature parity and ignored (a HIP buffer has neither).
class VertexBufferObject {
 public:
  template <class T>
  static VertexBufferObject new_(Context &ctx, const std::vector<T> &data, unsigned /*target*/ = 0, unsigned /*usage*/ = 0) {
    VertexBufferObject v;
    v.ctx_ = &ctx;
    ctx.check(tdt_buffer_create(ctx.raw(), data.data(), data.size() * sizeof(T), &v.buf_));
    return v;
  }
  tdt_buffer *id() const { return buf_; }   // vbo.rs:11 `id`
  void sub_data(size_t offset, size_t bytes, const void *data) const { ctx_->check(tdt_buffer_sub_data(buf_, offset, bytes, data)); }
  template <class T> std::vector<T> read(size_t count) const {
    std::vector<T> out(count);
    ctx_->check(tdt_buffer_read(buf_, 0, count * sizeof(T), out.data()));
    return out;
  }
 private:
  Context *ctx_ = nullptr;
  tdt_buffer *buf_ = nullptr;
};

// gl::BindBufferBase(target, slot, id) as main.rs:352-448 and octree.rs:67-144 call it
inline void bind_buffer_base(Context &ctx, int target, unsigned slot, const VertexBufferObject &vbo) {
  ctx.check(tdt_bind_buffer_base(ctx.raw(), target, slot, vbo.id()));
}

// renderer/vao.rs:32-74: vertex-attribute state has no effect on SSBO access ("vao might not be needed",
// main.rs:232); kept so that host code written against the reference compiles unchanged.
struct VertexAttributePointer { unsigned location, size, offset; };
class VertexArrayObject {
 public:
  template <class T> static VertexArrayObject new_(const std::vector<VertexAttributePointer> &, const tdt_buffer *, unsigned) { return {}; }
  template <class T> void append_vbo(const std::vector<VertexAttributePointer> &, const tdt_buffer *, unsigned) const {}
  void bind() const {}
  static void unbind() {}
};

// renderer/texture.rs:7-92
class Texture {
 public:
  static Texture new_2d(Context &ctx, int width, int height) {   // new_2d(TEXTURE0, 0, RGBA32F, RGBA, w, h) texture.rs:47-75
    Texture t;
    t.ctx_ = &ctx; t.w_ = width; t.h_ = height;
    ctx.check(tdt_image_create_rgba32f(ctx.raw(), width, height, &t.img_));
    t.bind();
    return t;
  }
  void bind() const { ctx_->check(tdt_bind_image(ctx_->raw(), 0, img_)); }   // BindImageTexture(unit 0)
  int width() const { return w_; }
  int height() const { return h_; }
  int depth() const { return 1; }
  std::vector<float> read() const {   // NEW: the reference never reads back (quad.frag samples the texture)
    std::vector<float> px((size_t)w_ * h_ * 4);
    ctx_->check(tdt_image_read(img_, px.data()));
    return px;
  }
  // NEW: the frame as the quad pass presents it (quad.frag:10), RGBA8, converted on the GPU
  void read_rgba8(bool top_down, uint8_t *dst) const { ctx_->check(tdt_image_read_rgba8(img_, top_down ? 1 : 0, dst)); }
 private:
  Context *ctx_ = nullptr;
  tdt_image *img_ = nullptr;
  int w_ = 0, h_ = 0;
};

// renderer/program.rs:9-174: the uniform-by-name setters (the program object itself is the kernel)
class Program {
 public:
  Program() = default;
  Program(Context &ctx, tdt_compute *c) : ctx_(&ctx), c_(c) {}
  void bind() const {}               // program.rs:21-25 glUseProgram: nothing to do
  static void unbind() {}
  tdt_compute *id() const { return c_; }
  void set_i32(const std::string &name, int32_t v) const { uniform(tdt_set_i32(c_, name.c_str(), v), name, "i32"); }                                  // :35-45
  void set_f32(const std::string &name, float v) const { uniform(tdt_set_f32(c_, name.c_str(), v), name, "f32"); }                                    // :61-71
  void set_vector3_f32(const std::string &name, const Vector3f &v) const { uniform(tdt_set_vec3f(c_, name.c_str(), v[0], v[1], v[2]), name, "vec3 f32"); }   // :73-83
  void set_vector3_i32(const std::string &name, const Vector3i &v) const { uniform(tdt_set_vec3i(c_, name.c_str(), v[0], v[1], v[2]), name, "vec3 i32"); }   // :48-58
 private:
  void uniform(int rc, const std::string &name, const char *type) const {   // register_uniform + var_into_typed, :144-165
    if (rc == TDT_ERR_VARIABLE_NOT_FOUND) throw InitializeErr{InitializeErr::VariableNotFound, 0, name, "", tdt_last_error(ctx_->raw())}.var_into_typed(type);
    ctx_->check(rc);
  }
  Context *ctx_ = nullptr;
  tdt_compute *c_ = nullptr;
};

// renderer/compute_shader.rs:10-38
class ComputeShader {
 public:
  Program program;
  // Shader::from_resources(res, "shaders/raytracer.comp" | "shaders/octree_update.comp") + Program::from_shaders +
  // ComputeShader::new (main.rs:156-160, 226-230): `kind` names which of the two compute programs
  static ComputeShader new_(Context &ctx, int kind = TDT_PROGRAM_RAYTRACER) {
    ComputeShader cs;
    tdt_compute *c = nullptr;
    ctx.check(tdt_compute_create(ctx.raw(), kind, &c));
    cs.program = Program(ctx, c);
    cs.ctx_ = &ctx;
    ctx.check(tdt_compute_group_size(c, cs.group_size_));     // COMPUTE_WORK_GROUP_SIZE, compute_shader.rs:18
    return cs;
  }
  void dispatch_compute(int width, int height, int depth) const {   // compute_shader.rs:28-38 (floor-div groups inside)
    ctx_->check(tdt_dispatch_compute(program.id(), width, height, depth));
  }
  const int *group_size() const { return group_size_; }
 private:
  Context *ctx_ = nullptr;
  int group_size_[3] = {0, 0, 0};
};

// renderer/octree.rs:10-183
class Octree {
 public:
  VertexArrayObject vao;
  static Octree new_(Vector3f min_point, float scale, int max_depth, int cell_count, int active_cell_count,
                     int max_traversal_iter, VertexArrayObject vao = {}) {                      // octree.rs:26-38
    Octree o;
    o.min_point_ = min_point; o.scale_ = scale; o.max_depth_ = max_depth; o.cell_count_ = cell_count;
    o.active_cell_count_ = active_cell_count; o.max_traversal_iter_ = max_traversal_iter; o.vao = vao;
    o.block_distance_ = scale / (float)(1u << (max_depth < 31 ? max_depth : 31));               // scale / 2^max_depth
    return o;
  }
  void init_global_buffers(Context &ctx) {                                                       // octree.rs:40-151
    floats_ = VertexBufferObject::new_<float>(ctx, {min_point_[0], min_point_[1], min_point_[2], 0.0f, scale_, 1.0f / scale_,
                                                    1.0f / (float)cell_count_});                // :44-50
    bind_buffer_base(ctx, TDT_SHADER_STORAGE_BUFFER, 6, floats_);
    ints_ = VertexBufferObject::new_<int32_t>(ctx, {max_depth_, max_traversal_iter_, cell_count_});   // :76-81
    bind_buffer_base(ctx, TDT_SHADER_STORAGE_BUFFER, 7, ints_);
    counter_ = VertexBufferObject::new_<int32_t>(ctx, {active_cell_count_});                    // :105-110
    bind_buffer_base(ctx, TDT_ATOMIC_COUNTER_BUFFER, 0, counter_);
    delta_ = VertexBufferObject::new_<float>(ctx, std::vector<float>(1000, 0.0f));              // :124-128
    bind_buffer_base(ctx, TDT_SHADER_STORAGE_BUFFER, 5, delta_);
  }
  float block_distance() const { return block_distance_; }
  float scale() const { return scale_; }
  Vector3f min_point() const { return min_point_; }
  bool point_inside(const Vector3f &p) const {                                                   // :165-168
    return p[0] >= min_point_[0] && p[1] >= min_point_[1] && p[2] >= min_point_[2] && p[0] <= min_point_[0] + scale_ &&
           p[1] <= min_point_[1] + scale_ && p[2] <= min_point_[2] + scale_;
  }
  void update_vbo(const std::vector<float> &delta, size_t len, const ComputeShader &update_compute) const {   // :170-183
    const float LOCAL_GROUP_SIZE_X = 32.0f * 32.0f;
    delta_.sub_data(0, len * sizeof(float), delta.data());       // BufferSubData of the generic SSBO target = the delta buffer (:144,:174)
    const float x_schedule = (float)len * 0.2f;
    const int dispatch_count = (int)((float)len * 0.2f);
    const float q = x_schedule / LOCAL_GROUP_SIZE_X;
    if (q - (float)(long long)q != 0.0f) update_compute.dispatch_compute(0, dispatch_count, 0);
    else update_compute.dispatch_compute(dispatch_count, 1, 1);
  }
  const VertexBufferObject &counter() const { return counter_; }
 private:
  Vector3f min_point_{};
  float scale_ = 1, block_distance_ = 0;
  int max_depth_ = 0, cell_count_ = 0, active_cell_count_ = 0, max_traversal_iter_ = 0;
  VertexBufferObject floats_, ints_, counter_, delta_;
};

// renderer/camera.rs:8-16
using CameraSettings = tdt_camera_settings;
inline CameraSettings camera_settings_from_ron(const std::string &text) {                         // main.rs:171, 493
  CameraSettings s;
  if (tdt_camera_settings_from_ron(text.data(), text.size(), &s)) throw InitializeErr{InitializeErr::GL, 0x0501, "", "", tdt_host_last_error()};
  return s;
}

// utility/mod.rs:6-26
enum class Direction { Front, Back, Rigth, Left, Up, Down };
inline Vector3f into_vector3(Direction d) {
  switch (d) {
    case Direction::Front: return {0.0f, 0.0f, -1.0f};
    case Direction::Back: return {0.0f, 0.0f, 1.0f};
    case Direction::Rigth: return {1.0f, 0.0f, 0.0f};
    case Direction::Left: return {-1.0f, 0.0f, 0.0f};
    case Direction::Up: return {0.0f, 1.0f, 0.0f};
    default: return {0.0f, -1.0f, 0.0f};
  }
}

// renderer/camera.rs:20-102: the controller state lives in libtdthost's tdt_camera (cgmath restated there)
class Camera {
 public:
  tdt_camera state{};
  Texture render_texture;
  const float *horizontal() const { return state.horizontal; }
  const float *vertical() const { return state.vertical; }
  const float *lower_left_corner() const { return state.lower_left_corner; }
  const float *origin() const { return state.origin; }
  int32_t image_width() const { return state.image_width; }
  int32_t image_height() const { return state.image_height; }
  const CameraSettings &settings() const { return state.settings; }
  void translate(const Program &program, const Vector3f &by, double deltatime) {                   // :40-43
    tdt_camera_translate(&state, by.data(), deltatime); propagate_changes(program);
  }
  void turn_pitch(const Program &program, float angle) { tdt_camera_turn_pitch(&state, angle); propagate_changes(program); }   // :46-53
  void turn_yaw(const Program &program, float angle) { tdt_camera_turn_yaw(&state, angle); propagate_changes(program); }       // :56-62
  void set_speed_to_normal() { tdt_camera_set_speed_to_normal(&state); }                             // :84-86
  void set_speed_to_sprint() { tdt_camera_set_speed_to_sprint(&state); }                             // :88-90
  Vector3f look_at_world_point(float distance) const {                                              // :92-94
    Vector3f p{}; tdt_camera_look_at_world_point(&state, distance, p.data()); return p;
  }
  void apply_settings(const Program &program, const CameraSettings &s) {                            // :96-101
    tdt_camera_apply_settings(&state, &s);
    program.set_i32("camera.samples_per_pixel", state.settings.samples_per_pixel);
    program.set_i32("camera.max_bounce", state.settings.max_bounce);
  }
 private:
  void propagate_changes(const Program &program) const {                                            // the uploads of :78-81
    program.set_vector3_f32("camera.horizontal", {state.horizontal[0], state.horizontal[1], state.horizontal[2]});
    program.set_vector3_f32("camera.vertical", {state.vertical[0], state.vertical[1], state.vertical[2]});
    program.set_vector3_f32("camera.lower_left_corner", {state.lower_left_corner[0], state.lower_left_corner[1], state.lower_left_corner[2]});
    program.set_vector3_f32("camera.origin", {state.origin[0], state.origin[1], state.origin[2]});
  }
  friend class CameraBuilder;
};

// renderer/camera.rs:104-237
class CameraBuilder {
 public:
  static CameraBuilder new_(float vertical_fov, int32_t image_width) {                           // :119-133
    CameraBuilder b;
    std::memset(&b.b_, 0, sizeof b.b_);
    b.b_.vertical_fov = vertical_fov; b.b_.image_width = image_width;
    return b;
  }
  CameraBuilder &with_aspect_ratio(float a) { b_.has_aspect_ratio = 1; b_.aspect_ratio = a; return *this; }
  CameraBuilder &with_viewport_height(float h) { b_.has_viewport_height = 1; b_.viewport_height = h; return *this; }
  CameraBuilder &with_origin(const Vector3f &o) { b_.has_origin = 1; b_.origin[0] = o[0]; b_.origin[1] = o[1]; b_.origin[2] = o[2]; return *this; }
  CameraBuilder &with_sample_per_pixel(int32_t n) { b_.has_samples_per_pixel = 1; b_.samples_per_pixel = n; return *this; }
  CameraBuilder &with_max_bounce(int32_t n) { b_.has_max_bounce = 1; b_.max_bounce = n; return *this; }
  CameraBuilder &with_turn_rate(float v) { b_.has_turn_rate = 1; b_.turn_rate = v; return *this; }
  CameraBuilder &with_normal_speed(float v) { b_.has_normal_speed = 1; b_.normal_speed = v; return *this; }
  CameraBuilder &with_sprint_speed(float v) { b_.has_sprint_speed = 1; b_.sprint_speed = v; return *this; }
  Camera build(Context &ctx, const Program &program) const {                                     // :135-196
    Camera c;
    if (tdt_camera_init(&b_, &c.state)) throw InitializeErr{InitializeErr::GL, 0x0501, "", "", tdt_host_last_error()};
    c.render_texture = Texture::new_2d(ctx, c.state.image_width, c.state.image_height);          // :158-165
    // initial_uniforms, camera.rs:241-253
    program.set_i32("camera.image_width", c.state.image_width);
    program.set_i32("camera.image_height", c.state.image_height);
    c.propagate_changes(program);
    program.set_i32("camera.samples_per_pixel", c.state.settings.samples_per_pixel);
    program.set_i32("camera.max_bounce", c.state.settings.max_bounce);
    return c;
  }
 private:
  tdt_camera_builder b_;
};

// Presentation: the frame the reference's quad pass would show (main.rs:582-600, quad.frag:10), as a PNG file
inline void present_png(Texture &texture, const std::string &path, bool with_alpha = false) {
  std::vector<uint8_t> frame(static_cast<size_t>(texture.width()) * texture.height() * 4);
  texture.read_rgba8(true, frame.data());
  if (tdt_png_write(path.c_str(), frame.data(), texture.width(), texture.height(), with_alpha ? 1 : 0))
    throw InitializeErr{InitializeErr::GL, 0x0501, "", "", tdt_host_last_error()};
}

}  // namespace renderer
