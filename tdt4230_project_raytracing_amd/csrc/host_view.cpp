// host_view.cpp — the callers either side of the trace that SURVEY.md §8f lists as rows 3 and 4 (pure host code):
//
//   §8f-3  camera controller + settings: Camera::{translate, turn_pitch, turn_yaw, orientation, propagate_changes,
//          look_at_world_point, apply_settings, set_speed_to_*} (src/renderer/camera.rs:39-102), the builder's
//          controller defaults (camera.rs:167-169) and the RON settings file (camera.rs:8-16, assets/settings/
//          camera.ron, read at main.rs:170-176 and re-read on change main.rs:492-494).
//          The vector / quaternion arithmetic of the reference lives in the cgmath crate (Cargo.lock: cgmath 0.18.0),
//          which is not vendored and cannot be built here (no Rust toolchain): its published formulas are restated
//          below (Quaternion * Quaternion, Quaternion * Vector3, InnerSpace::normalize = v * (1 / |v|), Vector3::cross)
//          in f32 and in the reference's call order.  PARITY UNPINNED: no fixture of the reference covers these
//          values; tests check the algebraic properties and the identity-orientation case against the builder.
//   §8f-4  presentation: what the reference's quad pass (assets/shaders/quad.vert, quad.frag:10; main.rs:113-153,
//          582-600) leaves in the window's RGBA8 back buffer, as a byte image, plus a PNG writer (the reference only
//          ever shows the frame in a window).  Pinned on llvmpipe through oracle/glref (tests/golden/present_*.npz).
#include <zlib.h>

#include <cctype>
#include <climits>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "tdt_host.h"

extern "C" void tdt_host_set_error(const char *msg);   // host_scene.cpp

namespace {

int fail(const std::string &m) { tdt_host_set_error(m.c_str()); return 1; }

// ---- cgmath 0.18.0 restated (f32; Quaternion::new(w, xi, yj, zk) order in the arrays) -------------------------------
struct V3 { float x, y, z; };
struct Q { float s, x, y, z; };
inline V3 operator*(V3 a, float k) { return {a.x * k, a.y * k, a.z * k}; }
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 neg(V3 a) { return {-a.x, -a.y, -a.z}; }
inline float dot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }          // mul_element_wise().sum()
inline V3 cross(V3 a, V3 b) { return {(a.y * b.z) - (a.z * b.y), (a.z * b.x) - (a.x * b.z), (a.x * b.y) - (a.y * b.x)}; }
inline V3 normalize(V3 a) { return a * (1.0f / std::sqrt(dot(a, a))); }             // normalize_to(1): v * (1 / magnitude)
inline Q qmul(Q l, Q r) {
  return {l.s * r.s - l.x * r.x - l.y * r.y - l.z * r.z,
          l.s * r.x + l.x * r.s + l.y * r.z - l.z * r.y,
          l.s * r.y + l.y * r.s + l.z * r.x - l.x * r.z,
          l.s * r.z + l.z * r.s + l.x * r.y - l.y * r.x};
}
inline Q qnormalize(Q q) {
  const float d = q.s * q.s + ((q.x * q.x + q.y * q.y) + q.z * q.z);                // s*s + v.dot(v)
  const float k = 1.0f / std::sqrt(d);
  return {q.s * k, q.x * k, q.y * k, q.z * k};
}
inline V3 rotate(Q q, V3 v) {                                                      // Quaternion * Vector3
  const V3 qv = {q.x, q.y, q.z};
  const V3 tmp = cross(qv, v) + (v * q.s);
  return (cross(qv, tmp) * 2.0f) + v;
}

inline Q load_q(const float a[4]) { return {a[0], a[1], a[2], a[3]}; }
inline void store_q(float a[4], Q q) { a[0] = q.s; a[1] = q.x; a[2] = q.y; a[3] = q.z; }
inline V3 load_v(const float a[3]) { return {a[0], a[1], a[2]}; }
inline void store_v(float a[3], V3 v) { a[0] = v.x; a[1] = v.y; a[2] = v.z; }

Q orientation(const tdt_camera *c) { return qnormalize(qmul(load_q(c->yaw), load_q(c->pitch))); }   // camera.rs:64-66

void propagate_changes(tdt_camera *c) {                                            // camera.rs:68-82
  const V3 forward = normalize(rotate(orientation(c), V3{0.f, 0.f, 1.f}));
  const V3 right = normalize(cross(V3{0.f, 1.f, 0.f}, forward));
  const V3 up = normalize(cross(forward, right));
  const V3 horizontal = right * c->viewport_width, vertical = up * c->viewport_height;
  const V3 llc = ((load_v(c->origin) - horizontal * 0.5f) - vertical * 0.5f) - forward;
  store_v(c->horizontal, horizontal); store_v(c->vertical, vertical); store_v(c->lower_left_corner, llc);
}

// ---- RON subset for `CameraSettings( name: number, ... )` (ron 0.6.4 grammar: optional struct name, whitespace,
// `//` and `/* */` comments, optional trailing comma, fields in any order) -------------------------------------------
struct Ron {
  const char *p, *e;
  void ws() {
    for (;;) {
      while (p < e && (*p == ' ' || *p == '\t' || *p == '\r' || *p == '\n')) p++;
      if (p + 1 < e && p[0] == '/' && p[1] == '/') { while (p < e && *p != '\n') p++; continue; }
      if (p + 1 < e && p[0] == '/' && p[1] == '*') {
        p += 2; int depth = 1;                                                     // RON block comments nest
        while (p + 1 < e && depth) { if (p[0] == '/' && p[1] == '*') { depth++; p += 2; } else if (p[0] == '*' && p[1] == '/') { depth--; p += 2; } else p++; }
        continue;
      }
      return;
    }
  }
  bool ident(std::string &out) {
    ws(); const char *b = p;
    while (p < e && (std::isalnum(static_cast<unsigned char>(*p)) || *p == '_')) p++;
    out.assign(b, p); return p > b;
  }
  bool lit(char c) { ws(); if (p < e && *p == c) { p++; return true; } return false; }
  bool number(std::string &out, bool &is_int) {
    ws(); const char *b = p; is_int = true;
    if (p < e && (*p == '+' || *p == '-')) p++;
    while (p < e && (std::isdigit(static_cast<unsigned char>(*p)) || *p == '_' || *p == '.' || *p == 'e' || *p == 'E' ||
                     ((*p == '+' || *p == '-') && (p[-1] == 'e' || p[-1] == 'E')))) {
      if (*p == '.' || *p == 'e' || *p == 'E') is_int = false;
      p++;
    }
    out.clear();
    for (const char *q = b; q < p; q++) if (*q != '_') out.push_back(*q);
    return !out.empty() && out != "+" && out != "-";
  }
};

inline uint8_t unorm8(float c) {
  // GL float -> UNORM8 as llvmpipe does it (and as the fixtures pin it): clamp to [0,1] with NaN -> 0, times 255,
  // round half to even
  const float v = !(c > 0.0f) ? 0.0f : (c > 1.0f ? 1.0f : c);
  return static_cast<uint8_t>(std::nearbyintf(v * 255.0f));
}

void put32(std::vector<uint8_t> &o, uint32_t v) { o.push_back(v >> 24); o.push_back(v >> 16); o.push_back(v >> 8); o.push_back(v); }
void chunk(std::vector<uint8_t> &o, const char type[4], const uint8_t *data, size_t n) {
  put32(o, static_cast<uint32_t>(n));
  const size_t at = o.size();
  o.insert(o.end(), type, type + 4);
  if (n) o.insert(o.end(), data, data + n);
  put32(o, static_cast<uint32_t>(crc32(0L, o.data() + at, static_cast<uInt>(n + 4))));
}

}  // namespace

extern "C" {

// ------------------------------------------------------------------------------------------- §8f-3 camera ----
int tdt_camera_init(const tdt_camera_builder *b, tdt_camera *cam) {                  // CameraBuilder::build camera.rs:135-196
  if (!b || !cam) return fail("null argument");
  tdt_camera_uniforms u;
  if (tdt_camera_build(b, &u) != 0) return 1;
  std::memset(cam, 0, sizeof *cam);
  // viewport_width / viewport_height as build() keeps them (camera.rs:138-141): horizontal = unit_x * width etc.
  cam->viewport_width = u.horizontal[0]; cam->viewport_height = u.vertical[1];
  std::memcpy(cam->horizontal, u.horizontal, sizeof u.horizontal); std::memcpy(cam->vertical, u.vertical, sizeof u.vertical);
  std::memcpy(cam->lower_left_corner, u.lower_left_corner, sizeof u.lower_left_corner); std::memcpy(cam->origin, u.origin, sizeof u.origin);
  cam->pitch[0] = 1.0f; cam->yaw[0] = 1.0f;                                          // Quaternion::new(1,0,0,0) camera.rs:178-179
  cam->image_width = u.image_width; cam->image_height = u.image_height;
  cam->settings.samples_per_pixel = u.samples_per_pixel; cam->settings.max_bounce = u.max_bounce;
  cam->settings.turn_rate = b->has_turn_rate ? b->turn_rate : 0.025f;                // camera.rs:167
  cam->settings.normal_speed = b->has_normal_speed ? b->normal_speed : 1.0f;          // camera.rs:168
  cam->settings.sprint_speed = b->has_sprint_speed ? b->sprint_speed : cam->settings.normal_speed * 2.0f;   // camera.rs:169
  cam->movement_speed = cam->settings.normal_speed;                                  // camera.rs:190
  return 0;
}

int tdt_camera_translate(tdt_camera *cam, const float by[3], double deltatime) {      // camera.rs:40-43
  if (!cam || !by) return fail("null argument");
  const V3 step = (load_v(by) * static_cast<float>(deltatime)) * cam->movement_speed;
  store_v(cam->origin, load_v(cam->origin) + rotate(orientation(cam), step));
  propagate_changes(cam);
  return 0;
}

int tdt_camera_turn_pitch(tdt_camera *cam, float angle) {                             // camera.rs:46-53
  if (!cam) return fail("null argument");
  const float h_angle = angle * cam->settings.turn_rate;
  store_q(cam->pitch, qmul(load_q(cam->pitch), qnormalize(Q{std::cos(h_angle), std::sin(h_angle), 0.0f, 0.0f})));
  propagate_changes(cam);
  return 0;
}

int tdt_camera_turn_yaw(tdt_camera *cam, float angle) {                               // camera.rs:56-62
  if (!cam) return fail("null argument");
  const float h_angle = angle * cam->settings.turn_rate;
  store_q(cam->yaw, qmul(load_q(cam->yaw), qnormalize(Q{std::cos(h_angle), 0.0f, std::sin(h_angle), 0.0f})));
  propagate_changes(cam);
  return 0;
}

void tdt_camera_set_speed_to_normal(tdt_camera *cam) { if (cam) cam->movement_speed = cam->settings.normal_speed; }   // camera.rs:84-86
void tdt_camera_set_speed_to_sprint(tdt_camera *cam) { if (cam) cam->movement_speed = cam->settings.sprint_speed; }   // camera.rs:88-90

int tdt_camera_look_at_world_point(const tdt_camera *cam, float distance, float out[3]) {   // camera.rs:92-94
  if (!cam || !out) return fail("null argument");
  store_v(out, rotate(orientation(cam), neg(V3{0.f, 0.f, 1.f}) * distance) + load_v(cam->origin));
  return 0;
}

int tdt_camera_apply_settings(tdt_camera *cam, const tdt_camera_settings *s) {        // camera.rs:96-101
  if (!cam || !s) return fail("null argument");
  cam->settings = *s;      // as in the reference, movement_speed keeps its old value until the next set_speed_to_*
  return 0;
}

int tdt_camera_get_uniforms(const tdt_camera *cam, tdt_camera_uniforms *out) {        // camera.rs:78-81, 241-253
  if (!cam || !out) return fail("null argument");
  out->image_width = cam->image_width; out->image_height = cam->image_height;
  std::memcpy(out->horizontal, cam->horizontal, sizeof out->horizontal); std::memcpy(out->vertical, cam->vertical, sizeof out->vertical);
  std::memcpy(out->lower_left_corner, cam->lower_left_corner, sizeof out->lower_left_corner); std::memcpy(out->origin, cam->origin, sizeof out->origin);
  out->samples_per_pixel = cam->settings.samples_per_pixel; out->max_bounce = cam->settings.max_bounce;
  return 0;
}

int tdt_camera_settings_from_ron(const char *text, size_t n, tdt_camera_settings *out) {   // main.rs:171, 493
  if (!text || !out) return fail("null argument");
  Ron r{text, text + n};
  std::string id;
  const char *save = r.p;
  if (r.ident(id)) { if (id != "CameraSettings") return fail("RON: expected `CameraSettings(`, found `" + id + "`"); } else r.p = save;
  if (!r.lit('(')) return fail("RON: expected `(`");
  bool seen[5] = {false, false, false, false, false};
  static const char *names[5] = {"samples_per_pixel", "max_bounce", "turn_rate", "normal_speed", "sprint_speed"};
  tdt_camera_settings s;
  std::memset(&s, 0, sizeof s);
  for (;;) {
    if (r.lit(')')) break;
    if (!r.ident(id)) return fail("RON: expected a field name");
    if (!r.lit(':')) return fail("RON: expected `:` after `" + id + "`");
    std::string num; bool is_int = false;
    if (!r.number(num, is_int)) return fail("RON: expected a number for `" + id + "`");
    int k = -1;
    for (int i = 0; i < 5; i++) if (id == names[i]) k = i;
    if (k >= 0) {
      if (seen[k]) return fail("RON: duplicate field `" + id + "`");
      seen[k] = true;
      if (k < 2) {
        if (!is_int) return fail("RON: `" + id + "` must be an integer");
        const long long v = std::strtoll(num.c_str(), nullptr, 10);
        if (v < INT32_MIN || v > INT32_MAX) return fail("RON: `" + id + "` out of range for i32");
        (k == 0 ? s.samples_per_pixel : s.max_bounce) = static_cast<int32_t>(v);
      } else {
        const float v = std::strtof(num.c_str(), nullptr);
        (k == 2 ? s.turn_rate : (k == 3 ? s.normal_speed : s.sprint_speed)) = v;
      }
    }                                                            // unknown fields are ignored (serde's default)
    if (r.lit(',')) continue;
    if (r.lit(')')) break;
    return fail("RON: expected `,` or `)`");
  }
  r.ws();
  if (r.p != r.e) return fail("RON: trailing characters");
  for (int i = 0; i < 5; i++) if (!seen[i]) return fail(std::string("RON: missing field `") + names[i] + "`");
  *out = s;
  return 0;
}

// -------------------------------------------------------------------------------------- §8f-4 presentation ----
int tdt_present_rgba8(const float *rgba, int w, int h, int top_down, uint8_t *dst) {
  if (!rgba || !dst || w <= 0 || h <= 0) return fail("bad argument");
  for (int y = 0; y < h; y++) {
    const float *src = rgba + static_cast<size_t>(y) * w * 4;
    uint8_t *d = dst + static_cast<size_t>(top_down ? h - 1 - y : y) * w * 4;
    for (int i = 0; i < w * 4; i++) d[i] = unorm8(src[i]);
  }
  return 0;
}

int tdt_png_encode(const uint8_t *rgba8, int w, int h, int with_alpha, uint8_t **out, size_t *len) {
  if (!rgba8 || !out || !len || w <= 0 || h <= 0) return fail("bad argument");
  const int ch = with_alpha ? 4 : 3;
  std::vector<uint8_t> raw(static_cast<size_t>(h) * (1 + static_cast<size_t>(w) * ch));
  size_t o = 0;
  for (int y = 0; y < h; y++) {
    raw[o++] = 0;                                                // filter type 0 (None)
    const uint8_t *s = rgba8 + static_cast<size_t>(y) * w * 4;
    for (int x = 0; x < w; x++, s += 4) { raw[o++] = s[0]; raw[o++] = s[1]; raw[o++] = s[2]; if (with_alpha) raw[o++] = s[3]; }
  }
  uLongf zn = compressBound(static_cast<uLong>(raw.size()));
  std::vector<uint8_t> z(zn);
  if (compress2(z.data(), &zn, raw.data(), static_cast<uLong>(raw.size()), 6) != Z_OK) return fail("zlib compress2 failed");
  std::vector<uint8_t> png = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
  uint8_t ihdr[13] = {0, 0, 0, 0, 0, 0, 0, 0, 8, static_cast<uint8_t>(with_alpha ? 6 : 2), 0, 0, 0};
  ihdr[0] = static_cast<uint8_t>(w >> 24); ihdr[1] = static_cast<uint8_t>(w >> 16); ihdr[2] = static_cast<uint8_t>(w >> 8); ihdr[3] = static_cast<uint8_t>(w);
  ihdr[4] = static_cast<uint8_t>(h >> 24); ihdr[5] = static_cast<uint8_t>(h >> 16); ihdr[6] = static_cast<uint8_t>(h >> 8); ihdr[7] = static_cast<uint8_t>(h);
  chunk(png, "IHDR", ihdr, sizeof ihdr);
  chunk(png, "IDAT", z.data(), zn);
  chunk(png, "IEND", nullptr, 0);
  uint8_t *buf = static_cast<uint8_t *>(std::malloc(png.size()));
  if (!buf) return fail("out of memory");
  std::memcpy(buf, png.data(), png.size());
  *out = buf; *len = png.size();
  return 0;
}

void tdt_host_free(void *p) { std::free(p); }

int tdt_png_write(const char *path, const uint8_t *rgba8, int w, int h, int with_alpha) {
  if (!path) return fail("null path");
  uint8_t *buf = nullptr; size_t n = 0;
  if (tdt_png_encode(rgba8, w, h, with_alpha, &buf, &n) != 0) return 1;
  FILE *f = std::fopen(path, "wb");
  if (!f) { std::free(buf); return fail(std::string("cannot open ") + path); }
  const bool ok = std::fwrite(buf, 1, n, f) == n;
  std::fclose(f); std::free(buf);
  return ok ? 0 : fail(std::string("short write to ") + path);
}

}  // extern "C"
