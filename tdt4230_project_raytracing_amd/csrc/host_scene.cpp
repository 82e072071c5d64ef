// libtdthost.so — host-side inputs of the trace (include/tdt_host.h).
//
// Restates, in C++, what the reference's Rust host computes before the first dispatch:
// camera uniforms (src/renderer/camera.rs:135-196), octree uniform payloads
// (src/renderer/octree.rs:44-50,76-81), the demo scene (src/main.rs:235-463, as data), and adds
// the deterministic synthetic-scene generators the benchmark configs need (the reference has
// no octree builder).  Integer-only hashing so every platform generates identical bytes.
#include "tdt_host.h"

#include <algorithm>
#include <array>
#include <cmath>
#include <cstring>
#include <deque>
#include <map>
#include <new>
#include <string>
#include <vector>

namespace {

thread_local std::string g_err;

constexpr uint32_t EMPTY = 0, PARENT = 1, LEAF = 2;          // octree.rs:6-8
constexpr uint32_t LAMBERTIAN = 0, METAL = 1, DIELECTRIC = 2; // renderer/mod.rs:20-24

}  // namespace

struct tdt_scene {
  std::vector<uint32_t> cells;       // binding 0
  std::vector<uint32_t> materials;   // binding 1
  std::vector<float> albedos;        // binding 2
  std::vector<float> metal;          // binding 3
  std::vector<float> dielectric;     // binding 4
  std::vector<float> octree_floats;  // binding 6
  std::vector<int32_t> octree_ints;  // binding 7
  int64_t counts[6] = {0, 0, 0, 0, 0, 0};
};

namespace {

// Octree::init_global_buffers, octree.rs:44-50 and 76-81
void set_octree_uniforms(tdt_scene &s, float mx, float my, float mz, float scale, int max_depth, int max_iter,
                         int cell_count) {
  s.octree_floats = {mx, my, mz, 0.0f, scale, 1.0f / scale, 1.0f / static_cast<float>(cell_count)};
  s.octree_ints = {max_depth, max_iter, cell_count};
}

void count_nodes(tdt_scene &s, size_t used_cells) {
  s.counts[0] = static_cast<int64_t>(used_cells);
  s.counts[1] = s.counts[2] = s.counts[3] = 0;
  for (size_t n = 0; n < used_cells * 8; n++) {
    uint32_t t = s.cells[n * 2 + 1];
    s.counts[t == PARENT ? 1 : (t == LEAF ? 2 : 3)]++;
  }
  s.counts[4] = static_cast<int64_t>(s.materials.size() / 3);
}

// ------------------------------------------------------------------ integer hashing ------
inline uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
inline uint64_t hash3(uint64_t seed, int64_t x, int64_t y, int64_t z) {
  return splitmix64(seed ^ splitmix64(static_cast<uint64_t>(x) * 0x9E3779B97F4A7C15ull ^
                                      splitmix64(static_cast<uint64_t>(y) * 0xC2B2AE3D27D4EB4Full ^
                                                 splitmix64(static_cast<uint64_t>(z) * 0x165667B19E3779F9ull))));
}

// ------------------------------------------------------------------ voxel grid -> octree --
struct Grid {
  int depth;
  int n;
  std::vector<uint8_t> v;  // 0 = empty, else material index + 1
  explicit Grid(int d) : depth(d), n(1 << d), v(static_cast<size_t>(1) << (3 * d), 0) {}
  inline size_t at(int x, int y, int z) const { return (static_cast<size_t>(x) * n + y) * n + z; }
  inline void set(int x, int y, int z, uint8_t m) {
    if (x >= 0 && y >= 0 && z >= 0 && x < n && y < n && z < n) v[at(x, y, z)] = m;
  }
};

constexpr uint8_t MIXED = 0xFF;

// Breadth-first linearisation into the reference's indirect-cell format
// (Node{value,type}, 8 nodes per cell, node index = cell*8 + x*4 + y*2 + z; raytracer.comp:184,375-376).
// A uniformly filled subtree becomes one LEAF at its own level; cells [0,K) are the top levels.
bool build_octree(const Grid &g, tdt_scene &s, int cell_count) {
  const int D = g.depth;
  // pyramid[l] has edge 2^l; pyramid[D] aliases the voxel grid
  std::vector<std::vector<uint8_t>> pyr(D + 1);
  for (int l = D - 1; l >= 1; l--) {
    const int n = 1 << l, nc = n << 1;
    const uint8_t *child = (l + 1 == D) ? g.v.data() : pyr[l + 1].data();
    pyr[l].assign(static_cast<size_t>(n) * n * n, 0);
    for (int x = 0; x < n; x++)
      for (int y = 0; y < n; y++)
        for (int z = 0; z < n; z++) {
          uint8_t first = child[(static_cast<size_t>(2 * x) * nc + 2 * y) * nc + 2 * z];
          uint8_t r = first;
          for (int c = 1; c < 8 && r != MIXED; c++) {
            uint8_t m = child[(static_cast<size_t>(2 * x + (c >> 2)) * nc + 2 * y + ((c >> 1) & 1)) * nc + 2 * z + (c & 1)];
            if (m != first) r = MIXED;
          }
          pyr[l][(static_cast<size_t>(x) * n + y) * n + z] = r;
        }
  }
  struct Item { int level, x, y, z; };  // a cell whose 8 nodes live at `level` (edge 2^level), base coords
  std::deque<Item> queue;
  queue.push_back({1, 0, 0, 0});
  size_t next_cell = 1;
  s.cells.clear();
  while (!queue.empty()) {
    Item it = queue.front();
    queue.pop_front();
    const int n = 1 << it.level;
    const uint8_t *lev = (it.level == D) ? g.v.data() : pyr[it.level].data();
    for (int c = 0; c < 8; c++) {
      int x = it.x + (c >> 2), y = it.y + ((c >> 1) & 1), z = it.z + (c & 1);
      uint8_t m = lev[(static_cast<size_t>(x) * n + y) * n + z];
      if (m == 0) { s.cells.push_back(0); s.cells.push_back(EMPTY); }
      else if (m != MIXED) { s.cells.push_back(static_cast<uint32_t>(m - 1)); s.cells.push_back(LEAF); }
      else {
        s.cells.push_back(static_cast<uint32_t>(next_cell)); s.cells.push_back(PARENT);
        next_cell++;
        queue.push_back({it.level + 1, 2 * x, 2 * y, 2 * z});
      }
    }
    if (next_cell > static_cast<size_t>(cell_count)) {
      g_err = "scene needs more than cell_count cells";
      return false;
    }
  }
  count_nodes(s, next_cell);
  int64_t occ = 0;
  for (uint8_t m : g.v) occ += (m != 0);
  s.counts[5] = occ;
  return true;
}

// ------------------------------------------------------------------ material tables -------
// The reference's own parameter values: fuzz {0.1,0.3,0.4,0.8} (main.rs:418-423), ior 1.2 (main.rs:439-441).
void make_materials(tdt_scene &s, uint64_t seed, int n_lambert, int n_metal, int n_dielectric) {
  s.metal = {0.1f, 0.3f, 0.4f, 0.8f};
  s.dielectric = {1.2f};
  const int n_albedo = 16;
  s.albedos.clear();
  for (int i = 0; i < n_albedo; i++)
    for (int c = 0; c < 3; c++) {
      int q = 1 + static_cast<int>(hash3(seed, i, c, 77) % 9);  // 0.1 .. 0.9
      s.albedos.push_back(static_cast<float>(q) / 10.0f);
    }
  s.materials.clear();
  int idx = 0;
  for (int i = 0; i < n_lambert; i++, idx++) { s.materials.insert(s.materials.end(), {LAMBERTIAN, 0u, static_cast<uint32_t>(idx % n_albedo)}); }
  for (int i = 0; i < n_metal; i++, idx++) { s.materials.insert(s.materials.end(), {METAL, static_cast<uint32_t>(i % 4), static_cast<uint32_t>(idx % n_albedo)}); }
  for (int i = 0; i < n_dielectric; i++, idx++) { s.materials.insert(s.materials.end(), {DIELECTRIC, 0u, static_cast<uint32_t>(idx % n_albedo)}); }
}

// camera of main.rs:165-168 in octree-normalised coordinates: origin (0,-0.1,-0.3), octree min
// (-0.5,-0.5,-1.0), scale 1  ->  (0.5, 0.4, 0.7); it looks towards -z.
void carve_camera_cavity(Grid &g, int radius_vox) {
  const int cx = g.n / 2, cy = (g.n * 2) / 5, cz = (g.n * 7) / 10;
  const int64_t r2 = static_cast<int64_t>(radius_vox) * radius_vox;
  for (int x = cx - radius_vox; x <= cx + radius_vox; x++)
    for (int y = cy - radius_vox; y <= cy + radius_vox; y++)
      for (int z = cz - radius_vox; z <= cz + radius_vox; z++) {
        int64_t dx = x - cx, dy = y - cy, dz = z - cz;
        if (dx * dx + dy * dy + dz * dz <= r2) g.set(x, y, z, 0);
      }
}

void fill_sphere(Grid &g, int cx, int cy, int cz, int r, int r_inner, uint8_t m) {
  const int64_t r2 = static_cast<int64_t>(r) * r, ri2 = static_cast<int64_t>(r_inner) * r_inner;
  for (int x = std::max(0, cx - r); x <= std::min(g.n - 1, cx + r); x++)
    for (int y = std::max(0, cy - r); y <= std::min(g.n - 1, cy + r); y++)
      for (int z = std::max(0, cz - r); z <= std::min(g.n - 1, cz + r); z++) {
        int64_t dx = x - cx, dy = y - cy, dz = z - cz, d2 = dx * dx + dy * dy + dz * dz;
        if (d2 <= r2 && (r_inner <= 0 || d2 >= ri2)) g.v[g.at(x, y, z)] = m;
      }
}

// config 1: integer-hash occupancy 35 %, all Lambertian, 4 albedos
void gen_hash_grid(Grid &g, tdt_scene &s, uint64_t seed) {
  s.metal = {0.1f, 0.3f, 0.4f, 0.8f};
  s.dielectric = {1.2f};
  s.albedos = {0.1f, 0.2f, 0.5f, 0.8f, 0.8f, 0.0f, 0.8f, 0.6f, 0.2f, 0.2f, 0.4f, 0.8f};  // 4 of main.rs:394-400
  s.materials = {LAMBERTIAN, 0, 0, LAMBERTIAN, 0, 1, LAMBERTIAN, 0, 2, LAMBERTIAN, 0, 3};
  for (int x = 0; x < g.n; x++)
    for (int y = 0; y < g.n; y++)
      for (int z = 0; z < g.n; z++) {
        uint64_t h = hash3(seed, x, y, z);
        if (h % 100 < 35) g.v[g.at(x, y, z)] = static_cast<uint8_t>(1 + ((h >> 32) & 3));
      }
  carve_camera_cavity(g, std::max(1, g.n / 6));
}

// configs 2-4: value-noise height-field floor (a shell a few voxels thick) + hash-placed solid
// spheres; 60 % Lambertian / 25 % metal / 15 % dielectric materials
void gen_terrain(Grid &g, tdt_scene &s, uint64_t seed) {
  make_materials(s, seed, 12, 5, 3);
  const int n_mat = 20, N = g.n;
  const int lattice = std::max(2, N / 8);      // coarse octave period
  const int fine = std::max(1, N / 32);        // fine octave period
  const int thickness = std::max(2, N / 64);
  auto lat = [&](uint64_t salt, int i, int k) { return static_cast<int64_t>(hash3(seed ^ salt, i, 0, k) & 0xFFFF); };
  auto noise = [&](uint64_t salt, int period, int x, int z) {  // bilinear value noise, 16.16 fixed point in [0,65535]
    int i = x / period, k = z / period;
    int64_t fx = ((static_cast<int64_t>(x % period)) << 16) / period, fz = ((static_cast<int64_t>(z % period)) << 16) / period;
    int64_t a = lat(salt, i, k), b = lat(salt, i + 1, k), c = lat(salt, i, k + 1), d = lat(salt, i + 1, k + 1);
    int64_t ab = a + (((b - a) * fx) >> 16), cd = c + (((d - c) * fx) >> 16);
    return ab + (((cd - ab) * fz) >> 16);
  };
  for (int x = 0; x < N; x++)
    for (int z = 0; z < N; z++) {
      // height in voxels: 6 % .. 30 % of the edge
      int64_t hn = (noise(1, lattice, x, z) * 3 + noise(2, fine, x, z)) >> 2;
      int h = static_cast<int>((static_cast<int64_t>(N) * (6 * 65536 + 24 * hn)) / (100 * 65536));
      uint8_t m = static_cast<uint8_t>(1 + hash3(seed ^ 3, x / std::max(1, N / 16), 0, z / std::max(1, N / 16)) % n_mat);
      for (int y = std::max(0, h - thickness); y <= h && y < N; y++) g.v[g.at(x, y, z)] = m;
    }
  const int n_spheres = 28;
  for (int i = 0; i < n_spheres; i++) {
    uint64_t h = hash3(seed ^ 4, i, 1, 2);
    int r = std::max(1, N / 40 + static_cast<int>((h & 0xFF) * static_cast<uint64_t>(N / 12) / 256));
    int cx = static_cast<int>(((h >> 8) & 0xFFFF) * static_cast<uint64_t>(N) >> 16);
    int cy = N / 5 + static_cast<int>(((h >> 24) & 0xFFFF) * static_cast<uint64_t>(N * 3 / 5) >> 16);
    int cz = static_cast<int>(((h >> 40) & 0xFFFF) * static_cast<uint64_t>(N * 13 / 20) >> 16);
    uint8_t m = static_cast<uint8_t>(1 + (hash3(seed ^ 5, i, 3, 4) % n_mat));
    fill_sphere(g, cx, cy, cz, r, 0, m);
  }
  carve_camera_cavity(g, std::max(2, N / 10));
}

// config 5: thin spherical shells with sponge holes (<= 2 % occupancy)
void gen_shells(Grid &g, tdt_scene &s, uint64_t seed) {
  make_materials(s, seed, 12, 5, 3);
  const int n_mat = 20, N = g.n;
  const int n_shells = 36;
  for (int i = 0; i < n_shells; i++) {
    uint64_t h = hash3(seed ^ 6, i, 5, 6);
    int r = N / 24 + static_cast<int>((h & 0xFF) * static_cast<uint64_t>(N / 7) / 256);
    int cx = static_cast<int>(((h >> 8) & 0xFFFF) * static_cast<uint64_t>(N) >> 16);
    int cy = static_cast<int>(((h >> 24) & 0xFFFF) * static_cast<uint64_t>(N) >> 16);
    int cz = static_cast<int>(((h >> 40) & 0xFFFF) * static_cast<uint64_t>(N * 13 / 20) >> 16);
    uint8_t m = static_cast<uint8_t>(1 + (hash3(seed ^ 7, i, 7, 8) % n_mat));
    fill_sphere(g, cx, cy, cz, r, std::max(1, r - 2), m);
  }
  // sponge: knock out one block in four
  const int blk = std::max(1, N / 64);
  for (int x = 0; x < N; x++)
    for (int y = 0; y < N; y++)
      for (int z = 0; z < N; z++)
        if (g.v[g.at(x, y, z)] && (hash3(seed ^ 8, x / blk, y / blk, z / blk) & 3) == 0) g.v[g.at(x, y, z)] = 0;
  carve_camera_cavity(g, std::max(2, N / 10));
}

}  // namespace

// ------------------------------------------------------------------ PLY point loader -------

struct tdt_ply {
  int64_t header_vertex = 0;
  std::vector<int> prop_id;        // 0..5 = x y z r g b   (Identifier, ply_point_loader.rs:52-59)
  std::vector<int> prop_is_pos;    // Type::Pos / Type::Uchar (:62-65)
  std::vector<int32_t> voxels;     // x, y, z, key
  std::map<uint32_t, std::array<uint8_t, 3>> albedos;
  int32_t min_point[3] = {INT32_MAX, INT32_MAX, INT32_MAX};   // :221
};

namespace {

// mod ply_reader (:321-339)
constexpr uint8_t CR = 0x0D, NL = 0x0A, SPACE = 0x20;
// expect(): index of the first mismatch over the ZIPPED (shorter) length, or -1 (None)
int64_t ply_expect(const char *expected, const uint8_t *b, size_t n) {
  for (size_t i = 0; expected[i] != 0 && i < n; i++)
    if (static_cast<uint8_t>(expected[i]) != b[i]) return static_cast<int64_t>(i);
  return -1;
}
int64_t seek_end_of_line(const uint8_t *b, size_t n) {
  for (size_t i = 0; i < n; i++) if (b[i] == CR || b[i] == NL) return static_cast<int64_t>(i);
  return -1;
}
int64_t seek_end_word(const uint8_t *b, size_t n) {
  for (size_t i = 0; i < n; i++) if (b[i] == SPACE || b[i] == CR || b[i] == NL) return static_cast<int64_t>(i);
  return -1;
}
// str::parse::<i64>() on ASCII: optional sign, at least one digit, nothing else; range-checked by the caller
bool parse_int(const uint8_t *b, size_t n, int64_t lo, int64_t hi, bool allow_minus, int64_t &out) {
  size_t i = 0; bool neg = false;
  if (n == 0) return false;
  if (b[0] == '+') i = 1; else if (b[0] == '-') { if (!allow_minus) return false; neg = true; i = 1; }
  if (i >= n) return false;
  int64_t v = 0;
  for (; i < n; i++) {
    if (b[i] < '0' || b[i] > '9') return false;
    v = v * 10 + (b[i] - '0');
    if (v > (int64_t{1} << 40)) return false;
  }
  v = neg ? -v : v;
  if (v < lo || v > hi) return false;
  out = v;
  return true;
}
// cantor_pair (:228-241): f64 arithmetic, `as u32` saturates
uint32_t cantor_pair(uint8_t a, uint8_t b, uint8_t c) {
  const double fa = a, fb = b, fc = c;
  const double fd = 0.5 * (fa + fb) * (fa + fb + 1.0) + fb;
  const double hash = 0.5 * (fd + fc) * (fd + fc + 1.0) + fc;
  return hash >= 4294967295.0 ? 0xFFFFFFFFu : static_cast<uint32_t>(hash);
}

bool ply_fail(const char *what, const char *state, size_t offset) {
  g_err = std::string("Unexpected ") + what + " at offset '" + std::to_string(offset) + "', State: " + state;   // Display, :27-29
  return false;
}

bool ply_parse_strict(const uint8_t *buf, size_t n, tdt_ply &out) {
  size_t off = 0;
  auto need = [&](size_t k) { return off + k <= n; };
  // HeaderSubstate::Ply / Format (:119-134)
  if (!need(5)) return ply_fail("end of file", "ReadHeader(Ply)", off);
  if (int64_t e = ply_expect("ply\r\n", buf + off, 5); e >= 0) return ply_fail("character", "ReadHeader(Ply)", static_cast<size_t>(e));
  off += 5;
  if (!need(18)) return ply_fail("end of file", "ReadHeader(Format)", off);
  if (int64_t e = ply_expect("format ascii 1.0\r\n", buf + off, 18); e >= 0) return ply_fail("format", "ReadHeader(Format)", static_cast<size_t>(e));
  off += 18;
  for (;;) {   // HeaderSubstate::InferLine (:135-154)
    if (off >= n) return ply_fail("end of file", "ReadHeader(InferLine)", off);
    int64_t end = seek_end_word(buf + off, n - off);
    if (end < 0) return ply_fail("end of file", "ReadHeader(InferLine)", off);
    const size_t w = static_cast<size_t>(end);
    if (ply_expect("property", buf + off, w) < 0) {                 // HeaderSubstate::Propery (:177-212)
      off += 9;
      if (off >= n) return ply_fail("end of file", "ReadHeader(Propery)", off);
      int is_pos;
      if (ply_expect("float", buf + off, n - off) < 0) { off += 6; is_pos = 1; }
      else if (ply_expect("uchar", buf + off, n - off) < 0) { off += 6; is_pos = 0; }
      else return ply_fail("type", "ReadHeader(Propery)", off);
      if (off >= n) return ply_fail("end of file", "ReadHeader(Propery)", off);
      int id;
      switch (buf[off]) { case 'x': id = 0; break; case 'y': id = 1; break; case 'z': id = 2; break;
                          case 'r': id = 3; break; case 'g': id = 4; break; case 'b': id = 5; break;
                          default: return ply_fail("variable", "ReadHeader(Propery)", off); }
      out.prop_id.push_back(id); out.prop_is_pos.push_back(is_pos);
      int64_t eol = seek_end_of_line(buf + off, n - off);
      if (eol < 0) return ply_fail("end of file", "ReadHeader(Propery)", off);
      off += static_cast<size_t>(eol) + 2;
    } else if (ply_expect("comment", buf + off, w) < 0) {           // HeaderSubstate::Comment (:155-160)
      off += 8;
      if (off > n) return ply_fail("end of file", "ReadHeader(Comment)", off);
      int64_t eol = seek_end_of_line(buf + off, n - off);
      if (eol < 0) return ply_fail("end of file", "ReadHeader(Comment)", off);
      off += static_cast<size_t>(eol) + 2;
    } else if (ply_expect("element", buf + off, w) < 0) {           // HeaderSubstate::Element (:161-176)
      off += 8;
      if (off > n) return ply_fail("end of file", "ReadHeader(Element)", off);
      if (int64_t e = ply_expect("vertex", buf + off, n - off); e >= 0) return ply_fail("character", "ReadHeader(Element)", static_cast<size_t>(e));
      off += 7;
      if (off > n) return ply_fail("end of file", "ReadHeader(Element)", off);
      int64_t eol = seek_end_of_line(buf + off, n - off);
      if (eol < 0) return ply_fail("end of file", "ReadHeader(Element)", off);
      int64_t v;
      if (!parse_int(buf + off, static_cast<size_t>(eol), 0, int64_t{1} << 40, false, v)) return ply_fail("vertex count", "ReadHeader(Element)", off);
      out.header_vertex = v;
      off += static_cast<size_t>(eol) + 2;
    } else if (ply_expect("end_header", buf + off, w) < 0) {        // (:149-153)
      off += 12;
      break;
    } else {
      return ply_fail("header keyword (the reference does not terminate here)", "ReadHeader(InferLine)", off);
    }
  }
  // ReadState::ReadPoint (:243-313)
  for (;;) {
    if (off >= n || seek_end_of_line(buf + off, n - off) < 0) break;     // EOF test (:247-250)
    int32_t pos[3] = {0, 0, 0};
    uint8_t albedo[3] = {0, 0, 0};
    uint32_t key = 0;
    for (size_t k = 0; k < out.prop_id.size(); k++) {
      if (off > n) return ply_fail("end of file", "ReadPoint", off);
      int64_t we = seek_end_word(buf + off, n - off);
      if (we < 0) return ply_fail("end of file", "ReadPoint", off);
      int64_t v;
      const int id = out.prop_id[k];
      if (out.prop_is_pos[k]) {
        if (!parse_int(buf + off, static_cast<size_t>(we), INT32_MIN, INT32_MAX, true, v)) return ply_fail("FloatParseError", "ReadPoint", off);
        if (id < 3) { pos[id] = static_cast<int32_t>(v); out.min_point[id] = std::min(out.min_point[id], pos[id]); }
      } else {
        if (!parse_int(buf + off, static_cast<size_t>(we), 0, 255, false, v)) return ply_fail("FloatParseError", "ReadPoint", off);
        if (id >= 3) albedo[id - 3] = static_cast<uint8_t>(v);
      }
      key = cantor_pair(albedo[0], albedo[1], albedo[2]);               // after EVERY property (:300)
      if (!out.albedos.count(key)) out.albedos[key] = {albedo[0], albedo[1], albedo[2]};
      off += static_cast<size_t>(we) + 1;
    }
    out.voxels.insert(out.voxels.end(), {pos[0], pos[1], pos[2], static_cast<int32_t>(key)});
    off += 1;
  }
  return true;
}

}  // namespace

// ==================================================================== C ABI ===============
extern "C" {

const char *tdt_host_last_error(void) { return g_err.c_str(); }
void tdt_host_set_error(const char *msg) { g_err = msg ? msg : ""; }   // for host_view.cpp

int tdt_camera_build(const tdt_camera_builder *b, tdt_camera_uniforms *out) {
  if (!b || !out) { g_err = "null argument"; return 1; }
  // camera.rs:135-196 — every step in f32, in the reference's order
  const float aspect_ratio = b->has_aspect_ratio ? b->aspect_ratio : 16.0f / 9.0f;
  const float pi = 3.14159265358979323846f;        // std::f32::consts::PI
  const float theta = b->vertical_fov * pi / 180.0f;
  const float h = std::tan(theta / 2.0f);
  const float viewport_height = (b->has_viewport_height ? b->viewport_height : 2.0f) * h;
  const float viewport_width = aspect_ratio * viewport_height;
  const float origin[3] = {b->has_origin ? b->origin[0] : 0.0f, b->has_origin ? b->origin[1] : 0.0f,
                           b->has_origin ? b->origin[2] : 0.0f};
  // forward = unit_z, right = unit_y x forward = unit_x, up = forward x right = unit_y (identity orientation)
  const float forward[3] = {0.0f, 0.0f, 1.0f}, right[3] = {1.0f, 0.0f, 0.0f}, up[3] = {0.0f, 1.0f, 0.0f};
  for (int i = 0; i < 3; i++) {
    out->horizontal[i] = right[i] * viewport_width;
    out->vertical[i] = up[i] * viewport_height;
    out->lower_left_corner[i] = ((origin[i] - out->horizontal[i] * 0.5f) - out->vertical[i] * 0.5f) - forward[i];
    out->origin[i] = origin[i];
  }
  out->image_width = b->image_width;
  // `(self.image_width as f32 / aspect_ratio) as i32`: Rust float->int casts truncate and saturate
  float hh = static_cast<float>(b->image_width) / aspect_ratio;
  out->image_height = (hh != hh) ? 0 : (hh >= 2147483648.0f ? INT32_MAX : (hh <= -2147483648.0f ? INT32_MIN : static_cast<int32_t>(hh)));
  out->samples_per_pixel = b->has_samples_per_pixel ? b->samples_per_pixel : 10;
  out->max_bounce = b->has_max_bounce ? b->max_bounce : 3;
  return 0;
}

int tdt_camera_reference_pose(int width, int height, int spp, int max_bounce, tdt_camera_uniforms *out) {
  tdt_camera_builder b;
  std::memset(&b, 0, sizeof b);
  b.vertical_fov = 90.0f; b.image_width = width;                                  // main.rs:165
  b.has_aspect_ratio = 1; b.aspect_ratio = static_cast<float>(width) / static_cast<float>(height);  // main.rs:166
  b.has_origin = 1; b.origin[0] = 0.0f; b.origin[1] = -0.1f; b.origin[2] = -0.3f;  // main.rs:167
  b.has_viewport_height = 1; b.viewport_height = 2.0f;                             // main.rs:168
  b.has_samples_per_pixel = 1; b.samples_per_pixel = spp;
  b.has_max_bounce = 1; b.max_bounce = max_bounce;
  // image_height comes out of build() as (width as f32 / aspect) as i32, exactly as in the
  // reference (camera.rs:154); it equals `height` for every size used here (tested)
  return tdt_camera_build(&b, out);
}

int tdt_scene_demo(tdt_scene **out) {
  if (!out) { g_err = "null argument"; return 1; }
  tdt_scene *s = new (std::nothrow) tdt_scene;
  if (!s) { g_err = "out of memory"; return 1; }
  const uint32_t E = EMPTY, P = PARENT, L = LEAF;
  auto cell = [&](std::initializer_list<uint32_t> nodes) { s->cells.insert(s->cells.end(), nodes); };
  auto chain = [&](uint32_t child) { for (int i = 0; i < 8; i++) { s->cells.push_back(child); s->cells.push_back(P); } };
  // main.rs:238-337 — (value, type) pairs, 8 per cell
  cell({1, P, 1, E, 1, E, 10, P, 1, E, 1, E, 1, E, 1, P});        // cell 0 (root)
  for (uint32_t c = 2; c <= 8; c++) chain(c);                      // cells 1..7: every child -> next cell
  cell({9, E, 9, P, 9, P, 9, E, 9, E, 9, P, 9, P, 9, P});         // cell 8
  cell({0, L, 2, L, 0, L, 1, L, 0, L, 3, L, 0, L, 0, L});         // cell 9
  cell({11, P, 11, P, 11, E, 11, E, 11, E, 11, E, 11, P, 11, P}); // cell 10
  for (uint32_t c = 12; c <= 18; c++) chain(c);                    // cells 11..17
  cell({7, L, 2, L, 6, L, 1, L, 0, L, 3, L, 0, L, 5, L});         // cell 18
  // main.rs:339-341: `for _ in 8 * 2 * 10..PRE_ALLOCATED_CELLS { push(EMPTY) }`
  for (int i = 8 * 2 * 10; i < 100000; i++) s->cells.push_back(E);
  s->materials = {LAMBERTIAN, 0, 0, LAMBERTIAN, 0, 1, DIELECTRIC, 0, 2, METAL, 0, 3, METAL, 1, 4,      // main.rs:363-379
                  METAL, 2, 5, METAL, 3, 6, DIELECTRIC, 0, 4, DIELECTRIC, 0, 5, LAMBERTIAN, 0, 6,
                  LAMBERTIAN, 0, 5, LAMBERTIAN, 0, 4, LAMBERTIAN, 0, 3};
  s->albedos = {0.1f, 0.2f, 0.5f, 0.8f, 0.8f, 0.0f, 0.8f, 0.8f, 0.8f, 0.8f, 0.6f, 0.2f,              // main.rs:394-401
                0.2f, 0.4f, 0.8f, 0.4f, 0.8f, 0.2f, 0.2f, 0.2f, 0.2f};
  s->metal = {0.1f, 0.3f, 0.4f, 0.8f};                                                                 // main.rs:418-423
  s->dielectric = {1.2f};                                                                              // main.rs:439-441
  set_octree_uniforms(*s, -0.5f, -0.5f, -1.0f, 1.0f, 10, 100, 100000);                                // main.rs:455-462
  count_nodes(*s, 19);
  *out = s;
  return 0;
}

int tdt_scene_generate(const tdt_scene_params *p, tdt_scene **out) {
  if (!p || !out) { g_err = "null argument"; return 1; }
  if (p->max_depth < 1 || p->max_depth > 9) { g_err = "max_depth must be 1..9"; return 1; }
  if (p->cell_count < 1 || (p->cell_count & (p->cell_count - 1)) != 0) { g_err = "cell_count must be a power of two"; return 1; }
  tdt_scene *s = new (std::nothrow) tdt_scene;
  if (!s) { g_err = "out of memory"; return 1; }
  try {
    Grid g(p->max_depth);
    switch (p->kind) {
      case TDT_SCENE_HASH_GRID: gen_hash_grid(g, *s, p->seed); break;
      case TDT_SCENE_TERRAIN: gen_terrain(g, *s, p->seed); break;
      case TDT_SCENE_SHELLS: gen_shells(g, *s, p->seed); break;
      default: g_err = "unknown scene kind"; delete s; return 1;
    }
    if (!build_octree(g, *s, p->cell_count)) { delete s; return 1; }
  } catch (const std::bad_alloc &) { g_err = "out of memory"; delete s; return 1; }
  set_octree_uniforms(*s, -0.5f, -0.5f, -1.0f, 1.0f, p->max_depth, p->max_iter, p->cell_count);   // main.rs:456-457 AABB
  *out = s;
  return 0;
}

int tdt_scene_config(int config, tdt_scene **out) {
  tdt_scene_params p;
  std::memset(&p, 0, sizeof p);
  p.seed = 0x5EED0001ull + static_cast<uint64_t>(config == 4 ? 3 : config);
  switch (config) {
    case 0: return tdt_scene_demo(out);
    case 1: p.kind = TDT_SCENE_HASH_GRID; p.max_depth = 3; p.cell_count = 128; p.max_iter = 100; break;
    case 2: p.kind = TDT_SCENE_TERRAIN; p.max_depth = 6; p.cell_count = 1 << 16; p.max_iter = 100; break;
    case 3: case 4: p.kind = TDT_SCENE_TERRAIN; p.max_depth = 8; p.cell_count = 1 << 20; p.max_iter = 256; break;
    case 5: p.kind = TDT_SCENE_SHELLS; p.max_depth = 9; p.cell_count = 1 << 20; p.max_iter = 512; break;
    default: g_err = "config must be 0..5"; return 1;
  }
  return tdt_scene_generate(&p, out);
}

int tdt_scene_from_blobs(const void *const blobs[8], const size_t bytes[8], tdt_scene **out) {
  if (!blobs || !bytes || !out) { g_err = "null argument"; return 1; }
  tdt_scene *s = new (std::nothrow) tdt_scene;
  if (!s) { g_err = "out of memory"; return 1; }
  auto cp = [&](auto &vec, int slot) {
    using T = typename std::remove_reference<decltype(vec)>::type::value_type;
    vec.resize(bytes[slot] / sizeof(T));
    if (!vec.empty()) std::memcpy(vec.data(), blobs[slot], vec.size() * sizeof(T));
  };
  cp(s->cells, 0); cp(s->materials, 1); cp(s->albedos, 2); cp(s->metal, 3); cp(s->dielectric, 4);
  cp(s->octree_floats, 6); cp(s->octree_ints, 7);
  count_nodes(*s, s->cells.size() / 16);
  *out = s;
  return 0;
}

void tdt_scene_destroy(tdt_scene *s) { delete s; }

const void *tdt_scene_blob(const tdt_scene *s, int slot, size_t *bytes) {
  const void *p = nullptr; size_t n = 0;
  if (s) switch (slot) {
    case 0: p = s->cells.data(); n = s->cells.size() * 4; break;
    case 1: p = s->materials.data(); n = s->materials.size() * 4; break;
    case 2: p = s->albedos.data(); n = s->albedos.size() * 4; break;
    case 3: p = s->metal.data(); n = s->metal.size() * 4; break;
    case 4: p = s->dielectric.data(); n = s->dielectric.size() * 4; break;
    case 6: p = s->octree_floats.data(); n = s->octree_floats.size() * 4; break;
    case 7: p = s->octree_ints.data(); n = s->octree_ints.size() * 4; break;
    default: break;
  }
  if (bytes) *bytes = n;
  return p;
}

int tdt_scene_counts(const tdt_scene *s, int64_t out[6]) {
  if (!s || !out) { g_err = "null argument"; return 1; }
  for (int i = 0; i < 6; i++) out[i] = s->counts[i];
  return 0;
}

int tdt_ply_parse(const void *data, size_t bytes, int strict_crlf, tdt_ply **out) {
  if (!data || !out) { g_err = "null argument"; return 1; }
  *out = nullptr;
  const uint8_t *b = static_cast<const uint8_t *>(data);
  std::vector<uint8_t> conv;
  if (!strict_crlf) {   // LF-only file: give every line the CRLF the reference grammar expects
    bool has_cr = false;
    for (size_t i = 0; i < bytes; i++) if (b[i] == CR) { has_cr = true; break; }
    if (!has_cr) {
      conv.reserve(bytes + bytes / 8);
      for (size_t i = 0; i < bytes; i++) { if (b[i] == NL) conv.push_back(CR); conv.push_back(b[i]); }
      b = conv.data(); bytes = conv.size();
    }
  }
  tdt_ply *p = new (std::nothrow) tdt_ply;
  if (!p) { g_err = "out of memory"; return 1; }
  try {
    if (!ply_parse_strict(b, bytes, *p)) { delete p; return 1; }
  } catch (const std::bad_alloc &) { g_err = "out of memory"; delete p; return 1; }
  *out = p;
  return 0;
}

void tdt_ply_destroy(tdt_ply *p) { delete p; }

int tdt_ply_info(const tdt_ply *p, int64_t *header_vertex, int64_t *n_voxels, int32_t min_point[3], int64_t *n_albedos) {
  if (!p) { g_err = "null argument"; return 1; }
  if (header_vertex) *header_vertex = p->header_vertex;
  if (n_voxels) *n_voxels = static_cast<int64_t>(p->voxels.size() / 4);
  if (min_point) for (int i = 0; i < 3; i++) min_point[i] = p->min_point[i];
  if (n_albedos) *n_albedos = static_cast<int64_t>(p->albedos.size());
  return 0;
}

const int32_t *tdt_ply_voxels(const tdt_ply *p) { return p ? p->voxels.data() : nullptr; }

int64_t tdt_ply_albedos(const tdt_ply *p, uint32_t *keys, uint8_t *rgb, int64_t capacity) {
  if (!p) return 0;
  int64_t i = 0;
  for (const auto &kv : p->albedos) {
    if (i >= capacity) break;
    if (keys) keys[i] = kv.first;
    if (rgb) { rgb[3 * i] = kv.second[0]; rgb[3 * i + 1] = kv.second[1]; rgb[3 * i + 2] = kv.second[2]; }
    i++;
  }
  return i;
}

int tdt_scene_from_ply(const tdt_ply *p, int max_iter, int z_up, tdt_scene **out) {
  if (!p || !out) { g_err = "null argument"; return 1; }
  *out = nullptr;
  const size_t nv = p->voxels.size() / 4;
  if (nv == 0) { g_err = "the PLY holds no voxels"; return 1; }
  int32_t mx[3] = {INT32_MIN, INT32_MIN, INT32_MIN};
  for (size_t i = 0; i < nv; i++) for (int a = 0; a < 3; a++) mx[a] = std::max(mx[a], p->voxels[4 * i + a]);
  int64_t ext[3];
  for (int a = 0; a < 3; a++) ext[a] = static_cast<int64_t>(mx[a]) - p->min_point[a] + 1;
  const int64_t emax = std::max(ext[0], std::max(ext[1], ext[2]));
  int depth = 1;
  while ((int64_t{1} << depth) < emax) depth++;
  if (depth > 9) { g_err = "model larger than 512 voxels on an edge"; return 1; }
  tdt_scene *s = new (std::nothrow) tdt_scene;
  if (!s) { g_err = "out of memory"; return 1; }
  try {
    // one Lambertian material per colour that a voxel really ends up with (not the polluted palette)
    std::map<uint32_t, uint32_t> mat_of_key;
    for (size_t i = 0; i < nv; i++) mat_of_key.emplace(static_cast<uint32_t>(p->voxels[4 * i + 3]), 0u);
    if (mat_of_key.size() > 254) { g_err = "more than 254 distinct colours"; delete s; return 1; }
    uint32_t m = 0;
    for (auto &kv : mat_of_key) {
      kv.second = m;
      const auto &rgb = p->albedos.at(kv.first);
      s->materials.insert(s->materials.end(), {LAMBERTIAN, 0u, m});
      for (int c = 0; c < 3; c++) s->albedos.push_back(static_cast<float>(rgb[c]) / 255.0f);
      m++;
    }
    s->metal = {0.1f, 0.3f, 0.4f, 0.8f};     // the reference's tables (main.rs:418-441); unused by Lambertian voxels
    s->dielectric = {1.2f};
    Grid g(depth);
    const int N = g.n;
    // file axes -> octree axes; centred in x, on the floor, at the far (z = 0) side
    const int ax = 0, ay = z_up ? 2 : 1, az = z_up ? 1 : 2;
    const int ox = static_cast<int>((N - ext[ax]) / 2), oy = 0, oz = 0;
    for (size_t i = 0; i < nv; i++) {
      const int x = p->voxels[4 * i + ax] - p->min_point[ax] + ox;
      const int y = p->voxels[4 * i + ay] - p->min_point[ay] + oy;
      const int z = p->voxels[4 * i + az] - p->min_point[az] + oz;
      g.set(x, y, z, static_cast<uint8_t>(1 + mat_of_key[static_cast<uint32_t>(p->voxels[4 * i + 3])]));
    }
    int cell_count = 1 << 10;
    for (;;) {
      if (build_octree(g, *s, cell_count)) break;
      if (cell_count >= (1 << 22)) { delete s; return 1; }
      cell_count <<= 1;
    }
    set_octree_uniforms(*s, -0.5f, -0.5f, -1.0f, 1.0f, depth, max_iter, cell_count);
  } catch (const std::bad_alloc &) { g_err = "out of memory"; delete s; return 1; }
  *out = s;
  return 0;
}

}  // extern "C"
