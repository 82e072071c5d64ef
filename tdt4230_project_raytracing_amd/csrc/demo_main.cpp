// tdt_demo — the reference's main.rs, headless: everything main.rs does between creating the GL context and the
// first `dispatch_compute` (main.rs:156-470, 579), written against include/renderer.hpp (the C++ mirror of
// `src/renderer`), then the frame is read back and written as a PFM instead of being blitted to a window.
//
//   tdt_demo [--size WxH] [--spp N] [--bounce N] [--edit x,y,z,type,value] [--device N] --out frame.pfm
//
// Defaults are the reference's: 1280x720 window (main.rs:26), camera.ron's spp 4 / max_bounce 6.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "renderer.hpp"

using namespace renderer;

int main(int argc, char **argv) {
  int W = 1280, H = 720, spp = 4, bounce = 6, device = 0;
  std::string out;
  std::vector<float> edit;
  for (int i = 1; i < argc; i++) {
    std::string a = argv[i];
    auto next = [&]() -> const char * { if (i + 1 >= argc) { std::fprintf(stderr, "missing value for %s\n", a.c_str()); std::exit(2); } return argv[++i]; };
    if (a == "--size") { if (std::sscanf(next(), "%dx%d", &W, &H) != 2) { std::fprintf(stderr, "--size WxH\n"); return 2; } }
    else if (a == "--spp") spp = std::atoi(next());
    else if (a == "--bounce") bounce = std::atoi(next());
    else if (a == "--device") device = std::atoi(next());
    else if (a == "--out") out = next();
    else if (a == "--edit") { float v[5]; if (std::sscanf(next(), "%f,%f,%f,%f,%f", v, v + 1, v + 2, v + 3, v + 4) != 5) { std::fprintf(stderr, "--edit x,y,z,type,value\n"); return 2; } edit.assign(v, v + 5); }
    else { std::fprintf(stderr, "unknown option %s\n", a.c_str()); return 2; }
  }
  try {
    Context ctx(device);
    // main.rs:156-160
    ComputeShader raytrace_program = ComputeShader::new_(ctx, TDT_PROGRAM_RAYTRACER);
    // main.rs:164-209 (camera.ron's values arrive as --spp / --bounce)
    Camera camera = CameraBuilder::new_(90.0f, W)
                        .with_aspect_ratio((float)W / (float)H)
                        .with_origin({0.0f, -0.1f, -0.3f})
                        .with_viewport_height(2.0f)
                        .with_sample_per_pixel(spp)
                        .with_max_bounce(bounce)
                        .with_turn_rate(0.05f).with_normal_speed(0.03f).with_sprint_speed(0.15f)
                        .build(ctx, raytrace_program.program);
    camera.render_texture.bind();                                                        // main.rs:216
    // main.rs:226-230
    ComputeShader octree_update_program = ComputeShader::new_(ctx, TDT_PROGRAM_OCTREE_UPDATE);
    // main.rs:238-450: the scene literal, one buffer per table, each bound to its shader-storage slot
    tdt_scene *scene = nullptr;
    if (tdt_scene_demo(&scene)) { std::fprintf(stderr, "%s\n", tdt_host_last_error()); return 1; }
    std::vector<VertexBufferObject> keep;
    for (unsigned slot = 0; slot <= 4; slot++) {
      size_t bytes = 0;
      const uint32_t *p = (const uint32_t *)tdt_scene_blob(scene, (int)slot, &bytes);
      keep.push_back(VertexBufferObject::new_<uint32_t>(ctx, std::vector<uint32_t>(p, p + bytes / 4)));
      bind_buffer_base(ctx, TDT_SHADER_STORAGE_BUFFER, slot, keep.back());
    }
    tdt_scene_destroy(scene);
    // main.rs:455-468
    Octree octree = Octree::new_({-0.5f, -0.5f, -1.0f}, 1.0f, 10, 100000, 19, 100);
    octree.init_global_buffers(ctx);
    if (!edit.empty()) {                                                                  // main.rs:555-569, one click
      std::vector<float> delta(500, 0.0f);
      std::copy(edit.begin(), edit.end(), delta.begin());
      octree.update_vbo(delta, 5, octree_update_program);
    }
    // main.rs:578-580
    octree.vao.bind();
    raytrace_program.dispatch_compute(camera.render_texture.width() + 1, camera.render_texture.height() + 1, camera.render_texture.depth());
    VertexArrayObject::unbind();
    const std::vector<float> px = camera.render_texture.read();
    unsigned long long h = 1469598103934665603ull;                                        // FNV-1a of the frame's bits
    for (float f : px) { uint32_t u; std::memcpy(&u, &f, 4); for (int b = 0; b < 4; b++) { h ^= (u >> (8 * b)) & 0xff; h *= 1099511628211ull; } }
    std::printf("%dx%d spp %d bounce %d fnv1a %016llx counter %d\n", camera.image_width, camera.image_height, spp, bounce, h,
                octree.counter().read<int32_t>(1)[0]);
    if (!out.empty()) {
      FILE *f = std::fopen(out.c_str(), "wb");
      if (!f) { std::perror(out.c_str()); return 1; }
      std::fprintf(f, "PF4\n%d %d\n-1.0\n", camera.image_width, camera.image_height);     // 4-channel little-endian float map, bottom row first
      std::fwrite(px.data(), sizeof(float), px.size(), f);
      std::fclose(f);
    }
  } catch (const InitializeErr &e) {
    std::fprintf(stderr, "InitializeErr: %s (%s)\n", e.to_string().c_str(), e.detail.c_str());
    return 1;
  }
  return 0;
}
