// tdt_demo — the reference's main.rs, headless: everything main.rs does between creating the GL context and the
// first `dispatch_compute` (main.rs:156-470, 579), written against include/renderer.hpp (the C++ mirror of
// `src/renderer`), then the frame is read back and written as a PFM instead of being blitted to a window.
//
//   tdt_demo [--size WxH] [--spp N] [--bounce N] [--settings camera.ron] [--move KEYS] [--edit x,y,z,type,value]
//            [--device N] [--out frame.pfm] [--png frame.png]
//
// Defaults are the reference's: 1280x720 window (main.rs:26), camera.ron's spp 4 / max_bounce 6 / controller rates.
// --move replays key presses through the camera controller (main.rs:506-546), one 1/60 s frame each:
//   w a s d = translate Front / Left / Back / Rigth, u j = Up / Down, q e = turn_yaw(-/+ 1), r f = turn_pitch(-/+ 1),
//   S / N = sprint / normal speed.  --png writes the frame as the quad pass would present it (main.rs:582-600).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "renderer.hpp"

using namespace renderer;

int main(int argc, char **argv) {
  int W = 1280, H = 720, spp = 4, bounce = 6, device = 0;
  std::string out, png, settings_path, moves;
  std::vector<float> edit;
  for (int i = 1; i < argc; i++) {
    std::string a = argv[i];
    auto next = [&]() -> const char * { if (i + 1 >= argc) { std::fprintf(stderr, "missing value for %s\n", a.c_str()); std::exit(2); } return argv[++i]; };
    if (a == "--size") { if (std::sscanf(next(), "%dx%d", &W, &H) != 2) { std::fprintf(stderr, "--size WxH\n"); return 2; } }
    else if (a == "--spp") spp = std::atoi(next());
    else if (a == "--bounce") bounce = std::atoi(next());
    else if (a == "--device") device = std::atoi(next());
    else if (a == "--out") out = next();
    else if (a == "--png") png = next();
    else if (a == "--settings") settings_path = next();
    else if (a == "--move") moves = next();
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
    if (!settings_path.empty()) {                                                          // main.rs:170-176 / 492-494
      FILE *f = std::fopen(settings_path.c_str(), "rb");
      if (!f) { std::perror(settings_path.c_str()); return 1; }
      std::string text; char buf[4096]; size_t n;
      while ((n = std::fread(buf, 1, sizeof buf, f)) > 0) text.append(buf, n);
      std::fclose(f);
      camera.apply_settings(raytrace_program.program, camera_settings_from_ron(text));
      spp = camera.settings().samples_per_pixel; bounce = camera.settings().max_bounce;
      camera.set_speed_to_normal();
    }
    for (char k : moves) {                                                                 // main.rs:506-546
      const double dt = 1.0 / 60.0;
      const Program &prog = raytrace_program.program;
      switch (k) {
        case 'w': camera.translate(prog, into_vector3(Direction::Front), dt); break;
        case 'a': camera.translate(prog, into_vector3(Direction::Left), dt); break;
        case 's': camera.translate(prog, into_vector3(Direction::Back), dt); break;
        case 'd': camera.translate(prog, into_vector3(Direction::Rigth), dt); break;
        case 'u': camera.translate(prog, into_vector3(Direction::Up), dt); break;
        case 'j': camera.translate(prog, into_vector3(Direction::Down), dt); break;
        case 'q': camera.turn_yaw(prog, -1.0f); break;
        case 'e': camera.turn_yaw(prog, 1.0f); break;
        case 'r': camera.turn_pitch(prog, -1.0f); break;
        case 'f': camera.turn_pitch(prog, 1.0f); break;
        case 'S': camera.set_speed_to_sprint(); break;
        case 'N': camera.set_speed_to_normal(); break;
        default: std::fprintf(stderr, "unknown key '%c' in --move\n", k); return 2;
      }
    }
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
    std::printf("%dx%d spp %d bounce %d fnv1a %016llx counter %d\n", camera.image_width(), camera.image_height(), spp, bounce, h,
                octree.counter().read<int32_t>(1)[0]);
    if (!out.empty()) {
      FILE *f = std::fopen(out.c_str(), "wb");
      if (!f) { std::perror(out.c_str()); return 1; }
      std::fprintf(f, "PF4\n%d %d\n-1.0\n", camera.image_width(), camera.image_height());     // 4-channel little-endian float map, bottom row first
      std::fwrite(px.data(), sizeof(float), px.size(), f);
      std::fclose(f);
    }
    if (!png.empty()) present_png(camera.render_texture, png);                           // instead of main.rs:582-600
  } catch (const InitializeErr &e) {
    std::fprintf(stderr, "InitializeErr: %s (%s)\n", e.to_string().c_str(), e.detail.c_str());
    return 1;
  }
  return 0;
}
