/*
 * glref — TEST INFRASTRUCTURE ONLY (never linked into or called by the product).
 *
 * Runs the reference's own GLSL compute shader, UNMODIFIED and loaded at run time from
 * $REF_DIR (default /root/reference) — assets/shaders/raytracer.comp — on the Mesa
 * llvmpipe software rasteriser that is installed in this image, with no X server:
 * the DRI "swrast" driver is dlopen()ed and asked for an OpenGL 4.5 core context the
 * same way libGL's drisw loader does (GL/internal/dri_interface.h, DRI_SWRast ext).
 *
 * The call sequence mirrors what the reference's Rust host does:
 *   Shader::from_source + Program::from_shaders   (src/renderer/shader.rs:46, program.rs:101)
 *   ComputeShader::new  -> COMPUTE_WORK_GROUP_SIZE (src/renderer/compute_shader.rs:15-26)
 *   VertexBufferObject::new + glBindBufferBase     (src/renderer/vbo.rs:32-55, main.rs:343-450,
 *                                                   octree.rs:44-100)
 *   Texture::new_2d RGBA32F + BindImageTexture     (src/renderer/texture.rs:47-75)
 *   Program::set_i32 / set_vector3_f32             (src/renderer/program.rs:35-83)
 *   ComputeShader::dispatch_compute                (src/renderer/compute_shader.rs:28-38)
 *   the presentation pass: quad program, vertex/index buffers, draw (src/main.rs:113-153, 582-600)
 *     — into an RGBA8 framebuffer object instead of a window's back buffer (there is no window)
 *
 * Nothing of the reference is copied here: the shader text is read from $REF_DIR when
 * glref_program() is called.  Build products go to oracle/_ref/ (git-ignored).
 */
#define GL_GLEXT_PROTOTYPES 0
#include <GL/glcorearb.h>
#include <GL/internal/dri_interface.h>
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

typedef void (*glproc_t)(void);
static glproc_t (*glapi_get_proc)(const char *);

#define GLFUNCS(X) \
  X(PFNGLGETSTRINGPROC, glGetString) \
  X(PFNGLGETERRORPROC, glGetError) \
  X(PFNGLCREATESHADERPROC, glCreateShader) \
  X(PFNGLSHADERSOURCEPROC, glShaderSource) \
  X(PFNGLCOMPILESHADERPROC, glCompileShader) \
  X(PFNGLGETSHADERIVPROC, glGetShaderiv) \
  X(PFNGLGETSHADERINFOLOGPROC, glGetShaderInfoLog) \
  X(PFNGLCREATEPROGRAMPROC, glCreateProgram) \
  X(PFNGLATTACHSHADERPROC, glAttachShader) \
  X(PFNGLLINKPROGRAMPROC, glLinkProgram) \
  X(PFNGLGETPROGRAMIVPROC, glGetProgramiv) \
  X(PFNGLGETPROGRAMINFOLOGPROC, glGetProgramInfoLog) \
  X(PFNGLUSEPROGRAMPROC, glUseProgram) \
  X(PFNGLDELETEPROGRAMPROC, glDeleteProgram) \
  X(PFNGLDELETESHADERPROC, glDeleteShader) \
  X(PFNGLGETUNIFORMLOCATIONPROC, glGetUniformLocation) \
  X(PFNGLPROGRAMUNIFORM1IPROC, glProgramUniform1i) \
  X(PFNGLPROGRAMUNIFORM1FPROC, glProgramUniform1f) \
  X(PFNGLPROGRAMUNIFORM3FPROC, glProgramUniform3f) \
  X(PFNGLGENBUFFERSPROC, glGenBuffers) \
  X(PFNGLDELETEBUFFERSPROC, glDeleteBuffers) \
  X(PFNGLBINDBUFFERPROC, glBindBuffer) \
  X(PFNGLBUFFERDATAPROC, glBufferData) \
  X(PFNGLBINDBUFFERBASEPROC, glBindBufferBase) \
  X(PFNGLGETBUFFERSUBDATAPROC, glGetBufferSubData) \
  X(PFNGLGENTEXTURESPROC, glGenTextures) \
  X(PFNGLDELETETEXTURESPROC, glDeleteTextures) \
  X(PFNGLACTIVETEXTUREPROC, glActiveTexture) \
  X(PFNGLBINDTEXTUREPROC, glBindTexture) \
  X(PFNGLTEXPARAMETERIPROC, glTexParameteri) \
  X(PFNGLTEXIMAGE2DPROC, glTexImage2D) \
  X(PFNGLBINDIMAGETEXTUREPROC, glBindImageTexture) \
  X(PFNGLGETTEXIMAGEPROC, glGetTexImage) \
  X(PFNGLDISPATCHCOMPUTEPROC, glDispatchCompute) \
  X(PFNGLMEMORYBARRIERPROC, glMemoryBarrier) \
  X(PFNGLFINISHPROC, glFinish) \
  X(PFNGLGETPROGRAMRESOURCEINDEXPROC, glGetProgramResourceIndex) \
  X(PFNGLGETPROGRAMRESOURCEIVPROC, glGetProgramResourceiv) \
  X(PFNGLTEXSUBIMAGE2DPROC, glTexSubImage2D) \
  X(PFNGLGENVERTEXARRAYSPROC, glGenVertexArrays) \
  X(PFNGLDELETEVERTEXARRAYSPROC, glDeleteVertexArrays) \
  X(PFNGLBINDVERTEXARRAYPROC, glBindVertexArray) \
  X(PFNGLVERTEXATTRIBPOINTERPROC, glVertexAttribPointer) \
  X(PFNGLENABLEVERTEXATTRIBARRAYPROC, glEnableVertexAttribArray) \
  X(PFNGLGENFRAMEBUFFERSPROC, glGenFramebuffers) \
  X(PFNGLDELETEFRAMEBUFFERSPROC, glDeleteFramebuffers) \
  X(PFNGLBINDFRAMEBUFFERPROC, glBindFramebuffer) \
  X(PFNGLFRAMEBUFFERTEXTURE2DPROC, glFramebufferTexture2D) \
  X(PFNGLCHECKFRAMEBUFFERSTATUSPROC, glCheckFramebufferStatus) \
  X(PFNGLVIEWPORTPROC, glViewport) \
  X(PFNGLCLEARCOLORPROC, glClearColor) \
  X(PFNGLCLEARPROC, glClear) \
  X(PFNGLDRAWELEMENTSPROC, glDrawElements) \
  X(PFNGLREADPIXELSPROC, glReadPixels) \
  X(PFNGLPIXELSTOREIPROC, glPixelStorei)

#define X(T, N) static T p_##N;
GLFUNCS(X)
#undef X

/* ---- DRI swrast loader callbacks: there is no window, so presentation is a no-op ---- */
static void ld_get_drawable_info(__DRIdrawable *d, int *x, int *y, int *w, int *h, void *p) {
  (void)d; (void)p; *x = 0; *y = 0; *w = 64; *h = 64;
}
static void ld_put_image(__DRIdrawable *d, int op, int x, int y, int w, int h, char *data, void *p) {
  (void)d; (void)op; (void)x; (void)y; (void)w; (void)h; (void)data; (void)p;
}
static void ld_get_image(__DRIdrawable *d, int x, int y, int w, int h, char *data, void *p) {
  (void)d; (void)x; (void)y; (void)p; memset(data, 0, (size_t)w * h * 4);
}
static void ld_put_image2(__DRIdrawable *d, int op, int x, int y, int w, int h, int stride, char *data, void *p) {
  (void)d; (void)op; (void)x; (void)y; (void)w; (void)h; (void)stride; (void)data; (void)p;
}
static void ld_get_image2(__DRIdrawable *d, int x, int y, int w, int h, int stride, char *data, void *p) {
  (void)d; (void)x; (void)y; (void)w; (void)p; memset(data, 0, (size_t)stride * h);
}
static const __DRIswrastLoaderExtension swrast_loader = {
  .base = { __DRI_SWRAST_LOADER, 3 },
  .getDrawableInfo = ld_get_drawable_info,
  .putImage = ld_put_image,
  .getImage = ld_get_image,
  .putImage2 = ld_put_image2,
  .getImage2 = ld_get_image2,
};
static const __DRIextension *loader_exts[] = { &swrast_loader.base, NULL };

static const __DRIcoreExtension *core;
static const __DRIswrastExtension *swrast;
static __DRIscreen *screen;
static __DRIcontext *ctx;
static __DRIdrawable *drawable;
static int g_ready;

static GLuint g_prog;
static GLuint g_tex;
static int g_tex_w, g_tex_h;
#define MAX_BUFS 64
static GLuint g_bufs[MAX_BUFS];
static int g_nbufs;

static char g_err[8192];
const char *glref_last_error(void) { return g_err; }

int glref_init(void) {
  if (g_ready) return 0;
  void *glapi = dlopen("libglapi.so.0", RTLD_NOW | RTLD_GLOBAL);
  if (!glapi) { snprintf(g_err, sizeof g_err, "llvmpipe unavailable: %s", dlerror()); return -1; }
  const char *drv = getenv("GLREF_SWRAST");
  if (!drv) drv = "/usr/lib/x86_64-linux-gnu/dri/swrast_dri.so";
  void *dri = dlopen(drv, RTLD_NOW | RTLD_GLOBAL);
  if (!dri) { snprintf(g_err, sizeof g_err, "llvmpipe unavailable: %s", dlerror()); return -1; }
  const __DRIextension **(*get_exts)(void) =
      (const __DRIextension **(*)(void))dlsym(dri, "__driDriverGetExtensions_swrast");
  if (!get_exts) { snprintf(g_err, sizeof g_err, "no __driDriverGetExtensions_swrast"); return -1; }
  const __DRIextension **exts = get_exts();
  for (int i = 0; exts[i]; i++) {
    if (!strcmp(exts[i]->name, __DRI_CORE)) core = (const __DRIcoreExtension *)exts[i];
    if (!strcmp(exts[i]->name, __DRI_SWRAST)) swrast = (const __DRIswrastExtension *)exts[i];
  }
  if (!core || !swrast || swrast->base.version < 4) {
    snprintf(g_err, sizeof g_err, "swrast driver lacks DRI_Core / DRI_SWRast v4"); return -1;
  }
  const __DRIconfig **configs = NULL;
  screen = swrast->createNewScreen2(0, loader_exts, exts, &configs, NULL);
  if (!screen || !configs || !configs[0]) { snprintf(g_err, sizeof g_err, "createNewScreen2 failed"); return -1; }
  uint32_t attribs[] = { __DRI_CTX_ATTRIB_MAJOR_VERSION, 4, __DRI_CTX_ATTRIB_MINOR_VERSION, 5 };
  unsigned err = 0;
  ctx = swrast->createContextAttribs(screen, __DRI_API_OPENGL_CORE, configs[0], NULL, 2, attribs, &err, NULL);
  if (!ctx) { snprintf(g_err, sizeof g_err, "createContextAttribs(4.5 core) failed: %u", err); return -1; }
  drawable = swrast->createNewDrawable(screen, configs[0], NULL);
  if (!drawable) { snprintf(g_err, sizeof g_err, "createNewDrawable failed"); return -1; }
  if (!core->bindContext(ctx, drawable, drawable)) { snprintf(g_err, sizeof g_err, "bindContext failed"); return -1; }
  glapi_get_proc = (glproc_t(*)(const char *))dlsym(glapi, "_glapi_get_proc_address");
  if (!glapi_get_proc) { snprintf(g_err, sizeof g_err, "no _glapi_get_proc_address"); return -1; }
#define X(T, N) p_##N = (T)glapi_get_proc(#N); if (!p_##N) { snprintf(g_err, sizeof g_err, "missing GL entry %s", #N); return -1; }
  GLFUNCS(X)
#undef X
  g_ready = 1;
  return 0;
}

const char *glref_renderer(void) { return g_ready ? (const char *)p_glGetString(GL_RENDERER) : ""; }
const char *glref_version(void) { return g_ready ? (const char *)p_glGetString(GL_VERSION) : ""; }

/* Shader::from_source + Program::from_shaders for ONE compute shader file. */
int glref_program(const char *path) {
  if (!g_ready) { snprintf(g_err, sizeof g_err, "glref_init not called"); return -1; }
  FILE *f = fopen(path, "rb");
  if (!f) { snprintf(g_err, sizeof g_err, "cannot open shader %s", path); return -1; }
  fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
  char *src = (char *)malloc((size_t)n + 1);
  if (fread(src, 1, (size_t)n, f) != (size_t)n) { fclose(f); free(src); snprintf(g_err, sizeof g_err, "short read"); return -1; }
  src[n] = 0; fclose(f);
  GLuint sh = p_glCreateShader(GL_COMPUTE_SHADER);
  const GLchar *srcs[1] = { src };
  p_glShaderSource(sh, 1, srcs, NULL);
  p_glCompileShader(sh);
  free(src);
  GLint ok = 0;
  p_glGetShaderiv(sh, GL_COMPILE_STATUS, &ok);
  if (!ok) { p_glGetShaderInfoLog(sh, sizeof g_err, NULL, g_err); return -2; }
  GLuint prog = p_glCreateProgram();
  p_glAttachShader(prog, sh);
  p_glLinkProgram(prog);
  p_glGetProgramiv(prog, GL_LINK_STATUS, &ok);
  if (!ok) { p_glGetProgramInfoLog(prog, sizeof g_err, NULL, g_err); return -3; }
  p_glDeleteShader(sh);
  if (g_prog) p_glDeleteProgram(g_prog);
  g_prog = prog;
  return 0;
}

int glref_group_size(int out[3]) {
  p_glGetProgramiv(g_prog, GL_COMPUTE_WORK_GROUP_SIZE, out);
  return (int)p_glGetError();
}

/* VertexBufferObject::new (glBufferData copies) + glBindBufferBase(SHADER_STORAGE_BUFFER, slot). */
int glref_ssbo(unsigned slot, const void *data, size_t bytes) {
  if (g_nbufs >= MAX_BUFS) { snprintf(g_err, sizeof g_err, "too many buffers"); return -1; }
  GLuint b = 0;
  p_glGenBuffers(1, &b);
  p_glBindBuffer(GL_SHADER_STORAGE_BUFFER, b);
  p_glBufferData(GL_SHADER_STORAGE_BUFFER, (GLsizeiptr)bytes, data, GL_DYNAMIC_COPY);
  p_glBindBufferBase(GL_SHADER_STORAGE_BUFFER, slot, b);
  g_bufs[g_nbufs++] = b;
  return (int)p_glGetError();
}

/* VertexBufferObject::new(.., gl::ATOMIC_COUNTER_BUFFER, ..) + gl::BindBufferBase(ATOMIC_COUNTER_BUFFER, slot)
 * (octree.rs:105-117); read it back with glref_ssbo_read(index) like any other buffer */
int glref_atomic_counter(unsigned slot, const void *data, size_t bytes) {
  if (g_nbufs >= MAX_BUFS) { snprintf(g_err, sizeof g_err, "too many buffers"); return -1; }
  GLuint b = 0;
  p_glGenBuffers(1, &b);
  p_glBindBuffer(GL_ATOMIC_COUNTER_BUFFER, b);
  p_glBufferData(GL_ATOMIC_COUNTER_BUFFER, (GLsizeiptr)bytes, data, GL_DYNAMIC_COPY);
  p_glBindBufferBase(GL_ATOMIC_COUNTER_BUFFER, slot, b);
  g_bufs[g_nbufs++] = b;
  return (int)p_glGetError();
}
int glref_buffer_count(void) { return g_nbufs; }

/* read back an SSBO bound earlier (used by test-only micro shaders) */
int glref_ssbo_read(int index, void *dst, size_t bytes) {
  if (index < 0 || index >= g_nbufs) return -1;
  p_glBindBuffer(GL_SHADER_STORAGE_BUFFER, g_bufs[index]);
  p_glGetBufferSubData(GL_SHADER_STORAGE_BUFFER, 0, (GLsizeiptr)bytes, dst);
  return (int)p_glGetError();
}

void glref_free_buffers(void) {
  if (g_nbufs) p_glDeleteBuffers(g_nbufs, g_bufs);
  g_nbufs = 0;
}

/* Texture::new_2d(TEXTURE0, 0, RGBA32F, RGBA, w, h); contents zero-filled (the reference
 * leaves them undefined) so never-written pixels are deterministic in goldens. */
int glref_image(int w, int h) {
  if (g_tex) { p_glDeleteTextures(1, &g_tex); g_tex = 0; }
  p_glGenTextures(1, &g_tex);
  p_glActiveTexture(GL_TEXTURE0);
  p_glBindTexture(GL_TEXTURE_2D, g_tex);
  p_glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_WRAP_S, GL_REPEAT);
  p_glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_WRAP_T, GL_REPEAT);
  p_glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MAG_FILTER, GL_NEAREST);
  p_glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MIN_FILTER, GL_NEAREST);
  float *zero = (float *)calloc((size_t)w * h * 4, sizeof(float));
  p_glTexImage2D(GL_TEXTURE_2D, 0, GL_RGBA32F, w, h, 0, GL_RGBA, GL_FLOAT, zero);
  free(zero);
  p_glBindImageTexture(0, g_tex, 0, GL_FALSE, 0, GL_READ_WRITE, GL_RGBA32F);
  g_tex_w = w; g_tex_h = h;
  return (int)p_glGetError();
}

int glref_image_read(float *dst) {
  p_glBindTexture(GL_TEXTURE_2D, g_tex);
  p_glGetTexImage(GL_TEXTURE_2D, 0, GL_RGBA, GL_FLOAT, dst);
  return (int)p_glGetError();
}

/* Program::set_i32 / set_f32 / set_vector3_f32: -1 location => VariableNotFound (program.rs:144-165) */
int glref_set_i32(const char *name, int v) {
  GLint loc = p_glGetUniformLocation(g_prog, name);
  if (loc < 0) { snprintf(g_err, sizeof g_err, "failed to locate uniform %s", name); return -4; }
  p_glProgramUniform1i(g_prog, loc, v);
  return 0;
}
int glref_set_f32(const char *name, float v) {
  GLint loc = p_glGetUniformLocation(g_prog, name);
  if (loc < 0) { snprintf(g_err, sizeof g_err, "failed to locate uniform %s", name); return -4; }
  p_glProgramUniform1f(g_prog, loc, v);
  return 0;
}
int glref_set_vec3f(const char *name, float x, float y, float z) {
  GLint loc = p_glGetUniformLocation(g_prog, name);
  if (loc < 0) { snprintf(g_err, sizeof g_err, "failed to locate uniform %s", name); return -4; }
  p_glProgramUniform3f(g_prog, loc, x, y, z);
  return 0;
}

/* ComputeShader::dispatch_compute(width, height, depth): floor-div by the group size, min 1
 * (compute_shader.rs:28-38).  Returns wall seconds for dispatch + glFinish, or <0. */
double glref_dispatch_compute(int width, int height, int depth) {
  int gs[3];
  p_glGetProgramiv(g_prog, GL_COMPUTE_WORK_GROUP_SIZE, gs);
  unsigned gx = (unsigned)(width / gs[0]);  if (gx < 1) gx = 1;
  unsigned gy = (unsigned)(height / gs[1]); if (gy < 1) gy = 1;
  unsigned gz = (unsigned)(depth / gs[2]);  if (gz < 1) gz = 1;
  p_glFinish();
  struct timespec t0, t1;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  p_glUseProgram(g_prog);
  p_glDispatchCompute(gx, gy, gz);
  p_glMemoryBarrier(GL_SHADER_IMAGE_ACCESS_BARRIER_BIT | GL_SHADER_STORAGE_BARRIER_BIT | GL_ATOMIC_COUNTER_BARRIER_BIT);
  p_glUseProgram(0);
  p_glFinish();
  clock_gettime(CLOCK_MONOTONIC, &t1);
  if (p_glGetError() != GL_NO_ERROR) return -1.0;
  return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}

/* layout introspection of a buffer variable, e.g. "indirect_cells[0].type" -> offset / stride */
int glref_buffer_variable(const char *name, int out_offset_stride[2]) {
  GLuint idx = p_glGetProgramResourceIndex(g_prog, GL_BUFFER_VARIABLE, name);
  if (idx == GL_INVALID_INDEX) return -1;
  GLenum props[2] = { GL_OFFSET, GL_TOP_LEVEL_ARRAY_STRIDE };
  GLint vals[2] = { -1, -1 };
  p_glGetProgramResourceiv(g_prog, GL_BUFFER_VARIABLE, idx, 2, props, 2, NULL, vals);
  out_offset_stride[0] = vals[0]; out_offset_stride[1] = vals[1];
  return 0;
}

/* Overwrites the render texture with caller data (W*H*4 floats, row 0 = bottom) — for presentation tests on
 * crafted values. */
int glref_image_write(const float *src) {
  p_glBindTexture(GL_TEXTURE_2D, g_tex);
  p_glTexSubImage2D(GL_TEXTURE_2D, 0, 0, 0, g_tex_w, g_tex_h, GL_RGBA, GL_FLOAT, src);
  return (int)p_glGetError();
}

static GLuint load_stage(GLenum kind, const char *path) {
  FILE *f = fopen(path, "rb");
  if (!f) { snprintf(g_err, sizeof g_err, "cannot open shader %s", path); return 0; }
  fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
  char *src = (char *)malloc((size_t)n + 1);
  if (fread(src, 1, (size_t)n, f) != (size_t)n) { fclose(f); free(src); snprintf(g_err, sizeof g_err, "short read"); return 0; }
  src[n] = 0; fclose(f);
  GLuint sh = p_glCreateShader(kind);
  const GLchar *srcs[1] = { src };
  p_glShaderSource(sh, 1, srcs, NULL);
  p_glCompileShader(sh);
  free(src);
  GLint ok = 0;
  p_glGetShaderiv(sh, GL_COMPILE_STATUS, &ok);
  if (!ok) { p_glGetShaderInfoLog(sh, sizeof g_err, NULL, g_err); return 0; }
  return sh;
}

/* The reference's presentation pass (main.rs:113-153 set-up, 582-600 per frame): Program::from_resources(
 * "shaders/quad") = quad.vert + quad.frag, the 4-vertex / 6-index quad, Clear, DrawElements — drawn into an
 * RGBA8 colour attachment of vw x vh pixels (the window's back buffer in the reference; glViewport(0,0,vw,vh),
 * main.rs:114), then read back with glReadPixels: out = vh*vw*4 bytes, row 0 = bottom. */
int glref_present(const char *vert_path, const char *frag_path, int vw, int vh, unsigned char *out) {
  if (!g_ready || !g_tex) { snprintf(g_err, sizeof g_err, "no context / render texture"); return -1; }
  GLuint vs = load_stage(GL_VERTEX_SHADER, vert_path); if (!vs) return -2;
  GLuint fs = load_stage(GL_FRAGMENT_SHADER, frag_path); if (!fs) return -2;
  GLuint prog = p_glCreateProgram();
  p_glAttachShader(prog, vs); p_glAttachShader(prog, fs);
  p_glLinkProgram(prog);
  GLint ok = 0;
  p_glGetProgramiv(prog, GL_LINK_STATUS, &ok);
  if (!ok) { p_glGetProgramInfoLog(prog, sizeof g_err, NULL, g_err); return -3; }
  p_glDeleteShader(vs); p_glDeleteShader(fs);
  static const GLuint indices[6] = { 0, 1, 2, 0, 1, 3 };                                   /* main.rs:121-124 */
  static const GLfloat vertices[20] = { -1.f, -1.f, 0.f, 0.f, 0.f,   1.f, 1.f, 0.f, 1.f, 1.f,   /* main.rs:141-146 */
                                        -1.f, 1.f, 0.f, 0.f, 1.f,    1.f, -1.f, 0.f, 1.f, 0.f };
  GLuint vao = 0, vbo = 0, ebo = 0, fbo = 0, color = 0;
  p_glGenVertexArrays(1, &vao); p_glBindVertexArray(vao);
  p_glGenBuffers(1, &vbo); p_glBindBuffer(GL_ARRAY_BUFFER, vbo);
  p_glBufferData(GL_ARRAY_BUFFER, sizeof vertices, vertices, GL_STATIC_DRAW);
  p_glVertexAttribPointer(0, 3, GL_FLOAT, GL_FALSE, 5 * sizeof(GLfloat), (const void *)0);                       /* vao.rs:45-58 */
  p_glEnableVertexAttribArray(0);
  p_glVertexAttribPointer(1, 2, GL_FLOAT, GL_FALSE, 5 * sizeof(GLfloat), (const void *)(3 * sizeof(GLfloat)));
  p_glEnableVertexAttribArray(1);
  p_glGenBuffers(1, &ebo); p_glBindBuffer(GL_ELEMENT_ARRAY_BUFFER, ebo);
  p_glBufferData(GL_ELEMENT_ARRAY_BUFFER, sizeof indices, indices, GL_STATIC_DRAW);
  p_glGenTextures(1, &color);
  p_glActiveTexture(GL_TEXTURE1);
  p_glBindTexture(GL_TEXTURE_2D, color);
  p_glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MIN_FILTER, GL_NEAREST);
  p_glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MAG_FILTER, GL_NEAREST);
  p_glTexImage2D(GL_TEXTURE_2D, 0, GL_RGBA8, vw, vh, 0, GL_RGBA, GL_UNSIGNED_BYTE, NULL);
  p_glGenFramebuffers(1, &fbo); p_glBindFramebuffer(GL_FRAMEBUFFER, fbo);
  p_glFramebufferTexture2D(GL_FRAMEBUFFER, GL_COLOR_ATTACHMENT0, GL_TEXTURE_2D, color, 0);
  int rc = 0;
  if (p_glCheckFramebufferStatus(GL_FRAMEBUFFER) != GL_FRAMEBUFFER_COMPLETE) { snprintf(g_err, sizeof g_err, "framebuffer incomplete"); rc = -4; }
  if (!rc) {
    p_glActiveTexture(GL_TEXTURE0);                      /* the render texture stays bound to unit 0 (texture.rs:55-56); */
    p_glBindTexture(GL_TEXTURE_2D, g_tex);               /* `ourTexture` is never set, i.e. unit 0 (quad.frag:6) */
    p_glViewport(0, 0, vw, vh);
    p_glClearColor(0.f, 0.f, 0.f, 1.f);                  /* main.rs:115 */
    p_glUseProgram(prog);
    p_glClear(GL_COLOR_BUFFER_BIT);
    p_glDrawElements(GL_TRIANGLES, 6, GL_UNSIGNED_INT, (const void *)0);
    p_glUseProgram(0);
    p_glFinish();
    p_glPixelStorei(GL_PACK_ALIGNMENT, 1);
    p_glReadPixels(0, 0, vw, vh, GL_RGBA, GL_UNSIGNED_BYTE, out);
    if (p_glGetError() != GL_NO_ERROR) { snprintf(g_err, sizeof g_err, "GL error in the presentation pass"); rc = -5; }
  }
  p_glBindFramebuffer(GL_FRAMEBUFFER, 0);
  p_glBindVertexArray(0);
  p_glDeleteFramebuffers(1, &fbo); p_glDeleteTextures(1, &color);
  p_glDeleteBuffers(1, &vbo); p_glDeleteBuffers(1, &ebo); p_glDeleteVertexArrays(1, &vao);
  p_glDeleteProgram(prog);
  p_glActiveTexture(GL_TEXTURE0);
  return rc;
}
