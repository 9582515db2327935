/*
 * llvmpipe_runner — TEST INFRASTRUCTURE, never part of the shipped path.
 *
 * Runs a GLSL 4.30 compute shader (the reference's shaders/compute.glsl, read
 * from the path given on the command line; nothing of it is compiled in or
 * copied) headless on Mesa's llvmpipe software rasteriser, with the same SSBO
 * bindings and uniforms the reference renderer sets before its
 * glDispatchCompute (reference src/renderer/renderer.py:176-246), and dumps
 * the RGBA32F image the shader stores (shaders/compute.glsl:673).
 *
 * No X server / EGL / OSMesa is needed: the swrast DRI driver is loaded
 * directly and a surfaceless OpenGL 4.3 core context is made current.
 *
 * Build (oracle/Makefile): gcc -O2 llvmpipe_runner.c -ldl -o _ref/llvmpipe_runner
 * Usage:
 *   llvmpipe_runner --shader S.glsl --w W --h H --bounces B --frames F0 F1
 *       --uniforms U.bin [--bvh f] [--tris f] [--spheres f] [--mats f]
 *       [--planes f] [--boxes f] --out OUT.bin [--reps R]
 * U.bin: 25 little-endian f32: inverse_view_projection[16] (memory order, sent
 * untransposed), eye[3], scene_min[3], scene_extent[3].
 * OUT.bin: for each frame f in [F0,F1): W*H*4 f32, row 0 = bottom row (GL).
 * stdout: one JSON line with per-frame milliseconds (glFinish-bounded).
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <GL/glcorearb.h>
#include <GL/internal/dri_interface.h>

static void die(const char *m) {
    fprintf(stderr, "llvmpipe_runner: %s\n", m);
    exit(2);
}

/* ---- swrast loader callbacks (never used: there is no drawable) ---- */
static void cb_info(__DRIdrawable *d, int *x, int *y, int *w, int *h, void *p) {
    (void)d; (void)p; *x = *y = 0; *w = *h = 1;
}
static void cb_put(__DRIdrawable *d, int op, int x, int y, int w, int h, char *data, void *p) {
    (void)d; (void)op; (void)x; (void)y; (void)w; (void)h; (void)data; (void)p;
}
static void cb_get(__DRIdrawable *d, int x, int y, int w, int h, char *data, void *p) {
    (void)d; (void)x; (void)y; (void)w; (void)h; (void)data; (void)p;
}

static void *(*get_proc)(const char *);
#define GLF(type, name) type name = (type)get_proc(#name); if (!name) die("missing " #name)

static void *read_file(const char *path, size_t *n) {
    FILE *f = fopen(path, "rb");
    if (!f) { fprintf(stderr, "cannot open %s\n", path); exit(2); }
    fseek(f, 0, SEEK_END);
    long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    char *buf = malloc((size_t)sz + 1);
    if (fread(buf, 1, (size_t)sz, f) != (size_t)sz) die("short read");
    buf[sz] = 0;
    fclose(f);
    *n = (size_t)sz;
    return buf;
}

static double now_ms(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

int main(int argc, char **argv) {
    const char *shader = NULL, *uni = NULL, *out = NULL;
    const char *buf_path[16] = {0};
    int W = 640, H = 360, f0 = 0, f1 = 1, reps = 1;
    float bounces = 2.0f;
    for (int i = 1; i < argc; i++) {
        if (!strcmp(argv[i], "--shader")) shader = argv[++i];
        else if (!strcmp(argv[i], "--uniforms")) uni = argv[++i];
        else if (!strcmp(argv[i], "--out")) out = argv[++i];
        else if (!strcmp(argv[i], "--w")) W = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--h")) H = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--bounces")) bounces = (float)atof(argv[++i]);
        else if (!strcmp(argv[i], "--frames")) { f0 = atoi(argv[++i]); f1 = atoi(argv[++i]); }
        else if (!strcmp(argv[i], "--reps")) reps = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--spheres")) buf_path[5] = argv[++i];
        else if (!strcmp(argv[i], "--planes")) buf_path[6] = argv[++i];
        else if (!strcmp(argv[i], "--boxes")) buf_path[7] = argv[++i];
        else if (!strcmp(argv[i], "--mats")) buf_path[9] = argv[++i];
        else if (!strcmp(argv[i], "--bvh")) buf_path[11] = argv[++i];
        else if (!strcmp(argv[i], "--tris")) buf_path[12] = argv[++i];
        else { fprintf(stderr, "unknown arg %s\n", argv[i]); return 2; }
    }
    if (!shader || !uni || !out) die("need --shader --uniforms --out");

    /* ---- bring up llvmpipe through the swrast DRI driver ---- */
    const char *drv = getenv("SWRAST_DRI");
    if (!drv) drv = "/usr/lib/x86_64-linux-gnu/dri/swrast_dri.so";
    void *h = dlopen(drv, RTLD_NOW | RTLD_GLOBAL);
    if (!h) { fprintf(stderr, "dlopen %s: %s\n", drv, dlerror()); return 3; }
    const __DRIextension **(*getext)(void) =
        (const __DRIextension **(*)(void))dlsym(h, "__driDriverGetExtensions_swrast");
    if (!getext) die("no __driDriverGetExtensions_swrast");
    const __DRIextension **exts = getext();
    const __DRIcoreExtension *core = NULL;
    const __DRIswrastExtension *sw = NULL;
    for (int i = 0; exts[i]; i++) {
        if (!strcmp(exts[i]->name, __DRI_CORE)) core = (const __DRIcoreExtension *)exts[i];
        if (!strcmp(exts[i]->name, __DRI_SWRAST)) sw = (const __DRIswrastExtension *)exts[i];
    }
    if (!core || !sw || sw->base.version < 4) die("swrast driver lacks DRI_Core/DRI_SWRast v4");
    static __DRIswrastLoaderExtension loader;
    memset(&loader, 0, sizeof loader);
    loader.base.name = __DRI_SWRAST_LOADER;
    loader.base.version = 1;
    loader.getDrawableInfo = cb_info;
    loader.putImage = cb_put;
    loader.getImage = cb_get;
    const __DRIextension *loader_exts[] = {&loader.base, NULL};
    const __DRIconfig **configs = NULL;
    __DRIscreen *scr = sw->createNewScreen2(0, loader_exts, exts, &configs, NULL);
    if (!scr) die("createNewScreen2 failed");
    uint32_t attrs[] = {__DRI_CTX_ATTRIB_MAJOR_VERSION, 4, __DRI_CTX_ATTRIB_MINOR_VERSION, 3};
    unsigned err = 0;
    __DRIcontext *ctx =
        sw->createContextAttribs(scr, __DRI_API_OPENGL_CORE, NULL, NULL, 2, attrs, &err, NULL);
    if (!ctx) { fprintf(stderr, "createContextAttribs failed err=%u\n", err); return 3; }
    if (!core->bindContext(ctx, NULL, NULL)) die("bindContext failed");
    void *ga = dlopen("libglapi.so.0", RTLD_NOW | RTLD_GLOBAL);
    if (!ga) die("dlopen libglapi.so.0 failed");
    get_proc = (void *(*)(const char *))dlsym(ga, "_glapi_get_proc_address");
    if (!get_proc) die("no _glapi_get_proc_address");

    GLF(PFNGLGETSTRINGPROC, glGetString);
    GLF(PFNGLCREATESHADERPROC, glCreateShader);
    GLF(PFNGLSHADERSOURCEPROC, glShaderSource);
    GLF(PFNGLCOMPILESHADERPROC, glCompileShader);
    GLF(PFNGLGETSHADERIVPROC, glGetShaderiv);
    GLF(PFNGLGETSHADERINFOLOGPROC, glGetShaderInfoLog);
    GLF(PFNGLCREATEPROGRAMPROC, glCreateProgram);
    GLF(PFNGLATTACHSHADERPROC, glAttachShader);
    GLF(PFNGLLINKPROGRAMPROC, glLinkProgram);
    GLF(PFNGLGETPROGRAMIVPROC, glGetProgramiv);
    GLF(PFNGLGETPROGRAMINFOLOGPROC, glGetProgramInfoLog);
    GLF(PFNGLUSEPROGRAMPROC, glUseProgram);
    GLF(PFNGLGENTEXTURESPROC, glGenTextures);
    GLF(PFNGLBINDTEXTUREPROC, glBindTexture);
    GLF(PFNGLTEXSTORAGE2DPROC, glTexStorage2D);
    GLF(PFNGLBINDIMAGETEXTUREPROC, glBindImageTexture);
    GLF(PFNGLGENBUFFERSPROC, glGenBuffers);
    GLF(PFNGLBINDBUFFERPROC, glBindBuffer);
    GLF(PFNGLBUFFERDATAPROC, glBufferData);
    GLF(PFNGLBINDBUFFERBASEPROC, glBindBufferBase);
    GLF(PFNGLGETUNIFORMLOCATIONPROC, glGetUniformLocation);
    GLF(PFNGLUNIFORMMATRIX4FVPROC, glUniformMatrix4fv);
    GLF(PFNGLUNIFORM3FPROC, glUniform3f);
    GLF(PFNGLUNIFORM1FPROC, glUniform1f);
    GLF(PFNGLDISPATCHCOMPUTEPROC, glDispatchCompute);
    GLF(PFNGLMEMORYBARRIERPROC, glMemoryBarrier);
    GLF(PFNGLFINISHPROC, glFinish);
    GLF(PFNGLGETTEXIMAGEPROC, glGetTexImage);
    GLF(PFNGLGETERRORPROC, glGetError);
    GLF(PFNGLCLEARTEXIMAGEPROC, glClearTexImage);

    fprintf(stderr, "GL_RENDERER: %s\nGL_VERSION: %s\n", (const char *)glGetString(GL_RENDERER),
            (const char *)glGetString(GL_VERSION));

    /* ---- program ---- */
    size_t slen;
    char *src = read_file(shader, &slen);
    GLuint sh = glCreateShader(GL_COMPUTE_SHADER);
    const char *srcs[1] = {src};
    glShaderSource(sh, 1, srcs, NULL);
    glCompileShader(sh);
    GLint ok = 0;
    glGetShaderiv(sh, GL_COMPILE_STATUS, &ok);
    if (!ok) {
        char log[8192];
        glGetShaderInfoLog(sh, sizeof log, NULL, log);
        fprintf(stderr, "compile failed:\n%s\n", log);
        return 4;
    }
    GLuint prog = glCreateProgram();
    glAttachShader(prog, sh);
    glLinkProgram(prog);
    glGetProgramiv(prog, GL_LINK_STATUS, &ok);
    if (!ok) {
        char log[8192];
        glGetProgramInfoLog(prog, sizeof log, NULL, log);
        fprintf(stderr, "link failed:\n%s\n", log);
        return 4;
    }
    glUseProgram(prog);

    /* ---- image + buffers ---- */
    GLuint tex;
    glGenTextures(1, &tex);
    glBindTexture(GL_TEXTURE_2D, tex);
    glTexStorage2D(GL_TEXTURE_2D, 1, GL_RGBA32F, W, H);
    glBindImageTexture(0, tex, 0, GL_FALSE, 0, GL_READ_WRITE, GL_RGBA32F);
    for (int b = 0; b < 16; b++) {
        if (!buf_path[b]) continue;
        size_t n;
        void *data = read_file(buf_path[b], &n);
        GLuint bo;
        glGenBuffers(1, &bo);
        glBindBuffer(GL_SHADER_STORAGE_BUFFER, bo);
        glBufferData(GL_SHADER_STORAGE_BUFFER, (GLsizeiptr)n, data, GL_STATIC_DRAW);
        glBindBufferBase(GL_SHADER_STORAGE_BUFFER, (GLuint)b, bo);
        glBindBuffer(GL_SHADER_STORAGE_BUFFER, 0);
        free(data);
    }
    size_t un;
    float *u = read_file(uni, &un);
    if (un != 25 * sizeof(float)) die("uniform file must hold 25 f32");
    glUniformMatrix4fv(glGetUniformLocation(prog, "inverse_view_projection"), 1, GL_FALSE, u);
    glUniform3f(glGetUniformLocation(prog, "eye"), u[16], u[17], u[18]);
    glUniform3f(glGetUniformLocation(prog, "scene_min"), u[19], u[20], u[21]);
    glUniform3f(glGetUniformLocation(prog, "scene_extent"), u[22], u[23], u[24]);
    glUniform1f(glGetUniformLocation(prog, "bounces"), bounces);
    GLint loc_time = glGetUniformLocation(prog, "time");

    FILE *fo = fopen(out, "wb");
    if (!fo) die("cannot open --out");
    size_t npx = (size_t)W * H;
    float *img = malloc(npx * 4 * sizeof(float));
    printf("{\"w\": %d, \"h\": %d, \"bounces\": %g, \"ms\": [", W, H, bounces);
    /* one untimed dispatch first: the driver JIT-compiles on first use */
    glUniform1f(loc_time, (float)f0);
    glDispatchCompute((GLuint)(W / 8), (GLuint)(H / 4), 1);
    glMemoryBarrier(GL_SHADER_IMAGE_ACCESS_BARRIER_BIT);
    glFinish();
    int first = 1;
    for (int f = f0; f < f1; f++) {
        double best = 1e300;
        for (int r = 0; r < reps; r++) {
            const float zero[4] = {0, 0, 0, 0};
            glClearTexImage(tex, 0, GL_RGBA, GL_FLOAT, zero);
            glUniform1f(loc_time, (float)f);
            glFinish();
            double t0 = now_ms();
            /* same group counts as reference renderer.py:241-245 */
            glDispatchCompute((GLuint)(W / 8), (GLuint)(H / 4), 1);
            glMemoryBarrier(GL_SHADER_IMAGE_ACCESS_BARRIER_BIT);
            glFinish();
            double dt = now_ms() - t0;
            if (dt < best) best = dt;
        }
        glGetTexImage(GL_TEXTURE_2D, 0, GL_RGBA, GL_FLOAT, img);
        if (fwrite(img, sizeof(float) * 4, npx, fo) != npx) die("short write");
        printf("%s%.3f", first ? "" : ", ", best);
        first = 0;
    }
    printf("], \"gl_error\": %u}\n", glGetError());
    fclose(fo);
    return 0;
}
