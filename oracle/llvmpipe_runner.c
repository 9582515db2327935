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
 *       [--screen-vs V.glsl --screen-fs F.glsl --acc-out ACC.bin [--recreate-at F]]
 * U.bin: 25 little-endian f32: inverse_view_projection[16] (memory order, sent
 * untransposed), eye[3], scene_min[3], scene_extent[3].
 * OUT.bin: for each frame f in [F0,F1): W*H*4 f32, row 0 = bottom row (GL).
 * With --screen-vs/--screen-fs/--acc-out every frame is followed by the
 * reference's accumulate pass (src/renderer/renderer.py:249-288: the screen
 * shader drawn over a quad into an RGBA32F framebuffer texture with "frames" =
 * frames accumulated so far, alpha blending on, then glCopyTexSubImage2D into
 * the previous-frame texture), and ACC.bin receives the framebuffer texture
 * after every frame, same layout as OUT.bin.  --recreate-at F: before frame F
 * the three textures are deleted and created again at the same size and the
 * framebuffer re-attached, as Renderer.update_size does (renderer.py:347-382);
 * the count of accumulated frames goes on, as the reference's does.
 * stdout: one JSON line with per-frame milliseconds (glFinish-bounded).
 *
 * Present mode (no compute dispatch):
 *   llvmpipe_runner --present-vs V.glsl --present-fs F.glsl --present-in IMG.bin
 *       --present-out OUT.bin --w W --h H
 * IMG.bin (W*H*4 f32, row 0 = bottom) is put into an RGBA32F texture made as
 * the reference makes its accumulation texture (renderer.py:399-433) with
 * glTexSubImage2D -- what a host does with an accumulation image it got from
 * elsewhere -- and the reference's present pass (renderer.py:290-300: the
 * upscale program drawn over the screen quad, blending on as everywhere) is
 * drawn into an RGBA8 framebuffer of the same size, which stands for the
 * window's default framebuffer.  OUT.bin receives its W*H*4 bytes, row 0 =
 * bottom.
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
    const char *screen_vs = NULL, *screen_fs = NULL, *acc_out = NULL;
    const char *present_vs = NULL, *present_fs = NULL, *present_in = NULL, *present_out = NULL;
    const char *buf_path[16] = {0};
    int W = 640, H = 360, f0 = 0, f1 = 1, reps = 1, recreate_at = -1;
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
        else if (!strcmp(argv[i], "--screen-vs")) screen_vs = argv[++i];
        else if (!strcmp(argv[i], "--screen-fs")) screen_fs = argv[++i];
        else if (!strcmp(argv[i], "--acc-out")) acc_out = argv[++i];
        else if (!strcmp(argv[i], "--recreate-at")) recreate_at = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--present-vs")) present_vs = argv[++i];
        else if (!strcmp(argv[i], "--present-fs")) present_fs = argv[++i];
        else if (!strcmp(argv[i], "--present-in")) present_in = argv[++i];
        else if (!strcmp(argv[i], "--present-out")) present_out = argv[++i];
        else if (!strcmp(argv[i], "--spheres")) buf_path[5] = argv[++i];
        else if (!strcmp(argv[i], "--planes")) buf_path[6] = argv[++i];
        else if (!strcmp(argv[i], "--boxes")) buf_path[7] = argv[++i];
        else if (!strcmp(argv[i], "--mats")) buf_path[9] = argv[++i];
        else if (!strcmp(argv[i], "--bvh")) buf_path[11] = argv[++i];
        else if (!strcmp(argv[i], "--tris")) buf_path[12] = argv[++i];
        else { fprintf(stderr, "unknown arg %s\n", argv[i]); return 2; }
    }
    const int present = present_vs || present_fs || present_in || present_out;
    if (present && !(present_vs && present_fs && present_in && present_out))
        die("--present-vs, --present-fs, --present-in and --present-out go together");
    if (!present && (!shader || !uni || !out)) die("need --shader --uniforms --out");
    int accumulate = screen_vs || screen_fs || acc_out;
    if (accumulate && !(screen_vs && screen_fs && acc_out))
        die("--screen-vs, --screen-fs and --acc-out go together");
    if (accumulate) reps = 1;

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
    GLF(PFNGLDELETETEXTURESPROC, glDeleteTextures);
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
    GLF(PFNGLGENVERTEXARRAYSPROC, glGenVertexArrays);
    GLF(PFNGLBINDVERTEXARRAYPROC, glBindVertexArray);
    GLF(PFNGLENABLEVERTEXATTRIBARRAYPROC, glEnableVertexAttribArray);
    GLF(PFNGLVERTEXATTRIBPOINTERPROC, glVertexAttribPointer);
    GLF(PFNGLGENFRAMEBUFFERSPROC, glGenFramebuffers);
    GLF(PFNGLBINDFRAMEBUFFERPROC, glBindFramebuffer);
    GLF(PFNGLFRAMEBUFFERTEXTURE2DPROC, glFramebufferTexture2D);
    GLF(PFNGLCHECKFRAMEBUFFERSTATUSPROC, glCheckFramebufferStatus);
    GLF(PFNGLDRAWARRAYSPROC, glDrawArrays);
    GLF(PFNGLACTIVETEXTUREPROC, glActiveTexture);
    GLF(PFNGLTEXPARAMETERIPROC, glTexParameteri);
    GLF(PFNGLUNIFORM1IPROC, glUniform1i);
    GLF(PFNGLCOPYTEXSUBIMAGE2DPROC, glCopyTexSubImage2D);
    GLF(PFNGLVIEWPORTPROC, glViewport);
    GLF(PFNGLENABLEPROC, glEnable);
    GLF(PFNGLBLENDFUNCPROC, glBlendFunc);
    GLF(PFNGLTEXSUBIMAGE2DPROC, glTexSubImage2D);
    GLF(PFNGLCLEARCOLORPROC, glClearColor);
    GLF(PFNGLCLEARPROC, glClear);

    fprintf(stderr, "GL_RENDERER: %s\nGL_VERSION: %s\n", (const char *)glGetString(GL_RENDERER),
            (const char *)glGetString(GL_VERSION));

    if (present) {
        const char *path[2] = {present_vs, present_fs};
        const GLenum kind[2] = {GL_VERTEX_SHADER, GL_FRAGMENT_SHADER};
        GLuint pprog = glCreateProgram();
        GLint pok = 0;
        for (int i = 0; i < 2; i++) {
            size_t slen;
            char *src = read_file(path[i], &slen);
            GLuint sh = glCreateShader(kind[i]);
            const char *srcs[1] = {src};
            glShaderSource(sh, 1, srcs, NULL);
            glCompileShader(sh);
            glGetShaderiv(sh, GL_COMPILE_STATUS, &pok);
            if (!pok) {
                char log[8192];
                glGetShaderInfoLog(sh, sizeof log, NULL, log);
                fprintf(stderr, "compile of %s failed:\n%s\n", path[i], log);
                return 4;
            }
            glAttachShader(pprog, sh);
            free(src);
        }
        glLinkProgram(pprog);
        glGetProgramiv(pprog, GL_LINK_STATUS, &pok);
        if (!pok) die("present program does not link");
        size_t nin;
        float *src_img = read_file(present_in, &nin);
        if (nin != (size_t)W * H * 16) die("--present-in must hold W*H*4 f32");
        GLuint ptex[2];
        glGenTextures(2, ptex);
        for (int i = 0; i < 2; i++) {
            glBindTexture(GL_TEXTURE_2D, ptex[i]);
            glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_WRAP_S, GL_CLAMP_TO_EDGE);
            glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_WRAP_T, GL_CLAMP_TO_EDGE);
            glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MAG_FILTER, GL_NEAREST);
            glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MIN_FILTER, GL_NEAREST);
            glTexStorage2D(GL_TEXTURE_2D, 1, i == 0 ? GL_RGBA32F : GL_RGBA8, W, H);
        }
        glBindTexture(GL_TEXTURE_2D, ptex[0]);
        glTexSubImage2D(GL_TEXTURE_2D, 0, 0, 0, W, H, GL_RGBA, GL_FLOAT, src_img);
        /* quad of reference src/utils/opengl_utils.py:8-57: x y z u v per vertex */
        static const float pquad[20] = {-1, 1, 0, 0, 1, -1, -1, 0, 0, 0, 1, 1, 0, 1, 1, 1, -1, 0, 1, 0};
        GLuint pvao, pvbo, pfbo;
        glGenVertexArrays(1, &pvao);
        glGenBuffers(1, &pvbo);
        glBindBuffer(GL_ARRAY_BUFFER, pvbo);
        glBufferData(GL_ARRAY_BUFFER, sizeof pquad, pquad, GL_STATIC_DRAW);
        glBindVertexArray(pvao);
        glEnableVertexAttribArray(0);
        glVertexAttribPointer(0, 3, GL_FLOAT, GL_FALSE, 5 * sizeof(float), (void *)0);
        glEnableVertexAttribArray(1);
        glVertexAttribPointer(1, 2, GL_FLOAT, GL_FALSE, 5 * sizeof(float), (void *)(3 * sizeof(float)));
        glGenFramebuffers(1, &pfbo);
        glBindFramebuffer(GL_FRAMEBUFFER, pfbo);
        glFramebufferTexture2D(GL_FRAMEBUFFER, GL_COLOR_ATTACHMENT0, GL_TEXTURE_2D, ptex[1], 0);
        if (glCheckFramebufferStatus(GL_FRAMEBUFFER) != GL_FRAMEBUFFER_COMPLETE) die("framebuffer incomplete");
        glClearColor(1.0f, 1.0f, 1.0f, 1.0f); /* renderer.py:85 */
        glClear(GL_COLOR_BUFFER_BIT);
        glEnable(GL_BLEND); /* renderer.py:87-89 */
        glBlendFunc(GL_SRC_ALPHA, GL_ONE_MINUS_SRC_ALPHA);
        glUseProgram(pprog); /* renderer.py:290-296 */
        glActiveTexture(GL_TEXTURE0);
        glBindTexture(GL_TEXTURE_2D, ptex[0]);
        glViewport(0, 0, W, H);
        glDrawArrays(GL_TRIANGLE_STRIP, 0, 4);
        glFinish();
        unsigned char *bytes = malloc((size_t)W * H * 4);
        glBindTexture(GL_TEXTURE_2D, ptex[1]);
        glGetTexImage(GL_TEXTURE_2D, 0, GL_RGBA, GL_UNSIGNED_BYTE, bytes);
        FILE *fp = fopen(present_out, "wb");
        if (!fp) die("cannot open --present-out");
        if (fwrite(bytes, 4, (size_t)W * H, fp) != (size_t)W * H) die("short write");
        fclose(fp);
        printf("{\"w\": %d, \"h\": %d, \"present\": true, \"gl_error\": %u}\n", W, H, glGetError());
        return 0;
    }

    /* ---- programs ---- */
    const char *stage_path[3] = {shader, screen_vs, screen_fs};
    const GLenum stage_kind[3] = {GL_COMPUTE_SHADER, GL_VERTEX_SHADER, GL_FRAGMENT_SHADER};
    GLuint stage[3] = {0, 0, 0};
    GLint ok = 0;
    for (int i = 0; i < (accumulate ? 3 : 1); i++) {
        size_t slen;
        char *src = read_file(stage_path[i], &slen);
        stage[i] = glCreateShader(stage_kind[i]);
        const char *srcs[1] = {src};
        glShaderSource(stage[i], 1, srcs, NULL);
        glCompileShader(stage[i]);
        glGetShaderiv(stage[i], GL_COMPILE_STATUS, &ok);
        if (!ok) {
            char log[8192];
            glGetShaderInfoLog(stage[i], sizeof log, NULL, log);
            fprintf(stderr, "compile of %s failed:\n%s\n", stage_path[i], log);
            return 4;
        }
        free(src);
    }
    GLuint prog = glCreateProgram(), screen_prog = 0;
    glAttachShader(prog, stage[0]);
    glLinkProgram(prog);
    glGetProgramiv(prog, GL_LINK_STATUS, &ok);
    if (ok && accumulate) {
        screen_prog = glCreateProgram();
        glAttachShader(screen_prog, stage[1]);
        glAttachShader(screen_prog, stage[2]);
        glLinkProgram(screen_prog);
        glGetProgramiv(screen_prog, GL_LINK_STATUS, &ok);
    }
    if (!ok) {
        char log[8192];
        glGetProgramInfoLog(screen_prog ? screen_prog : prog, sizeof log, NULL, log);
        fprintf(stderr, "link failed:\n%s\n", log);
        return 4;
    }
    glUseProgram(prog);

    /* ---- image + buffers ---- */
    /* three RGBA32F textures as reference renderer.py:95-96,153 and :399-433 makes them:
       tex[0] = compute output, tex[1] = previous accumulation, tex[2] = framebuffer colour */
    GLuint tex3[3];
    glGenTextures(3, tex3);
    for (int i = 0; i < 3; i++) {
        glBindTexture(GL_TEXTURE_2D, tex3[i]);
        glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_WRAP_S, GL_CLAMP_TO_EDGE);
        glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_WRAP_T, GL_CLAMP_TO_EDGE);
        glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MAG_FILTER, GL_NEAREST);
        glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MIN_FILTER, GL_NEAREST);
        glTexStorage2D(GL_TEXTURE_2D, 1, GL_RGBA32F, W, H);
    }
    GLuint tex = tex3[0], fbo = 0, vao = 0;
    glBindTexture(GL_TEXTURE_2D, tex);
    glBindImageTexture(0, tex, 0, GL_FALSE, 0, GL_READ_WRITE, GL_RGBA32F);
    if (accumulate) {
        /* quad of reference src/utils/opengl_utils.py:8-57: x y z u v per vertex */
        static const float quad[20] = {-1, 1, 0, 0, 1, -1, -1, 0, 0, 0, 1, 1, 0, 1, 1, 1, -1, 0, 1, 0};
        GLuint vbo;
        glGenVertexArrays(1, &vao);
        glGenBuffers(1, &vbo);
        glBindBuffer(GL_ARRAY_BUFFER, vbo);
        glBufferData(GL_ARRAY_BUFFER, sizeof quad, quad, GL_STATIC_DRAW);
        glBindVertexArray(vao);
        glEnableVertexAttribArray(0);
        glVertexAttribPointer(0, 3, GL_FLOAT, GL_FALSE, 5 * sizeof(float), (void *)0);
        glEnableVertexAttribArray(1);
        glVertexAttribPointer(1, 2, GL_FLOAT, GL_FALSE, 5 * sizeof(float), (void *)(3 * sizeof(float)));
        glGenFramebuffers(1, &fbo);
        glBindFramebuffer(GL_FRAMEBUFFER, fbo);
        glFramebufferTexture2D(GL_FRAMEBUFFER, GL_COLOR_ATTACHMENT0, GL_TEXTURE_2D, tex3[2], 0);
        if (glCheckFramebufferStatus(GL_FRAMEBUFFER) != GL_FRAMEBUFFER_COMPLETE) die("framebuffer incomplete");
        glViewport(0, 0, W, H);
        glEnable(GL_BLEND); /* renderer.py:87-89 */
        glBlendFunc(GL_SRC_ALPHA, GL_ONE_MINUS_SRC_ALPHA);
        glActiveTexture(GL_TEXTURE1);
        glBindTexture(GL_TEXTURE_2D, tex3[1]);
        glActiveTexture(GL_TEXTURE0);
        glBindTexture(GL_TEXTURE_2D, tex);
    }
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
    FILE *fa = accumulate ? fopen(acc_out, "wb") : NULL;
    if (accumulate && !fa) die("cannot open --acc-out");
    int accumulated = 0;
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
        if (accumulate && f == recreate_at) {
            glDeleteTextures(3, tex3);
            glGenTextures(3, tex3);
            const GLenum unit[3] = {GL_TEXTURE0, GL_TEXTURE1, GL_TEXTURE0}; /* renderer.py:367-369 */
            for (int i = 0; i < 3; i++) {
                glActiveTexture(unit[i]);
                glBindTexture(GL_TEXTURE_2D, tex3[i]);
                glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_WRAP_S, GL_CLAMP_TO_EDGE);
                glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_WRAP_T, GL_CLAMP_TO_EDGE);
                glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MAG_FILTER, GL_NEAREST);
                glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MIN_FILTER, GL_NEAREST);
                glTexStorage2D(GL_TEXTURE_2D, 1, GL_RGBA32F, W, H);
            }
            tex = tex3[0];
            glBindFramebuffer(GL_FRAMEBUFFER, fbo);
            glFramebufferTexture2D(GL_FRAMEBUFFER, GL_COLOR_ATTACHMENT0, GL_TEXTURE_2D, tex3[2], 0);
            if (glCheckFramebufferStatus(GL_FRAMEBUFFER) != GL_FRAMEBUFFER_COMPLETE) die("framebuffer incomplete");
            glBindFramebuffer(GL_FRAMEBUFFER, 0);
            /* the next render() binds the compute image again (renderer.py:176-180); unit 1 keeps the
               new previous-frame texture bound above */
            glActiveTexture(GL_TEXTURE0);
            glBindTexture(GL_TEXTURE_2D, tex);
            glBindImageTexture(0, tex, 0, GL_FALSE, 0, GL_READ_WRITE, GL_RGBA32F);
        }
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
        glBindTexture(GL_TEXTURE_2D, tex);
        glGetTexImage(GL_TEXTURE_2D, 0, GL_RGBA, GL_FLOAT, img);
        if (fwrite(img, sizeof(float) * 4, npx, fo) != npx) die("short write");
        if (accumulate) {
            glUseProgram(screen_prog);
            glUniform1i(glGetUniformLocation(screen_prog, "tex"), 0);
            glUniform1i(glGetUniformLocation(screen_prog, "old_tex"), 1);
            glUniform1f(glGetUniformLocation(screen_prog, "frames"), (float)accumulated);
            glUniform1f(glGetUniformLocation(screen_prog, "denoise"), 0.0f);
            glBindFramebuffer(GL_FRAMEBUFFER, fbo);
            glBindVertexArray(vao);
            glDrawArrays(GL_TRIANGLE_STRIP, 0, 4);
            accumulated++;
            glBindTexture(GL_TEXTURE_2D, tex3[1]);
            glCopyTexSubImage2D(GL_TEXTURE_2D, 0, 0, 0, 0, 0, W, H);
            glFinish();
            glBindTexture(GL_TEXTURE_2D, tex3[2]);
            glGetTexImage(GL_TEXTURE_2D, 0, GL_RGBA, GL_FLOAT, img);
            if (fwrite(img, sizeof(float) * 4, npx, fa) != npx) die("short write");
            glBindTexture(GL_TEXTURE_2D, tex);
            glUseProgram(prog);
        }
        printf("%s%.3f", first ? "" : ", ", best);
        first = 0;
    }
    printf("], \"gl_error\": %u}\n", glGetError());
    fclose(fo);
    if (fa) fclose(fa);
    return 0;
}
