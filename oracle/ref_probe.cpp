// ref_probe.cpp -- harness around the REFERENCE's own host code, compiled where it lies.
//
// TEST INFRASTRUCTURE ONLY (see oracle/pt_oracle.h).  This file contains no reference code:
// it #includes the reference headers from /root/reference/src and is linked with the
// reference translation units scene.cpp, utilities.cpp, image.cpp and
// stb_image/stb_image_write.c, all compiled from /root/reference by oracle/Makefile into
// oracle/_ref/ (git-ignored).  Those four are the only reference TUs that build in this
// image without writing stand-in headers: <cuda_runtime.h> comes from the CUDA header set
// that ships inside the image's triton package, GLM is vendored by the reference.  The
// kernel-side sources (raytraceKernel.cu, intersections.h, interactions.h) additionally
// need <cutil_math.h> and a CUDA Thrust next to <cuda_runtime.h>, which the image cannot
// provide -- they are NOT built.
//
// Usage: ref_probe scene <file>                     -> JSON dump of the parsed scene
//        ref_probe transform tx ty tz rx ry rz sx sy sz -> JSON transform + inverse
//        ref_probe glm                              -> JSON of vector-op known answers
//        ref_probe image <W> <H> <divisor> <gamma> <in.f32> <out.bmp|out.png>
//        ref_probe tokens "<line>"                  -> JSON list of tokens
#include <cmath>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

#include "scene.h"
#include "image.h"
#include "utilities.h"

static unsigned bits(float f) { unsigned u; memcpy(&u, &f, 4); return u; }

static void put_f(FILE *o, float f) { fprintf(o, "%u", bits(f)); }

static void put_v3(FILE *o, glm::vec3 v) {
    fprintf(o, "["); put_f(o, v.x); fprintf(o, ","); put_f(o, v.y); fprintf(o, ","); put_f(o, v.z); fprintf(o, "]");
}

static void put_mat(FILE *o, cudaMat4 m) {
    glm::vec4 r[4] = {m.x, m.y, m.z, m.w};
    fprintf(o, "[");
    for (int i = 0; i < 4; i++) {
        for (int j = 0; j < 4; j++) { put_f(o, r[i][j]); if (i != 3 || j != 3) fprintf(o, ","); }
    }
    fprintf(o, "]");
}

static int cmd_scene(const char *path) {
    // the parser chats on stdout; keep the JSON on stderr-free channel: write to fd 3 style file
    std::stringstream sink;
    std::streambuf *old = std::cout.rdbuf(sink.rdbuf());
    scene *s = new scene(std::string(path));
    std::cout.rdbuf(old);
    FILE *o = stdout;
    camera &c = s->renderCam;
    int F = c.frames;
    fprintf(o, "{\"floats_are\":\"binary32 bit patterns\",\n \"materials\":[");
    for (size_t i = 0; i < s->materials.size(); i++) {
        material &m = s->materials[i];
        const float *f = (const float *)&m;
        fprintf(o, "%s[", i ? "," : "");
        for (int k = 0; k < 16; k++) { put_f(o, f[k]); if (k != 15) fprintf(o, ","); }
        fprintf(o, "]");
    }
    fprintf(o, "],\n \"objects\":[");
    for (size_t i = 0; i < s->objects.size(); i++) {
        geom &g = s->objects[i];
        fprintf(o, "%s\n  {\"type\":%d,\"materialid\":%d,\"frames\":[", i ? "," : "", (int)g.type, g.materialid);
        for (int f = 0; f < F; f++) {
            fprintf(o, "%s{\"translation\":", f ? "," : ""); put_v3(o, g.translations[f]);
            fprintf(o, ",\"rotation\":"); put_v3(o, g.rotations[f]);
            fprintf(o, ",\"scale\":"); put_v3(o, g.scales[f]);
            fprintf(o, ",\"transform\":"); put_mat(o, g.transforms[f]);
            fprintf(o, ",\"inverseTransform\":"); put_mat(o, g.inverseTransforms[f]);
            fprintf(o, "}");
        }
        fprintf(o, "]}");
    }
    fprintf(o, "],\n \"camera\":{\"resolution\":["); put_f(o, c.resolution.x); fprintf(o, ","); put_f(o, c.resolution.y);
    fprintf(o, "],\"fov\":["); put_f(o, c.fov.x); fprintf(o, ","); put_f(o, c.fov.y);
    fprintf(o, "],\"iterations\":%u,\"frames\":%d,\"imageName\":\"%s\",\"positions\":[", c.iterations, F, c.imageName.c_str());
    for (int f = 0; f < F; f++) { if (f) fprintf(o, ","); put_v3(o, c.positions[f]); }
    fprintf(o, "],\"views\":[");
    for (int f = 0; f < F; f++) { if (f) fprintf(o, ","); put_v3(o, c.views[f]); }
    fprintf(o, "],\"ups\":[");
    for (int f = 0; f < F; f++) { if (f) fprintf(o, ","); put_v3(o, c.ups[f]); }
    fprintf(o, "]},\n \"sizeof\":{\"material\":%zu,\"geom\":%zu,\"staticGeom\":%zu,\"cameraData\":%zu,\"camera\":%zu,\"ray\":%zu,\"cudaMat4\":%zu}}\n",
            sizeof(material), sizeof(geom), sizeof(staticGeom), sizeof(cameraData), sizeof(camera), sizeof(ray), sizeof(cudaMat4));
    return 0;
}

static int cmd_transform(char **a) {
    glm::vec3 t(atof(a[0]), atof(a[1]), atof(a[2])), r(atof(a[3]), atof(a[4]), atof(a[5])), s(atof(a[6]), atof(a[7]), atof(a[8]));
    glm::mat4 m = utilityCore::buildTransformationMatrix(t, r, s);
    printf("{\"transform\":"); put_mat(stdout, utilityCore::glmMat4ToCudaMat4(m));
    printf(",\"inverseTransform\":"); put_mat(stdout, utilityCore::glmMat4ToCudaMat4(glm::inverse(m)));
    printf("}\n");
    return 0;
}

static int cmd_glm() {
    // deterministic pseudo-random inputs (simple LCG of our own; inputs are echoed)
    unsigned st = 12345u;
    printf("{\"cases\":[");
    for (int i = 0; i < 64; i++) {
        float v[6];
        for (int k = 0; k < 6; k++) { st = st * 1664525u + 1013904223u; v[k] = ((float)(st >> 8) / 16777216.0f) * 20.0f - 10.0f; }
        glm::vec3 a(v[0], v[1], v[2]), b(v[3], v[4], v[5]);
        glm::vec3 n = glm::normalize(a), c = glm::cross(a, b);
        printf("%s{\"a\":", i ? "," : ""); put_v3(stdout, a); printf(",\"b\":"); put_v3(stdout, b);
        printf(",\"normalize_a\":"); put_v3(stdout, n);
        printf(",\"cross\":"); put_v3(stdout, c);
        printf(",\"dot\":"); put_f(stdout, glm::dot(a, b));
        printf(",\"length_a\":"); put_f(stdout, glm::length(a));
        printf(",\"distance\":"); put_f(stdout, glm::distance(a, b));
        printf("}");
    }
    printf("]}\n");
    return 0;
}

static int cmd_image(char **a) {
    int W = atoi(a[0]), H = atoi(a[1]);
    int divisor = atoi(a[2]);
    float gamma = (float)atof(a[3]);
    FILE *f = fopen(a[4], "rb");
    if (!f) return 2;
    std::vector<float> buf((size_t)W * H * 3);
    if (fread(buf.data(), sizeof(float), buf.size(), f) != buf.size()) return 3;
    fclose(f);
    image img(W, H);
    for (int x = 0; x < W; x++)
        for (int y = 0; y < H; y++) {
            int idx = x + y * W;
            img.writePixelRGB(x, y, glm::vec3(buf[3 * idx], buf[3 * idx + 1], buf[3 * idx + 2]));
        }
    gammaSettings g;
    g.applyGamma = true; g.gamma = gamma; g.divisor = divisor;
    img.setGammaSettings(g);
    img.saveImageRGB(std::string(a[5]));
    return 0;
}

static int cmd_tokens(const char *line) {
    std::vector<std::string> t = utilityCore::tokenizeString(std::string(line));
    printf("[");
    for (size_t i = 0; i < t.size(); i++) printf("%s\"%s\"", i ? "," : "", t[i].c_str());
    printf("]\n");
    return 0;
}

int main(int argc, char **argv) {
    if (argc >= 3 && !strcmp(argv[1], "scene")) return cmd_scene(argv[2]);
    if (argc >= 11 && !strcmp(argv[1], "transform")) return cmd_transform(argv + 2);
    if (argc >= 2 && !strcmp(argv[1], "glm")) return cmd_glm();
    if (argc >= 8 && !strcmp(argv[1], "image")) return cmd_image(argv + 2);
    if (argc >= 3 && !strcmp(argv[1], "tokens")) return cmd_tokens(argv[2]);
    fprintf(stderr, "usage: ref_probe scene|transform|glm|image|tokens ...\n");
    return 64;
}
