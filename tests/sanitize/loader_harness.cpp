// loader_harness.cpp -- host-side scene I/O (csrc/pt_scene.cpp: no HIP inside) under AddressSanitizer +
// UndefinedBehaviorSanitizer.  Loads every scene file named on the command line, exercises every accessor
// for every frame, builds transforms, converts and saves a small image as BMP and PNG.  Malformed files must
// come back as PT_ERR_PARSE / PT_ERR_IO with a message -- never a crash, a leak or an out-of-bounds access.
// Built and driven by tests/test_sanitizers.py (CPU only; the GPU build cannot use sanitizers on this pool).
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "ptmi355.h"

int main(int argc, char **argv) {
    int loaded = 0, rejected = 0;
    const std::string outdir = argc > 1 ? argv[1] : "/tmp";
    for (int i = 2; i < argc; ++i) {
        pt_scene *s = nullptr;
        const int rc = pt_scene_load(argv[i], &s);
        if (rc != PT_OK) {
            if ((rc != PT_ERR_PARSE && rc != PT_ERR_IO) || !pt_last_error() || !*pt_last_error() || s != nullptr) {
                fprintf(stderr, "bad failure for %s: rc=%d\n", argv[i], rc);
                return 2;
            }
            rejected++;
            continue;
        }
        int G = 0, M = 0, F = 0, N = 0;
        if (pt_scene_counts(s, &G, &M, &F, &N) != PT_OK) return 3;
        (void)pt_scene_image_name(s);
        std::vector<pt_geom> g(G > 0 ? G : 1);
        std::vector<pt_material> m(M > 0 ? M : 1);
        for (int f = -1; f <= F; ++f) {                       // incl. out-of-range frames: must be refused, not read
            pt_camera cam;
            const int r = pt_scene_flatten(s, f, g.data(), m.data(), &cam);
            if ((f < 0 || f >= F) ? r == PT_OK : r != PT_OK) { fprintf(stderr, "flatten frame %d of %s: rc=%d\n", f, argv[i], r); pt_scene_free(s); return 4; }
            float T[16], Ti[16];
            for (int o = -1; o <= G; ++o) (void)pt_scene_object_matrices(s, o, f, T, Ti);
        }
        // MESH objects: every accessor, in and out of range
        const int nm = pt_scene_mesh_count(s);
        for (int k = -1; k <= nm; ++k) {
            pt_mesh me;
            memset(&me, 0, sizeof me);
            const int r = pt_scene_mesh(s, k, &me);
            if ((k < 0 || k >= nm) ? r == PT_OK : r != PT_OK) { fprintf(stderr, "pt_scene_mesh(%d) of %s: rc=%d\n", k, argv[i], r); pt_scene_free(s); return 9; }
            if (r == PT_OK) {
                long long sum = 0;
                for (int t = 0; t < 3 * me.ntriangles; ++t) {
                    if (me.indices[t] < 0 || me.indices[t] >= me.nvertices) { fprintf(stderr, "index out of range in %s\n", argv[i]); pt_scene_free(s); return 10; }
                    sum += (long long)me.vertices[3 * me.indices[t]];          // touches every referenced vertex
                }
                (void)sum;
            }
        }
        pt_scene_free(s);
        loaded++;
    }
    // image path: gamma/clamp/u8 and both writers, odd width for the BMP row padding
    const int W = 7, H = 5;
    std::vector<float> img((size_t)W * H * 3);
    for (size_t k = 0; k < img.size(); ++k) img[k] = (float)k * 0.37f - 3.0f;       // negative, > 1 and in range
    img[4] = 1e30f; img[5] = -1e30f;
    std::vector<uint8_t> u8((size_t)W * H * 3);
    if (pt_image_to_u8(img.data(), W, H, 3, 1.0f / 2.2f, u8.data()) != PT_OK) return 5;
    if (pt_image_save((outdir + "/asan.bmp").c_str(), img.data(), W, H, 3, 1.0f / 2.2f) != PT_OK) return 6;
    if (pt_image_save((outdir + "/asan.png").c_str(), img.data(), W, H, 3, 1.0f / 2.2f) != PT_OK) return 7;
    if (pt_image_save((outdir + "/no/such/dir/x.bmp").c_str(), img.data(), W, H, 3, 1.0f / 2.2f) == PT_OK) return 8;
    const float t[3] = {1, 2, 3}, r[3] = {10, 20, 30}, sc[3] = {0.5f, 2, 0};             // singular scale: no crash
    float T[16], Ti[16];
    (void)pt_build_transform(t, r, sc, T, Ti);
    pt_config cfg;
    pt_config_default(&cfg);
    printf("loaded %d rejected %d abi %d\n", loaded, rejected, pt_abi_version());
    return 0;
}
