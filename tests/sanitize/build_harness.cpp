// ASan / UBSan harness for the host-only scene builders (csrc/pt_build.cpp): every scene file on the command line is loaded with
// the library's own loader and run through the grid builder (narrow or wide references by primitive count, three densities) with
// a few thousand probe rays, the camera-fan probe, the cluster builder and -- for scenes with meshes -- the BVH builder.
// TEST INFRASTRUCTURE: compiled by tests/test_sanitizers.py with hipcc --offload-host-only (no device code, no device needed).
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "ptmi355.h"
#include "../../project2-pathtracer_amd/csrc/pt_build.hpp"

int main(int argc, char **argv) {
    int scenes = 0;
    for (int a = 1; a < argc; ++a) {
        pt_scene *s = nullptr;
        if (pt_scene_load(argv[a], &s) != PT_OK) { fprintf(stderr, "load failed: %s: %s\n", argv[a], pt_last_error()); return 2; }
        int G = 0, M = 0, frames = 0, iters = 0;
        pt_scene_counts(s, &G, &M, &frames, &iters);
        std::vector<pt_geom> g(G);
        std::vector<pt_material> m(M);
        pt_camera cam;
        if (pt_scene_flatten(s, 0, g.data(), m.data(), &cam) != PT_OK) return 3;
        // probe rays: from the eye towards a lattice of points, and the reverse
        std::vector<float> rays;
        uint32_t st = 12345u;
        auto rnd = [&]() { st = st * 1664525u + 1013904223u; return (float)(st >> 8) / 16777216.0f; };
        for (int k = 0; k < 3000; ++k) {
            float o[3] = {cam.position[0] + 2.0f * (rnd() - 0.5f), cam.position[1] + 2.0f * (rnd() - 0.5f), cam.position[2] + 2.0f * (rnd() - 0.5f)};
            float d[3] = {rnd() - 0.5f, rnd() - 0.5f, -rnd() - 0.05f};
            if (k & 1) { for (int c = 0; c < 3; ++c) { o[c] += 12.0f * d[c]; d[c] = -d[c]; } }
            if (k % 97 == 0) d[0] = 0.0f;
            if (k % 389 == 0) d[1] = d[2] = 0.0f, d[0] = 1.0f;
            for (int c = 0; c < 3; ++c) rays.push_back(o[c]);
            for (int c = 0; c < 3; ++c) rays.push_back(d[c]);
        }
        const int nrays = (int)rays.size() / 6;
        const size_t words = (size_t)(G > 256 ? (G + 31) / 32 : 8);
        std::vector<uint32_t> sets((size_t)nrays * words), info(24);
        for (int density : {0, 1, 16}) {
            if (pth::grid_probe(g.data(), G, density, rays.data(), nrays, sets.data(), info.data()) != PT_OK) { fprintf(stderr, "grid_probe: %s\n", pt_last_error()); return 4; }
            if (info[3] != 0) { fprintf(stderr, "%s: a primitive listed twice (density %d)\n", argv[a], density); return 5; }
        }
        const int nfans = nrays / 64;
        std::vector<float> fans((size_t)nfans * 64 * 6);
        for (int f = 0; f < nfans; ++f)
            for (int l = 0; l < 64; ++l) {
                float *r = &fans[((size_t)f * 64 + l) * 6];
                for (int c = 0; c < 3; ++c) r[c] = cam.position[c];
                const float dx = -0.4f + 0.8f * (float)f / (float)nfans + 0.0007f * (float)l, dy = 0.3f * (rnd() - 0.5f) * (f & 1 ? 0.0f : 1.0f) + 0.01f * (float)(f % 7);
                const float n = 1.0f / std::sqrt(dx * dx + dy * dy + 1.0f);
                r[3] = dx * n; r[4] = dy * n; r[5] = -n;
            }
        uint32_t finfo[4] = {0, 0, 0, 0};
        if (pth::fan_probe(g.data(), G, fans.data(), nfans, sets.data(), finfo) != PT_OK) return 6;
        std::vector<ptd::GeomRec> rec(G);
        for (int i = 0; i < G; ++i) { memset(&rec[i], 0, sizeof(ptd::GeomRec)); rec[i].type = g[i].type; pth::world_bounds(g[i], &rec[i]); }
        if (G > 32 && G <= 256) { pth::ClusterBuild cb; (void)pth::build_clusters(rec, G, 0, &cb); (void)pth::build_clusters(rec, G, 16, &cb); }
        const int nm = pt_scene_mesh_count(s);
        for (int k = 0; k < nm; ++k) {
            pt_mesh pm;
            if (pt_scene_mesh(s, k, &pm) != PT_OK) return 7;
            pth::HostMesh hm;
            hm.geom_index = pm.geom_index;
            hm.v.assign(pm.vertices, pm.vertices + 3 * (size_t)pm.nvertices);
            hm.idx.assign(pm.indices, pm.indices + 3 * (size_t)pm.ntriangles);
            uint32_t tri_offset = 0;
            const std::vector<unsigned char> blob = pth::build_mesh_blob(hm, &tri_offset);
            if (blob.size() < tri_offset) return 8;
            pth::mesh_world_bounds(g[pm.geom_index], hm, &rec[pm.geom_index]);
        }
        pt_scene_free(s);
        scenes++;
    }
    printf("built %d\n", scenes);
    return 0;
}
