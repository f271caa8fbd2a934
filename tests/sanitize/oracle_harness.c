/* oracle_harness.c -- the CPU oracle (oracle/pt_oracle.c) under AddressSanitizer + UndefinedBehaviorSanitizer:
 * a small hand-built room (diffuse walls, mirror / glass / diffuse spheres, cube and sphere emitters, a
 * rotated cube) rendered with every option on -- AA, thin lens, both camera modes, direct light sampling,
 * row sharding -- plus the pool trace, the flat reference kernel and the image conversions.  Float -> integer
 * conversions (hash of a float seed), array indexing by material / primitive ids and the accumulation
 * loops are what the sanitizers watch.  Built and run by tests/test_sanitizers.py. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "pt_oracle.h"

static orc_geom make(int type, int mat, float tx, float ty, float tz, float rx, float ry, float rz, float sx, float sy, float sz) {
    orc_geom g;
    memset(&g, 0, sizeof g);
    g.type = type; g.materialid = mat;
    const float t[3] = {tx, ty, tz}, r[3] = {rx, ry, rz}, s[3] = {sx, sy, sz};
    if (orc_build_transform(t, r, s, g.transform, g.inverseTransform) != 0) { fprintf(stderr, "singular transform\n"); exit(2); }
    return g;
}

int main(void) {
    orc_material m[6];
    memset(m, 0, sizeof m);
    for (int i = 0; i < 6; i++) { m[i].color[0] = 0.8f; m[i].color[1] = 0.7f; m[i].color[2] = 0.6f; m[i].specularColor[0] = m[i].specularColor[1] = m[i].specularColor[2] = 1.0f; }
    m[1].hasReflective = 1.0f;
    m[2].hasRefractive = 1.0f; m[2].indexOfRefraction = 1.5f;
    m[3].emittance = 12.0f;
    m[4].emittance = 3.0f; m[4].color[1] = 0.2f;
    orc_geom g[10];
    int n = 0;
    g[n++] = make(1, 0, 0, 0, 0, 0, 0, 90, .01f, 10, 10);        /* floor */
    g[n++] = make(1, 0, 0, 5, -5, 0, 90, 0, .01f, 10, 10);       /* back wall */
    g[n++] = make(1, 0, 0, 10, 0, 0, 0, 90, .01f, 10, 10);       /* ceiling */
    g[n++] = make(1, 5, -5, 5, 0, 0, 0, 0, .01f, 10, 10);        /* left */
    g[n++] = make(1, 0, 5, 5, 0, 0, 0, 0, .01f, 10, 10);         /* right */
    g[n++] = make(0, 1, -2, 5, -1, 0, 0, 0, 3, 3, 3);            /* mirror sphere */
    g[n++] = make(0, 2, 2, 3, 1, 0, 0, 0, 2.5f, 2.5f, 2.5f);     /* glass sphere */
    g[n++] = make(1, 2, 0, 1.5f, 2, 20, 30, 40, 1.5f, 1.5f, 1.5f);/* glass cube, rotated */
    g[n++] = make(1, 3, 0, 10, 0, 0, 0, 0, 3, .3f, 3);           /* cube light */
    g[n++] = make(0, 4, -3, 8, 2, 0, 0, 0, 1, 1, 1);             /* sphere light */
    orc_camera cam;
    memset(&cam, 0, sizeof cam);
    const int W = 40, H = 30;
    cam.resolution[0] = (float)W; cam.resolution[1] = (float)H;
    cam.position[1] = 4.5f; cam.position[2] = 12.0f;
    cam.view[2] = -1.0f; cam.up[1] = 1.0f;
    cam.fov[1] = 25.0f; cam.fov[0] = 32.0f;
    float *img = (float *)calloc((size_t)W * H * 3, sizeof(float));
    uint64_t live[65];
    double total = 0;
    for (int variant = 0; variant < 6; variant++) {
        orc_config cfg;
        memset(&cfg, 0, sizeof cfg);
        cfg.max_depth = variant == 5 ? 64 : 6;
        cfg.camera_mode = variant & 1;
        cfg.antialias = variant >= 2;
        cfg.aperture = (variant & 1) ? 0.25f : 0.0f;
        cfg.focal_distance = 12.0f;
        cfg.row_offset = variant == 4 ? 1 : 0;
        cfg.row_stride = variant == 4 ? 3 : 1;
        cfg.direct_light = variant >= 3;
        if (orc_render(g, n, m, 6, &cam, &cfg, 1 + 7 * variant, 3, img, live, 2) != 0) return 3;
        for (int k = 0; k <= cfg.max_depth; k++) total += (double)live[k];
        uint32_t *pix = (uint32_t *)malloc(sizeof(uint32_t) * W * H);
        float *f = (float *)malloc(sizeof(float) * 9 * W * H);
        for (int b = 0; b <= 3; b++)
            total += orc_trace_pool(g, n, m, 6, &cam, &cfg, 2, b, f, f + W * H, f + 2 * W * H, f + 3 * W * H, f + 4 * W * H,
                                    f + 5 * W * H, f + 6 * W * H, f + 7 * W * H, f + 8 * W * H, pix);
        free(pix); free(f);
    }
    int *hit = (int *)malloc(sizeof(int) * W * H);
    if (orc_raycast_flat(g, n, m, 6, &cam, img, hit, 2) != 0) return 4;
    uint8_t *u8 = (uint8_t *)malloc((size_t)W * H * 3);
    img[0] = INFINITY; img[1] = -5.0f; img[2] = NAN;                   /* conversions must not trap */
    orc_image_to_u8(img, W * H, 3.0f, 1.0f / 2.2f, u8);
    uint8_t px[4];
    for (int i = 0; i < W * H; i++) orc_display_pixel(img + 3 * i, px);
    /* samplers with awkward seeds: negative, huge, fractional */
    const float seeds[6] = {0.0f, 1.5f, 16777215.0f, 4294967040.0f, 3.0e9f, 123456.7f};
    for (int i = 0; i < 6; i++) {
        float Q[3], inv;
        (void)orc_sample_light(&g[8], seeds[i], Q, &inv);
        (void)orc_sample_light(&g[9], seeds[i], Q, &inv);
        float out[3];
        orc_rng_from_thread(800.0f, 800.0f, seeds[i] > 1e6f ? 3.0f : seeds[i], 17, 5, out);
    }
    /* a MESH primitive (octahedron, 8 triangles) in place of the rotated glass cube: registry, brute-force test, render */
    {
        static const float ov[18] = {.5f, 0, 0, -.5f, 0, 0, 0, .5f, 0, 0, -.5f, 0, 0, 0, .5f, 0, 0, -.5f};
        static const int oi[24] = {0, 2, 4, 2, 1, 4, 1, 3, 4, 3, 0, 4, 2, 0, 5, 1, 2, 5, 3, 1, 5, 0, 3, 5};
        g[7].type = 2;
        const int gi[1] = {7}, nv[1] = {6}, nt[1] = {8};
        const float *vp[1] = {ov};
        const int *ip[1] = {oi};
        if (orc_set_meshes(gi, vp, nv, ip, nt, 1) != 0) return 5;
        orc_config cfg;
        memset(&cfg, 0, sizeof cfg);
        cfg.max_depth = 6; cfg.row_stride = 1; cfg.direct_light = 1;
        float *img2 = (float *)calloc((size_t)W * H * 3, sizeof(float));
        if (orc_render(g, n, m, 6, &cam, &cfg, 1, 3, img2, live, 2) != 0) return 6;
        for (int k = 0; k <= 6; k++) total += (double)live[k];
        float P[3], N[3];
        int tri = -1;
        const float o[3] = {0, 1.5f, 12}, d[3] = {0, 0, -1};
        total += orc_mesh_test(&g[7], ov, oi, 8, o, d, P, N, &tri) > 0 ? 1 : 0;
        (void)orc_set_meshes(NULL, NULL, NULL, NULL, NULL, 0);
        free(img2);
    }
    free(hit); free(u8); free(img);
    printf("ok %.0f\n", total);
    return 0;
}
