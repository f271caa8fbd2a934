// ptrender.cpp -- headless command-line renderer on top of the C ABI (include/ptmi355.h): the
// product's own equivalent of the reference's viewer loop, for machines where the reference's
// sources (and OpenGL) are absent.  Same command line and observable behaviour as
// /root/reference/src/main.cpp:
//   * arguments are `key=value`; `scene=<file>` is required ("Error: scene file needed!" and
//     exit status 0 without it, main.cpp:31-47); `frame=<n>` renders only that frame
//     (single-frame mode, :39-42,157-161), an out-of-range frame falls back to 0 with the
//     reference's warning (:55-58);
//   * without `frame=` every frame from 0 to frames-1 is rendered in turn, the accumulator is
//     cleared between frames (:163-173);
//   * each frame runs camera.iterations iterations (1-based index = sample seed, :108-110), is
//     divided by the iteration count, gamma 1/2.2 corrected and saved as X.<frame>.bmp / .png where
//     X is the scene's FILE entry (:136-156), followed by "Saved frame <n> to <file>".
// Options the reference has no channel for are taken as further key=value arguments, falling
// back to the adaptor's environment variables:
//   depth= (PT_MAX_DEPTH)  mode=pathtrace|reference (PT_MODE)  camera= (PT_CAMERA_MODE)  aa= (PT_AA)
//   aperture= (PT_APERTURE)  focal= (PT_FOCAL_DIST)  gpus= (PT_NGPU)  device= (PT_DEVICE)
//   devices=0,1,.. (PT_DEVICES: explicit ordinals, may repeat)  direct=0|1 (PT_DIRECT_LIGHT)  streams=<n> (PT_STREAMS, default 2)
//   iterations=<n> overrides the scene's ITERATIONS, out=<dir> redirects the output directory,
//   ordering=/batch= pass through to pt_config, raw=1 also writes <file>.f32 (the float sum).
// Several GPUs: one context per device, rows interleaved, every device's work enqueued before any
// is awaited; the owned rows are gathered at the end of the frame (bit-identical to one GPU).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "ptmi355.h"

namespace {

[[noreturn]] void die(const char *what) {
    // error convention of the reference's device code (src/raytraceKernel.cu:20-26)
    fprintf(stderr, "Cuda error: %s: %s.\n", what, pt_last_error());
    exit(EXIT_FAILURE);
}

struct Options {
    std::string scene, outdir, devices;
    int frame = 0;
    bool single_frame = false;
    int depth, mode, camera_mode, aa, gpus, device, iterations = 0, ordering = 2, batch = -1, raw = 0, direct = 0, streams = 2;
    float aperture, focal;
};

const char *env_or(const char *name, const char *dflt) {
    const char *v = getenv(name);
    return (v && *v) ? v : dflt;
}

// first occurrence only, like utilityCore::replaceString (src/utilities.cpp:22-28)
void replace_first(std::string &s, const std::string &from, const std::string &to) {
    const size_t at = s.find(from);
    if (at != std::string::npos) s.replace(at, from.size(), to);
}

}  // namespace

int main(int argc, char **argv) {
    Options o;
    o.depth = atoi(env_or("PT_MAX_DEPTH", "8"));
    o.mode = !strcmp(env_or("PT_MODE", "pathtrace"), "reference") ? 1 : 0;
    o.camera_mode = atoi(env_or("PT_CAMERA_MODE", "0"));
    o.aa = atoi(env_or("PT_AA", "0"));
    o.aperture = (float)atof(env_or("PT_APERTURE", "0"));
    o.focal = (float)atof(env_or("PT_FOCAL_DIST", "0"));
    o.gpus = atoi(env_or("PT_NGPU", "1"));
    o.device = atoi(env_or("PT_DEVICE", "0"));
    o.devices = env_or("PT_DEVICES", "");
    o.direct = atoi(env_or("PT_DIRECT_LIGHT", "0"));
    o.streams = atoi(env_or("PT_STREAMS", "2"));
    for (int i = 1; i < argc; ++i) {
        const std::string arg(argv[i]);
        const size_t eq = arg.find('=');
        const std::string key = arg.substr(0, eq), val = eq == std::string::npos ? "" : arg.substr(eq + 1);
        if (key == "scene") o.scene = val;
        else if (key == "frame") { o.frame = atoi(val.c_str()); o.single_frame = true; }
        else if (key == "out") o.outdir = val;
        else if (key == "depth") o.depth = atoi(val.c_str());
        else if (key == "mode") o.mode = val == "reference" ? 1 : 0;
        else if (key == "camera") o.camera_mode = atoi(val.c_str());
        else if (key == "aa") o.aa = atoi(val.c_str());
        else if (key == "aperture") o.aperture = (float)atof(val.c_str());
        else if (key == "focal") o.focal = (float)atof(val.c_str());
        else if (key == "gpus") o.gpus = atoi(val.c_str());
        else if (key == "device") o.device = atoi(val.c_str());
        else if (key == "devices") o.devices = val;
        else if (key == "iterations") o.iterations = atoi(val.c_str());
        else if (key == "ordering") o.ordering = atoi(val.c_str());
        else if (key == "batch") o.batch = atoi(val.c_str());
        else if (key == "raw") o.raw = atoi(val.c_str());
        else if (key == "direct") o.direct = atoi(val.c_str());
        else if (key == "streams") o.streams = atoi(val.c_str());
        // unknown keys are ignored, like main.cpp's argument loop
    }
    if (o.scene.empty()) {
        printf("Error: scene file needed!\n");
        return 0;
    }
    pt_scene *scene = nullptr;
    if (pt_scene_load(o.scene.c_str(), &scene) != PT_OK) {
        // the reference prints the parser's message and carries on into a crash; fail cleanly instead
        fprintf(stderr, "%s\n", pt_last_error());
        return EXIT_FAILURE;
    }
    int G = 0, M = 0, frames = 0, iterations = 0;
    pt_scene_counts(scene, &G, &M, &frames, &iterations);
    if (o.iterations > 0) iterations = o.iterations;
    if (o.frame >= frames || o.frame < 0) {
        printf("Warning: Specified target frame is out of range, defaulting to frame 0.\n");
        o.frame = 0;
    }
    if (o.gpus < 1) o.gpus = 1;

    std::vector<int> ordinals;
    for (const char *p = o.devices.c_str(); *p;) {
        ordinals.push_back(atoi(p));
        while (*p && *p != ',') ++p;
        if (*p == ',') ++p;
    }
    std::vector<pt_context *> ctxs;
    for (int r = 0; r < o.gpus; ++r) {
        pt_config cfg;
        pt_config_default(&cfg);
        cfg.device = r < (int)ordinals.size() ? ordinals[r] : o.device + r;
        cfg.mode = o.mode;
        cfg.max_depth = o.depth;
        cfg.camera_mode = o.camera_mode;
        cfg.antialias = o.aa;
        cfg.aperture = o.aperture;
        cfg.focal_distance = o.focal;
        cfg.direct_light = o.direct;
        cfg.streams = o.streams;
        cfg.row_offset = r;
        cfg.row_stride = o.gpus;
        if (o.ordering >= 0) cfg.ordering = o.ordering;
        if (o.batch >= 0) cfg.batch = o.batch;
        pt_context *c = nullptr;
        if (pt_create(&cfg, &c) != PT_OK) die("pt_create");
        ctxs.push_back(c);
    }

    // MESH objects whose .obj the loader found: registered once, used by every upload
    std::vector<pt_mesh> meshes((size_t)pt_scene_mesh_count(scene));
    for (size_t k = 0; k < meshes.size(); ++k)
        if (pt_scene_mesh(scene, (int)k, &meshes[k]) != PT_OK) die("pt_scene_mesh");
    if (!meshes.empty()) {
        for (pt_context *c : ctxs)
            if (pt_set_meshes(c, meshes.data(), (int)meshes.size()) != PT_OK) die("pt_set_meshes");
        printf("Loaded %zu mesh(es)\n", meshes.size());
    }

    std::vector<pt_geom> geoms(G > 0 ? G : 1);
    std::vector<pt_material> materials(M > 0 ? M : 1);
    const int last = o.single_frame ? o.frame : frames - 1;
    for (int frame = o.frame; frame <= last; ++frame) {
        pt_camera cam;
        if (pt_scene_flatten(scene, frame, geoms.data(), materials.data(), &cam) != PT_OK) die("pt_scene_flatten");
        const int W = (int)cam.resolution[0], H = (int)cam.resolution[1];
        for (pt_context *c : ctxs) {
            if (pt_upload_scene(c, geoms.data(), G, materials.data(), M, &cam) != PT_OK) die("pt_upload_scene");
            if (pt_set_image(c, nullptr) != PT_OK) die("pt_set_image");            // cleared accumulator (main.cpp:168-170)
        }
        // reference mode overwrites instead of accumulating: one pass shows the same picture as n
        const int first = 1, count = iterations;
        for (pt_context *c : ctxs)                                                 // asynchronous on every device
            if (pt_render(c, first, count) != PT_OK) die("pt_render");
        std::vector<float> image((size_t)W * H * 3);
        if (ctxs.size() == 1) {
            if (pt_get_image(ctxs[0], image.data()) != PT_OK) die("Kernel failed!");
        } else {
            // every GPU sends only the rows it owns (1/N of the frame over its own PCIe link)
            for (pt_context *c : ctxs)
                if (pt_get_rows(c, image.data()) != PT_OK) die("Kernel failed!");
        }
        std::string filename = pt_scene_image_name(scene);
        const std::string n = std::to_string(frame);
        replace_first(filename, ".bmp", "." + n + ".bmp");
        replace_first(filename, ".png", "." + n + ".png");
        if (!o.outdir.empty()) {
            const size_t slash = filename.find_last_of('/');
            filename = o.outdir + "/" + (slash == std::string::npos ? filename : filename.substr(slash + 1));
        }
        if (pt_image_save(filename.c_str(), image.data(), W, H, iterations, (float)(1.0 / 2.2)) != PT_OK) die("pt_image_save");
        printf("Saved frame %s to %s\n", n.c_str(), filename.c_str());
        if (o.raw) {
            FILE *f = fopen((filename + ".f32").c_str(), "wb");
            if (f) { fwrite(image.data(), sizeof(float), image.size(), f); fclose(f); }
        }
    }
    for (pt_context *c : ctxs) pt_destroy(c);                                      // cudaDeviceReset (main.cpp:159,171)
    pt_scene_free(scene);
    return 0;
}
