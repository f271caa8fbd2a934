// dropin_driver.cpp -- headless stand-in for the reference's GL viewer loop, used ONLY to show that
// the reference's own host code drives the HIP library through cudaRaytraceCore unchanged.
//
// It links the REFERENCE's scene.cpp / utilities.cpp / image.cpp / stb_image_write.c (compiled from
// /root/reference where they lie, oracle/Makefile) and the adaptor TU, and repeats what
// runCuda() does in /root/reference/src/main.cpp:103-176 without the OpenGL calls: pack geoms and
// materials into fresh arrays every iteration (:114-122), call cudaRaytraceCore (:126), then copy
// camera::image into the reference's `image` class, apply gamma 1/2.2 with divisor = iterations
// (:136-147), name the file X.<frame>.bmp (:148-154) and save.  TEST INFRASTRUCTURE.
//
// usage: dropin_driver scene=<file> [frame=<n>] [out=<dir>] [pbo=0]   (pbo=0: pass pos = NULL like a run
//        without a mapped display buffer)
#include <cstdio>
#include <cstring>
#include <iostream>
#include <sstream>
#include <string>

#include "scene.h"
#include "image.h"
#include "utilities.h"

void cudaRaytraceCore(uchar4 *pos, camera *renderCam, int frame, int iterations, material *materials,
                      int numberOfMaterials, geom *geoms, int numberOfGeoms);
extern "C" void ptmi355_adaptor_reset(void);

int main(int argc, char **argv) {
    scene *renderScene = NULL;
    int targetFrame = 0;
    std::string outdir;
    bool use_pbo = true;
    for (int i = 1; i < argc; i++) {
        std::string header, data;
        std::istringstream liness(argv[i]);
        std::getline(liness, header, '=');
        std::getline(liness, data, '=');
        if (header == "scene") renderScene = new scene(data);
        else if (header == "frame") targetFrame = atoi(data.c_str());
        else if (header == "out") outdir = data;
        else if (header == "pbo") use_pbo = atoi(data.c_str()) != 0;
    }
    if (!renderScene) { std::cout << "Error: scene file needed!" << std::endl; return 0; }
    camera *renderCam = &renderScene->renderCam;
    if (targetFrame >= renderCam->frames) targetFrame = 0;
    const int W = (int)renderCam->resolution.x, H = (int)renderCam->resolution.y;
    uchar4 *pbo = new uchar4[(size_t)W * H];
    for (int iterations = 1; iterations <= (int)renderCam->iterations; ++iterations) {
        geom *geoms = new geom[renderScene->objects.size()];
        material *materials = new material[renderScene->materials.size()];
        for (size_t i = 0; i < renderScene->objects.size(); i++) geoms[i] = renderScene->objects[i];
        for (size_t i = 0; i < renderScene->materials.size(); i++) materials[i] = renderScene->materials[i];
        cudaRaytraceCore(use_pbo ? pbo : NULL, renderCam, targetFrame, iterations, materials, (int)renderScene->materials.size(),
                         geoms, (int)renderScene->objects.size());
        delete[] geoms;
        delete[] materials;
    }
    image outputImage(W, H);
    for (int x = 0; x < W; x++)
        for (int y = 0; y < H; y++) outputImage.writePixelRGB(x, y, renderCam->image[x + y * W]);
    gammaSettings gamma;
    gamma.applyGamma = true;
    gamma.gamma = 1.0 / 2.2;
    gamma.divisor = renderCam->iterations;
    outputImage.setGammaSettings(gamma);
    std::string filename = renderCam->imageName;
    std::stringstream out;
    out << targetFrame;
    utilityCore::replaceString(filename, ".bmp", "." + out.str() + ".bmp");
    utilityCore::replaceString(filename, ".png", "." + out.str() + ".png");
    if (!outdir.empty()) {
        size_t slash = filename.find_last_of('/');
        filename = outdir + "/" + (slash == std::string::npos ? filename : filename.substr(slash + 1));
    }
    outputImage.saveImageRGB(filename);
    std::cout << "Saved frame " << out.str() << " to " << filename << std::endl;
    // raw accumulator + last display buffer next to it, for the parity test
    FILE *f = fopen((filename + ".f32").c_str(), "wb");
    if (f) { fwrite(renderCam->image, sizeof(glm::vec3), (size_t)W * H, f); fclose(f); }
    f = fopen((filename + ".pbo").c_str(), "wb");
    if (f) { fwrite(pbo, 4, (size_t)W * H, f); fclose(f); }
    ptmi355_adaptor_reset();
    return 0;
}
