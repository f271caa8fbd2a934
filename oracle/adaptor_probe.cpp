// adaptor_probe.cpp -- calls the adaptor's cudaRaytraceCore the way the reference's runCuda() does, but changes the
// scene in the MIDDLE of a frame, which the unchanged main.cpp never does and the reference handles for free (it
// round-trips camera::image on every call, /root/reference/src/raytraceKernel.cu:176,215).  TEST INFRASTRUCTURE: no
// reference code in this file; it includes the reference's scene.h and links its scene.cpp / utilities.cpp compiled
// against the product's shim (oracle/Makefile, _ref/adaptor_probe).
//
// usage: adaptor_probe <scene file> <iterations> <change_at> <out.f32>
//   iterations 1..change_at-1 render the scene as loaded; from iteration change_at on, material 0's colour is
//   (0.2, 0.9, 0.4) and object 5 has moved up by 1.  The raw camera::image after the last iteration goes to out.f32.
#include <cstdio>
#include <cstdlib>
#include <string>

#include "scene.h"

void cudaRaytraceCore(uchar4 *pos, camera *renderCam, int frame, int iterations, material *materials, int numberOfMaterials,
                      geom *geoms, int numberOfGeoms);
extern "C" void ptmi355_adaptor_reset(void);

int main(int argc, char **argv) {
    if (argc < 5) { fprintf(stderr, "usage: adaptor_probe <scene> <iterations> <change_at> <out.f32>\n"); return 2; }
    scene *sc = new scene(std::string(argv[1]));
    const int iterations = atoi(argv[2]), change_at = atoi(argv[3]);
    camera *cam = &sc->renderCam;
    cam->iterations = (unsigned)iterations;
    const int W = (int)cam->resolution.x, H = (int)cam->resolution.y;
    for (int it = 1; it <= iterations; ++it) {
        if (it == change_at) {
            sc->materials[0].color = glm::vec3(0.2f, 0.9f, 0.4f);
            geom &g = sc->objects[5];
            g.translations[0].y += 1.0f;
            glm::mat4 t = utilityCore::buildTransformationMatrix(g.translations[0], g.rotations[0], g.scales[0]);
            g.transforms[0] = utilityCore::glmMat4ToCudaMat4(t);
            g.inverseTransforms[0] = utilityCore::glmMat4ToCudaMat4(glm::inverse(t));
        }
        geom *geoms = new geom[sc->objects.size()];
        material *materials = new material[sc->materials.size()];
        for (size_t i = 0; i < sc->objects.size(); i++) geoms[i] = sc->objects[i];
        for (size_t i = 0; i < sc->materials.size(); i++) materials[i] = sc->materials[i];
        cudaRaytraceCore(NULL, cam, 0, it, materials, (int)sc->materials.size(), geoms, (int)sc->objects.size());
        delete[] geoms;
        delete[] materials;
    }
    FILE *f = fopen(argv[4], "wb");
    if (!f) return 3;
    fwrite(cam->image, sizeof(glm::vec3), (size_t)W * H, f);
    fclose(f);
    ptmi355_adaptor_reset();
    return 0;
}
