// pt_scene.cpp -- host-only half of libptmi355.so: error string, config defaults, camera basis,
// scene-file loader, transform builder and image writer.  No HIP calls in this TU, so these
// entry points work (and are tested) on a machine without a GPU.
//
// Reference behaviour restated here (paths relative to /root/reference):
//   scene grammar                src/scene.cpp:9-263 (summary: SURVEY.md section 5)
//   buildTransformationMatrix    src/utilities.cpp:70-77, glmMat4ToCudaMat4 :79-86, with GLM
//                                0.9.3.4 translate/rotate/scale/inverse semantics
//   fov from FOVY + resolution   src/scene.cpp:201-205
//   gamma/clamp/u8 + file naming src/image.cpp:40-87, src/main.cpp:143-154
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "../../include/ptmi355.h"
#include "pt_host.hpp"

namespace {

thread_local char g_error[512] = "";

const float kPI = 3.1415926535897932384626422832795028841971f;   // src/utilities.h:20

struct Vec3 { float v[3]; };

// ---- 4x4 float matrix, column vectors like glm::mat4 ------------------------------------
struct Mat4 {
    float col[4][4];   // col[c][r]
    static Mat4 identity() {
        Mat4 m;
        for (int c = 0; c < 4; ++c)
            for (int r = 0; r < 4; ++r) m.col[c][r] = (c == r) ? 1.0f : 0.0f;
        return m;
    }
};

// operator* of glm::mat4: every result column is ((A0*b0 + A1*b1) + A2*b2) + A3*b3
Mat4 multiply(const Mat4 &a, const Mat4 &b) {
    Mat4 out;
    for (int c = 0; c < 4; ++c) {
        const float *bc = b.col[c];
        for (int r = 0; r < 4; ++r) {
            float acc = a.col[0][r] * bc[0] + a.col[1][r] * bc[1];
            acc = acc + a.col[2][r] * bc[2];
            acc = acc + a.col[3][r] * bc[3];
            out.col[c][r] = acc;
        }
    }
    return out;
}

Mat4 translation_of(const float t[3]) {        // glm::translate(identity, t)
    Mat4 id = Mat4::identity(), out = id;
    for (int r = 0; r < 4; ++r) {
        float acc = id.col[0][r] * t[0] + id.col[1][r] * t[1];
        acc = acc + id.col[2][r] * t[2];
        out.col[3][r] = acc + id.col[3][r];
    }
    return out;
}

Mat4 rotation_of(float degrees, int axis_index) {   // glm::rotate(identity, degrees, unit axis)
    const float pi = 3.1415926535897932384626433832795f;
    const float a = degrees * (pi / 180.0f);
    const float c = cosf(a), s = sinf(a);
    float axis[3] = {0.0f, 0.0f, 0.0f};
    axis[axis_index] = 1.0f;
    // normalize(axis) = axis * (1/sqrt(dot)) -- exact for a unit basis vector, kept for fidelity
    const float inv = 1.0f / sqrtf(axis[0] * axis[0] + axis[1] * axis[1] + axis[2] * axis[2]);
    for (float &x : axis) x = x * inv;
    float temp[3];
    for (int k = 0; k < 3; ++k) temp[k] = (1.0f - c) * axis[k];
    float rot[3][3];
    rot[0][0] = c + temp[0] * axis[0];
    rot[0][1] = 0 + temp[0] * axis[1] + s * axis[2];
    rot[0][2] = 0 + temp[0] * axis[2] - s * axis[1];
    rot[1][0] = 0 + temp[1] * axis[0] - s * axis[2];
    rot[1][1] = c + temp[1] * axis[1];
    rot[1][2] = 0 + temp[1] * axis[2] + s * axis[0];
    rot[2][0] = 0 + temp[2] * axis[0] + s * axis[1];
    rot[2][1] = 0 + temp[2] * axis[1] - s * axis[0];
    rot[2][2] = c + temp[2] * axis[2];
    Mat4 id = Mat4::identity(), out;
    for (int k = 0; k < 3; ++k)
        for (int r = 0; r < 4; ++r) {
            float acc = id.col[0][r] * rot[k][0] + id.col[1][r] * rot[k][1];
            out.col[k][r] = acc + id.col[2][r] * rot[k][2];
        }
    for (int r = 0; r < 4; ++r) out.col[3][r] = id.col[3][r];
    return out;
}

Mat4 scaling_of(const float s[3]) {            // glm::scale(identity, s)
    Mat4 id = Mat4::identity(), out;
    for (int r = 0; r < 4; ++r) {
        for (int k = 0; k < 3; ++k) out.col[k][r] = id.col[k][r] * s[k];
        out.col[3][r] = id.col[3][r];
    }
    return out;
}

// glm::inverse(mat4): cofactors from 2x2 sub-determinants, divided by the determinant taken
// along the first column (src/glm/core/func_matrix.inl:523-583)
Mat4 inverse_of(const Mat4 &M) {
    auto e = [&](int c, int r) { return M.col[c][r]; };
    auto sub = [&](int c0, int r0, int c1, int r1, int c2, int r2, int c3, int r3) {
        return e(c0, r0) * e(c1, r1) - e(c2, r2) * e(c3, r3);
    };
    const float c00 = sub(2, 2, 3, 3, 3, 2, 2, 3), c02 = sub(1, 2, 3, 3, 3, 2, 1, 3), c03 = sub(1, 2, 2, 3, 2, 2, 1, 3);
    const float c04 = sub(2, 1, 3, 3, 3, 1, 2, 3), c06 = sub(1, 1, 3, 3, 3, 1, 1, 3), c07 = sub(1, 1, 2, 3, 2, 1, 1, 3);
    const float c08 = sub(2, 1, 3, 2, 3, 1, 2, 2), c10 = sub(1, 1, 3, 2, 3, 1, 1, 2), c11 = sub(1, 1, 2, 2, 2, 1, 1, 2);
    const float c12 = sub(2, 0, 3, 3, 3, 0, 2, 3), c14 = sub(1, 0, 3, 3, 3, 0, 1, 3), c15 = sub(1, 0, 2, 3, 2, 0, 1, 3);
    const float c16 = sub(2, 0, 3, 2, 3, 0, 2, 2), c18 = sub(1, 0, 3, 2, 3, 0, 1, 2), c19 = sub(1, 0, 2, 2, 2, 0, 1, 2);
    const float c20 = sub(2, 0, 3, 1, 3, 0, 2, 1), c22 = sub(1, 0, 3, 1, 3, 0, 1, 1), c23 = sub(1, 0, 2, 1, 2, 0, 1, 1);
    const float f0[4] = {c00, c00, c02, c03}, f1[4] = {c04, c04, c06, c07}, f2[4] = {c08, c08, c10, c11};
    const float f3[4] = {c12, c12, c14, c15}, f4[4] = {c16, c16, c18, c19}, f5[4] = {c20, c20, c22, c23};
    const float v0[4] = {e(1, 0), e(0, 0), e(0, 0), e(0, 0)}, v1[4] = {e(1, 1), e(0, 1), e(0, 1), e(0, 1)};
    const float v2[4] = {e(1, 2), e(0, 2), e(0, 2), e(0, 2)}, v3[4] = {e(1, 3), e(0, 3), e(0, 3), e(0, 3)};
    const float sa[4] = {1, -1, 1, -1}, sb[4] = {-1, 1, -1, 1};
    Mat4 inv;
    for (int k = 0; k < 4; ++k) {
        inv.col[0][k] = sa[k] * ((v1[k] * f0[k] - v2[k] * f1[k]) + v3[k] * f2[k]);
        inv.col[1][k] = sb[k] * ((v0[k] * f0[k] - v2[k] * f3[k]) + v3[k] * f4[k]);
        inv.col[2][k] = sa[k] * ((v0[k] * f1[k] - v1[k] * f3[k]) + v3[k] * f5[k]);
        inv.col[3][k] = sb[k] * ((v0[k] * f2[k] - v1[k] * f4[k]) + v2[k] * f5[k]);
    }
    float det = e(0, 0) * inv.col[0][0] + e(0, 1) * inv.col[1][0];
    det = det + e(0, 2) * inv.col[2][0];
    det = det + e(0, 3) * inv.col[3][0];
    for (int c = 0; c < 4; ++c)
        for (int r = 0; r < 4; ++r) inv.col[c][r] = inv.col[c][r] / det;
    return inv;
}

void to_rows(const Mat4 &m, float out[16]) {   // glmMat4ToCudaMat4: row r -> out[4r..4r+3]
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) out[4 * r + c] = m.col[c][r];
}

void build_pair(const float t[3], const float r[3], const float s[3], float xf[16], float inv[16]) {
    Mat4 rot = multiply(multiply(rotation_of(r[0], 0), rotation_of(r[1], 1)), rotation_of(r[2], 2));
    Mat4 full = multiply(multiply(translation_of(t), rot), scaling_of(s));
    to_rows(full, xf);
    to_rows(inverse_of(full), inv);
}

std::vector<std::string> tokens_of(const std::string &line) {   // utilityCore::tokenizeString
    std::istringstream ss(line);
    std::vector<std::string> out;
    std::string tok;
    while (ss >> tok) out.push_back(tok);
    return out;
}

float num(const std::vector<std::string> &t, size_t i) { return i < t.size() ? (float)atof(t[i].c_str()) : 0.0f; }

struct ObjectFrames {
    int type = 0, material = 0;
    std::string mesh_file;                     // MESH: the `<name>.obj` of the type line
    std::vector<Vec3> trans, rot, scale;
    std::vector<std::vector<float>> xf, inv;   // 16 floats each
};

}  // namespace

struct LoadedMesh {
    int geom_index = 0;
    std::vector<float> vertices;               // x y z
    std::vector<int> indices;                  // 3 per triangle, 0-based
};

struct pt_scene {
    std::vector<pt_material> materials;
    std::vector<ObjectFrames> objects;
    std::vector<LoadedMesh> meshes;
    float res[2] = {0, 0};
    float fov[2] = {0, 0};
    int iterations = 0;
    std::string image_name;
    std::vector<Vec3> eye, view, up;
};

namespace pth {

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof g_error, fmt, ap);
    va_end(ap);
}

void camera_basis(const pt_camera *cam, const pt_config *cfg, ptd::CamRec *out) {
    const float *E = cam->position, *C = cam->view, *U = cam->up;
    auto crossf = [](const float *x, const float *y, float *o) {
        o[0] = x[1] * y[2] - y[1] * x[2];
        o[1] = x[2] * y[0] - y[2] * x[0];
        o[2] = x[0] * y[1] - y[0] * x[1];
    };
    auto lengthf = [](const float *v) { return sqrtf((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]); };
    float A[3], B[3];
    crossf(C, U, A);
    crossf(A, C, B);
    const float CD = lengthf(C);
    const float kh = CD * tanf(cam->fov[0] * (kPI / 180.0f));
    const float kv = CD * tanf(-cam->fov[1] * (kPI / 180.0f));
    const float la = lengthf(A), lb = lengthf(B);
    for (int k = 0; k < 3; ++k) {
        out->E[k] = E[k];
        out->M[k] = E[k] + C[k];
        out->H[k] = (A[k] * kh) / la;
        out->V[k] = (B[k] * kv) / lb;
        out->Cn[k] = C[k] / CD;
        out->Ah[k] = A[k] / la;
        out->Bh[k] = B[k] / lb;
    }
    out->W = (int)cam->resolution[0];
    out->Hh = (int)cam->resolution[1];
    out->wm1 = cam->resolution[0] - 1.0f;
    out->hm1 = cam->resolution[1] - 1.0f;
    out->aperture = cfg->aperture;
    out->focal = cfg->focal_distance;
    out->camera_mode = cfg->camera_mode;
    out->antialias = cfg->antialias;
    out->row_offset = cfg->row_offset;
    out->row_stride = cfg->row_stride;
    auto magic = [](unsigned int d, unsigned int *m, unsigned int *sh) {
        unsigned int L = 0;
        while ((1u << L) < d) ++L;
        *sh = 28u + L;
        *m = (unsigned int)(((1ull << *sh) / d) + 1ull);
    };
    magic((unsigned int)(out->W > 0 ? out->W : 1), &out->mW, &out->shW);
    magic((unsigned int)(out->row_stride > 0 ? out->row_stride : 1), &out->mS, &out->shS);
}

}  // namespace pth

// Wavefront OBJ subset for MESH objects: `v x y z [w]` and `f i j k ...` (polygons fanned around their first
// vertex; `i/t/n` forms use the vertex index; negative indices count from the end).  Everything else is ignored.
// The reference never opens the file (src/scene.cpp:55-64 only tags the object), so a missing file is not an
// error: the object then stays a MESH without data and is skipped, as in the reference.  0 ok / 1 absent / -1 bad.
static int read_obj(const std::string &path, LoadedMesh *m) {
    std::ifstream in(path.c_str());
    if (!in.good()) return 1;
    std::string line;
    while (std::getline(in, line)) {
        if (!line.empty() && line[line.size() - 1] == '\r') line.erase(line.size() - 1);
        std::istringstream ls(line);
        std::string tag;
        ls >> tag;
        if (tag == "v") {
            float x = 0, y = 0, z = 0;
            ls >> x >> y >> z;
            if (ls.fail()) { pth::set_error("%s: malformed vertex line '%s'", path.c_str(), line.c_str()); return -1; }
            m->vertices.push_back(x); m->vertices.push_back(y); m->vertices.push_back(z);
        } else if (tag == "f") {
            std::vector<int> poly;
            std::string w;
            while (ls >> w) {
                const int raw = atoi(w.c_str());                   // stops at the first '/'
                const int nv = (int)(m->vertices.size() / 3);
                const int idx = raw > 0 ? raw - 1 : nv + raw;
                if (raw == 0 || idx < 0 || idx >= nv) { pth::set_error("%s: face index %s out of range (%d vertices so far)", path.c_str(), w.c_str(), nv); return -1; }
                poly.push_back(idx);
            }
            if (poly.size() < 3) { pth::set_error("%s: face with %zu vertices", path.c_str(), poly.size()); return -1; }
            for (size_t k = 1; k + 1 < poly.size(); ++k) {
                m->indices.push_back(poly[0]); m->indices.push_back(poly[k]); m->indices.push_back(poly[k + 1]);
            }
        }
    }
    if (m->indices.empty()) { pth::set_error("%s: no faces", path.c_str()); return -1; }
    return 0;
}

extern "C" {

int pt_abi_version(void) { return PTMI355_ABI_VERSION; }

const char *pt_last_error(void) { return g_error; }

void pt_config_default(pt_config *cfg) {
    if (!cfg) return;
    memset(cfg, 0, sizeof *cfg);
    cfg->max_depth = 8;
    cfg->row_stride = 1;
    cfg->ordering = 2;           // whole paths, one launch per group of iterations
    cfg->streams = 2;            // two streams per GPU: one's launch tails are filled by the other's blocks
}

int pt_build_transform(const float t[3], const float r[3], const float s[3], float xf[16], float inv[16]) {
    if (!t || !r || !s || !xf || !inv) { pth::set_error("pt_build_transform: null argument"); return PT_ERR_ARGUMENT; }
    build_pair(t, r, s, xf, inv);
    return PT_OK;
}

// ---- scene file -----------------------------------------------------------------------

static int parse_frames(std::ifstream &in, const char *k0, const char *k1, const char *k2,
                        std::vector<Vec3> &a, std::vector<Vec3> &b, std::vector<Vec3> &c) {
    int frames = 0;
    std::string line;
    std::getline(in, line);
    while (!line.empty() && in.good()) {
        std::vector<std::string> t = tokens_of(line);
        if (t.size() < 2 || t[0] != "frame" || atoi(t[1].c_str()) != frames) {
            pth::set_error("ERROR: Incorrect frame count!");
            return -1;
        }
        for (int i = 0; i < 3; ++i) {
            std::getline(in, line);
            t = tokens_of(line);
            if (t.empty()) continue;
            Vec3 v = {{num(t, 1), num(t, 2), num(t, 3)}};
            if (t[0] == k0) a.push_back(v);
            else if (t[0] == k1) b.push_back(v);
            else if (t[0] == k2) c.push_back(v);
        }
        frames++;
        std::getline(in, line);
    }
    if ((int)a.size() != frames || (int)b.size() != frames || (int)c.size() != frames) {
        pth::set_error("scene: a frame block lacks one of %s/%s/%s", k0, k1, k2);
        return -1;
    }
    return frames;
}

int pt_scene_load(const char *path, pt_scene **out) {
    if (!path || !out) { pth::set_error("pt_scene_load: null argument"); return PT_ERR_ARGUMENT; }
    *out = nullptr;
    std::ifstream in(path);
    if (!in.is_open()) { pth::set_error("pt_scene_load: cannot open %s", path); return PT_ERR_IO; }
    pt_scene *s = new pt_scene();
    bool have_camera = false;
    std::string line;
    while (in.good()) {
        std::getline(in, line);
        if (line.empty()) continue;
        std::vector<std::string> t = tokens_of(line);
        if (t.empty()) continue;
        if (t[0] == "MATERIAL") {
            const int id = t.size() > 1 ? atoi(t[1].c_str()) : -1;
            if (id != (int)s->materials.size()) {
                pth::set_error("ERROR: MATERIAL ID does not match expected number of materials (got %d, expected %zu)", id, s->materials.size());
                delete s;
                return PT_ERR_PARSE;
            }
            pt_material m;
            memset(&m, 0, sizeof m);
            for (int i = 0; i < 10; ++i) {           // exactly ten property lines, any order
                std::getline(in, line);
                std::vector<std::string> p = tokens_of(line);
                if (p.empty()) continue;
                const std::string &k = p[0];
                if (k == "RGB") { m.color[0] = num(p, 1); m.color[1] = num(p, 2); m.color[2] = num(p, 3); }
                else if (k == "SPECEX") m.specularExponent = num(p, 1);
                else if (k == "SPECRGB") { m.specularColor[0] = num(p, 1); m.specularColor[1] = num(p, 2); m.specularColor[2] = num(p, 3); }
                else if (k == "REFL") m.hasReflective = num(p, 1);
                else if (k == "REFR") m.hasRefractive = num(p, 1);
                else if (k == "REFRIOR") m.indexOfRefraction = num(p, 1);
                else if (k == "SCATTER") m.hasScatter = num(p, 1);
                else if (k == "ABSCOEFF") { m.absorptionCoefficient[0] = num(p, 1); m.absorptionCoefficient[1] = num(p, 2); m.absorptionCoefficient[2] = num(p, 3); }
                else if (k == "RSCTCOEFF") m.reducedScatterCoefficient = num(p, 1);
                else if (k == "EMITTANCE") m.emittance = num(p, 1);
            }
            s->materials.push_back(m);
        } else if (t[0] == "OBJECT") {
            const int id = t.size() > 1 ? atoi(t[1].c_str()) : -1;
            if (id != (int)s->objects.size()) {
                pth::set_error("ERROR: OBJECT ID does not match expected number of objects (got %d, expected %zu)", id, s->objects.size());
                delete s;
                return PT_ERR_PARSE;
            }
            ObjectFrames o;
            std::getline(in, line);                 // the type must be the WHOLE line (scene.cpp:48-69)
            if (line == "sphere") o.type = 0;
            else if (line == "cube") o.type = 1;
            else {
                const size_t dot = line.find('.');
                std::string ext = dot == std::string::npos ? "" : line.substr(dot + 1);
                const size_t dot2 = ext.find('.');
                if (dot2 != std::string::npos) ext = ext.substr(0, dot2);
                if (ext == "obj") { o.type = 2; o.mesh_file = line; }
                else {
                    pth::set_error("ERROR: %s is not a valid object type!", line.c_str());
                    delete s;
                    return PT_ERR_PARSE;
                }
            }
            std::getline(in, line);                 // material <id>
            std::vector<std::string> p = tokens_of(line);
            o.material = p.size() > 1 ? atoi(p[1].c_str()) : 0;
            const int frames = parse_frames(in, "TRANS", "ROTAT", "SCALE", o.trans, o.rot, o.scale);
            if (frames < 0) { delete s; return PT_ERR_PARSE; }
            for (int f = 0; f < frames; ++f) {
                std::vector<float> xf(16), inv(16);
                build_pair(o.trans[f].v, o.rot[f].v, o.scale[f].v, xf.data(), inv.data());
                o.xf.push_back(xf);
                o.inv.push_back(inv);
            }
            s->objects.push_back(o);
        } else if (t[0] == "CAMERA") {
            float fovy = 0.0f;
            for (int i = 0; i < 4; ++i) {            // exactly four property lines
                std::getline(in, line);
                std::vector<std::string> p = tokens_of(line);
                if (p.empty()) continue;
                if (p[0] == "RES") { s->res[0] = (float)atoi(p.size() > 1 ? p[1].c_str() : "0"); s->res[1] = (float)atoi(p.size() > 2 ? p[2].c_str() : "0"); }
                else if (p[0] == "FOVY") fovy = num(p, 1);
                else if (p[0] == "ITERATIONS") s->iterations = p.size() > 1 ? atoi(p[1].c_str()) : 0;
                else if (p[0] == "FILE") s->image_name = p.size() > 1 ? p[1] : "";
            }
            const int frames = parse_frames(in, "EYE", "VIEW", "UP", s->eye, s->view, s->up);
            if (frames < 0) { delete s; return PT_ERR_PARSE; }
            // fov from FOVY and the aspect ratio (scene.cpp:201-205), binary32 throughout
            const float yscaled = tanf(fovy * (kPI / 180.0f));
            const float xscaled = (yscaled * s->res[0]) / s->res[1];
            const float fovx = (atanf(xscaled) * 180.0f) / kPI;
            s->fov[0] = fovx;
            s->fov[1] = fovy;
            have_camera = true;
        }
    }
    if (!have_camera || s->eye.empty()) { pth::set_error("pt_scene_load: %s has no CAMERA block", path); delete s; return PT_ERR_PARSE; }
    if (s->objects.empty() || s->materials.empty()) { pth::set_error("pt_scene_load: %s has no objects or no materials", path); delete s; return PT_ERR_PARSE; }
    // the reference indexes every per-frame array of every object with the camera's frame number
    // (raytraceKernel.cu:184-188): an object with fewer frames than the camera is an out-of-bounds read there
    for (size_t i = 0; i < s->objects.size(); ++i)
        if (s->objects[i].xf.size() < s->eye.size()) {
            pth::set_error("pt_scene_load: object %zu has %zu frame(s), the camera has %zu", i, s->objects[i].xf.size(), s->eye.size());
            delete s;
            return PT_ERR_PARSE;
        }
    for (size_t i = 0; i < s->objects.size(); ++i)
        if (s->objects[i].material < 0 || s->objects[i].material >= (int)s->materials.size()) {
            pth::set_error("pt_scene_load: object %zu references material %d (have %zu)", i, s->objects[i].material, s->materials.size());
            delete s;
            return PT_ERR_PARSE;
        }
    // MESH objects: `<name>.obj` relative to the scene file's directory
    {
        std::string dir(path);
        const size_t slash = dir.find_last_of('/');
        dir = slash == std::string::npos ? "" : dir.substr(0, slash + 1);
        for (size_t i = 0; i < s->objects.size(); ++i) {
            if (s->objects[i].type != 2) continue;
            LoadedMesh m;
            m.geom_index = (int)i;
            const int rc = read_obj(dir + s->objects[i].mesh_file, &m);
            if (rc < 0) { delete s; return PT_ERR_PARSE; }
            if (rc == 0) s->meshes.push_back(m);
        }
    }
    *out = s;
    return PT_OK;
}

int pt_scene_mesh_count(const pt_scene *s) { return s ? (int)s->meshes.size() : 0; }

int pt_scene_mesh(const pt_scene *s, int k, pt_mesh *mesh) {
    if (!s || !mesh || k < 0 || k >= (int)s->meshes.size()) { pth::set_error("pt_scene_mesh: bad argument"); return PT_ERR_ARGUMENT; }
    const LoadedMesh &m = s->meshes[(size_t)k];
    mesh->geom_index = m.geom_index;
    mesh->vertices = m.vertices.data(); mesh->nvertices = (int)(m.vertices.size() / 3);
    mesh->indices = m.indices.data(); mesh->ntriangles = (int)(m.indices.size() / 3);
    return PT_OK;
}

void pt_scene_free(pt_scene *s) { delete s; }

int pt_scene_counts(const pt_scene *s, int *ngeoms, int *nmaterials, int *nframes, int *iterations) {
    if (!s) { pth::set_error("pt_scene_counts: null scene"); return PT_ERR_ARGUMENT; }
    if (ngeoms) *ngeoms = (int)s->objects.size();
    if (nmaterials) *nmaterials = (int)s->materials.size();
    if (nframes) *nframes = (int)s->eye.size();
    if (iterations) *iterations = s->iterations;
    return PT_OK;
}

const char *pt_scene_image_name(const pt_scene *s) { return s ? s->image_name.c_str() : ""; }

int pt_scene_flatten(const pt_scene *s, int frame, pt_geom *geoms, pt_material *materials, pt_camera *camera) {
    if (!s || frame < 0 || frame >= (int)s->eye.size()) { pth::set_error("pt_scene_flatten: frame %d out of range", frame); return PT_ERR_ARGUMENT; }
    if (materials) memcpy(materials, s->materials.data(), s->materials.size() * sizeof(pt_material));
    if (geoms) {
        for (size_t i = 0; i < s->objects.size(); ++i) {
            const ObjectFrames &o = s->objects[i];
            if (frame >= (int)o.xf.size()) { pth::set_error("pt_scene_flatten: object %zu has no frame %d", i, frame); return PT_ERR_ARGUMENT; }
            geoms[i].type = o.type;
            geoms[i].materialid = o.material;
            memcpy(geoms[i].transform, o.xf[frame].data(), 48);          // rows x,y,z
            memcpy(geoms[i].inverseTransform, o.inv[frame].data(), 48);
        }
    }
    if (camera) {
        camera->resolution[0] = s->res[0]; camera->resolution[1] = s->res[1];
        memcpy(camera->position, s->eye[frame].v, 12);
        memcpy(camera->view, s->view[frame].v, 12);
        memcpy(camera->up, s->up[frame].v, 12);
        camera->fov[0] = s->fov[0]; camera->fov[1] = s->fov[1];
    }
    return PT_OK;
}

int pt_scene_object_matrices(const pt_scene *s, int object, int frame, float xf[16], float inv[16]) {
    if (!s || object < 0 || object >= (int)s->objects.size() || frame < 0 || frame >= (int)s->objects[object].xf.size()) {
        pth::set_error("pt_scene_object_matrices: index out of range");
        return PT_ERR_ARGUMENT;
    }
    if (xf) memcpy(xf, s->objects[object].xf[frame].data(), 64);
    if (inv) memcpy(inv, s->objects[object].inv[frame].data(), 64);
    return PT_OK;
}

// ---- image output -----------------------------------------------------------------------

int pt_image_to_u8(const float *rgb, int w, int h, int divisor, float gamma, uint8_t *out) {
    if (!rgb || !out || w < 1 || h < 1) { pth::set_error("pt_image_to_u8: bad argument"); return PT_ERR_ARGUMENT; }
    const float div = (float)divisor;
    const size_t n = (size_t)w * h * 3;
    for (size_t i = 0; i < n; ++i) {
        float v = powf(rgb[i] / div, gamma) * 255.0f;     // image::applyGamma, then *255
        if (v < 0.0f) v = 0.0f;                           // utilityCore::clamp(f, 0, 255)
        else if (v > 255.0f) v = 255.0f;
        out[i] = (unsigned char)v;
    }
    return PT_OK;
}

static void put_le(std::vector<unsigned char> &b, uint32_t v, int bytes) {
    for (int i = 0; i < bytes; ++i) b.push_back((unsigned char)(v >> (8 * i)));
}

static uint32_t crc32_of(const unsigned char *p, size_t n, uint32_t crc) {
    static uint32_t table[256];
    static bool ready = false;
    if (!ready) {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            table[i] = c;
        }
        ready = true;
    }
    crc = ~crc;
    for (size_t i = 0; i < n; ++i) crc = table[(crc ^ p[i]) & 0xFF] ^ (crc >> 8);
    return ~crc;
}

static void png_chunk(std::vector<unsigned char> &f, const char *tag, const std::vector<unsigned char> &data) {
    for (int i = 3; i >= 0; --i) f.push_back((unsigned char)(data.size() >> (8 * i)));
    std::vector<unsigned char> body(tag, tag + 4);
    body.insert(body.end(), data.begin(), data.end());
    f.insert(f.end(), body.begin(), body.end());
    const uint32_t crc = crc32_of(body.data(), body.size(), 0);
    for (int i = 3; i >= 0; --i) f.push_back((unsigned char)(crc >> (8 * i)));
}

int pt_image_save(const char *path, const float *rgb, int w, int h, int divisor, float gamma) {
    if (!path) { pth::set_error("pt_image_save: null path"); return PT_ERR_ARGUMENT; }
    std::vector<uint8_t> px((size_t)w * h * 3);
    int rc = pt_image_to_u8(rgb, w, h, divisor, gamma, px.data());
    if (rc) return rc;
    std::string name(path);
    // image::saveImageRGB picks BMP when the name ends in "bmp" (optionally followed by '\r'), else PNG
    std::string probe = name;
    if (!probe.empty() && probe.back() == '\r') probe.pop_back();
    const bool bmp = probe.size() >= 3 && probe.compare(probe.size() - 3, 3, "bmp") == 0;
    std::vector<unsigned char> f;
    if (bmp) {
        const int pad = (-w * 3) & 3;
        f.push_back('B'); f.push_back('M');
        put_le(f, 14 + 40 + (w * 3 + pad) * h, 4); put_le(f, 0, 2); put_le(f, 0, 2); put_le(f, 14 + 40, 4);
        put_le(f, 40, 4); put_le(f, w, 4); put_le(f, h, 4); put_le(f, 1, 2); put_le(f, 24, 2);
        for (int i = 0; i < 6; ++i) put_le(f, 0, 4);
        for (int y = h - 1; y >= 0; --y) {               // bottom-up rows, BGR
            for (int x = 0; x < w; ++x) {
                const uint8_t *p = &px[((size_t)y * w + x) * 3];
                f.push_back(p[2]); f.push_back(p[1]); f.push_back(p[0]);
            }
            for (int k = 0; k < pad; ++k) f.push_back(0);
        }
    } else {
        const unsigned char sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
        f.assign(sig, sig + 8);
        std::vector<unsigned char> ihdr;
        for (int i = 3; i >= 0; --i) ihdr.push_back((unsigned char)((uint32_t)w >> (8 * i)));
        for (int i = 3; i >= 0; --i) ihdr.push_back((unsigned char)((uint32_t)h >> (8 * i)));
        ihdr.push_back(8); ihdr.push_back(2); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);
        png_chunk(f, "IHDR", ihdr);
        // zlib stream of stored (uncompressed) deflate blocks: filter byte 0 + RGB per row
        std::vector<unsigned char> raw;
        raw.reserve((size_t)h * (w * 3 + 1));
        for (int y = 0; y < h; ++y) {
            raw.push_back(0);
            raw.insert(raw.end(), &px[(size_t)y * w * 3], &px[(size_t)y * w * 3] + (size_t)w * 3);
        }
        std::vector<unsigned char> z;
        z.push_back(0x78); z.push_back(0x01);
        size_t pos = 0;
        uint32_t s1 = 1, s2 = 0;
        for (unsigned char c : raw) { s1 = (s1 + c) % 65521u; s2 = (s2 + s1) % 65521u; }
        do {
            const size_t n = raw.size() - pos > 65535 ? 65535 : raw.size() - pos;
            z.push_back(pos + n == raw.size() ? 1 : 0);
            z.push_back((unsigned char)(n & 0xFF)); z.push_back((unsigned char)(n >> 8));
            z.push_back((unsigned char)(~n & 0xFF)); z.push_back((unsigned char)((~n >> 8) & 0xFF));
            z.insert(z.end(), raw.begin() + pos, raw.begin() + pos + n);
            pos += n;
        } while (pos < raw.size());
        const uint32_t adler = (s2 << 16) | s1;
        for (int i = 3; i >= 0; --i) z.push_back((unsigned char)(adler >> (8 * i)));
        png_chunk(f, "IDAT", z);
        png_chunk(f, "IEND", std::vector<unsigned char>());
    }
    FILE *fp = fopen(path, "wb");
    if (!fp) { pth::set_error("pt_image_save: cannot open %s for writing", path); return PT_ERR_IO; }
    const size_t wr = fwrite(f.data(), 1, f.size(), fp);
    fclose(fp);
    if (wr != f.size()) { pth::set_error("pt_image_save: short write to %s", path); return PT_ERR_IO; }
    return PT_OK;
}

}  // extern "C"
