// pt_build.cpp -- host-only scene builders (declared in pt_build.hpp): culling bounds, k_path_w's grid, the clusters of the
// per-bounce many-primitive variant, the mesh BVH.  Pure host code: the CPU-side tests exercise it without a device.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <vector>

#include "pt_build.hpp"
#include "pt_host.hpp"

using namespace ptd;
using namespace ptk;

namespace pth {

// Conservative world-space AABB of a primitive for the culling pass (double precision, then
// inflated).  Box: the 8 transformed corners of [-.5,.5]^3.  Sphere (an ellipsoid after the affine
// map): centre +- 0.5*|row_k of the linear part|.  The inflation has to stay below RAY_BIAS_AMOUNT
// (2e-4) at scene scale, or every scattered ray would keep the wall it just left as a candidate;
// it has to exceed the few-ulp slop of the exact object-space tests (about 1e-6 at |x| ~ 10).
void world_bounds(const pt_geom &src, GeomRec *dst) {
    double lo[3], hi[3], maxabs = 0.0, maxrow = 0.0;
    const float *m = src.transform;
    for (int k = 0; k < 3; ++k) {
        const double a = m[4 * k], b = m[4 * k + 1], c3 = m[4 * k + 2], t = m[4 * k + 3];
        double ext;
        if (src.type == 0) ext = 0.5 * std::sqrt(a * a + b * b + c3 * c3);
        else ext = 0.5 * (std::fabs(a) + std::fabs(b) + std::fabs(c3));
        lo[k] = t - ext; hi[k] = t + ext;
        maxabs = std::fmax(maxabs, std::fmax(std::fabs(lo[k]), std::fabs(hi[k])));
        maxrow = std::fmax(maxrow, std::sqrt(a * a + b * b + c3 * c3));
    }
    const double infl = 3e-5 + 4e-6 * maxabs;
    if (src.type == 0) {
        // bounding sphere of the ellipsoid: centre, R = largest semi-axis <= 0.5 * largest row norm...
        // the exact bound is 0.5 * the largest singular value; 0.5 * Frobenius norm is a safe over-estimate
        // for non-uniform scales and equals 0.5*s*sqrt(3) only for... keep it tight for the common
        // uniform case: rows of equal norm and orthogonal -> R = 0.5 * row norm.
        double fro = 0.0, ortho = 0.0;
        for (int k = 0; k < 3; ++k)
            for (int j = 0; j < 3; ++j) fro += (double)m[4 * k + j] * m[4 * k + j];
        for (int k = 0; k < 3; ++k)
            for (int l = k + 1; l < 3; ++l) {
                double dotp = 0.0;
                for (int j = 0; j < 3; ++j) dotp += (double)m[4 * k + j] * m[4 * l + j];
                ortho = std::fmax(ortho, std::fabs(dotp));
            }
        double r0 = 0.0, r1 = 1e300;
        for (int k = 0; k < 3; ++k) {
            const double rn = std::sqrt((double)m[4 * k] * m[4 * k] + (double)m[4 * k + 1] * m[4 * k + 1] + (double)m[4 * k + 2] * m[4 * k + 2]);
            r0 = std::fmax(r0, rn); r1 = std::fmin(r1, rn);
        }
        const bool uniform = (r0 - r1) <= 1e-5 * r0 && ortho <= 1e-5 * r0 * r0;
        const double R = (uniform ? 0.5 * r0 : 0.5 * std::sqrt(fro)) + infl;
        dst->bmin[0] = m[3]; dst->bmin[1] = m[7]; dst->bmin[2] = m[11];
        dst->bmin[3] = std::nextafterf((float)(R * R * (1.0 + 1e-5)), INFINITY);
        dst->bmax[0] = dst->bmax[1] = dst->bmax[2] = 0.0f;
        dst->bmax[3] = std::nextafterf((float)R, INFINITY);
    } else {
        for (int k = 0; k < 3; ++k) {
            dst->bmin[k] = std::nextafterf((float)(lo[k] - infl), -INFINITY);
            dst->bmax[k] = std::nextafterf((float)(hi[k] + infl), INFINITY);
        }
        dst->bmin[3] = dst->bmax[3] = 0.0f;
    }
    // the sphere test reports the point 1e-4 (object space, along the ray) in front of the surface
    dst->slack = src.type == 0 ? (float)(1.5e-4 * maxrow + 1e-5) : 1e-5f;
}


// ---- k_path_w: uniform grid over the small analytic primitives (GridArgs, pt_kernels.hpp) ---------------------
// Cell size: about `density` cells per small primitive over the box of their bounds (cubic cells, at most kGridMaxCells).
// A primitive whose bound (grown by the margin below) meets more than kGridBigCells cells is BIG: it stays out of the
// grid and every ray tests its bound.  Too many big ones (each costs every ray a bound test, and more than 8 hit
// candidates of a ray overflow its list): the grid is rebuilt coarser, until few are left or the grid is one cell.
// Margin: cells list a primitive over its conservative bound grown by 2e-3 cell sizes + 1e-5 of the largest
// coordinate.  The walk's float error (boundaries rebuilt from integer indices, distances as (b - o) * 1/d) stays
// below 1e-5 of the scene's diagonal for rays starting within `reach` = 8 diagonals of the centre, a tenth of the margin;
// farther or non-finite rays are not walked at all (the kernel gives them the reference loop).
void build_grid(const std::vector<GeomRec> &g, int G, int density, size_t max_bytes, bool wide_refs, GridBuild *out) {
    const uint64_t max_cells = wide_refs ? kGridMaxCellsWide : kGridMaxCells;
    const size_t max_refs = wide_refs ? (size_t)kGridMaxRefsWide : (size_t)8191;
    struct Box { double lo[3], hi[3]; int id; bool sphere; };
    std::vector<Box> prims;
    double maxabs = 0.0;
    for (int i = 0; i < G; ++i) {
        if (g[i].type != 0 && g[i].type != 1) continue;                  // MESH without data: in no list, like the reference's empty branch
        Box b;
        b.id = i; b.sphere = g[i].type == 0;
        for (int k = 0; k < 3; ++k) {
            b.lo[k] = b.sphere ? (double)g[i].bmin[k] - (double)g[i].bmax[3] : (double)g[i].bmin[k];
            b.hi[k] = b.sphere ? (double)g[i].bmin[k] + (double)g[i].bmax[3] : (double)g[i].bmax[k];
            maxabs = std::fmax(maxabs, std::fmax(std::fabs(b.lo[k]), std::fabs(b.hi[k])));
        }
        prims.push_back(b);
    }
    GridArgs ga;
    memset(&ga, 0, sizeof ga);
    std::vector<char> big(prims.size(), 0);
    std::vector<std::vector<uint32_t>> lists;                             // per cell: primitive | flags << 24
    double dens = density > 0 ? (double)density : density < 0 ? 0.125 : 4.0;
    int n[3] = {1, 1, 1};
    double gmin[3] = {0, 0, 0}, h[3] = {1, 1, 1};
    for (int attempt = 0; attempt < 12; ++attempt) {
        std::fill(big.begin(), big.end(), 0);
        size_t nbig = 0;
        for (int pass = 0; pass < 3; ++pass) {                             // bounds of the small ones -> cells -> who is big -> again
            double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
            size_t m = 0;
            for (size_t q = 0; q < prims.size(); ++q) {
                if (big[q]) continue;
                m++;
                for (int k = 0; k < 3; ++k) { lo[k] = std::fmin(lo[k], prims[q].lo[k]); hi[k] = std::fmax(hi[k], prims[q].hi[k]); }
            }
            if (m == 0) { for (int k = 0; k < 3; ++k) { lo[k] = -1.0; hi[k] = 1.0; } }
            double ext[3], vol = 1.0, emax = 0.0;
            for (int k = 0; k < 3; ++k) { ext[k] = hi[k] - lo[k]; emax = std::fmax(emax, ext[k]); }
            if (!(emax > 0.0)) emax = 1.0;
            for (int k = 0; k < 3; ++k) { if (ext[k] < 1e-3 * emax) ext[k] = 1e-3 * emax; vol *= ext[k]; }
            double cell = std::cbrt(vol / (dens * (double)(m > 0 ? m : 1)));
            for (;;) {
                uint64_t total = 1;
                for (int k = 0; k < 3; ++k) { n[k] = (int)std::ceil(ext[k] / cell); if (n[k] < 1) n[k] = 1; total *= (uint64_t)n[k]; }
                if (total <= max_cells) break;
                cell *= 1.1;
            }
            for (int k = 0; k < 3; ++k) {
                const double pad = 4e-3 * cell + 4e-5 * maxabs;               // the grid's box reaches beyond every listed bound and its margin
                h[k] = (ext[k] + 2.0 * pad) / (double)n[k];
                gmin[k] = lo[k] - pad;
                // the kernel's numbers are these floats
                h[k] = (double)(float)h[k]; gmin[k] = (double)(float)gmin[k];
            }
            nbig = 0;
            for (size_t q = 0; q < prims.size(); ++q) {
                uint64_t cellsq = 1;
                for (int k = 0; k < 3; ++k) {
                    const double mg = 2e-3 * h[k] + 1e-5 * maxabs;
                    int c0 = (int)std::floor((prims[q].lo[k] - mg - gmin[k]) / h[k]), c1 = (int)std::floor((prims[q].hi[k] + mg - gmin[k]) / h[k]);
                    c0 = std::max(c0, 0); c1 = std::min(c1, n[k] - 1);
                    cellsq *= (uint64_t)(c1 >= c0 ? c1 - c0 + 1 : 1);
                }
                // a primitive once found big stays big for this attempt (the grid only gets finer as they leave)
                if (cellsq > (uint64_t)kGridBigCells || big[q]) { big[q] = 1; nbig++; }
            }
        }
        if (nbig <= 8 || dens < 0.02) break;
        dens *= 0.5;                                                       // coarser cells: fewer primitives are big
    }
    // the lists
    const uint32_t ncells = (uint32_t)((uint64_t)n[0] * (uint64_t)n[1] * (uint64_t)n[2]);
    lists.assign(ncells, {});
    std::vector<uint32_t> bigs;
    for (size_t q = 0; q < prims.size(); ++q) {
        if (big[q]) { bigs.push_back((uint32_t)prims[q].id); continue; }
        int c0[3], c1[3];
        for (int k = 0; k < 3; ++k) {
            const double mg = 2e-3 * h[k] + 1e-5 * maxabs;
            c0[k] = (int)std::floor((prims[q].lo[k] - mg - gmin[k]) / h[k]); c1[k] = (int)std::floor((prims[q].hi[k] + mg - gmin[k]) / h[k]);
            c0[k] = std::min(std::max(c0[k], 0), n[k] - 1); c1[k] = std::min(std::max(c1[k], 0), n[k] - 1);
        }
        for (int z = c0[2]; z <= c1[2]; ++z)
            for (int y = c0[1]; y <= c1[1]; ++y)
                for (int x = c0[0]; x <= c1[0]; ++x) {
                    uint32_t flags = prims[q].sphere ? 0x80u : 0u;
                    if (x == c0[0]) flags |= 1u; if (y == c0[1]) flags |= 2u; if (z == c0[2]) flags |= 4u;
                    if (x == c1[0]) flags |= 8u; if (y == c1[1]) flags |= 16u; if (z == c1[2]) flags |= 32u;
                    lists[(size_t)x + (size_t)n[0] * ((size_t)y + (size_t)n[1] * (size_t)z)].push_back((uint32_t)prims[q].id | (flags << 24));
                }
    }
    size_t nrefs = 0;
    for (uint32_t cI = 0; cI < ncells; ++cI) nrefs += lists[cI].size();
    // narrow: cells[ncells] (first ref | count << 16), refs[nrefs] 16 bits (id | flags << 8), big[nbig] bytes
    // wide:   cells[ncells] (first ref, 0xFFFFFFFF = empty), refs[nrefs] 32 bits (id | flags << 24), big[nbig] 32 bits
    const size_t nrefs_even = (nrefs + 1) & ~(size_t)1;
    size_t bytes = wide_refs ? (size_t)ncells * 4 + nrefs * 4 + bigs.size() * 4 : (size_t)ncells * 4 + nrefs_even * 2 + bigs.size();
    bytes = (bytes + 15) & ~(size_t)15;
    if (((!wide_refs && bytes > max_bytes) || nrefs > max_refs) && ncells > 1) {   // no room beside the tables in LDS, or beyond the reference index of a pair entry: coarser
        const int next = (int)std::floor(dens * 0.7);
        if (next >= 1 && density != 1) { build_grid(g, G, next, max_bytes, wide_refs, out); return; }
        if (density != -1) { build_grid(g, G, -1, max_bytes, wide_refs, out); return; }   // -1: one cell per 8 primitives, the coarsest the builder makes
    }
    out->blob.assign(bytes, 0);
    {
        uint32_t *cellrec = reinterpret_cast<uint32_t *>(out->blob.data());
        uint16_t *r16 = reinterpret_cast<uint16_t *>(out->blob.data() + (size_t)ncells * 4);
        uint32_t *r32 = reinterpret_cast<uint32_t *>(out->blob.data() + (size_t)ncells * 4);
        size_t at = 0;
        for (uint32_t cI = 0; cI < ncells; ++cI) {
            const std::vector<uint32_t> &l = lists[cI];
            cellrec[cI] = wide_refs ? (l.empty() ? 0xFFFFFFFFu : (uint32_t)at) : ((uint32_t)at | ((uint32_t)l.size() << 16));
            for (size_t k = 0; k < l.size(); ++k) {
                uint32_t flags = l[k] >> 24;
                if (k + 1 == l.size()) flags |= 0x40u;                       // the cell's last reference
                if (wide_refs) r32[at++] = (l[k] & 0xFFFFFFu) | (flags << 24);
                else r16[at++] = (uint16_t)((l[k] & 0xFFu) | (flags << 8));
            }
        }
        if (wide_refs) { uint32_t *b32 = r32 + nrefs; for (size_t k = 0; k < bigs.size(); ++k) b32[k] = bigs[k]; }
        else { unsigned char *b8 = out->blob.data() + (size_t)ncells * 4 + nrefs_even * 2; for (size_t k = 0; k < bigs.size(); ++k) b8[k] = (unsigned char)bigs[k]; }
    }
    double diag2 = 0.0;
    for (int k = 0; k < 3; ++k) {
        ga.gmin[k] = (float)gmin[k]; ga.h[k] = (float)h[k]; ga.inv_h[k] = (float)(1.0 / h[k]); ga.n[k] = n[k];
        ga.centre[k] = (float)(gmin[k] + 0.5 * h[k] * n[k]);
        diag2 += (h[k] * n[k]) * (h[k] * n[k]);
    }
    ga.reach = (float)(8.0 * std::sqrt(diag2));
    {   // the survivors' bins by walk length: short <= 0.75 x the mean cell count per axis, middle <= 1.4 x (swept on configs[3]: DESIGN.md appendix B)
        const double navg = (n[0] + n[1] + n[2]) / 3.0;
#ifndef PT_BIN1_PCT
#define PT_BIN1_PCT 75
#endif
#ifndef PT_BIN2_PCT
#define PT_BIN2_PCT 140
#endif
        const uint32_t b1 = (uint32_t)std::max(2.0, std::floor(0.01 * PT_BIN1_PCT * navg + 0.5));
        ga.bin1 = b1; ga.bin2 = std::max(b1 + 1u, (uint32_t)std::floor(0.01 * PT_BIN2_PCT * navg + 0.5));
    }
    ga.ncells = ncells; ga.nrefs = (uint32_t)nrefs; ga.nbig = (uint32_t)bigs.size();
    ga.blob_bytes = (uint32_t)bytes;
    ga.blob = nullptr;
    out->ga = ga;
}


// ---- k_bounce_seg<WIDE>: spatial clusters for the two-level culling of 33..256 primitives ---------------------
// The primitives of each type are sorted into clusters by recursive median splits of their centres along the widest axis
// (left parts whole numbers of clusters); a cluster record = the union of its members' bounds + first member / count.
bool build_clusters(const std::vector<GeomRec> &g, int G, int preferred_size, ClusterBuild *out) {
    std::vector<ClusterRec> recs;
    std::vector<unsigned char> ids;
    int csize = preferred_size > 0 ? preferred_size : PT_CLUSTER;
    if (csize < 1) csize = 1;
    if (csize > kClusterMax) csize = kClusterMax;
    for (; csize <= kClusterMax; ++csize) {                // the per-lane cluster mask: 64 bits in all
        int nb = 0, ns = 0;
        for (int i = 0; i < G; ++i) { if (g[i].type == 1) nb++; else if (g[i].type == 0) ns++; }
        const int cb = (nb + csize - 1) / csize, cs = (ns + csize - 1) / csize;
        if (cb + cs <= 64) break;
    }
    if (csize > kClusterMax) csize = kClusterMax;
    out->nbc = out->nsc = 0;
    for (int pass = 0; pass < 2; ++pass) {
        const int type = pass == 0 ? 1 : 0;
        std::vector<int> prim;
        for (int i = 0; i < G; ++i) if (g[i].type == type) prim.push_back(i);
        auto lo_of = [&](int i, int k) { return type == 1 ? g[i].bmin[k] : g[i].bmin[k] - g[i].bmax[3]; };
        auto hi_of = [&](int i, int k) { return type == 1 ? g[i].bmax[k] : g[i].bmin[k] + g[i].bmax[3]; };
        std::vector<std::pair<int, int>> stack{{0, (int)prim.size()}};
        std::vector<std::pair<int, int>> leaves;
        while (!stack.empty()) {
            const std::pair<int, int> r = stack.back();
            stack.pop_back();
            const int first = r.first, count = r.second;
            if (count <= 0) continue;
            if (count <= csize) { leaves.push_back(r); continue; }
            int axis = 0;
            float ext = -1.0f;
            for (int k = 0; k < 3; ++k) {
                float cmin = 3e38f, cmax = -3e38f;
                for (int q = first; q < first + count; ++q) { const float cc = lo_of(prim[q], k) + hi_of(prim[q], k); cmin = std::fmin(cmin, cc); cmax = std::fmax(cmax, cc); }
                if (cmax - cmin > ext) { ext = cmax - cmin; axis = k; }
            }
            int half = ((count / 2 + csize - 1) / csize) * csize;
            if (half >= count) half = count - csize > 0 ? count - csize : count / 2;
            std::nth_element(prim.begin() + first, prim.begin() + first + half, prim.begin() + first + count, [&](int x, int y) {
                const float cx = lo_of(x, axis) + hi_of(x, axis), cy = lo_of(y, axis) + hi_of(y, axis);
                return cx < cy || (cx == cy && x < y);
            });
            stack.push_back({first + half, count - half});
            stack.push_back({first, half});
        }
        for (const std::pair<int, int> &lf : leaves) {
            ClusterRec r;
            for (int k = 0; k < 3; ++k) { r.bmin[k] = 3e38f; r.bmax[k] = -3e38f; }
            r.first = (int)ids.size(); r.count = lf.second;
            for (int q = lf.first; q < lf.first + lf.second; ++q) {
                ids.push_back((unsigned char)prim[q]);
                for (int k = 0; k < 3; ++k) { r.bmin[k] = std::fmin(r.bmin[k], lo_of(prim[q], k)); r.bmax[k] = std::fmax(r.bmax[k], hi_of(prim[q], k)); }
            }
            recs.push_back(r);
            if (pass == 0) out->nbc++; else out->nsc++;
        }
    }
    if (out->nbc + out->nsc > 64) return false;
    const size_t idbytes = (ids.size() + 15) & ~(size_t)15;
    out->blob.assign(recs.size() * sizeof(ClusterRec) + idbytes, 0);
    memcpy(out->blob.data(), recs.data(), recs.size() * sizeof(ClusterRec));
    memcpy(out->blob.data() + recs.size() * sizeof(ClusterRec), ids.data(), ids.size());
    return true;
}

// ---- MESH: threaded BVH over the triangles of one mesh (object space), built at upload ----------------------
// Cuts chosen by the surface-area heuristic over the centroid order of each axis, <= 4 triangles per leaf, nodes in depth-first
// order with skip links per direction octant (traversal needs no stack and visits the near child first).  Boxes are the exact float min/max of the member vertices,
// inflated by 1e-5 * (1 + largest |coordinate|): the slab test adds its own relative margins (cull_box).
struct MeshBuild {
    static constexpr int kSweepDepth = 48;      // lopsided cuts (degenerate meshes) end here: below, medians keep the recursion at log n
    static constexpr int kSweepMax = 1 << 16;   // nodes above this many triangles are cut at the median (a sweep of every split costs n log n per node)
    const float *v;
    const int *idx;
    std::vector<int> order;                  // triangle permutation (leaf ranges index into it)
    std::vector<MeshNode> nodes;
    std::vector<float> cen;                  // 3 per triangle
    void bounds(int first, int count, float lo[3], float hi[3]) const {
        for (int k = 0; k < 3; ++k) { lo[k] = 3e38f; hi[k] = -3e38f; }
        for (int i = first; i < first + count; ++i)
            for (int c = 0; c < 3; ++c) {
                const float *p = v + 3 * idx[3 * order[i] + c];
                for (int k = 0; k < 3; ++k) { lo[k] = std::fmin(lo[k], p[k]); hi[k] = std::fmax(hi[k], p[k]); }
            }
    }
    int emit(int first, int count, int depth = 0) {
        const int id = (int)nodes.size();
        nodes.emplace_back();
        float lo[3], hi[3];
        bounds(first, count, lo, hi);
        float maxabs = 0.0f;
        for (int k = 0; k < 3; ++k) maxabs = std::fmax(maxabs, std::fmax(std::fabs(lo[k]), std::fabs(hi[k])));
        const float infl = 1e-5f * (1.0f + maxabs);
        for (int k = 0; k < 3; ++k) { nodes[id].bmin[k] = lo[k] - infl; nodes[id].bmax[k] = hi[k] + infl; }
        nodes[id].far = 0;
        for (int o = 0; o < 8; ++o) nodes[id].skip[o] = -1;
        if (count <= 4) { nodes[id].leaf = first | (count << 27); return id; }
        nodes[id].leaf = -1;
        int cut_axis = 0;
        // where to cut: the surface-area heuristic over every split of the centroid order of each axis (cost = area x count of the
        // two sides), ties to the lower axis and the earlier split; above kSweepMax triangles the median along the widest axis
        int half = count / 2;
        auto by_axis = [&](int axis) {
            return [this, axis](int x, int y) { return cen[3 * x + axis] < cen[3 * y + axis] || (cen[3 * x + axis] == cen[3 * y + axis] && x < y); };
        };
        if (count <= kSweepMax && depth < kSweepDepth) {
            int best_axis = 0;
            double best_cost = 1e300;
            std::vector<double> right(count);
            for (int k = 0; k < 3; ++k) {
                std::sort(order.begin() + first, order.begin() + first + count, by_axis(k));
                float lo2[3] = {3e38f, 3e38f, 3e38f}, hi2[3] = {-3e38f, -3e38f, -3e38f};
                auto grow = [&](int i) {
                    for (int c = 0; c < 3; ++c) {
                        const float *p = v + 3 * idx[3 * order[first + i] + c];
                        for (int q = 0; q < 3; ++q) { lo2[q] = std::fmin(lo2[q], p[q]); hi2[q] = std::fmax(hi2[q], p[q]); }
                    }
                };
                auto area = [&]() {
                    const double dx = (double)hi2[0] - lo2[0], dy = (double)hi2[1] - lo2[1], dz = (double)hi2[2] - lo2[2];
                    return dx * dy + dy * dz + dz * dx;
                };
                for (int i = count - 1; i >= 1; --i) { grow(i); right[i] = area(); }
                for (int q = 0; q < 3; ++q) { lo2[q] = 3e38f; hi2[q] = -3e38f; }
                for (int i = 1; i < count; ++i) {
                    grow(i - 1);
                    const double cost = area() * i + right[i] * (count - i);
                    if (cost < best_cost) { best_cost = cost; best_axis = k; half = i; }
                }
            }
            std::sort(order.begin() + first, order.begin() + first + count, by_axis(best_axis));
            cut_axis = best_axis;
        } else {
            int axis = 0;
            float ext = -1.0f;
            for (int k = 0; k < 3; ++k) {
                float cmin = 3e38f, cmax = -3e38f;
                for (int i = first; i < first + count; ++i) { cmin = std::fmin(cmin, cen[3 * order[i] + k]); cmax = std::fmax(cmax, cen[3 * order[i] + k]); }
                if (cmax - cmin > ext) { ext = cmax - cmin; axis = k; }
            }
            std::nth_element(order.begin() + first, order.begin() + first + half, order.begin() + first + count, by_axis(axis));
            cut_axis = axis;
        }
        emit(first, half, depth + 1);                                   // the lower child: the next node
        const int right = emit(first + half, count - half, depth + 1);
        nodes[id].far = cut_axis | (right << 2);
        return id;
    }
};

// [MeshNode x nnodes | MeshTri x ntris] for one mesh; *tri_offset = byte offset of the triangles
std::vector<unsigned char> build_mesh_blob(const HostMesh &hm, uint32_t *tri_offset) {
    MeshBuild mb;
    mb.v = hm.v.data(); mb.idx = hm.idx.data();
    const int nt = (int)(hm.idx.size() / 3);
    mb.order.resize(nt); mb.cen.resize((size_t)3 * nt);
    for (int t = 0; t < nt; ++t) {
        mb.order[t] = t;
        for (int k = 0; k < 3; ++k)
            mb.cen[3 * t + k] = (hm.v[3 * hm.idx[3 * t] + k] + hm.v[3 * hm.idx[3 * t + 1] + k] + hm.v[3 * hm.idx[3 * t + 2] + k]) * (1.0f / 3.0f);
    }
    mb.emit(0, nt);
    // where a ray of direction octant o goes on after a node: the near child's successor is the far child, the far child's is
    // its parent's (parents precede their children in the node order)
    for (size_t n = 0; n < mb.nodes.size(); ++n) {
        if (mb.nodes[n].leaf >= 0) continue;
        const int axis = mb.nodes[n].far & 3, lo_child = (int)n + 1, hi_child = mb.nodes[n].far >> 2;
        for (int o = 0; o < 8; ++o) {
            const bool neg = (o >> axis) & 1;
            const int near_c = neg ? hi_child : lo_child, far_c = neg ? lo_child : hi_child;
            mb.nodes[near_c].skip[o] = far_c;
            mb.nodes[far_c].skip[o] = mb.nodes[n].skip[o];
        }
    }
    const size_t nn = mb.nodes.size();
    *tri_offset = (uint32_t)(nn * sizeof(MeshNode));
    std::vector<unsigned char> blob(nn * sizeof(MeshNode) + (size_t)nt * sizeof(MeshTri), 0);
    memcpy(blob.data(), mb.nodes.data(), mb.nodes.size() * sizeof(MeshNode));
    MeshTri *tris = reinterpret_cast<MeshTri *>(blob.data() + *tri_offset);
    for (int i = 0; i < nt; ++i) {
        const int t = mb.order[i];
        const float *p0 = &hm.v[3 * hm.idx[3 * t]], *p1 = &hm.v[3 * hm.idx[3 * t + 1]], *p2 = &hm.v[3 * hm.idx[3 * t + 2]];
        const f3 v0 = mk(p0[0], p0[1], p0[2]);
        const f3 e1 = mk(p1[0], p1[1], p1[2]) - v0, e2 = mk(p2[0], p2[1], p2[2]) - v0;     // the kernels' own float subtraction
        const f3 ng = cross(e1, e2);
        MeshTri &r = tris[i];
        r.v0[0] = v0.x; r.v0[1] = v0.y; r.v0[2] = v0.z; r.index = t;
        r.e1[0] = e1.x; r.e1[1] = e1.y; r.e1[2] = e1.z;
        r.e2[0] = e2.x; r.e2[1] = e2.y; r.e2[2] = e2.z;
        r.ng[0] = ng.x; r.ng[1] = ng.y; r.ng[2] = ng.z;
    }
    return blob;
}

// conservative world-space AABB of a mesh primitive (double precision, inflated like the cubes')
void mesh_world_bounds(const pt_geom &src, const HostMesh &hm, GeomRec *dst) {
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300}, maxabs = 0.0, maxrow = 0.0;
    const float *m = src.transform;
    for (size_t i = 0; i + 2 < hm.v.size(); i += 3)
        for (int k = 0; k < 3; ++k) {
            const double w = (double)m[4 * k] * hm.v[i] + (double)m[4 * k + 1] * hm.v[i + 1] + (double)m[4 * k + 2] * hm.v[i + 2] + (double)m[4 * k + 3];
            lo[k] = std::fmin(lo[k], w); hi[k] = std::fmax(hi[k], w);
        }
    for (int k = 0; k < 3; ++k) {
        maxabs = std::fmax(maxabs, std::fmax(std::fabs(lo[k]), std::fabs(hi[k])));
        maxrow = std::fmax(maxrow, std::sqrt((double)m[4 * k] * m[4 * k] + (double)m[4 * k + 1] * m[4 * k + 1] + (double)m[4 * k + 2] * m[4 * k + 2]));
    }
    const double infl = 3e-5 + 4e-6 * maxabs;
    for (int k = 0; k < 3; ++k) {
        dst->bmin[k] = std::nextafterf((float)(lo[k] - infl), -INFINITY);
        dst->bmax[k] = std::nextafterf((float)(hi[k] + infl), INFINITY);
    }
    dst->slack = (float)(1.5e-4 * maxrow + 1e-5);        // the hit point sits 1e-4 (object space) in front of the surface
}

// ---- pt_debug_grid_probe: k_path_w's spatial index on the host ------------------------------------------------
// Builds the grid exactly as pt_upload_scene does (narrow references up to 256 primitives, wide ones beyond) and walks the
// rays with the kernel's own functions.  out_sets: ceil(G / 32) words per ray (at least 8), bit p = primitive p has its bound tested.
int grid_probe(const pt_geom *geoms, int G, int density, const float *rays, int nrays, uint32_t *out_sets, uint32_t *out_info) {
    if (!geoms || G < 1 || G > (1 << 24) || !rays || nrays < 0 || !out_sets || !out_info) { set_error("pt_debug_grid_probe: bad argument"); return PT_ERR_ARGUMENT; }
    std::vector<GeomRec> g(G);
    for (int i = 0; i < G; ++i) { memset(&g[i], 0, sizeof(GeomRec)); g[i].type = geoms[i].type; world_bounds(geoms[i], &g[i]); }
    const bool wide = G > 256;
    GridBuild gb;
    build_grid(g, G, density, 24u * 1024u, wide, &gb);
    const GridArgs &ga = gb.ga;
    const uint32_t *cells = reinterpret_cast<const uint32_t *>(gb.blob.data());
    const uint16_t *refs16 = reinterpret_cast<const uint16_t *>(cells + ga.ncells);
    const uint32_t *refs32 = cells + ga.ncells;
    const unsigned char *bigs8 = reinterpret_cast<const unsigned char *>(refs16 + ((ga.nrefs + 1u) & ~1u));
    const uint32_t *bigs32 = refs32 + ga.nrefs;
    const size_t words = (size_t)std::max(8, (G + 31) / 32);
    uint32_t dups = 0, unwalked = 0, maxtrips = 0, est_worst = 0;
    uint64_t est_abs = 0;
    uint64_t trips_total = 0, entries = 0, news = 0;
    for (int r = 0; r < nrays; ++r) {
        uint32_t *set = out_sets + (size_t)r * words;
        for (size_t k = 0; k < words; ++k) set[k] = 0u;
        for (uint32_t k = 0; k < ga.nbig; ++k) { const uint32_t p = wide ? bigs32[k] : bigs8[k]; set[p >> 5] |= 1u << (p & 31); }
        const f3 o = mk(rays[6 * r], rays[6 * r + 1], rays[6 * r + 2]), d = mk(rays[6 * r + 3], rays[6 * r + 4], rays[6 * r + 5]);
        if (!grid_walk_sane(ga, o, d)) { unwalked++; for (size_t k = 0; k < words; ++k) set[k] = 0xFFFFFFFFu; continue; }     // the kernel tests every primitive
        auto grcp = [](float x) { const float ax = std::fabs(x); const float gg = ax < 1e-30f ? std::copysign(1e-30f, x) : x; return 1.0f / gg; };
        const f3 inv = mk(grcp(d.x), grcp(d.y), grcp(d.z));
        GridWalk w = grid_walk_begin(ga, o, d, inv, true);
        const uint32_t cap = (uint32_t)(ga.n[0] + ga.n[1] + ga.n[2]) + 2u;
        uint32_t trips = 0;
        while (w.walking && trips < cap) {
            trips++;
            const uint32_t rec = cells[grid_walk_cell(w)];
            if (wide ? rec != 0xFFFFFFFFu : (rec >> 16) != 0u) {
                entries++;
                for (uint32_t k = wide ? rec : (rec & 0xFFFFu);; ++k) {     // the kernel's CELLS stage: one reference per entry, the next one re-queued
                    const uint32_t ref = wide ? refs32[k] : (uint32_t)refs16[k];
                    const uint32_t flags = wide ? ref >> 24 : ref >> 8;
                    if (grid_flags_new(flags, w.emask)) {
                        const uint32_t p = wide ? (ref & 0xFFFFFFu) : (ref & 0xFFu);
                        if (set[p >> 5] & (1u << (p & 31))) dups++;
                        set[p >> 5] |= 1u << (p & 31);
                        news++;
                    }
                    if (flags & 0x40u) break;
                }
            }
            grid_walk_step(w);
        }
        if (w.walking) dups += 1000000u;                                   // the step bound must never cut a walk short
        {   // the length estimate the survivors are sorted by: never short of the walk by more than the ties can explain
            const uint32_t est = grid_walk_length(ga, o, d, inv);
            const uint32_t err = est > trips ? est - trips : trips - est;
            if (err > est_worst) est_worst = err;
            est_abs += err;
        }
        trips_total += trips;
        if (trips > maxtrips) maxtrips = trips;
    }
    out_info[0] = ga.ncells; out_info[1] = ga.nrefs; out_info[2] = ga.nbig; out_info[3] = dups; out_info[4] = unwalked;
    out_info[5] = (uint32_t)ga.n[0]; out_info[6] = (uint32_t)ga.n[1]; out_info[7] = (uint32_t)ga.n[2];
    out_info[8] = (uint32_t)(nrays ? trips_total / (uint64_t)nrays : 0); out_info[9] = maxtrips;
    out_info[10] = (uint32_t)(nrays ? (100 * entries) / (uint64_t)nrays : 0); out_info[11] = (uint32_t)(nrays ? (100 * news) / (uint64_t)nrays : 0);
    out_info[12] = ga.blob_bytes; out_info[13] = ga.bin1; out_info[14] = ga.bin2;
    out_info[15] = est_worst; out_info[16] = (uint32_t)(nrays ? (100 * est_abs) / (uint64_t)nrays : 0);
    return PT_OK;
}

// ---- pt_debug_fan_probe: the cone test of k_path_w's camera groups on the host --------------------------------------
// rays: nfans x 64 x (origin, direction); out_sets: max(8, ceil(G / 32)) words per fan, bit p = the cone of the fan meets primitive
// p's bound (all ones when the fan gets no cone: the kernel then walks the grid); out_info[0] = fans that got a cone.
int fan_probe(const pt_geom *geoms, int G, const float *rays, int nfans, uint32_t *out_sets, uint32_t *out_info) {
    if (!geoms || G < 1 || !rays || nfans < 0 || !out_sets || !out_info) { set_error("pt_debug_fan_probe: bad argument"); return PT_ERR_ARGUMENT; }
    std::vector<GeomRec> g(G);
    for (int i = 0; i < G; ++i) { memset(&g[i], 0, sizeof(GeomRec)); g[i].type = geoms[i].type; world_bounds(geoms[i], &g[i]); }
    const size_t words = (size_t)std::max(8, (G + 31) / 32);
    uint32_t cones = 0;
    for (int f = 0; f < nfans; ++f) {
        const float *r = rays + (size_t)f * 64 * 6;
        uint32_t *set = out_sets + (size_t)f * words;
        const f3 e = mk(r[0], r[1], r[2]);
        const f3 df = mk(r[3], r[4], r[5]), dl = mk(r[63 * 6 + 3], r[63 * 6 + 4], r[63 * 6 + 5]);
        const f3 ax = fan_axis(df, dl), nr = fan_normal(df, dl);
        float md = 1.0f, mo = 0.0f;
        bool same = true;
        for (int l = 0; l < 64; ++l) {
            const float *q = r + 6 * l;
            const float dt = __builtin_fmaf(q[5], ax.z, __builtin_fmaf(q[4], ax.y, q[3] * ax.x));
            const float of = std::fabs(__builtin_fmaf(q[5], nr.z, __builtin_fmaf(q[4], nr.y, q[3] * nr.x)));
            if (!(dt == dt) || !(of == of)) { md = -1.0f; continue; }
            md = std::fmin(md, dt); mo = std::fmax(mo, of);
            same = same && q[0] == e.x && q[1] == e.y && q[2] == e.z;
        }
        FanCone cone;
        if (!(fan_finish(e, ax, md, nr, mo, cone) && same)) { for (size_t k = 0; k < words; ++k) set[k] = 0xFFFFFFFFu; continue; }
        cones++;
        for (size_t k = 0; k < words; ++k) set[k] = 0u;
        for (int p = 0; p < G; ++p)
            if ((g[p].type == 0 || g[p].type == 1) && fan_meets(g[p].bmin, g[p].bmax, g[p].type == 0, cone)) set[p >> 5] |= 1u << (p & 31);
    }
    out_info[0] = cones;
    return PT_OK;
}

}  // namespace pth
