// pt_build.hpp -- host-only scene builders of libptmi355.so: conservative culling bounds, k_path_w's uniform grid,
// the spatial clusters of the per-bounce many-primitive variant, the threaded BVH of a MESH primitive.  Nothing here
// touches a device; pt_api.hip uploads what these return, and the CPU-side tests reach them through
// pt_debug_grid_probe (tests/test_grid_cpu.py) without a GPU.
#pragma once

#include <cstdint>
#include <vector>

#include "../../include/ptmi355.h"
#include "pt_kernels.hpp"

namespace pth {

// conservative world-space bound of an analytic primitive for the culling passes (GeomRec::bmin / bmax / slack)
void world_bounds(const pt_geom &src, ptd::GeomRec *dst);

// k_path_w's uniform grid over the small analytic primitives (layout: GridArgs, pt_kernels.hpp).
//   wide_refs = false: 16-bit references, byte primitive ids (<= 256 primitives), at most kGridMaxCells cells and 8 191 references,
//                      coarsened until the blob fits `max_bytes` (it is staged in LDS beside the geometry table)
//   wide_refs = true : 32-bit references (24-bit ids), at most kGridMaxCellsWide cells and kGridMaxRefsWide references; `max_bytes`
//                      is not a limit (a blob that does not fit in LDS is read from global memory)
struct GridBuild { ptk::GridArgs ga; std::vector<unsigned char> blob; };
void build_grid(const std::vector<ptd::GeomRec> &g, int G, int density, size_t max_bytes, bool wide_refs, GridBuild *out);

// two-level culling of k_bounce_seg<WIDE> (33..256 primitives): spatial clusters of <= kClusterMax primitives of one type.
// Returns false when more than 64 clusters would be needed (the per-lane cluster mask has 64 bits).
struct ClusterBuild { std::vector<unsigned char> blob; int nbc = 0, nsc = 0; };
bool build_clusters(const std::vector<ptd::GeomRec> &g, int G, int preferred_size, ClusterBuild *out);

// MESH primitives (pt_set_meshes): host copy, device blob [MeshNode x nnodes | MeshTri x ntris], world bound
struct HostMesh { int geom_index; std::vector<float> v; std::vector<int> idx; };
std::vector<unsigned char> build_mesh_blob(const HostMesh &hm, uint32_t *tri_offset);
void mesh_world_bounds(const pt_geom &src, const HostMesh &hm, ptd::GeomRec *dst);

// the walk of k_path_w on the host (pt_debug_grid_probe): see include/ptmi355.h
int grid_probe(const pt_geom *geoms, int G, int density, const float *rays, int nrays, uint32_t *out_sets, uint32_t *out_info);
// the cone test of k_path_w's camera groups on the host (pt_debug_fan_probe)
int fan_probe(const pt_geom *geoms, int G, const float *rays, int nfans, uint32_t *out_sets, uint32_t *out_info);

}  // namespace pth
