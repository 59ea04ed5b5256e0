// rt_internal.h -- data layout shared by the host scene builder and the HIP kernels.
#ifndef RT_INTERNAL_H
#define RT_INTERNAL_H

#include <cstdint>
#include <string>
#include <vector>

#include "../../include/rt_amd.h"

// Constants of the reference path (model/model.cuh:21-25, camera/camera.cuh:5,
// octtree/octtree.cu:17, ray/ray.cuh:10,16, octtree/octtree.cuh:23).
#define RT_INF 64.0f
#define RT_EPS 0.001f
#define RT_NEAR 0.1f
#define RT_LEAF_CAP 50
#define RT_LEAF_PAD 64   // records behind the last leaf's that a walk may still request
#define RT_LEAF_ASK 0x100u   // bit of a leaf's RtWaveNode::meta, see there
#define RT_LEAF_ASK_MIN 3    // fewest triangles of a leaf worth asking about (asking costs about as much as two and a half triangle tests)
#define RT_RAY_STACK 10
#define RT_OCT_STACK 20
#define RT_MAX_LEVELS 3

// RT_TEX_*: the colour of the material comes from a texture (resources.json `brdf_color_texture_material`); the reference
// declares such a material (material.cuh:10, material.cu:263-301) but never loads an image or shades with one.
enum RtMatType { RT_LAMBERT = 0, RT_PHONG = 1, RT_REFLECTIVE = 2, RT_DIELECTRIC = 3, RT_TEX_LAMBERT = 4, RT_TEX_PHONG = 5 };
#define RT_TEX_BILINEAR 1u   // flags of a texture material (bits of RtMaterial::att[1]): filtering, and whether the ambient light sees the texture
#define RT_TEX_AMBIENT 2u
enum RtLightType { RT_PUNCTUAL = 0, RT_SPOT = 1, RT_AMBIENT = 2, RT_DIRECTIONAL = 3 };

// 96 B, passed to kernels by value (lives in SGPRs / kernarg, never re-read from HBM per pixel).
struct RtCamera {
  float u[3], v[3], w[3], pos[3];
  float left, right, bottom, top;
  float scene_refr, bg[3];
  int32_t width, height, max_depth, shadows;
};

struct RtMaterial {   // 32 B
  float col[3]; float param;
  float att[3]; int32_t type;   // texture materials: att[0] = bits of the texture's index, att[1] = bits of the RT_TEX_* flags
};

// A texture: `width * height` texels (R | G << 8 | B << 16, row 0 on top) starting at texel `first` of the scene's texel array.
struct RtTexture { uint32_t first, width, height, pad; };

struct RtLight {      // 48 B
  float col[3]; int32_t type;
  float pos[3]; float angle;
  float dir[3]; float cos_half;   // cosf(angle/2), evaluated once on the host (lights.cu:289,343 evaluate it per call)
};

struct RtSphere {     // 32 B
  float c[3], r;
  int32_t mat_first, mat_count, pad0, pad1;
};

// Octree node, 32 B = two 16-byte loads.  Children of a branch are eight
// consecutive records starting at `first`.
//   a = ini.x ini.y ini.z bits(first)   first: branch -> index of child 0, leaf -> offset into leaf_refs
//   b = end.x end.y end.z bits(count)   count: branch -> -1, leaf -> number of triangles (0..50)
struct RtNode { float a[4]; float b[4]; };

// The same node as the wave-synchronous kernel reads it, 32 B = one 8-dword scalar load.  The two planes of an
// axis sit in adjacent dwords so that a packed-fp32 instruction can take them straight from an SGPR pair.
//   p.x q.x p.y q.y | p.z q.z first meta
//   meta: leaf -> number of triangles (0..50), | RT_LEAF_ASK when a wave had better put its bundle of rays in front of the
//         leaf's triangles before testing them (rt_math.h, rt_cull_test); branch -> 0x80000000 | bit c set when child c is
//         not an empty leaf; either way bits 24..26 hold the record's own position among its eight siblings (0 for the root)
// The array is stored RT_NODE_VARIANTS = 8 times over (variant v of node i is record v * n + i, n counting the pad
// record, so that a walk stays inside one dense copy):  in variant v the pair of axis k is
// (lo, hi) when bit k of v is clear and (hi, lo) when it is set, lo / hi being the min / max corner
// (ini = lo.x lo.y hi.z, end = hi.x hi.y lo.z).  A wave whose rays all have the same direction signs walks the
// variant in which every pair is already (near plane, far plane) and saves the six min / max per node test; a trace
// only ever touches one variant, so the cache footprint of a trace is that of a single copy.
#define RT_NODE_VARIANTS 8
struct RtWaveNode { float lohi[6]; int32_t first; uint32_t meta; };

// Triangle intersection record, 48 B = three 16-byte loads:
//   v0.xyz, e1.xyz, e2.xyz (e1 = v1 - v0, e2 = v2 - v0 rounded to float, the first
//   six operations of triangle.cu:118-120), then mesh index and two pads.
struct RtTriGeom { float v0[3]; float e1[3]; float e2[3]; int32_t mesh; int32_t pad[2]; };

// Leaf-ordered copy of the intersection record, 64 B, read with wave-uniform addresses into SGPRs.  The pairs that
// the first stage of the triangle test multiplies with packed-fp32 instructions sit in even-aligned adjacent dwords, so
// they are used straight from SGPR pairs (e2 is stored as its three cyclic pairs):
//   v0.x v0.y v0.z id | e1.x e1.y e1.z n.x | e2.x e2.y e2.y e2.z | e2.z e2.x n.y n.z
// n = e1 x e2 with the rounding of the third cofactor column of triangle.cu:121-133 (m02, m12, m22): it depends on the
// triangle alone, so the host forms it once (same float operations, no contraction) instead of every ray that gets
// as far as the t numerator.
struct alignas(64) RtTriLeaf { float v0[3]; int32_t id; float e1[3]; float nx; float e2xy[2]; float e2yz[2]; float e2zx[2]; float ny, nz; };

// Triangle shading record, 64 B, read once per closest hit: v1, v2 and the three vertex normals.
struct RtTriShade { float v1[3]; float v2[3]; float n0[3]; float n1[3]; float n2[3]; float pad; };

struct rt_scene {
  RtCamera cam;
  std::vector<RtMaterial> materials;
  std::vector<RtLight> lights;
  std::vector<RtSphere> spheres;
  std::vector<int32_t> mat_index;                 // concatenated material lists (spheres, then meshes)
  std::vector<int32_t> mesh_mat_first, mesh_mat_count;
  std::vector<RtNode> nodes;                      // nodes[0] = root (always a branch) when non-empty
  std::vector<RtWaveNode> wnodes;                 // nodes in the wave kernel's layout: RT_NODE_VARIANTS copies of nodes.size() records
  std::vector<uint32_t> node_skip;                // per node: where a walk in the fixed child order goes when it is done with the node's subtree (see rt_host.cpp)
  std::vector<int32_t> leaf_refs;
  std::vector<RtTriGeom> tri_geom;
  std::vector<RtTriLeaf> tri_leaf;                // intersection records repeated in leaf order (index space of leaf_refs), RT_LEAF_PAD pad records, then the RtCullRec array
  uint32_t cull_offset = 0;                       // byte offset of that array from the start of tri_leaf (0: no mesh)
  std::vector<RtTriShade> tri_shade;
  std::vector<float> tri_verts, tri_normals;      // n*9 each, as the reference uploads them
  std::vector<float> tri_uv;                      // n*6: texture coordinates of the three corners (parsing.cu:54-63,102-108; zeros without `vt` lines)
  std::vector<RtTexture> textures;                // the textures some material of the scene uses
  std::vector<uint32_t> texels;
  rt_scene_info info;
  int32_t max_pushes_per_hit = 0;                 // most rays one hit can push: 1 per reflective, 2 per dielectric material of a primitive
  int32_t dielectric_mesh = 0;                    // some mesh carries a dielectric material
  // the arrays behind rt_scene_describe (filled on request)
  std::vector<rt_light_desc> desc_lights;
  std::vector<rt_material_desc> desc_materials;
  std::vector<rt_sphere_desc> desc_spheres;
  std::vector<int32_t> desc_tri_mesh;
};

void rt_set_error(const std::string& msg);

#endif
