// Host-side builders for the two search structures the kernels traverse:
//   * LBVH over all scene triangles (Morton-ordered, binary radix split, leaves of <= 4 triangles), the
//     stand-in for Embree's BVH (/root/reference/src/scene.cpp:8-46, trianglemesh.cpp:107-143);
//   * the kd-tree of the global gradient cache, built exactly like nanoflann's KDTreeSingleIndexAdaptor
//     with leaf size 10 (/root/reference/src/global_cache.h:85-92; nanoflann.hpp:867-1007) so that the
//     reference's traversal-order-dependent "first 5 matches" query returns the same points.
#pragma once
#include <vector>

#include "../device/dchain.h"
#include "scene.h"

namespace lmc {

struct LbvhResult {
    std::vector<lmcd::BvhNode> nodes;
    std::vector<lmcd::LeafTri> leafTris;
    int depth = 0;
};
LbvhResult BuildLbvh(const std::vector<lmcd::TriData> &tris);
// same node / leaf format from a top-down binned-SAH build (fewer node visits per ray than the Morton tree)
LbvhResult BuildSahBvh(const std::vector<lmcd::TriData> &tris, int maxLeaf = 4);
// the binary tree behind the scene's: SAH unless LMC_BVH=lbvh (A/B switch; hits do not depend on the tree)
LbvhResult BuildSceneBvh(const std::vector<lmcd::TriData> &tris);
// the tree the renderer uploads: the binary tree with (up to) four children per node.  Every node takes its two children and
// keeps replacing the inner child with the largest surface area by that child's own two children until it has four.
// stackNeed = the largest number of entries the depth-first traversal can have pending.
struct Bvh4Result {
    std::vector<lmcd::BvhNode4> nodes;
    std::vector<lmcd::BvhNode4Q> qnodes;  // the same nodes with quantised child boxes (dscene.h BvhNode4Q), each verified to contain the exact box
    std::vector<lmcd::LeafTri> leafTris;
    int depth = 0, stackNeed = 0;
};
Bvh4Result CollapseToBvh4(const LbvhResult &bvh);
// Do the quantised nodes suit this tree?  What they cannot represent is a FLAT child away from its node's lower face (a wall, a table top): it gets a
// thickness of one or two steps, and every ray that leaves such a surface enters its box again (veach-door: 1.22 -> 1.71 leaf visits per ray, the torus
// scene, whose only flat surface is the floor in the lower face of the root: 1.03 -> 1.04; tests/helpers/bvh_stats.cpp).  Returns the share of the leaf
// children's surface area that belongs to such thickened flat children; an analysis aid (tests/helpers/bvh_stats.cpp prints it):
// host/context.cpp UploadScene chooses the node format of the scene's hot launches by it (torus 0.0006, veach-door 0.55).  0 for a tree without quantised nodes.
double ThickenedFlatLeafShare(const Bvh4Result &t);
// t.qnodes from t.nodes (called by CollapseToBvh4); leaves t.qnodes EMPTY when a box cannot be represented conservatively (never seen)
void QuantizeBvh4(Bvh4Result &t);

struct KdTreeResult {
    std::vector<lmcd::KdNode> nodes;
    std::vector<int> vind;
    std::vector<float> rootLow, rootHigh;
    int depth = 0;
};
KdTreeResult BuildKdTree(const float *pts, int n, int dim);
// Which m coordinates the existence grid of a cache dim is laid over.  The test is exact for any choice (a row within the query radius lies within
// one cell of the query in EVERY coordinate); the choice decides how many candidate rows a query scans.  The first coordinates -- screen position
// and first bounce -- are where both the cache rows and the queries cluster (profiles/r05_c_query_filter_study.txt: with coordinates 0..3 71-93 %
// of the queries of the headline workload scan candidates and a wave scans 6-33 rows; with the m coordinates in which the ROWS collide least,
// measured by the sum of squared occupancies of the G one-dimensional cells, 33-48 % and 3-4.5 rows).  Deterministic: integer scores, ties to
// the lower index, result ascending.  LMC_GRID_COORDS=first restores coordinates 0 .. m-1 (A/B).
void ChooseGridCoords(const float *pts, int n, int dim, int m, int *coord);
// dilated uniform grid over m coordinates of the cache points (DCacheDim::gridStart / gridRows / gridCoord):
// start[G^m + 1]; rows = the points of every cell's 3^m neighbourhood, dim floats each
struct CacheGrid {
    int G = 0, m = 0;
    int coord[4] = {0, 1, 2, 3};
    std::vector<int> start;
    std::vector<float> rows;
    bool Exists(const float *q, int dim) const;  // the kernel's test, on the host
};
CacheGrid BuildCacheGrid(const float *pts, int n, int dim, int m, const int *coord = nullptr);  // coord == nullptr: ChooseGridCoords

}  // namespace lmc
