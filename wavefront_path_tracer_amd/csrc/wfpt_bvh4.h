// wfpt_bvh4.h -- the four-wide, quantised node of the traversal for scenes beyond LDS, shared by the host code that
// builds it (wfpt_host.cpp: collapse_bvh4, plain C++) and the device code that walks it (wfpt_kernels.hip). Internal.
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <vector>

#include "wfpt.h"

namespace wfpt {

// Four-wide node, QUANTISED to one 64-byte line: the children's boxes as 8-bit offsets in the frame of their union
// (origin + q * 2^e per axis), then the child words. A child word is the index of a Node4 (inner child),
// kLeafFlag | count << 28 | first primitive (leaf child) or kEmptyChild (a leaf of no primitives). The traversal of scenes beyond LDS is bound by
// the rate of random line fetches (tools/microbench_node_fetch.hip: ~60 G lines/s whatever the record size up to 128 B),
// so a node should be ONE line: the 128-byte float version cost two L2 requests per visit. The quantised box encloses
// the true one (the host rounds lower planes down and upper planes up UNDER THE DEVICE'S OWN dequantisation arithmetic,
// one fma), so the box test stays conservative and the hits unchanged (what a traversal may change).
struct Node4 {
    float origin[3];   // lower corner of the union of the children's boxes
    uint8_t exp[3];    // per-axis scale 2^(exp - 127), as the biased exponent of a float
    uint8_t pad0;
    uint8_t qlo[3][4]; // [axis][child]: lower plane = origin + qlo * scale, rounded down
    uint8_t qhi[3][4]; // upper plane = origin + qhi * scale, rounded up
    uint32_t child[4];
    uint16_t scale_hi[3]; // the same scales as the upper 16 bits of their float (a power of two has no more): one shift / mask to decode
    uint16_t pad1;
};
static_assert(sizeof(Node4) == 64 && offsetof(Node4, qlo) == 16 && offsetof(Node4, child) == 40 && offsetof(Node4, scale_hi) == 56,
              "a four-wide node is one 64-byte line");
constexpr uint32_t kLeafFlag = 0x80000000u;
// an absent child: a leaf of no primitives behind an inverted box (qlo = 255, qhi = 0), so the traversal needs no test for it
constexpr uint32_t kEmptyChild = kLeafFlag;
constexpr uint32_t kLeafCountShift = 28, kLeafMaxCount = 6, kLeafFirstMask = (1u << kLeafCountShift) - 1u; // count 7 would make an all-ones word possible

// Collapses the binary tree (reference numbering, siblings at (2k, 2k+1)) into quantised four-wide nodes, numbered breadth
// first, every child box grown by `margin` per axis before it is quantised. Returns false when a leaf cannot be written as a
// child word or a box is not finite: the caller then keeps the binary traversal. `depth4` = levels of four-wide nodes below
// the root node.
bool collapse_bvh4(const wfpt_bvh_node *nodes, uint32_t n_nodes, const float margin[3], std::vector<Node4> &out, uint32_t &depth4);

} // namespace wfpt
