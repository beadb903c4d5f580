// Scene commit: flatten a renderer configuration into the HBM layout of
// device_scene.h.  Runs once per scene on the host.
//
// What must agree with the reference (the kernels consume these tables and the
// image depends on them): world-space baking of meshes and their tangent
// frames (reference src/rtcore/scene.cpp:15-111,247-324), unit rectangle /
// cube tessellation (scene.cpp:196-245), analytic shape constants
// (scene.cpp:326-472), the LBVH topology — it fixes the traversal order and
// the area-weighted light sampling (src/rtcore/accel/bvh_builder.cpp:74-207),
// light tables (src/renderer/renderer.cpp:271-304), per-BSDF constants
// (src/renderer/bsdfs/bsdf.cpp:112-186), medium constants
// (src/renderer/medium/medium.cpp:6-39), emitter constants and env-map tables
// (src/renderer/emitters/emitter.cpp:122-175, envmap.cpp:20-68,
// renderer.cpp:597-606), the Kulla-Conty LUT (kulla_conty.cpp:12-80).
#include "commit.hpp"
#include "measurement_env.hpp"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>

#include "../bsdfs.h"
#include "../textures.h"
#include "../vecmath.h"
#include "matrix.hpp"

namespace mcpt
{
namespace
{

// math.cpp:148-166: frame with the given z axis, as a matrix (cylinders).
Mat4f FrameAroundAxis(V3 up)
{
    V3 c;
    if (std::sqrt(D(sqr(up.x) + sqr(up.z))) > D(kEpsFloat))
    {
        const float k = static_cast<float>(1.0 / std::sqrt(D(sqr(up.x) + sqr(up.z))));
        c = V3{-up.z * k, 0, up.x * k};
    }
    else
    {
        const float k = static_cast<float>(1.0 / std::sqrt(D(sqr(up.y) + sqr(up.z))));
        c = V3{0, -up.z * k, up.y * k};
    }
    const V3 b = normalize(cross(c, up));
    Mat4f r = Identity();
    r.m[0] = b.x, r.m[1] = b.y, r.m[2] = b.z;
    r.m[4] = c.x, r.m[5] = c.y, r.m[6] = c.z;
    r.m[8] = up.x, r.m[9] = up.y, r.m[10] = up.z;
    return r;
}

V3 Load3(const float *p) { return V3{p[0], p[1], p[2]}; }
Vec3f Store3(V3 v) { return Vec3f{v.x, v.y, v.z}; }
float4 Pack(V3 v, float w) { return float4{v.x, v.y, v.z, w}; }
float Bits(uint32_t u)
{
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}

// ---- bounding boxes ---------------------------------------------------------
struct Bounds
{
    V3 lo{kMaxFloat, kMaxFloat, kMaxFloat}, hi{kLowestFloat, kLowestFloat, kLowestFloat}; // aabb.cpp:8
    void Add(V3 p) { lo = vmin(p, lo), hi = vmax(p, hi); }
    void Add(const Bounds &b) { lo = vmin(b.lo, lo), hi = vmax(b.hi, hi); }
};

// ---- LBVH -------------------------------------------------------------------
// Same keys, same sort, same split rule as bvh_builder.cpp, emitted directly in
// the stackless layout: pre-order array + skip links instead of child links.
struct TreeNode
{
    Bounds box;
    float area = 0;
    uint32_t object = kNoObject; // leaves only
    uint32_t skip = kEndOfTree;  // tree-local
};

uint32_t Dilate10(uint32_t v) // bvh_builder.cpp:14-21
{
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}

uint32_t MortonCode(V3 unit) // bvh_builder.cpp:38-48
{
    const float x = fminf(fmaxf(unit.x * 1024.0f, 0.0f), 1023.0f),
                y = fminf(fmaxf(unit.y * 1024.0f, 0.0f), 1023.0f),
                z = fminf(fmaxf(unit.z * 1024.0f, 0.0f), 1023.0f);
    return Dilate10(static_cast<uint32_t>(x)) * 4 + Dilate10(static_cast<uint32_t>(y)) * 2 +
           Dilate10(static_cast<uint32_t>(z));
}

int CommonPrefix(uint64_t a, uint64_t b) // clz(a ^ b), 64 when equal (bvh_builder.cpp:23-34)
{
    const uint64_t x = a ^ b;
    return x == 0 ? 64 : __builtin_clzll(x);
}

class LinearBvh
{
public:
    LinearBvh(const std::vector<Bounds> &boxes, const std::vector<float> &areas)
        : boxes_(boxes), areas_(areas)
    {
        const uint32_t n = static_cast<uint32_t>(boxes.size());
        Bounds all;
        for (const Bounds &b : boxes)
            all.Add(b);
        const V3 extent = all.hi - all.lo;
        keys_.resize(n);
        order_.resize(n);
        for (uint32_t i = 0; i < n; ++i)
        {
            const V3 centre = (boxes[i].lo + boxes[i].hi) * 0.5f;
            keys_[i] = (static_cast<uint64_t>(MortonCode((centre - all.lo) / extent)) << 32) | i;
            order_[i] = i;
        }
        std::sort(order_.begin(), order_.end(),
                  [&](uint32_t a, uint32_t b) { return keys_[a] < keys_[b]; });
        nodes_.reserve(n ? 2 * static_cast<size_t>(n) - 1 : 0);
        if (n)
            Emit(0, n, kEndOfTree);
    }

    std::vector<TreeNode> nodes_;

private:
    // bvh_builder.cpp:172-206
    uint32_t SplitPoint(uint32_t first, uint32_t last) const
    {
        const uint64_t a = keys_[order_[first]], z = keys_[order_[last - 1]];
        if (a == z)
            return (first + last) >> 1;
        const int common = CommonPrefix(a, z);
        uint32_t split = first, step = last - first;
        do
        {
            step = (step + 1) >> 1;
            const uint32_t probe = split + step;
            if (probe < last && CommonPrefix(a, keys_[order_[probe]]) > common)
                split = probe;
        } while (step > 1);
        return split;
    }

    // Pre-order emission (bvh_builder.cpp:143-170).  `skip` is where the
    // traversal continues after this subtree: the right sibling for a left
    // child, the parent's skip for a right child.
    uint32_t Emit(uint32_t begin, uint32_t end, uint32_t skip)
    {
        const uint32_t id = static_cast<uint32_t>(nodes_.size());
        nodes_.emplace_back();
        nodes_[id].skip = skip;
        if (begin + 1 == end)
        {
            const uint32_t obj = order_[begin];
            nodes_[id].object = obj;
            nodes_[id].box = boxes_[obj];
            nodes_[id].area = areas_[obj];
            return id;
        }
        const uint32_t mid = SplitPoint(begin, end) + 1;
        // the left subtree has 2*(mid-begin)-1 nodes, so the right child's
        // index is known before recursing
        const uint32_t right_id = id + 1 + (2 * (mid - begin) - 1);
        const uint32_t l = Emit(begin, mid, right_id);
        const uint32_t r = Emit(mid, end, skip);
        nodes_[id].area = nodes_[l].area + nodes_[r].area;
        nodes_[id].box.lo = vmin(nodes_[l].box.lo, nodes_[r].box.lo); // aabb.cpp:50-53
        nodes_[id].box.hi = vmax(nodes_[l].box.hi, nodes_[r].box.hi);
        return id;
    }

    const std::vector<Bounds> &boxes_;
    const std::vector<float> &areas_;
    std::vector<uint64_t> keys_;
    std::vector<uint32_t> order_;
};


// ---- ordered-walk hierarchy ---------------------------------------------------
// A second tree over ALL primitives of ALL instances, used only for ray queries
// (light sampling keeps the reference's LBVH above: its topology is part of the
// sampling distribution).  Binned-SAH binary tree, one primitive per leaf, fat
// nodes holding both children's boxes (device_scene.h).  Leaf boxes are the
// reference's leaf boxes and interior boxes exact unions, so the slab test on
// the path to a primitive succeeds exactly when it does in the reference's
// TLAS/BLAS pair (the test is monotone in the box) — the two hierarchies accept
// the same primitives, only the visiting order and the number of visited nodes
// differ.  The depth is bounded (kWalkDepthMax) because it sizes the per-lane
// traversal stack.
int g_walk_tree_strategy = 0; // SetWalkTreeStrategyForTesting
float g_walk_tie_scale = 1.0f; // SetWalkTieScaleForTesting: shrinks the tie radius, to build a scene that lies OUTSIDE the ordered
                               // walk's bounds on purpose (the self-guard of mcpt_renderer_create must catch it: tests/test_gpu_parity.py)

// ---- 4-wide, exact form of the ordered-walk hierarchy (device_scene.h: pool_nodes) --------------
// For the wavefront-cooperative pool walk of small scenes (pool_walk.h): a (ray, node) item of that walk costs a fixed
// amount of bookkeeping whatever the node holds, and a walk lasts at least as many steps as the tree is deep — four
// children per item halve both.  The children's boxes are the binary hierarchy's own (a child here is a child or
// grandchild there): a primitive is reachable under exactly the same conditions, interior boxes that disappear were
// supersets of what they held (the slab test is monotone in the box).

void BuildPoolNodes(FlatScene &fs)
{
    IntegratorRec &ig = fs.integrator;
    fs.pool_nodes.clear();
    ig.n_pool_nodes = 0, ig.pool_depth = 0;
    const uint32_t n_binary = ig.n_walk_nodes;
    if (n_binary == 0)
    {
        fs.pool_nodes.assign(8, float4{0, 0, 0, 0});
        return;
    }
    struct Child
    {
        uint32_t ref; // binary node index, or kWalkLeaf | slot
        Bounds box;
    };
    auto children_of = [&](uint32_t node, Child out[2]) -> int
    {
        const float4 *n = &fs.walk_nodes[4 * size_t(node)];
        int k = 0;
        uint32_t ref0, ref1;
        std::memcpy(&ref0, &n[0].w, 4), std::memcpy(&ref1, &n[1].w, 4);
        Bounds b0, b1;
        b0.lo = V3{n[0].x, n[0].y, n[0].z}, b0.hi = V3{n[1].x, n[1].y, n[1].z};
        b1.lo = V3{n[2].x, n[2].y, n[2].z}, b1.hi = V3{n[3].x, n[3].y, n[3].z};
        if (!(b0.lo.x > b0.hi.x)) // (not the "never entered" box of the top node's second child)
            out[k++] = Child{ref0, b0};
        if (!(b1.lo.x > b1.hi.x))
            out[k++] = Child{ref1, b1};
        return k;
    };
    auto area = [](const Bounds &b)
    {
        const double dx = double(b.hi.x) - b.lo.x, dy = double(b.hi.y) - b.lo.y, dz = double(b.hi.z) - b.lo.z;
        return dx * dy + dy * dz + dz * dx;
    };
    static const int order_env = []
    {
        const char *e = mcpt::MeasurementEnv("MCPT_POOL_ORDER");
        return e == nullptr ? -1 : std::string(e) == "sorted" ? 1 : 0;
    }();
    const bool sorted_children = order_env >= 0 ? order_env == 1 : (n_binary <= 1024u && ig.n_prims <= 1024u);
    struct Todo
    {
        uint32_t binary, depth;
    };
    std::vector<Todo> todo{{0u, 1u}};
    for (size_t k = 0; k < todo.size(); ++k)
    {
        Child kids[4];
        int n = children_of(todo[k].binary, kids);
        for (;;) // open the inner child of largest surface while the result still fits four
        {
            int pick = -1;
            double best = -1.0;
            for (int i = 0; i < n; ++i)
                if (!(kids[i].ref & kWalkLeaf) && area(kids[i].box) > best)
                    best = area(kids[i].box), pick = i;
            if (pick < 0 || n >= 4)
                break;
            Child grand[2];
            const int g = children_of(kids[pick].ref, grand);
            if (n - 1 + g > 4)
                break;
            kids[pick] = kids[n - 1], --n;
            for (int i = 0; i < g; ++i)
                kids[n++] = grand[i];
        }
        // ORDER of the children in the record = the order the pool walk lists the entered ones in; its lists are LIFO, so the last
        // one is looked at first.  Scenes small enough for the LDS form of the walk (pool_walk.h: 10-bit references): largest box
        // first, i.e. the smallest — the likeliest to end in a hit that shrinks the ray's bound — is taken first: cornell-box 512 x
        // 512 spp 256, 41.5 -> 39.4 ms, same frame (one pixel per lane; no change at 256 x 256).  Larger scenes keep the order the
        // collapse produces (the opened child's children last): sorted, dragon/scene.xml +-0, matpreview 1.3-2.5 % slower
        // (EXPERIMENTS R4-9; MCPT_POOL_ORDER=built|sorted overrides for measurements).
        if (sorted_children)
            std::stable_sort(kids, kids + n, [&](const Child &a, const Child &b) { return area(a.box) > area(b.box); });
        float plane[6][4];
        // (an unused child's reference names a real slot: a ray with NaN components passes every box test, this one's too)
        uint32_t refs[4] = {kWalkLeaf, kWalkLeaf, kWalkLeaf, kWalkLeaf};
        for (int i = 0; i < 4; ++i)
        {
            // unused child: the inverted box that no ray enters
            const V3 lo = i < n ? kids[i].box.lo : V3{kMaxFloat, kMaxFloat, kMaxFloat}, hi = i < n ? kids[i].box.hi : V3{-kMaxFloat, -kMaxFloat, -kMaxFloat};
            plane[0][i] = lo.x, plane[1][i] = lo.y, plane[2][i] = lo.z, plane[3][i] = hi.x, plane[4][i] = hi.y, plane[5][i] = hi.z;
            if (i >= n)
                continue;
            if (kids[i].ref & kWalkLeaf)
                refs[i] = kids[i].ref;
            else
            {
                refs[i] = static_cast<uint32_t>(todo.size());
                todo.push_back(Todo{kids[i].ref, todo[k].depth + 1u});
            }
        }
        ig.pool_depth = std::max(ig.pool_depth, todo[k].depth);
        for (int a = 0; a < 6; ++a)
            fs.pool_nodes.push_back(float4{plane[a][0], plane[a][1], plane[a][2], plane[a][3]});
        float4 r;
        std::memcpy(&r, refs, sizeof r);
        fs.pool_nodes.push_back(r);
        fs.pool_nodes.push_back(float4{0, 0, 0, 0});
    }
    ig.n_pool_nodes = static_cast<uint32_t>(todo.size());
}

// ---- 4-wide, quantised form of the ordered-walk hierarchy (device_scene.h: wide_nodes) ----------
// The mesh walk is bound by the number of 64-byte node records it pulls through L2 / Infinity Cache (a lean trace kernel
// moves ~150 G records/s = 9.4 TB/s whatever its occupancy: DESIGN.md section 6), so the lever is records per ray: a
// node that holds FOUR children in the same 64 bytes halves them.  The binary tree above is collapsed — a node takes its
// grandchildren, largest surface first, until it has four children — and the children's boxes are stored as 8-bit
// offsets on a per-node grid (origin + 2^e * q per axis).  Decoded boxes CONTAIN the binary tree's boxes, rounding
// included: the quantiser verifies `origin + scale * q` with the device's own float operations and moves q outwards until
// it does.  The slab test is monotone in the box, so every primitive the exact hierarchy reaches is reached here too;
// what a larger box lets through in addition is stopped at the primitive by the exact leaf-box test
// (traversal.h, test_slot: kLeafCheck), as it already is for slivers' grown boxes.  Answers unchanged.
#ifndef MCPT_WIDE_TREELET
#define MCPT_WIDE_TREELET 0
#endif
constexpr uint32_t kWideTreelet = MCPT_WIDE_TREELET; // records per treelet of wide_nodes (0: breadth-first numbering)

// One 64-byte record of the quantised form: the boxes of `n` <= 4 children on the record's own grid (origin = their common lower corner,
// scale = the power of two per axis that spans the extent in 255 steps), decoded planes verified with the device's own operations.
struct WideChild
{
    uint32_t ref; // binary node index, or kWalkLeaf | slot
    Bounds box;
};
void EmitWideRecord(const WideChild *kids, int n, const uint32_t refs[4], std::vector<uint4> &out)
{
    Bounds all;
    for (int i = 0; i < n; ++i)
        all.Add(kids[i].box);
    const float origin[3] = {n ? all.lo.x : 0.0f, n ? all.lo.y : 0.0f, n ? all.lo.z : 0.0f}, top[3] = {n ? all.hi.x : 0.0f, n ? all.hi.y : 0.0f, n ? all.hi.z : 0.0f};
    uint32_t expo[3];
    uint8_t qlo[4][3] = {}, qhi[4][3] = {};
    for (int a = 0; a < 3; ++a)
    {
        const float extent = top[a] - origin[a];
        int e = extent > 0.0f ? std::max(-126, std::ilogb(extent) - 9) : -126;
        for (;; ++e)
        {
            const float scale = std::ldexp(1.0f, e);
            bool fits = origin[a] + scale * 255.0f >= top[a];
            for (int i = 0; i < n && fits; ++i)
            {
                const float lo = comp(kids[i].box.lo, a), hi = comp(kids[i].box.hi, a);
                int ql = static_cast<int>(std::floor((double(lo) - origin[a]) / scale)), qh = static_cast<int>(std::ceil((double(hi) - origin[a]) / scale));
                ql = std::min(std::max(ql, 0), 255), qh = std::min(std::max(qh, 0), 255);
                while (ql > 0 && !(origin[a] + scale * static_cast<float>(ql) <= lo))
                    --ql;
                while (qh < 255 && !(origin[a] + scale * static_cast<float>(qh) >= hi))
                    ++qh;
                fits = origin[a] + scale * static_cast<float>(ql) <= lo && origin[a] + scale * static_cast<float>(qh) >= hi;
                qlo[i][a] = static_cast<uint8_t>(ql), qhi[i][a] = static_cast<uint8_t>(qh);
            }
            if (fits || e >= 127)
                break;
        }
        expo[a] = static_cast<uint32_t>(e + 127);
    }
    auto bits = [](float f)
    {
        uint32_t u;
        std::memcpy(&u, &f, 4);
        return u;
    };
    auto plane_word = [&](const uint8_t q[4][3], int a)
    { return uint32_t(q[0][a]) | (uint32_t(q[1][a]) << 8) | (uint32_t(q[2][a]) << 16) | (uint32_t(q[3][a]) << 24); };
    // unused child slots: an inverted box (255 .. 0) and the "done" reference; `n` in the top byte of word 3
    for (int i = n; i < 4; ++i)
        for (int a = 0; a < 3; ++a)
            qlo[i][a] = 255, qhi[i][a] = 0;
    out.push_back(uint4{bits(origin[0]), bits(origin[1]), bits(origin[2]), expo[0] | (expo[1] << 8) | (expo[2] << 16) | (uint32_t(n) << 24)});
    out.push_back(uint4{refs[0], refs[1], refs[2], refs[3]});
    out.push_back(uint4{plane_word(qlo, 0), plane_word(qlo, 1), plane_word(qlo, 2), plane_word(qhi, 0)});
    out.push_back(uint4{plane_word(qhi, 1), plane_word(qhi, 2), 0u, 0u});
}

void BuildWideNodes(FlatScene &fs)
{
    IntegratorRec &ig = fs.integrator;
    fs.wide_nodes.clear();
    ig.n_wide_nodes = 0, ig.wide_stack = 1;
    const uint32_t n_binary = ig.n_walk_nodes;
    if (n_binary == 0)
    {
        fs.wide_nodes.assign(4, uint4{0, 0, 0, 0});
        return;
    }
    struct Child
    {
        uint32_t ref; // binary node index, or kWalkLeaf | slot
        Bounds box;
    };
    auto children_of = [&](uint32_t node, Child out[2]) -> int
    {
        const float4 *n = &fs.walk_nodes[4 * size_t(node)];
        int k = 0;
        uint32_t ref0, ref1;
        std::memcpy(&ref0, &n[0].w, 4), std::memcpy(&ref1, &n[1].w, 4);
        Bounds b0, b1;
        b0.lo = V3{n[0].x, n[0].y, n[0].z}, b0.hi = V3{n[1].x, n[1].y, n[1].z};
        b1.lo = V3{n[2].x, n[2].y, n[2].z}, b1.hi = V3{n[3].x, n[3].y, n[3].z};
        if (!(b0.lo.x > b0.hi.x)) // (not the "never entered" box of the top node's second child)
            out[k++] = Child{ref0, b0};
        if (!(b1.lo.x > b1.hi.x))
            out[k++] = Child{ref1, b1};
        return k;
    };
    auto area = [](const Bounds &b)
    {
        const double dx = double(b.hi.x) - b.lo.x, dy = double(b.hi.y) - b.lo.y, dz = double(b.hi.z) - b.lo.z;
        return dx * dy + dy * dz + dz * dx;
    };
    // breadth-first over the wide nodes: node k of `todo` becomes wide node k
    struct Todo
    {
        uint32_t binary;     // the binary node whose subtree this wide node covers ...
        uint32_t stack_above; // entries on a walk's stack when it arrives here (worst case over the path)
    };
    std::vector<Todo> todo{{0u, 1u}};
    std::vector<uint4> &out = fs.wide_nodes;
    uint32_t worst_stack = 1;
    // Two 64-byte records share a 128-byte cache line, the inner children of a node get consecutive numbers (breadth-first), and
    // the pool walk (pool_walk.h) lists the children a ray entered next to each other: neighbouring lanes of its next step read
    // sibling records with ONE instruction.  MCPT_WIDE_ALIGN=1 starts the inner children of a node with two or more of them on an
    // even number (a padding record nobody references fills the odd slot before), so that four siblings lie in two lines, never
    // three.  Measured (round 5, profiles/r05_experiments/quantised_pool_nodes_ab.json): no gain — dragon/scene.xml +-0 inside its
    // spread, matpreview 1.2-1.6 % slower (more records under the same caches) — so it is off unless asked for.
    static const bool align_pairs = []
    {
        const char *e = mcpt::MeasurementEnv("MCPT_WIDE_ALIGN");
        return e != nullptr && std::atoi(e) != 0;
    }();
    constexpr uint32_t kPadding = 0xFFFFFFFFu;
    for (size_t k = 0; k < todo.size(); ++k)
    {
        if (todo[k].binary == kPadding)
        {
            for (int v = 0; v < 4; ++v)
                out.push_back(v == 1 ? uint4{kWalkDone, kWalkDone, kWalkDone, kWalkDone} : uint4{0u, 0u, 0u, 0u});
            continue;
        }
        Child kids[4];
        int n = children_of(todo[k].binary, kids);
        // take grandchildren, largest surface first, while there is room
        for (;;)
        {
            int pick = -1;
            double best = -1.0;
            for (int i = 0; i < n; ++i)
                if (!(kids[i].ref & kWalkLeaf) && area(kids[i].box) > best)
                    best = area(kids[i].box), pick = i;
            if (pick < 0 || n >= 4)
                break;
            Child grand[2];
            const int g = children_of(kids[pick].ref, grand);
            if (n - 1 + g > 4)
                break;
            kids[pick] = kids[n - 1], --n;
            for (int i = 0; i < g; ++i)
                kids[n++] = grand[i];
        }
        // the node's grid: origin = the children's common lower corner, scale = the power of two that spans the extent in 255 steps
        Bounds all;
        for (int i = 0; i < n; ++i)
            all.Add(kids[i].box);
        const float origin[3] = {all.lo.x, all.lo.y, all.lo.z}, top[3] = {all.hi.x, all.hi.y, all.hi.z};
        uint32_t expo[3];
        uint8_t qlo[4][3] = {}, qhi[4][3] = {};
        for (int a = 0; a < 3; ++a)
        {
            // smallest scale = 2^e whose grid spans the extent in 255 steps AND whose decoded planes — evaluated exactly
            // like the device evaluates them, origin + scale * float(q) — enclose every child
            const float extent = top[a] - origin[a];
            int e = extent > 0.0f ? std::max(-126, std::ilogb(extent) - 9) : -126;
            for (;; ++e)
            {
                const float scale = std::ldexp(1.0f, e);
                bool fits = origin[a] + scale * 255.0f >= top[a];
                for (int i = 0; i < n && fits; ++i)
                {
                    const float lo = comp(kids[i].box.lo, a), hi = comp(kids[i].box.hi, a);
                    int ql = static_cast<int>(std::floor((double(lo) - origin[a]) / scale)), qh = static_cast<int>(std::ceil((double(hi) - origin[a]) / scale));
                    ql = std::min(std::max(ql, 0), 255), qh = std::min(std::max(qh, 0), 255);
                    while (ql > 0 && !(origin[a] + scale * static_cast<float>(ql) <= lo))
                        --ql;
                    while (qh < 255 && !(origin[a] + scale * static_cast<float>(qh) >= hi))
                        ++qh;
                    fits = origin[a] + scale * static_cast<float>(ql) <= lo && origin[a] + scale * static_cast<float>(qh) >= hi;
                    qlo[i][a] = static_cast<uint8_t>(ql), qhi[i][a] = static_cast<uint8_t>(qh);
                }
                if (fits || e >= 127)
                    break;
            }
            expo[a] = static_cast<uint32_t>(e + 127);
        }
        uint32_t refs[4] = {kWalkDone, kWalkDone, kWalkDone, kWalkDone};
        const uint32_t pushes = n > 0 ? static_cast<uint32_t>(n - 1) : 0u;
        worst_stack = std::max(worst_stack, todo[k].stack_above + pushes);
        int inner = 0;
        for (int i = 0; i < n; ++i)
            inner += (kids[i].ref & kWalkLeaf) ? 0 : 1;
        if (align_pairs && inner >= 2 && (todo.size() & 1u))
            todo.push_back(Todo{kPadding, 0u});
        for (int i = 0; i < n; ++i)
        {
            if (kids[i].ref & kWalkLeaf)
                refs[i] = kids[i].ref;
            else
            {
                refs[i] = static_cast<uint32_t>(todo.size());
                todo.push_back(Todo{kids[i].ref, todo[k].stack_above + pushes});
            }
        }
        auto bits = [](float f)
        {
            uint32_t u;
            std::memcpy(&u, &f, 4);
            return u;
        };
        auto plane_word = [&](const uint8_t q[4][3], int a)
        { return uint32_t(q[0][a]) | (uint32_t(q[1][a]) << 8) | (uint32_t(q[2][a]) << 16) | (uint32_t(q[3][a]) << 24); };
        // unused child slots: an inverted box (255 .. 0) and the "done" reference; `n` in the top byte of word 3
        for (int i = n; i < 4; ++i)
            for (int a = 0; a < 3; ++a)
                qlo[i][a] = 255, qhi[i][a] = 0;
        out.push_back(uint4{bits(origin[0]), bits(origin[1]), bits(origin[2]), expo[0] | (expo[1] << 8) | (expo[2] << 16) | (uint32_t(n) << 24)});
        out.push_back(uint4{refs[0], refs[1], refs[2], refs[3]});
        out.push_back(uint4{plane_word(qlo, 0), plane_word(qlo, 1), plane_word(qlo, 2), plane_word(qhi, 0)});
        out.push_back(uint4{plane_word(qhi, 1), plane_word(qhi, 2), 0u, 0u});
    }
    ig.n_wide_nodes = static_cast<uint32_t>(todo.size());
    ig.wide_stack = worst_stack + 1;
    // TREELET ORDER (round 6; VERDICT round 4 / 5: "lay wide_nodes out in depth-first treelet order").  Breadth-first numbering puts
    // siblings side by side — which the pool walk's node steps like — and a node's children a whole level away from it: a ray's path
    // from the top touches one record per level, each in another part of a 20-50 MB array.  Here the records are renumbered in
    // TREELETS: a treelet is a subtree's top `kTreelet` records in breadth-first order (siblings stay adjacent), the subtrees hanging
    // below it follow as treelets of their own, depth first — so the levels a walk descends through next lie within 2-4 KB of the
    // record it stands at.  A pure renumbering: the same records and references, the same walk, the same answers.
    // Measured (EXPERIMENTS R6-5); MCPT_TREELET=<records per treelet, 0 = breadth-first> overrides in experiment builds.
    static const uint32_t treelet = []
    {
        const char *e = MeasurementEnv("MCPT_TREELET");
        return e != nullptr ? static_cast<uint32_t>(std::atoi(e)) : kWideTreelet;
    }();
    const uint32_t n_nodes = ig.n_wide_nodes;
    if (treelet > 1 && n_nodes > treelet)
    {
        auto inner_children = [&](uint32_t node, uint32_t kids[4]) -> int
        {
            const uint4 refs = out[4 * size_t(node) + 1];
            const uint32_t r[4] = {refs.x, refs.y, refs.z, refs.w};
            int n = 0;
            for (uint32_t v : r)
                if (v != kWalkDone && !(v & kWalkLeaf))
                    kids[n++] = v;
            return n;
        };
        std::vector<uint32_t> order, roots{0u}, queue, frontier; // order[new] = old
        order.reserve(n_nodes);
        std::vector<uint8_t> placed(n_nodes, 0);
        while (!roots.empty())
        {
            const uint32_t root = roots.back();
            roots.pop_back();
            queue.assign(1, root), frontier.clear();
            uint32_t count = 0;
            for (size_t head = 0; head < queue.size(); ++head)
            {
                const uint32_t node = queue[head];
                if (count >= treelet)
                {
                    frontier.push_back(node);
                    continue;
                }
                order.push_back(node), placed[node] = 1, ++count;
                uint32_t kids[4];
                const int n = inner_children(node, kids);
                for (int i = 0; i < n; ++i)
                    queue.push_back(kids[i]);
            }
            for (size_t i = frontier.size(); i-- > 0;) // (depth first: the first subtree below the treelet comes right behind it)
                roots.push_back(frontier[i]);
        }
        // (padding records of MCPT_WIDE_ALIGN are referenced by nobody: they go to the end)
        for (uint32_t node = 0; node < n_nodes; ++node)
            if (!placed[node])
                order.push_back(node);
        std::vector<uint32_t> renumbered(n_nodes);
        for (uint32_t k = 0; k < n_nodes; ++k)
            renumbered[order[k]] = k;
        std::vector<uint4> moved(out.size());
        for (uint32_t k = 0; k < n_nodes; ++k)
        {
            for (int v = 0; v < 4; ++v)
                moved[4 * size_t(k) + v] = out[4 * size_t(order[k]) + v];
            uint4 &refs = moved[4 * size_t(k) + 1];
            uint32_t *r[4] = {&refs.x, &refs.y, &refs.z, &refs.w};
            for (uint32_t *v : r)
                if (*v != kWalkDone && !(*v & kWalkLeaf))
                    *v = renumbered[*v];
        }
        out.swap(moved);
    }
}

// ---- PAIRS of quantised records (experiment builds, -DMCPT_POOL_PAIRS=1; EXPERIMENTS R6-10) ------------------------------------
// A path of dragon/scene.xml's slowest tiles is a chain of dependent steps — item, ray record, node record through L2, pushes — and
// the frame ends with such chains.  What shortens a chain is fewer steps per ray query: here a node item names a PAIR of 64-byte
// records in one 128-byte line — the collapse (to at most four children each, as above) of the two children of a binary node —, so
// that one step looks at up to EIGHT descendants three binary levels down, two to eight lanes per item (pool_walk.h).  A half whose
// binary child is a leaf holds that leaf alone.  Inner references name pairs.  Same boxes, same reachability, same answers.
void BuildWidePairs(FlatScene &fs)
{
    IntegratorRec &ig = fs.integrator;
    fs.wide_nodes.clear();
    ig.n_wide_nodes = 0, ig.wide_stack = 1;
    const uint32_t n_binary = ig.n_walk_nodes;
    if (n_binary == 0)
    {
        fs.wide_nodes.assign(8, uint4{0, 0, 0, 0});
        return;
    }
    auto children_of = [&](uint32_t node, WideChild out[2]) -> int
    {
        const float4 *n = &fs.walk_nodes[4 * size_t(node)];
        int k = 0;
        uint32_t ref0, ref1;
        std::memcpy(&ref0, &n[0].w, 4), std::memcpy(&ref1, &n[1].w, 4);
        Bounds b0, b1;
        b0.lo = V3{n[0].x, n[0].y, n[0].z}, b0.hi = V3{n[1].x, n[1].y, n[1].z};
        b1.lo = V3{n[2].x, n[2].y, n[2].z}, b1.hi = V3{n[3].x, n[3].y, n[3].z};
        if (!(b0.lo.x > b0.hi.x)) // (not the "never entered" box of the top node's second child)
            out[k++] = WideChild{ref0, b0};
        if (!(b1.lo.x > b1.hi.x))
            out[k++] = WideChild{ref1, b1};
        return k;
    };
    auto area = [](const Bounds &b)
    {
        const double dx = double(b.hi.x) - b.lo.x, dy = double(b.hi.y) - b.lo.y, dz = double(b.hi.z) - b.lo.z;
        return dx * dy + dy * dz + dz * dx;
    };
    std::vector<uint32_t> todo{0u}; // pair k covers the two children of binary node todo[k]
    std::vector<uint4> &out = fs.wide_nodes;
    for (size_t k = 0; k < todo.size(); ++k)
    {
        WideChild top[2];
        const int n_top = children_of(todo[k], top);
        for (int h = 0; h < 2; ++h)
        {
            WideChild kids[4];
            int n = 0;
            if (h < n_top)
            {
                if (top[h].ref & kWalkLeaf)
                    kids[n++] = top[h];
                else
                {
                    n = children_of(top[h].ref, kids);
                    for (;;) // take grandchildren, largest surface first, while there is room
                    {
                        int pick = -1;
                        double best = -1.0;
                        for (int i = 0; i < n; ++i)
                            if (!(kids[i].ref & kWalkLeaf) && area(kids[i].box) > best)
                                best = area(kids[i].box), pick = i;
                        if (pick < 0 || n >= 4)
                            break;
                        WideChild grand[2];
                        const int g = children_of(kids[pick].ref, grand);
                        if (n - 1 + g > 4)
                            break;
                        kids[pick] = kids[n - 1], --n;
                        for (int i = 0; i < g; ++i)
                            kids[n++] = grand[i];
                    }
                }
            }
            uint32_t refs[4] = {kWalkDone, kWalkDone, kWalkDone, kWalkDone};
            for (int i = 0; i < n; ++i)
            {
                if (kids[i].ref & kWalkLeaf)
                    refs[i] = kids[i].ref;
                else
                {
                    refs[i] = static_cast<uint32_t>(todo.size());
                    todo.push_back(kids[i].ref);
                }
            }
            EmitWideRecord(kids, n, refs, out);
        }
    }
    ig.n_wide_nodes = static_cast<uint32_t>(todo.size());
}

class WalkTreeBuilder
{
public:
    WalkTreeBuilder(const std::vector<Bounds> &boxes, std::vector<float4> &nodes, std::vector<uint32_t> &slot_prim)
        : boxes_(boxes), nodes_(nodes), slot_prim_(slot_prim)
    {
        const uint32_t n = static_cast<uint32_t>(boxes.size());
        order_.resize(n);
        centre_.resize(n);
        for (uint32_t i = 0; i < n; ++i)
        {
            order_[i] = i;
            centre_[i] = (boxes[i].lo + boxes[i].hi) * 0.5f;
        }
    }

    // Returns the depth of the tree (interior levels, top node included).
    uint32_t Build()
    {
        nodes_.clear();
        slot_prim_.clear();
        if (order_.empty())
            return 0;
        // A subtree over k primitives has k - 1 interior nodes and k leaf slots, so node
        // ids (pre-order) and slots are known before a subtree is built: the arrays are
        // sized up front and large subtrees are built by separate threads in place.
        const uint32_t n = static_cast<uint32_t>(order_.size());
        nodes_.resize(4 * static_cast<size_t>(n)); // top node + n - 1 interior nodes
        slot_prim_.resize(n);
        threads_left_ = static_cast<int>(std::max(1u, std::thread::hardware_concurrency())) - 1;
        Bounds all;
        uint32_t depth = 0;
        // Depth budget.  The traversal stacks live in LDS at 1 KiB per level and workgroup, and
        // six workgroups per CU (the occupancy the latency-bound mesh kernels want) fit 26 levels
        // including the sentinel: aim for at most 24 interior levels below the top node whenever
        // that leaves a few levels of slack over a balanced tree, otherwise the hard bound.
        const uint32_t balanced = Log2Ceil(n);
        const uint32_t budget = balanced + 4 <= 24 ? 24u : kWalkDepthMax - 1;
        const uint32_t root = Emit(0, n, budget, 1, all, depth);
        Bounds none; // lo = +max, hi = -max: never entered
        nodes_[0] = Pack(all.lo, Bits(root)), nodes_[1] = Pack(all.hi, Bits(root));
        nodes_[2] = Pack(none.lo, 0.0f), nodes_[3] = Pack(none.hi, 0.0f);
        return depth + 1;
    }

private:
    static float4 Pack(V3 v, float w) { return float4{v.x, v.y, v.z, w}; }
    static float Bits(uint32_t u)
    {
        float f;
        std::memcpy(&f, &u, 4);
        return f;
    }
    static uint32_t Log2Ceil(uint32_t n)
    {
        uint32_t l = 0;
        while ((1ull << l) < n)
            ++l;
        return l;
    }
    static double HalfArea(const Bounds &b)
    {
        const double dx = double(b.hi.x) - b.lo.x, dy = double(b.hi.y) - b.lo.y, dz = double(b.hi.z) - b.lo.z;
        return dx * dy + dy * dz + dz * dx;
    }

    // Splits order_[begin, end) in place; returns the first index of the right part.
    uint32_t Split(uint32_t begin, uint32_t end, uint32_t budget)
    {
        const uint32_t count = end - begin;
        Bounds cb;
        for (uint32_t i = begin; i < end; ++i)
            cb.Add(centre_[order_[i]]);
        const V3 extent = cb.hi - cb.lo;
        const int wide = (extent.x >= extent.y && extent.x >= extent.z) ? 0 : (extent.y >= extent.z ? 1 : 2);
        auto median = [&]()
        {
            const uint32_t mid = begin + count / 2;
            std::nth_element(order_.begin() + begin, order_.begin() + mid, order_.begin() + end,
                             [&](uint32_t a, uint32_t b)
                             {
                                 const float ca = comp(centre_[a], wide), cb2 = comp(centre_[b], wide);
                                 return ca < cb2 || (ca == cb2 && a < b);
                             });
            return mid;
        };
        if (count <= 2 || budget <= Log2Ceil(count) || g_walk_tree_strategy == 2)
            return count == 2 ? begin + 1 : median();
        if (g_walk_tree_strategy == 1 && count <= 4096)
        {
            // exact sweep over every split position of every axis (test strategy)
            double best = 1e300;
            uint32_t best_left = 0;
            std::vector<uint32_t> sorted(order_.begin() + begin, order_.begin() + end), best_order;
            std::vector<double> right_area(count);
            for (int axis = 0; axis < 3; ++axis)
            {
                std::sort(sorted.begin(), sorted.end(),
                          [&](uint32_t a, uint32_t b)
                          {
                              const float ca = comp(centre_[a], axis), cb2 = comp(centre_[b], axis);
                              return ca < cb2 || (ca == cb2 && a < b);
                          });
                Bounds acc;
                for (uint32_t k = count; k-- > 1;)
                {
                    acc.Add(boxes_[sorted[k]]);
                    right_area[k] = HalfArea(acc);
                }
                acc = Bounds();
                for (uint32_t k = 1; k < count; ++k)
                {
                    acc.Add(boxes_[sorted[k - 1]]);
                    if (Log2Ceil(std::max(k, count - k)) > budget - 1)
                        continue;
                    const double cost = HalfArea(acc) * k + right_area[k] * (count - k);
                    if (cost < best)
                        best = cost, best_left = k, best_order = sorted;
                }
            }
            if (best_order.empty())
                return median();
            std::copy(best_order.begin(), best_order.end(), order_.begin() + begin);
            return begin + best_left;
        }
        constexpr int kBins = 16;
        double best_cost = 1e300;
        int best_axis = -1, best_bin = 0;
        for (int axis = 0; axis < 3; ++axis)
        {
            const float lo = comp(cb.lo, axis), width = comp(extent, axis);
            if (!(width > 0.0f))
                continue;
            Bounds bin_box[kBins];
            uint32_t bin_count[kBins] = {};
            const float scale = kBins / width;
            for (uint32_t i = begin; i < end; ++i)
            {
                const uint32_t prim = order_[i];
                int b = static_cast<int>((comp(centre_[prim], axis) - lo) * scale);
                b = b < 0 ? 0 : (b >= kBins ? kBins - 1 : b);
                bin_box[b].Add(boxes_[prim]);
                ++bin_count[b];
            }
            double right_area[kBins];
            uint32_t right_count[kBins];
            Bounds acc;
            uint32_t n = 0;
            for (int b = kBins - 1; b > 0; --b)
            {
                if (bin_count[b])
                    acc.Add(bin_box[b]);
                n += bin_count[b];
                right_area[b] = n ? HalfArea(acc) : 0.0;
                right_count[b] = n;
            }
            acc = Bounds();
            n = 0;
            for (int b = 0; b + 1 < kBins; ++b)
            {
                if (bin_count[b])
                    acc.Add(bin_box[b]);
                n += bin_count[b];
                if (n == 0 || right_count[b + 1] == 0)
                    continue;
                const double cost = HalfArea(acc) * n + right_area[b + 1] * right_count[b + 1];
                if (cost < best_cost)
                    best_cost = cost, best_axis = axis, best_bin = b;
            }
        }
        if (best_axis < 0)
            return median();
        const float lo = comp(cb.lo, best_axis), scale = kBins / comp(extent, best_axis);
        const auto first_right = std::partition(order_.begin() + begin, order_.begin() + end,
                                                [&](uint32_t prim)
                                                {
                                                    int b = static_cast<int>((comp(centre_[prim], best_axis) - lo) * scale);
                                                    b = b < 0 ? 0 : (b >= kBins ? kBins - 1 : b);
                                                    return b <= best_bin;
                                                });
        const uint32_t mid = static_cast<uint32_t>(first_right - order_.begin());
        const uint32_t larger = std::max(mid - begin, end - mid);
        if (mid == begin || mid == end || Log2Ceil(larger) > budget - 1)
            return median();
        return mid;
    }

    // Builds the subtree over order_[begin, end); its interior nodes get the ids
    // [id, id + (end - begin) - 1), its leaves the slots [begin, end).
    uint32_t Emit(uint32_t begin, uint32_t end, uint32_t budget, uint32_t id, Bounds &box, uint32_t &depth)
    {
        if (begin + 1 == end)
        {
            slot_prim_[begin] = order_[begin];
            box = boxes_[order_[begin]];
            depth = 0;
            return kWalkLeaf | begin;
        }
        const uint32_t mid = Split(begin, end, budget);
        // left subtree: (mid - begin) - 1 interior nodes right after this one
        const uint32_t id_left = id + 1, id_right = id + 1 + (mid - begin - 1);
        Bounds b0, b1;
        uint32_t d0 = 0, d1 = 0, r0 = 0, r1 = 0;
        bool spawned = false;
        std::thread helper;
        if (end - begin >= kParallelGrain)
        {
            std::lock_guard<std::mutex> lock(mutex_);
            if (threads_left_ > 0)
                --threads_left_, spawned = true;
        }
        if (spawned)
        {
            helper = std::thread([&]() { r0 = Emit(begin, mid, budget - 1, id_left, b0, d0); });
            r1 = Emit(mid, end, budget - 1, id_right, b1, d1);
            helper.join();
            std::lock_guard<std::mutex> lock(mutex_);
            ++threads_left_;
        }
        else
        {
            r0 = Emit(begin, mid, budget - 1, id_left, b0, d0);
            r1 = Emit(mid, end, budget - 1, id_right, b1, d1);
        }
        if (g_walk_tree_strategy == 3)
            std::swap(b0, b1), std::swap(r0, r1);
        nodes_[4 * size_t(id)] = Pack(b0.lo, Bits(r0)), nodes_[4 * size_t(id) + 1] = Pack(b0.hi, Bits(r1));
        nodes_[4 * size_t(id) + 2] = Pack(b1.lo, 0.0f), nodes_[4 * size_t(id) + 3] = Pack(b1.hi, 0.0f);
        box.lo = vmin(b0.lo, b1.lo), box.hi = vmax(b0.hi, b1.hi);
        depth = 1 + std::max(d0, d1);
        return id;
    }

    static constexpr uint32_t kParallelGrain = 1u << 14;
    std::mutex mutex_;
    int threads_left_ = 0;
    const std::vector<Bounds> &boxes_;
    std::vector<float4> &nodes_;
    std::vector<uint32_t> &slot_prim_;
    std::vector<uint32_t> order_;
    std::vector<V3> centre_;
};

// ---- meshes -----------------------------------------------------------------
struct MeshSource
{
    std::vector<V2> uv;
    std::vector<V3> pos, nrm, tan, bit;
    std::vector<uint32_t> idx;
};

MeshSource RectangleSource() // scene.cpp:196-212
{
    MeshSource m;
    m.uv = {{0, 0}, {1, 0}, {1, 1}, {0, 1}};
    m.pos = {{-1, -1, 0}, {1, -1, 0}, {1, 1, 0}, {-1, 1, 0}};
    m.nrm = {{0, 0, 1}, {0, 0, 1}, {0, 0, 1}, {0, 0, 1}};
    m.idx = {0, 1, 2, 2, 3, 0};
    return m;
}

// scene.cpp:214-245.  24 vertices = 6 faces x 4 corners.  Corner signs per
// face are generated from the face's axis triple instead of a literal table:
// face order -y, +y, +x, +z, -x, -z.
MeshSource CubeSource()
{
    struct Face
    {
        int axis;          // fixed axis
        float side;        // its sign
        float a[4], b[4];  // the two remaining coordinates, per corner
    };
    // remaining axes are taken in (x,y,z) order with the fixed one removed
    static const Face faces[6] = {
        {1, -1, {1, 1, -1, -1}, {-1, 1, 1, -1}},   // -y : (x, z)
        {1, +1, {1, -1, -1, 1}, {-1, -1, 1, 1}},   // +y : (x, z)
        {0, +1, {-1, 1, 1, -1}, {-1, -1, 1, 1}},   // +x : (y, z)
        {2, +1, {1, 1, -1, -1}, {-1, 1, 1, -1}},   // +z : (x, y)
        {0, -1, {-1, 1, 1, -1}, {1, 1, -1, -1}},   // -x : (y, z)
        {2, -1, {1, 1, -1, -1}, {1, -1, -1, 1}},   // -z : (x, y)
    };
    static const float corner_uv[4][2] = {{0, 1}, {1, 1}, {1, 0}, {0, 0}};
    MeshSource m;
    for (int f = 0; f < 6; ++f)
    {
        const Face &F = faces[f];
        for (int c = 0; c < 4; ++c)
        {
            float p[3];
            int slot = 0;
            for (int ax = 0; ax < 3; ++ax)
                p[ax] = (ax == F.axis) ? F.side : (slot++ == 0 ? F.a[c] : F.b[c]);
            float n[3] = {0, 0, 0};
            n[F.axis] = F.side;
            m.pos.push_back({p[0], p[1], p[2]});
            m.nrm.push_back({n[0], n[1], n[2]});
            m.uv.push_back({corner_uv[c][0], corner_uv[c][1]});
        }
        const uint32_t o = 4 * f;
        const uint32_t tri[6] = {o, o + 1, o + 2, o + 3, o, o + 2};
        m.idx.insert(m.idx.end(), tri, tri + 6);
    }
    return m;
}

MeshSource MeshFromRecord(const mcsd::Instance &in)
{
    MeshSource m;
    for (size_t k = 0; k + 1 < in.texcoords.size(); k += 2)
        m.uv.push_back({in.texcoords[k], in.texcoords[k + 1]});
    auto load = [](const std::vector<float> &src, std::vector<V3> &dst)
    {
        for (size_t k = 0; k + 2 < src.size(); k += 3)
            dst.push_back({src[k], src[k + 1], src[k + 2]});
    };
    load(in.positions, m.pos), load(in.normals, m.nrm), load(in.tangents, m.tan),
        load(in.bitangents, m.bit);
    m.idx = in.indices;
    return m;
}

struct TriangleRecord
{
    V2 uv[3];
    V3 p[3], n[3], t[3], b[3];
};

// Bake to world space (scene.cpp:247-284) and build per-triangle records with
// the reference's tangent-frame rules (scene.cpp:15-111).  The "area" is
// |e1 x e2|, i.e. twice the triangle area, exactly as the reference stores it
// (scene.cpp:49) — light sampling and MIS use it consistently.
// fn(begin, end) over [0, count) on several threads when the range is large.
template <class Fn>
void ParallelFor(size_t count, Fn fn)
{
    constexpr size_t kGrain = 1u << 15;
    const size_t workers = std::min<size_t>(std::max(1u, std::thread::hardware_concurrency()), count / kGrain);
    if (workers <= 1)
    {
        fn(size_t(0), count);
        return;
    }
    std::vector<std::thread> pool;
    for (size_t w = 0; w < workers; ++w)
        pool.emplace_back(fn, count * w / workers, count * (w + 1) / workers);
    for (std::thread &t : pool)
        t.join();
}

void BakeTriangles(MeshSource m, const Mat4f &to_world, std::vector<TriangleRecord> &tris,
                   std::vector<float> &areas)
{
    if (m.idx.empty())
        throw std::runtime_error("cannot find vertex index info when adding instance to scene.");
    if (m.pos.empty())
        throw std::runtime_error("cannot find vertex position info when adding instance to scene.");
    for (V3 &p : m.pos)
        p = transform_point(to_world, p);
    if (!m.nrm.empty())
    {
        const Mat4f normal_to_world = Inverted(Transposed(to_world));
        for (V3 &n : m.nrm)
            n = transform_dir(normal_to_world, n);
    }
    for (V3 &t : m.tan)
        t = transform_dir(to_world, t);
    for (V3 &b : m.bit)
        b = transform_dir(to_world, b);

    const size_t count = m.idx.size() / 3;
    tris.resize(count);
    areas.resize(count);
    const size_t n_vert = m.pos.size();
    for (const uint32_t id : m.idx)
        if (id >= n_vert)
            throw std::runtime_error("mesh index out of range.");
    for (const std::vector<V3> *attr : {&m.nrm, &m.tan, &m.bit})
        if (!attr->empty() && attr->size() < n_vert)
            throw std::runtime_error("mesh attribute array shorter than the vertex array.");
    if (!m.uv.empty() && m.uv.size() < n_vert)
        throw std::runtime_error("mesh attribute array shorter than the vertex array.");
    ParallelFor(count, [&](size_t first, size_t last) {
    for (size_t i = first; i < last; ++i)
    {
        TriangleRecord &tri = tris[i];
        const uint32_t *id = &m.idx[3 * i];
        if (m.uv.empty())
            tri.uv[0] = {0, 0}, tri.uv[1] = {1, 0}, tri.uv[2] = {1, 1};
        else
            for (int j = 0; j < 3; ++j)
                tri.uv[j] = m.uv.at(id[j]);
        for (int j = 0; j < 3; ++j)
            tri.p[j] = m.pos[id[j]];
        const V3 e1 = tri.p[1] - tri.p[0], e2 = tri.p[2] - tri.p[0];
        const V3 ng = cross(e1, e2);
        areas[i] = length(ng);
        if (m.nrm.empty())
        {
            const V3 flat = normalize(ng);
            tri.n[0] = tri.n[1] = tri.n[2] = flat;
        }
        else
        {
            for (int j = 0; j < 3; ++j)
                tri.n[j] = m.nrm.at(id[j]);
        }
        if (m.tan.empty() && m.bit.empty())
        {
            const V2 d1 = tri.uv[1] - tri.uv[0], d2 = tri.uv[2] - tri.uv[0];
            const float r = 1.0f / (d1.v * d2.u - d1.u * d2.v);
            const V3 tangent = normalize((d1.v * e2 - d2.v * e1) * r);
            for (int j = 0; j < 3; ++j)
            {
                tri.b[j] = normalize(cross(tri.n[j], tangent));
                tri.t[j] = normalize(cross(tri.b[j], tri.n[j]));
            }
        }
        else if (m.tan.empty())
        {
            for (int j = 0; j < 3; ++j)
            {
                tri.b[j] = m.bit.at(id[j]);
                tri.t[j] = normalize(cross(tri.b[j], tri.n[j]));
                tri.b[j] = normalize(cross(tri.n[j], tri.t[j]));
            }
        }
        else
        {
            for (int j = 0; j < 3; ++j)
            {
                tri.t[j] = m.tan.at(id[j]);
                tri.b[j] = normalize(cross(tri.n[j], tri.t[j]));
                tri.t[j] = normalize(cross(tri.b[j], tri.n[j]));
            }
        }
    }
    });
}

// ---- Kulla-Conty ------------------------------------------------------------
void ComputeKullaContyRow(int i, float *brdf, float *albedo)
{
    constexpr uint32_t kSamples = 1024;
    constexpr float kSampleStep = 1.0f / kSamples;
    const float step = 1.0f / kLutRes;
    const V3 n = V3{0.0f, 0.0f, 1.0f};
    float albedo_sum = 0.0f;
    const float alpha = step * (static_cast<float>(i) + 0.5f);
    for (int j = kLutRes - 1; j >= 0; --j)
    {
        const float mu = step * (static_cast<float>(j) + 0.5f);
        const V3 view = V3{-sqrtf(1.f - mu * mu), 0.0f, -mu};
        float acc = 0.0f; // directional albedo of the single-scatter lobe (kulla_conty.cpp:12-36)
        for (uint32_t s = 0; s < kSamples; ++s)
        {
            V3 h;
            float pdf_h;
            ggx_sample_iso(s * kSampleStep, radical_inverse2(s), alpha, h, pdf_h);
            const V3 l = reflect(view, h);
            const float g = smith_g1_iso(alpha, -view, h) * smith_g1_iso(alpha, l, h);
            const float n_v = dot(n, -view), n_l = dot(n, l), n_h = dot(n, h), h_v = dot(h, -view);
            if (n_l > 0.0f && n_h > 0.0f && h_v > 0.0f)
                acc += (h_v * g) / (n_v * n_h);
        }
        const float e = fminf(acc * kSampleStep, 1.0f);
        brdf[i * kLutRes + j] = e;
        float acc2 = 0.0f; // kulla_conty.cpp:38-58
        for (uint32_t s = 0; s < kSamples; ++s)
        {
            V3 h;
            float pdf_h;
            ggx_sample_iso(s * kSampleStep, radical_inverse2(s), alpha, h, pdf_h);
            const V3 l = reflect(view, h);
            const float n_l = dot(n, l), n_h = dot(n, h), h_v = dot(-view, h);
            if (n_l > 0.0f && n_h > 0.0f && h_v > 0.0f)
                acc2 += e * n_l;
        }
        albedo_sum += acc2 * 2.0f * kSampleStep;
    }
    albedo[i] = albedo_sum * step;
}

// bsdf.cpp:12-38
float AverageFresnelDielectric(float eta)
{
    if (eta < 1.0)
        return -1.4399f * sqr(eta) + 0.7099f * eta + 0.6681f + 0.0636f / eta;
    const float i1 = 1.0f / eta, i2 = i1 * i1, i3 = i2 * i1, i4 = i3 * i1, i5 = i4 * i1;
    return 0.919317f - 3.4793f * i1 + 6.75335f * i2 - 7.80989f * i3 + 4.98554f * i4 - 1.36881f * i5;
}

// bsdf.cpp:40-52
V3 AverageFresnelConductor(V3 r, V3 g)
{
    return splat(0.087237f) + 0.0230685f * g - 0.0864902f * g * g + 0.0774594f * g * g * g +
           0.782654f * r - 0.136432f * r * r + 0.278708f * r * r * r + 0.19744f * g * r +
           0.0360605f * g * g * r - 0.2586f * g * r * r;
}

std::string TextureError(uint32_t id) { return "cannot find texture (id " + std::to_string(id) + ")."; }

} // namespace

void KullaContyTables(const float **brdf, const float **albedo)
{
    static std::once_flag once;
    static std::vector<float> s_brdf, s_albedo;
    std::call_once(once, []()
                   {
                       s_brdf.assign(kLutRes * kLutRes, 0.0f);
                       s_albedo.assign(kLutRes, 0.0f);
                       unsigned workers = std::max(1u, std::thread::hardware_concurrency());
                       std::vector<std::thread> pool;
                       for (unsigned t = 0; t < workers; ++t)
                           pool.emplace_back([t, workers]()
                                             {
                                                 for (int i = kLutRes - 1 - static_cast<int>(t); i >= 0;
                                                      i -= static_cast<int>(workers))
                                                     ComputeKullaContyRow(i, s_brdf.data(), s_albedo.data());
                                             });
                       for (std::thread &t : pool)
                           t.join();
                   });
    *brdf = s_brdf.data();
    *albedo = s_albedo.data();
}

DeviceScene FlatScene::HostView() const
{
    DeviceScene d{};
    d.camera = camera, d.integrator = integrator, d.features = features;
    d.nodes = nodes.data(), d.node_area = node_area.data();
    d.walk_nodes = walk_nodes.data(), d.walk_prims = walk_prims.data();
    d.wide_nodes = wide_nodes.data();
    d.pool_nodes = pool_nodes.data();
    d.tri_pos = tri_pos.data(), d.tri_attr = tri_attr.data();
    d.instances = instances.data(), d.analytic = analytic.data();
    d.light_inst = light_inst.data(), d.light_cdf = light_cdf.data();
    d.textures = textures.data(), d.texels = texels.data();
    d.bsdfs = bsdfs.data(), d.media = media.data(), d.emitters = emitters.data();
    d.env_tables = env_tables.data();
    d.lut_brdf = lut_brdf.data(), d.lut_albedo = lut_albedo.data();
    return d;
}

size_t FlatScene::GeometryBytes() const
{
    return nodes.size() * sizeof(float4) + walk_nodes.size() * sizeof(float4) + pool_nodes.size() * sizeof(float4) + walk_prims.size() * sizeof(float4) +
           tri_pos.size() * sizeof(float4) + tri_attr.size() * sizeof(float4);
}

void SetWalkTreeStrategyForTesting(int strategy) { g_walk_tree_strategy = strategy; }
void SetWalkTieScaleForTesting(float scale) { g_walk_tie_scale = scale; }

void BuildReferenceLbvh(uint32_t n, const float *boxes, const float *areas, std::vector<float4> &nodes,
                        std::vector<float> &node_area)
{
    std::vector<Bounds> b(n);
    for (uint32_t i = 0; i < n; ++i)
    {
        b[i].lo = V3{boxes[6 * i], boxes[6 * i + 1], boxes[6 * i + 2]};
        b[i].hi = V3{boxes[6 * i + 3], boxes[6 * i + 4], boxes[6 * i + 5]};
    }
    const std::vector<float> a(areas, areas + n);
    const LinearBvh tree(b, a);
    nodes.clear(), node_area.clear();
    for (const TreeNode &t : tree.nodes_)
    {
        nodes.push_back(Pack(t.box.lo, Bits(t.skip)));
        nodes.push_back(Pack(t.box.hi, Bits(t.object)));
        node_area.push_back(t.area);
    }
}

FlatScene CommitScene(const mcsd::Scene &in, LbvhAccelerator *lbvh)
{
    FlatScene fs;
    const auto t_begin = std::chrono::steady_clock::now();
    auto seconds_since = [](std::chrono::steady_clock::time_point t)
    { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t).count(); };

    // ---- camera (camera.cpp:26-37; fov_y is linear in the angle, quirk Q4) --
    {
        const mcsd::Camera &c = in.camera;
        if (c.width <= 0 || c.height <= 0 || c.spp == 0)
            throw std::runtime_error("invalid film size or sample count.");
        fs.camera.width = c.width, fs.camera.height = c.height, fs.camera.spp = c.spp;
        fs.camera.spp_inv = 1.0f / c.spp;
        const V3 eye = Load3(c.eye), look_at = Load3(c.look_at), up0 = Load3(c.up);
        const float fov_y = c.fov_x * c.height / c.width;
        const V3 front = normalize(look_at - eye);
        const V3 right = normalize(cross(front, up0));
        const V3 up = normalize(cross(right, front));
        fs.camera.eye = Store3(eye), fs.camera.front = Store3(front);
        fs.camera.dx = Store3(right * tanf(c.fov_x * 0.5f * 0.01745329251994329576923690768489f));
        fs.camera.dy = Store3(up * tanf(fov_y * 0.5f * 0.01745329251994329576923690768489f));
    }
    // ToRadians(0.5f * fov) = (0.5f * fov) * k; written above as fov * 0.5f * k,
    // the same two roundings in the same order.

    IntegratorRec &ig = fs.integrator;
    ig.volpath = in.integrator.type == MCSD_INTEGRATOR_VOLPATH;
    ig.hide_emitters = in.integrator.hide_emitters != 0;
    ig.pdf_rr = in.integrator.pdf_rr;
    ig.rr_scale = in.integrator.pdf_rr; // renderer.cpp:634: the "reciprocal" holds pdf_rr itself
    ig.depth_rr = in.integrator.depth_rr;
    ig.depth_max = in.integrator.depth_max;
    ig.id_sun = ig.id_envmap = kNone;
    if (ig.volpath)
        fs.features |= kFeatVolPath;

    // ---- geometry: one BLAS per instance ------------------------------------
    const auto t_geometry = std::chrono::steady_clock::now();
    const uint32_t n_inst = static_cast<uint32_t>(in.instances.size());
    std::vector<std::vector<TreeNode>> blas(n_inst);
    std::vector<uint32_t> prim_base(n_inst);
    std::vector<Bounds> prim_box;    // by global primitive index: the reference's leaf boxes
    std::vector<uint32_t> prim_inst; // owning instance
    uint32_t n_prims = 0;
    fs.instances.resize(n_inst);
    for (uint32_t i = 0; i < n_inst; ++i)
    {
        const mcsd::Instance &s = in.instances[i];
        InstanceRec &rec = fs.instances[i];
        rec = InstanceRec{};
        rec.analytic = kNone;
        const Mat4f to_world = Load(s.to_world);
        std::vector<Bounds> boxes;
        std::vector<float> areas;
        prim_base[i] = n_prims;
        auto add_triangles = [&](MeshSource src)
        {
            std::vector<TriangleRecord> tris;
            BakeTriangles(std::move(src), to_world, tris, areas);
            boxes.resize(tris.size());
            for (size_t k = 0; k < tris.size(); ++k)
            {
                const TriangleRecord &t = tris[k];
                for (int j = 0; j < 3; ++j)
                {
                    boxes[k].Add(t.p[j]); // triangle.cpp:9-15
                    fs.tri_pos.push_back(Pack(t.p[j], 0.0f));
                }
                const float w[9] = {t.uv[0].u, t.uv[0].v, t.uv[1].u, t.uv[1].v, t.uv[2].u, t.uv[2].v, 0, 0, 0};
                for (int j = 0; j < 3; ++j)
                    fs.tri_attr.push_back(Pack(t.n[j], w[j]));
                for (int j = 0; j < 3; ++j)
                    fs.tri_attr.push_back(Pack(t.t[j], w[3 + j]));
                for (int j = 0; j < 3; ++j)
                    fs.tri_attr.push_back(Pack(t.b[j], 0.0f));
            }
            rec.kind = kInstTriangles;
            n_prims += static_cast<uint32_t>(tris.size());
        };
        auto add_analytic = [&](uint32_t kind, const AnalyticRec &a, const Bounds &box, float area)
        {
            rec.kind = kind;
            rec.analytic = static_cast<uint32_t>(fs.analytic.size());
            fs.analytic.push_back(a);
            boxes = {box};
            areas = {area};
            // keep the global primitive numbering dense: one (unused) slot
            for (int j = 0; j < 3; ++j)
                fs.tri_pos.push_back(float4{0, 0, 0, 0});
            for (int j = 0; j < 9; ++j)
                fs.tri_attr.push_back(float4{0, 0, 0, 0});
            n_prims += 1;
            fs.features |= kFeatAnalytic;
        };
        switch (s.type)
        {
        case MCSD_INST_RECTANGLE:
            add_triangles(RectangleSource());
            break;
        case MCSD_INST_CUBE:
            add_triangles(CubeSource());
            break;
        case MCSD_INST_MESHES:
            add_triangles(MeshFromRecord(s));
            break;
        case MCSD_INST_SPHERE: // scene.cpp:326-372, sphere.cpp:9-15
        {
            AnalyticRec a{};
            a.radius = s.sphere_radius;
            const V3 centre = Load3(s.sphere_center);
            a.center = Store3(centre);
            a.to_world = to_world, a.to_local = Inverted(to_world);
            a.normal_to_world = Inverted(Transposed(to_world));
            Bounds box;
            box.Add(transform_point(to_world, centre + a.radius));
            box.Add(transform_point(to_world, centre - a.radius));
            const V3 cw = transform_point(to_world, centre),
                     bw = transform_point(to_world, centre + V3{a.radius, 0.0f, 0.0f});
            add_analytic(kInstSphere, a, box, 4.0f * kPi * sqr(length(cw - bw)));
            break;
        }
        case MCSD_INST_DISK: // scene.cpp:374-416, disk.cpp:9-15
        {
            AnalyticRec a{};
            a.to_world = to_world, a.to_local = Inverted(to_world);
            a.normal_to_world = Inverted(Transposed(to_world));
            Bounds box;
            box.Add(transform_point(to_world, V3{-0.5f, -0.5f, 0}));
            box.Add(transform_point(to_world, V3{0.5f, 0.5f, 0}));
            const V3 cw = transform_point(to_world, V3{0, 0, 0}),
                     bw = transform_point(to_world, V3{0.5f, 0, 0});
            add_analytic(kInstDisk, a, box, kPi * sqr(length(cw - bw)));
            break;
        }
        case MCSD_INST_CYLINDER: // scene.cpp:418-472, cylinder.cpp:9-19
        {
            AnalyticRec a{};
            const V3 p0 = Load3(s.cyl_p0), p1 = Load3(s.cyl_p1);
            Mat4f m = FrameAroundAxis(normalize(p1 - p0));
            m = Multiply(TranslationMatrix(p0), m);
            m = Multiply(to_world, m);
            const V3 origin_w = transform_point(m, V3{0, 0, 0});
            a.length = length(transform_point(m, V3{0, 0, length(p1 - p0)}) - origin_w);
            a.radius = length(transform_point(m, V3{s.cyl_radius, 0, 0}) - origin_w);
            a.to_world = m, a.to_local = Inverted(m), a.normal_to_world = Inverted(Transposed(m));
            Bounds box;
            box.Add(transform_point(m, V3{a.radius, a.radius, 0}));
            box.Add(transform_point(m, V3{-a.radius, -a.radius, 0}));
            box.Add(transform_point(m, V3{a.radius, a.radius, a.length}));
            box.Add(transform_point(m, V3{-a.radius, -a.radius, a.length}));
            add_analytic(kInstCylinder, a, box, k2Pi * sqr(a.radius));
            break;
        }
        default:
            throw std::runtime_error("unknow instance type.");
        }
        if (lbvh && boxes.size() >= lbvh->min_prims)
        {
            std::vector<float> flat_boxes(6 * boxes.size());
            for (size_t k = 0; k < boxes.size(); ++k)
            {
                float *b = &flat_boxes[6 * k];
                b[0] = boxes[k].lo.x, b[1] = boxes[k].lo.y, b[2] = boxes[k].lo.z;
                b[3] = boxes[k].hi.x, b[4] = boxes[k].hi.y, b[5] = boxes[k].hi.z;
            }
            std::vector<float4> tree_nodes;
            std::vector<float> tree_area;
            lbvh->Build(static_cast<uint32_t>(boxes.size()), flat_boxes.data(), areas.data(), tree_nodes, tree_area);
            blas[i].resize(tree_area.size());
            for (size_t k = 0; k < tree_area.size(); ++k)
            {
                TreeNode &t = blas[i][k];
                const float4 a = tree_nodes[2 * k], b = tree_nodes[2 * k + 1];
                t.box.lo = V3{a.x, a.y, a.z}, t.box.hi = V3{b.x, b.y, b.z};
                t.area = tree_area[k];
                std::memcpy(&t.skip, &a.w, 4), std::memcpy(&t.object, &b.w, 4);
            }
        }
        else
        {
            blas[i] = std::move(LinearBvh(boxes, areas).nodes_);
        }
        prim_box.insert(prim_box.end(), boxes.begin(), boxes.end());
        prim_inst.insert(prim_inst.end(), boxes.size(), i);
        rec.prim_base = prim_base[i];
        rec.bsdf = s.id_bsdf;
        rec.medium_int = s.id_medium_int, rec.medium_ext = s.id_medium_ext;
        rec.area_light = kNone;
        if (s.id_medium_int != kNone || s.id_medium_ext != kNone)
            fs.features |= kFeatVolPath;
    }

    // ---- TLAS in front, then the BLAS trees (scene.cpp:474-533) -------------
    std::vector<TreeNode> tlas;
    if (n_inst)
    {
        std::vector<Bounds> boxes(n_inst);
        std::vector<float> areas(n_inst);
        for (uint32_t i = 0; i < n_inst; ++i)
        {
            boxes[i] = blas[i][0].box;
            areas[i] = blas[i][0].area;
            fs.instances[i].pdf_area = 1.0f / areas[i];
        }
        tlas = std::move(LinearBvh(boxes, areas).nodes_);
    }
    auto append_tree = [&](const std::vector<TreeNode> &tree, uint32_t object_base)
    {
        const uint32_t base = static_cast<uint32_t>(fs.node_area.size());
        for (const TreeNode &n : tree)
        {
            const uint32_t skip = n.skip == kEndOfTree ? kEndOfTree : n.skip + base;
            const uint32_t object = n.object == kNoObject ? kNoObject : n.object + object_base;
            fs.nodes.push_back(Pack(n.box.lo, Bits(skip)));
            fs.nodes.push_back(Pack(n.box.hi, Bits(object)));
            fs.node_area.push_back(n.area);
        }
        return base;
    };
    append_tree(tlas, 0);
    ig.n_tlas_nodes = static_cast<uint32_t>(tlas.size());
    for (uint32_t i = 0; i < n_inst; ++i)
        fs.instances[i].blas_root = append_tree(blas[i], prim_base[i]);
    ig.n_nodes = static_cast<uint32_t>(fs.node_area.size());
    ig.n_instances = n_inst;
    ig.n_prims = n_prims;

    fs.seconds_lbvh = seconds_since(t_geometry);

    // ---- ordered-walk hierarchy over all primitives ---------------------------
    {
        const auto t_walk = std::chrono::steady_clock::now();
        // rank = position in the reference's visiting order (TLAS pre-order, then the
        // instance's BLAS pre-order): decides between primitives at equal distance
        std::vector<uint32_t> rank(n_prims, 0);
        uint32_t counter = 0;
        for (const TreeNode &t : tlas)
            if (t.object != kNoObject)
                for (const TreeNode &b : blas[t.object])
                    if (b.object != kNoObject)
                        rank[prim_base[t.object] + b.object] = counter++;
        float coordinate = std::max({fabsf(fs.camera.eye.x), fabsf(fs.camera.eye.y), fabsf(fs.camera.eye.z)});
        for (const Bounds &b : prim_box)
            coordinate = std::max({coordinate, fabsf(b.lo.x), fabsf(b.lo.y), fabsf(b.lo.z), fabsf(b.hi.x), fabsf(b.hi.y),
                                   fabsf(b.hi.z)});
        // Leaf boxes of the hierarchy = the reference's leaf boxes, except for SLIVERS.  The
        // reference's triangle distance is a mean of the vertices' depths weighted by the edge
        // functions; in a sliver those are differences of nearly equal products and the computed
        // distance can sit far in front of the triangle's plane — classroom has a 40 x 0.04 one whose
        // distance comes out 7e-4 short at coordinates of 20 (600 roundings), and the reference
        // then prefers it to a surface that is really nearer.  To give the same answer the walk has
        // to reach such a triangle although its box starts behind the current best hit: its leaf box
        // is grown by the error it can make, roundings at the scene's scale times the aspect ratio
        // (longest edge / height).  Supersets only ever add visits.
        std::vector<Bounds> leaf_box = prim_box;
        std::vector<uint8_t> sliver(n_prims, 0);
        float largest_grow = 0.0f;
        for (uint32_t p = 0; p < n_prims; ++p)
        {
            if (fs.instances[prim_inst[p]].kind != kInstTriangles)
                continue;
            const float4 *v = &fs.tri_pos[3 * static_cast<size_t>(p)];
            const V3 a{v[0].x, v[0].y, v[0].z}, b{v[1].x, v[1].y, v[1].z}, c{v[2].x, v[2].y, v[2].z};
            const V3 ab = b - a, ac = c - a, bc = c - b;
            const float longest2 = std::max({dot(ab, ab), dot(ac, ac), dot(bc, bc)}), area2 = length(cross(ab, ac));
            if (!(area2 > 0.0f))
                continue; // degenerate: never hit (triangle.cpp:66-68)
            const float aspect = longest2 / area2; // longest edge / the height over it
            if (aspect <= 128.0f)
                continue; // (measured roundings-per-aspect below: 0.3; up to here the tie radius covers it)
            const float grow = std::min(kEpsFloat * coordinate * aspect, 0.01f * coordinate);
            leaf_box[p].lo = leaf_box[p].lo - V3{grow, grow, grow}, leaf_box[p].hi = leaf_box[p].hi + V3{grow, grow, grow};
            largest_grow = std::max(largest_grow, grow);
            sliver[p] = 1; // ... and every comparison it takes part in is decided like the reference would (kWalkSliver)
        }
        std::vector<uint32_t> slot_prim;
        ig.walk_depth = WalkTreeBuilder(leaf_box, fs.walk_nodes, slot_prim).Build() + 1; // + sentinel entry
        ig.n_walk_nodes = static_cast<uint32_t>(fs.walk_nodes.size() / 4);
        // measured: on meshes a wavefront should stop waiting for its last
        // few searching lanes (matpreview +30 % at 8..16), in box-like scenes it should not
        // (sweep on the matpreview scene and a 0.8 M-triangle scene: flat optimum around 6..8 / 10..12)
        ig.walk_break = n_prims >= 2048 ? 8u : 0u;
        ig.walk_hold = n_prims >= 2048 ? 12u : 0u;
        // The triangle test's distance carries an ABSOLUTE error of a few roundings at the magnitude
        // of the vertices' offsets from the ray origin (it is a weighted mean of those), whatever the
        // distance itself is; a leaf box's entry distance does not.  Hits closer together than that
        // are "tied": the walk must visit both and decide like the reference (traversal.h, test_slot).
        // Offsets are at most (largest |coordinate| of the geometry or the eye) * 2.
        ig.walk_tie = 5e-6f * coordinate;
        // (SetWalkTieScaleForTesting: 1 in production — no environment variable reaches the walk's correctness bound)
        ig.walk_tie *= g_walk_tie_scale;
        ig.walk_extent = coordinate + largest_grow; // (every box plane lies within the geometry's coordinates, grown sliver boxes included)
        ig.walk_sliver_reach = largest_grow > 0.0f ? std::max(ig.walk_tie, largest_grow) : 0.0f; // 0: no slivers
        fs.walk_prims.reserve(3 * slot_prim.size());
        for (const uint32_t prim : slot_prim)
        {
            const float4 *p = &fs.tri_pos[3 * static_cast<size_t>(prim)];
            fs.walk_prims.push_back(float4{p[0].x, p[0].y, p[0].z, Bits(prim)});
            fs.walk_prims.push_back(float4{p[1].x, p[1].y, p[1].z, Bits(prim_inst[prim])});
            fs.walk_prims.push_back(float4{p[2].x, p[2].y, p[2].z, Bits(rank[prim] | (sliver[prim] ? kWalkSliver : 0u))});
        }
#if MCPT_POOL_PAIRS
        BuildWidePairs(fs);
#else
        BuildWideNodes(fs);
#endif
        BuildPoolNodes(fs);
        fs.seconds_walk = seconds_since(t_walk);
    }

    // ---- light tables (renderer.cpp:271-304): weights are NOT normalised ----
    std::vector<float> weights;
    for (uint32_t i = 0; i < n_inst; ++i)
    {
        const uint32_t b = in.instances[i].id_bsdf;
        if (b < in.bsdfs.size() && in.bsdfs[b].type == MCSD_BSDF_AREA_LIGHT)
        {
            fs.instances[i].area_light = static_cast<uint32_t>(fs.light_inst.size());
            fs.light_inst.push_back(i);
            weights.push_back(in.bsdfs[b].weight);
        }
    }
    fs.light_cdf.assign(weights.size() + 1, 0.0f);
    for (size_t k = 0; k < weights.size(); ++k)
        fs.light_cdf[k + 1] = weights[k] + fs.light_cdf[k];
    ig.n_area_lights = static_cast<uint32_t>(weights.size());

    // ---- textures (renderer.cpp:371-431): one shared texel pool -------------
    for (const mcsd::Texture &t : in.textures)
    {
        TextureRec o{};
        o.to_uv = Identity();
        switch (t.type)
        {
        case MCSD_TEX_CONSTANT:
            o.kind = kTexConstant;
            o.color = Vec3f{t.color[0], t.color[1], t.color[2]};
            break;
        case MCSD_TEX_CHECKERBOARD:
            o.kind = kTexChecker;
            o.color0 = Vec3f{t.color0[0], t.color0[1], t.color0[2]};
            o.color1 = Vec3f{t.color1[0], t.color1[1], t.color1[2]};
            o.to_uv = Load(t.to_uv);
            fs.features |= kFeatTextures;
            break;
        case MCSD_TEX_BITMAP:
            o.kind = kTexBitmap;
            o.width = t.width, o.height = t.height, o.channel = t.channel;
            if (t.width <= 0 || t.height <= 0 || (t.channel != 1 && t.channel != 3 && t.channel != 4))
                throw std::runtime_error("unsupported bitmap size or channel count.");
            o.to_uv = Load(t.to_uv);
            o.texel_base = static_cast<uint32_t>(fs.texels.size());
            fs.texels.insert(fs.texels.end(), t.data.begin(), t.data.end());
            fs.features |= kFeatTextures;
            break;
        default:
            throw std::runtime_error("unknow texture type.");
        }
        fs.textures.push_back(o);
    }
    const size_t n_tex = fs.textures.size();
    auto check = [&](uint32_t id, bool allow_none)
    {
        if (id == kNone && allow_none)
            return;
        if (id >= n_tex)
            throw std::runtime_error(TextureError(id));
    };

    // ---- BSDF constants (bsdf.cpp:112-186) ----------------------------------
    bool needs_lut = false;
    for (const mcsd::Bsdf &b : in.bsdfs)
    {
        BsdfRec o{};
        o.twosided = b.twosided != 0;
        o.opacity = b.id_opacity, o.bump = b.id_bump_map;
        o.tex0 = o.tex1 = o.tex2 = o.tex3 = kNone;
        o.reflectivity = 1.0f, o.eta = 1.0f, o.eta_inv = 1.0f, o.f_avg = 1.0f, o.f_avg_inv = 1.0f;
        check(o.opacity, true), check(o.bump, true);
        if (o.opacity != kNone || o.bump != kNone)
            fs.features |= kFeatTextures;
        if (o.opacity != kNone)
            ig.has_masks = 1; // the mask test draws random numbers DURING the walk (bsdf.cpp:272-276)
        switch (b.type)
        {
        case MCSD_BSDF_AREA_LIGHT:
            o.kind = kBsdfAreaLight;
            o.tex0 = b.id_radiance;
            check(o.tex0, false);
            break;
        case MCSD_BSDF_DIFFUSE:
            o.kind = kBsdfDiffuse;
            o.tex0 = b.id_diffuse_reflectance;
            check(o.tex0, false);
            break;
        case MCSD_BSDF_ROUGH_DIFFUSE:
            // use_fast_approx is parsed but never reaches the committed BSDF in
            // the reference (bsdf.cpp:139-144): the full Oren-Nayar model runs.
            o.kind = kBsdfRoughDiffuse;
            fs.integrator.has_reflectors = 1, fs.integrator.has_non_conductor = 1;
            o.tex0 = b.id_diffuse_reflectance, o.tex1 = b.id_roughness;
            check(o.tex0, false), check(o.tex1, false);
            fs.features |= kFeatMicrofacet;
            break;
        case MCSD_BSDF_CONDUCTOR:
            o.kind = kBsdfConductor;
            fs.integrator.has_reflectors = 1;
            o.tex0 = b.id_roughness_u, o.tex1 = b.id_roughness_v, o.tex2 = b.id_specular_reflectance;
            check(o.tex0, false), check(o.tex1, false), check(o.tex2, false);
            o.reflectivity3 = Vec3f{b.reflectivity[0], b.reflectivity[1], b.reflectivity[2]};
            o.f_avg3 = Store3(AverageFresnelConductor(Load3(b.reflectivity), Load3(b.edgetint)));
            needs_lut = true;
            fs.features |= kFeatMicrofacet;
            break;
        case MCSD_BSDF_DIELECTRIC:
        case MCSD_BSDF_THIN_DIELECTRIC:
            o.kind = b.type == MCSD_BSDF_DIELECTRIC ? kBsdfDielectric : kBsdfThinDielectric;
            fs.integrator.has_transmission = 1, fs.integrator.has_non_conductor = 1;
            if (b.type == MCSD_BSDF_THIN_DIELECTRIC)
                fs.integrator.has_reflectors = 1;
            if (b.type == MCSD_BSDF_DIELECTRIC)
            {
                o.f_avg = AverageFresnelDielectric(b.eta);
                o.f_avg_inv = AverageFresnelDielectric(1.0f / b.eta);
                needs_lut = true;
            }
            o.twosided = 1;
            o.tex0 = b.id_roughness_u, o.tex1 = b.id_roughness_v;
            o.tex2 = b.id_specular_reflectance, o.tex3 = b.id_specular_transmittance;
            check(o.tex0, false), check(o.tex1, false), check(o.tex2, false), check(o.tex3, false);
            o.eta = b.eta, o.eta_inv = 1.0f / b.eta;
            o.reflectivity = sqr(b.eta - 1.0f) / sqr(b.eta + 1.0f);
            // (the multiple-scattering term's transmitted share: constants of the record and the side, bsdfs.h)
            if (b.type == MCSD_BSDF_DIELECTRIC)
                o.ms_ratio_t = dielectric_ms_ratio_t(o, o.eta), o.ms_ratio_t_inside = dielectric_ms_ratio_t(o, o.eta_inv);
            fs.features |= kFeatMicrofacet;
            break;
        case MCSD_BSDF_PLASTIC:
            o.kind = kBsdfPlastic;
            fs.integrator.has_reflectors = 1, fs.integrator.has_non_conductor = 1;
            o.tex0 = b.id_roughness, o.tex1 = b.id_diffuse_reflectance, o.tex2 = b.id_specular_reflectance;
            check(o.tex0, false), check(o.tex1, false), check(o.tex2, false);
            o.reflectivity = sqr(b.eta - 1.0f) / sqr(b.eta + 1.0f);
            o.f_avg = AverageFresnelDielectric(b.eta);
            fs.features |= kFeatMicrofacet;
            break;
        default:
            throw std::runtime_error("unknow BSDF type.");
        }
        fs.bsdfs.push_back(o);
    }
    for (const InstanceRec &rec : fs.instances)
        if (rec.bsdf != kNone && rec.bsdf >= fs.bsdfs.size())
            throw std::runtime_error("cannot find BSDF (id " + std::to_string(rec.bsdf) + ").");

    // ---- Kulla-Conty LUT: only conductors and dielectrics read it -----------
    fs.lut_brdf.assign(kLutRes * kLutRes, 0.0f);
    fs.lut_albedo.assign(kLutRes, 0.0f);
    if (needs_lut)
    {
        const float *b, *a;
        KullaContyTables(&b, &a);
        fs.lut_brdf.assign(b, b + kLutRes * kLutRes);
        fs.lut_albedo.assign(a, a + kLutRes);
    }

    // ---- media (medium.cpp:6-39) --------------------------------------------
    for (const mcsd::Medium &m : in.media)
    {
        MediumRec o{};
        const V3 sa = Load3(m.sigma_a), ss = Load3(m.sigma_s);
        o.sigma_s = Store3(ss);
        const V3 st = sa + ss;
        o.sigma_t = Store3(st);
        const V3 albedo = ss / st;
        for (int d = 0; d < 3; ++d)
            if (comp(albedo, d) > o.sampling_weight && comp(st, d) > 0)
                o.sampling_weight = comp(albedo, d);
        if (o.sampling_weight > 0 && o.sampling_weight < 0.5f)
            o.sampling_weight = 0.5f;
        o.hg = m.phase_type == MCSD_PHASE_HG;
        o.g = Vec3f{m.g[0], m.g[1], m.g[2]};
        fs.media.push_back(o);
    }
    for (const InstanceRec &rec : fs.instances)
        for (uint32_t id : {rec.medium_int, rec.medium_ext})
            if (id != kNone && id >= fs.media.size())
                throw std::runtime_error("cannot find medium (id " + std::to_string(id) + ").");

    // ---- emitters (emitter.cpp:122-175, renderer.cpp:520-620) ---------------
    for (size_t i = 0; i < in.emitters.size(); ++i)
    {
        const mcsd::Emitter &e = in.emitters[i];
        EmitterRec o{};
        o.texture = kNone;
        o.to_world = Identity(), o.to_local = Identity();
        switch (e.type)
        {
        case MCSD_EMIT_POINT:
            o.kind = kEmitPoint;
            o.position = Vec3f{e.position[0], e.position[1], e.position[2]};
            o.intensity = Vec3f{e.intensity[0], e.intensity[1], e.intensity[2]};
            break;
        case MCSD_EMIT_SPOT:
            o.kind = kEmitSpot;
            o.cutoff = e.cutoff_angle;
            o.cos_cutoff = cosf(e.cutoff_angle);
            o.uv_factor = tanf(e.cutoff_angle);
            o.cos_beam = cosf(e.beam_width);
            o.transition_rcp = 1.0f / (e.cutoff_angle - e.beam_width);
            o.intensity = Vec3f{e.intensity[0], e.intensity[1], e.intensity[2]};
            o.texture = e.id_texture;
            check(o.texture, true);
            o.to_world = Load(e.to_world);
            o.position = Store3(transform_point(o.to_world, V3{0, 0, 0}));
            o.to_local = Inverted(o.to_world);
            break;
        case MCSD_EMIT_DIRECTIONAL:
            o.kind = kEmitDirectional;
            o.direction = Vec3f{e.direction[0], e.direction[1], e.direction[2]};
            o.radiance = Vec3f{e.radiance[0], e.radiance[1], e.radiance[2]};
            break;
        case MCSD_EMIT_SUN:
            o.kind = kEmitSun;
            o.cos_cutoff = e.cos_cutoff_angle;
            o.texture = e.id_texture;
            check(o.texture, false);
            o.direction = Vec3f{e.direction[0], e.direction[1], e.direction[2]};
            o.radiance = Vec3f{e.radiance[0], e.radiance[1], e.radiance[2]};
            ig.id_sun = static_cast<uint32_t>(i);
            break;
        case MCSD_EMIT_ENVMAP:
        {
            o.kind = kEmitEnvMap;
            o.texture = e.id_radiance;
            check(o.texture, false);
            o.to_world = Load(e.to_world);
            o.to_local = Inverted(o.to_world);
            const TextureRec &tex = fs.textures[o.texture];
            if (tex.kind != kTexBitmap)
                throw std::runtime_error("radiance texture '" + std::to_string(o.texture) +
                                         "' for emitter '" + std::to_string(i) + "' is not a bitmap.");
            // envmap.cpp:20-68.  The tables are stored [row cdf | row weights |
            // column cdfs] (renderer.cpp:597-606) but addressed through the
            // offsets of emitter.cpp:166-175, which name them differently
            // (quirk Q7): reproduce both halves of that.
            const int w = tex.width, h = tex.height;
            const float w_inv = 1.0f / w, h_inv = 1.0f / h;
            std::vector<float> cdf_rows(h + 1), weight_rows(h), cdf_cols(static_cast<size_t>(w + 1) * h);
            float sum_row = 0.0f;
            for (int y = 0; y < h; ++y)
            {
                float sum_col = 0.0f;
                float *col = &cdf_cols[static_cast<size_t>(y) * (w + 1)];
                for (int x = 0; x < w; ++x)
                {
                    const V3 rgb = texture_color(fs.textures.data(), fs.texels.data(), o.texture,
                                                 V2{x * w_inv, y * h_inv});
                    sum_col += luminance(rgb);
                    col[x + 1] = sum_col;
                }
                col[w] = 1.0f;
                const float norm_col = 1.0f / sum_col;
                for (int x = 1; x < w; ++x)
                    col[w - x] *= norm_col;
                const float weight = sinf((y + 0.5f) * kPi / h);
                weight_rows[y] = weight;
                sum_row += sum_col * weight;
                cdf_rows[y + 1] = sum_row;
            }
            cdf_rows[h] = 1.0f;
            const float norm_row = 1.0f / sum_row;
            for (int y = 1; y < h; ++y)
                cdf_rows[h - y] *= norm_row;
            if (!std::isfinite(sum_row))
                throw std::runtime_error("The environment map contains an invalid floating point value (nan/inf).");
            o.normalization = static_cast<float>(1.0 / D(sum_row * (k2Pi * w_inv) * (kPi * h_inv)));
            o.width = w, o.height = h;
            const uint32_t base = static_cast<uint32_t>(fs.env_tables.size());
            fs.env_tables.insert(fs.env_tables.end(), cdf_rows.begin(), cdf_rows.end());
            fs.env_tables.insert(fs.env_tables.end(), weight_rows.begin(), weight_rows.end());
            fs.env_tables.insert(fs.env_tables.end(), cdf_cols.begin(), cdf_cols.end());
            o.cdf_cols = base;
            o.cdf_rows = base + static_cast<uint32_t>(h) + 1;
            o.weight_rows = base + static_cast<uint32_t>(h + 1) + static_cast<uint32_t>(h);
            ig.id_envmap = static_cast<uint32_t>(i);
            break;
        }
        case MCSD_EMIT_CONSTANT:
            o.kind = kEmitConstant;
            o.radiance = Vec3f{e.radiance[0], e.radiance[1], e.radiance[2]};
            ig.id_envmap = static_cast<uint32_t>(i);
            break;
        default:
            throw std::runtime_error("unknow emitter type.");
        }
        fs.emitters.push_back(o);
    }
    ig.n_emitters = static_cast<uint32_t>(fs.emitters.size());
    if (ig.n_emitters)
        fs.features |= kFeatEmitters;
    if (ig.n_emitters)
        for (const EmitterRec &e : fs.emitters)
            if (e.texture != kNone)
                fs.features |= kFeatTextures;

    // never hand out null pointers for empty tables
    if (fs.analytic.empty())
        fs.analytic.push_back(AnalyticRec{});
    if (fs.light_inst.empty())
        fs.light_inst.push_back(kNone);
    if (fs.textures.empty())
        fs.textures.push_back(TextureRec{});
    if (fs.texels.empty())
        fs.texels.push_back(0.0f);
    if (fs.bsdfs.empty())
        fs.bsdfs.push_back(BsdfRec{});
    if (fs.media.empty())
        fs.media.push_back(MediumRec{});
    if (fs.emitters.empty())
        fs.emitters.push_back(EmitterRec{});
    if (fs.env_tables.empty())
        fs.env_tables.push_back(0.0f);
    if (fs.nodes.empty())
    {
        fs.nodes.assign(2, float4{0, 0, 0, 0});
        fs.node_area.push_back(0.0f);
        fs.tri_pos.assign(3, float4{0, 0, 0, 0});
        fs.tri_attr.assign(9, float4{0, 0, 0, 0});
        fs.instances.push_back(InstanceRec{});
    }
    if (fs.walk_nodes.empty())
        fs.walk_nodes.assign(4, float4{0, 0, 0, 0});
    if (fs.walk_prims.empty())
        fs.walk_prims.assign(3, float4{0, 0, 0, 0});
    if (fs.wide_nodes.empty())
        fs.wide_nodes.assign(4, uint4{0, 0, 0, 0});
    if (fs.pool_nodes.empty())
        fs.pool_nodes.assign(8, float4{0, 0, 0, 0});
    fs.seconds_total = seconds_since(t_begin);
    return fs;
}

} // namespace mcpt
