// Device build of the reference-topology LBVH.
//
// Produces, on the GPU, exactly the tree the host builder of commit.cpp
// (`LinearBvh`) and the reference (src/rtcore/accel/bvh_builder.cpp:74-207)
// produce: 64-bit keys (30-bit Morton code of the box centre << 32 | primitive
// index), sorted; the hierarchy is the binary radix tree of the sorted keys —
// the reference's top-down "largest index that shares a longer prefix with the
// first key" split rule defines the same tree as Karras' bottom-up construction
// (HPG 2012) because the keys are distinct.  The tree is emitted directly in the
// stackless layout of device_scene.h (pre-order numbering + skip links):
//     index(node) = 2 * first_leaf(node) + (left turns on the root->node path)
//     skip(node)  = index(node) + 2 * leaves(node) - 1, or "end" when the node's
//                   range ends at the last leaf
// Boxes are exact min/max unions and node areas left + right sums in tree order,
// so every output word is bit-identical to the host builder's.
//
// Kernels (all one thread per element, 256 per workgroup):
//   bounds   wavefront-shuffle + atomic min/max on order-preserving integer keys
//   keys     centre -> unit cube -> Morton code (same float ops as commit.cpp)
//   sort     rocPRIM / hipCUB radix sort of the 64-bit keys
//   topology Karras: range, split, children and parents of internal node i
//   emit     per node: walk to the root counting left turns -> pre-order index
//   fit      leaves walk up; the second arrival at a node merges its children
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include "../vecmath.h"
#include "lbvh_build.h"

namespace mcpt
{
namespace
{

constexpr int kThreads = 256;

// float <-> unsigned keys whose integer order is the float order
__device__ __forceinline__ uint32_t OrderedBits(float f)
{
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float FromOrderedBits(uint32_t k)
{
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k);
}

struct BuildState
{
    uint32_t bounds[6]; // ordered-bit min of lo.xyz, max of hi.xyz
};

__global__ void init_bounds(BuildState *st)
{
    if (threadIdx.x < 3)
        st->bounds[threadIdx.x] = 0xFFFFFFFFu;
    else if (threadIdx.x < 6)
        st->bounds[threadIdx.x] = 0u;
}

__global__ void reduce_bounds(uint32_t n, const float *__restrict__ boxes, BuildState *st)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    // aabb.cpp:8: the empty box is (+max, -max); out-of-range lanes contribute it
    float v[6] = {kMaxFloat, kMaxFloat, kMaxFloat, kLowestFloat, kLowestFloat, kLowestFloat};
    if (i < n)
        for (int k = 0; k < 6; ++k)
            v[k] = boxes[6 * static_cast<size_t>(i) + k];
    for (int k = 0; k < 6; ++k)
    {
        float x = v[k];
        for (int off = 32; off > 0; off >>= 1)
        {
            const float y = __shfl_xor(x, off);
            x = k < 3 ? fminf(x, y) : fmaxf(x, y);
        }
        if ((threadIdx.x & 63) == 0)
        {
            if (k < 3)
                atomicMin(&st->bounds[k], OrderedBits(x));
            else
                atomicMax(&st->bounds[k], OrderedBits(x));
        }
    }
}

__device__ __forceinline__ uint32_t Dilate10(uint32_t v) // bvh_builder.cpp:14-21
{
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}

__global__ void make_keys(uint32_t n, const float *__restrict__ boxes, const BuildState *st, uint64_t *__restrict__ keys)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    const V3 all_lo = V3{FromOrderedBits(st->bounds[0]), FromOrderedBits(st->bounds[1]), FromOrderedBits(st->bounds[2])};
    const V3 all_hi = V3{FromOrderedBits(st->bounds[3]), FromOrderedBits(st->bounds[4]), FromOrderedBits(st->bounds[5])};
    const float *b = boxes + 6 * static_cast<size_t>(i);
    const V3 lo = V3{b[0], b[1], b[2]}, hi = V3{b[3], b[4], b[5]};
    const V3 extent = all_hi - all_lo;
    const V3 centre = (lo + hi) * 0.5f;
    const V3 unit = (centre - all_lo) / extent; // reciprocal-multiply, as the host builder
    const float x = fminf(fmaxf(unit.x * 1024.0f, 0.0f), 1023.0f), y = fminf(fmaxf(unit.y * 1024.0f, 0.0f), 1023.0f),
                z = fminf(fmaxf(unit.z * 1024.0f, 0.0f), 1023.0f);
    const uint32_t code = Dilate10(static_cast<uint32_t>(x)) * 4 + Dilate10(static_cast<uint32_t>(y)) * 2 +
                          Dilate10(static_cast<uint32_t>(z));
    keys[i] = (static_cast<uint64_t>(code) << 32) | i;
}

// Topology of the binary radix tree.  Internal node i in [0, n-2], leaf j in [0, n-1].
// A child reference is a leaf position when its high bit is set.
constexpr uint32_t kLeafRef = 0x80000000u;

struct Topology
{
    uint32_t *first, *last;      // per internal node: leaf range
    uint32_t *left, *right;      // per internal node: child references
    uint32_t *parent_internal;   // per internal node (root: kNone)
    uint32_t *parent_leaf;       // per leaf
    uint32_t *left_child_flag_internal; // 1 when the node is its parent's left child
    uint32_t *left_child_flag_leaf;
};

__device__ __forceinline__ int Prefix(const uint64_t *keys, uint32_t n, int i, int j)
{
    if (j < 0 || j >= static_cast<int>(n))
        return -1;
    return __clzll(static_cast<long long>(keys[i] ^ keys[j])); // keys are distinct
}

__global__ void build_topology(uint32_t n, const uint64_t *__restrict__ keys, Topology t)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= static_cast<int>(n) - 1)
        return;
    // direction of the range and its other end (Karras 2012, algorithm of fig. 4)
    const int d = Prefix(keys, n, i, i + 1) - Prefix(keys, n, i, i - 1) >= 0 ? 1 : -1;
    const int delta_min = Prefix(keys, n, i, i - d);
    int l_max = 2;
    while (Prefix(keys, n, i, i + l_max * d) > delta_min)
        l_max <<= 1;
    int l = 0;
    for (int s = l_max >> 1; s >= 1; s >>= 1)
        if (Prefix(keys, n, i, i + (l + s) * d) > delta_min)
            l += s;
    const int j = i + l * d;
    const int delta_node = Prefix(keys, n, i, j);
    int s = 0;
    for (int step = (l + 1) >> 1;; step = (step + 1) >> 1)
    {
        if (Prefix(keys, n, i, i + (s + step) * d) > delta_node)
            s += step;
        if (step == 1)
            break;
    }
    const int gamma = i + s * d + (d < 0 ? -1 : 0); // last leaf of the left part
    const int lo = d > 0 ? i : j, hi = d > 0 ? j : i;
    t.first[i] = static_cast<uint32_t>(lo), t.last[i] = static_cast<uint32_t>(hi);
    if (lo == gamma)
    {
        t.left[i] = kLeafRef | static_cast<uint32_t>(gamma);
        t.parent_leaf[gamma] = static_cast<uint32_t>(i), t.left_child_flag_leaf[gamma] = 1;
    }
    else
    {
        t.left[i] = static_cast<uint32_t>(gamma);
        t.parent_internal[gamma] = static_cast<uint32_t>(i), t.left_child_flag_internal[gamma] = 1;
    }
    if (hi == gamma + 1)
    {
        t.right[i] = kLeafRef | static_cast<uint32_t>(gamma + 1);
        t.parent_leaf[gamma + 1] = static_cast<uint32_t>(i), t.left_child_flag_leaf[gamma + 1] = 0;
    }
    else
    {
        t.right[i] = static_cast<uint32_t>(gamma + 1);
        t.parent_internal[gamma + 1] = static_cast<uint32_t>(i), t.left_child_flag_internal[gamma + 1] = 0;
    }
    if (i == 0)
        t.parent_internal[0] = kNone;
}

// Pre-order slot of every node + its links.  Thread k < n-1: internal node k;
// thread n-1+j: leaf j.
__global__ void emit_nodes(uint32_t n, const uint64_t *__restrict__ keys, Topology t, uint32_t *__restrict__ slot_internal,
                           uint32_t *__restrict__ slot_leaf, float4 *__restrict__ nodes, float *__restrict__ node_area,
                           const float *__restrict__ boxes, const float *__restrict__ areas)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= 2 * n - 1)
        return;
    const bool is_leaf = k >= n - 1;
    const uint32_t id = is_leaf ? k - (n - 1) : k;
    const uint32_t first = is_leaf ? id : t.first[id], last = is_leaf ? id : t.last[id];
    uint32_t left_turns = 0;
    uint32_t flag = is_leaf ? t.left_child_flag_leaf[id] : t.left_child_flag_internal[id];
    uint32_t up = is_leaf ? t.parent_leaf[id] : t.parent_internal[id];
    if (n == 1)
        up = kNone;
    while (up != kNone)
    {
        left_turns += flag;
        flag = t.left_child_flag_internal[up];
        up = t.parent_internal[up];
    }
    const uint32_t slot = 2 * first + left_turns;
    const uint32_t size = 2 * (last - first + 1) - 1;
    const uint32_t skip = last == n - 1 ? kEndOfTree : slot + size;
    if (is_leaf)
    {
        slot_leaf[id] = slot;
        const uint32_t object = static_cast<uint32_t>(keys[id] & 0xFFFFFFFFull);
        const float *b = boxes + 6 * static_cast<size_t>(object);
        nodes[2 * static_cast<size_t>(slot)] = float4{b[0], b[1], b[2], __uint_as_float(skip)};
        nodes[2 * static_cast<size_t>(slot) + 1] = float4{b[3], b[4], b[5], __uint_as_float(object)};
        node_area[slot] = areas[object];
    }
    else
    {
        slot_internal[id] = slot;
        // box and area are filled by fit_boxes; links now
        nodes[2 * static_cast<size_t>(slot)].w = __uint_as_float(skip);
        nodes[2 * static_cast<size_t>(slot) + 1].w = __uint_as_float(kNoObject);
    }
}

__device__ __forceinline__ float LoadCoherent(const float *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float4 LoadCoherent(const float4 *p)
{
    const float *f = reinterpret_cast<const float *>(p);
    return float4{LoadCoherent(f), LoadCoherent(f + 1), LoadCoherent(f + 2), LoadCoherent(f + 3)};
}

// Bottom-up: every leaf walks towards the root; the second thread to reach an
// internal node owns it (both children are complete) and merges them.
__global__ void fit_boxes(uint32_t n, Topology t, const uint32_t *__restrict__ slot_internal,
                          const uint32_t *__restrict__ slot_leaf, uint32_t *__restrict__ arrivals, float4 *nodes,
                          float *node_area)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n || n == 1)
        return;
    uint32_t node = t.parent_leaf[j];
    while (node != kNone)
    {
        __threadfence();
        if (atomicAdd(&arrivals[node], 1u) == 0)
            return; // the sibling subtree is not finished yet
        __threadfence();
        const uint32_t l = t.left[node], r = t.right[node];
        const uint32_t sl = (l & kLeafRef) ? slot_leaf[l & ~kLeafRef] : slot_internal[l];
        const uint32_t sr = (r & kLeafRef) ? slot_leaf[r & ~kLeafRef] : slot_internal[r];
        // children may have been written by other CUs during this kernel: device-scope loads
        const float4 llo = LoadCoherent(nodes + 2 * static_cast<size_t>(sl)),
                     lhi = LoadCoherent(nodes + 2 * static_cast<size_t>(sl) + 1),
                     rlo = LoadCoherent(nodes + 2 * static_cast<size_t>(sr)),
                     rhi = LoadCoherent(nodes + 2 * static_cast<size_t>(sr) + 1);
        const uint32_t s = slot_internal[node];
        float4 &out_lo = nodes[2 * static_cast<size_t>(s)], &out_hi = nodes[2 * static_cast<size_t>(s) + 1];
        out_lo.x = fminf(llo.x, rlo.x), out_lo.y = fminf(llo.y, rlo.y), out_lo.z = fminf(llo.z, rlo.z); // aabb.cpp:50-53
        out_hi.x = fmaxf(lhi.x, rhi.x), out_hi.y = fmaxf(lhi.y, rhi.y), out_hi.z = fmaxf(lhi.z, rhi.z);
        node_area[s] = LoadCoherent(node_area + sl) + LoadCoherent(node_area + sr);
        node = t.parent_internal[node];
    }
}

struct Scratch
{
    void *ptr = nullptr;
    ~Scratch()
    {
        if (ptr)
            (void)hipFree(ptr);
    }
};

inline uint32_t Blocks(uint32_t n) { return (n + kThreads - 1) / kThreads; }

} // namespace

#define MCPT_TRY(expr)                \
    do                                \
    {                                 \
        const hipError_t e_ = (expr); \
        if (e_ != hipSuccess)         \
            return e_;                \
    } while (0)

hipError_t BuildLbvhOnDevice(uint32_t n, const float *boxes_dev, const float *areas_dev, float4 *nodes_dev,
                             float *node_area_dev, hipStream_t stream)
{
    if (n == 0)
        return hipSuccess;
    // one allocation: state | keys in | keys out | topology (8 arrays) | slots (2) | arrivals | sort temp
    size_t sort_bytes = 0;
    MCPT_TRY(hipcub::DeviceRadixSort::SortKeys(nullptr, sort_bytes, static_cast<uint64_t *>(nullptr),
                                               static_cast<uint64_t *>(nullptr), static_cast<int>(n), 0, 64, stream));
    auto align = [](size_t x) { return (x + 255) & ~size_t(255); };
    const size_t words = n; // arrays of n uint32 (internal arrays use n-1 of them)
    size_t at = 0;
    auto take = [&](size_t bytes)
    {
        const size_t here = at;
        at += align(bytes);
        return here;
    };
    const size_t o_state = take(sizeof(BuildState)), o_k0 = take(8 * size_t(n)), o_k1 = take(8 * size_t(n));
    size_t o_topo[8];
    for (size_t &o : o_topo)
        o = take(4 * words);
    const size_t o_si = take(4 * words), o_sl = take(4 * words), o_arr = take(4 * words), o_sort = take(sort_bytes);
    Scratch scratch;
    MCPT_TRY(hipMalloc(&scratch.ptr, at));
    char *base = static_cast<char *>(scratch.ptr);
    BuildState *st = reinterpret_cast<BuildState *>(base + o_state);
    uint64_t *keys_in = reinterpret_cast<uint64_t *>(base + o_k0), *keys = reinterpret_cast<uint64_t *>(base + o_k1);
    Topology t;
    uint32_t **fields[8] = {&t.first, &t.last, &t.left, &t.right, &t.parent_internal, &t.parent_leaf,
                            &t.left_child_flag_internal, &t.left_child_flag_leaf};
    for (int k = 0; k < 8; ++k)
        *fields[k] = reinterpret_cast<uint32_t *>(base + o_topo[k]);
    uint32_t *slot_internal = reinterpret_cast<uint32_t *>(base + o_si), *slot_leaf = reinterpret_cast<uint32_t *>(base + o_sl),
             *arrivals = reinterpret_cast<uint32_t *>(base + o_arr);

    hipLaunchKernelGGL(init_bounds, dim3(1), dim3(64), 0, stream, st);
    hipLaunchKernelGGL(reduce_bounds, dim3(Blocks(n)), dim3(kThreads), 0, stream, n, boxes_dev, st);
    hipLaunchKernelGGL(make_keys, dim3(Blocks(n)), dim3(kThreads), 0, stream, n, boxes_dev, st, keys_in);
    MCPT_TRY(hipcub::DeviceRadixSort::SortKeys(base + o_sort, sort_bytes, keys_in, keys, static_cast<int>(n), 0, 64, stream));
    MCPT_TRY(hipMemsetAsync(arrivals, 0, 4 * words, stream));
    if (n > 1)
        hipLaunchKernelGGL(build_topology, dim3(Blocks(n - 1)), dim3(kThreads), 0, stream, n, keys, t);
    hipLaunchKernelGGL(emit_nodes, dim3(Blocks(2 * n - 1)), dim3(kThreads), 0, stream, n, keys, t, slot_internal, slot_leaf,
                       nodes_dev, node_area_dev, boxes_dev, areas_dev);
    hipLaunchKernelGGL(fit_boxes, dim3(Blocks(n)), dim3(kThreads), 0, stream, n, t, slot_internal, slot_leaf, arrivals,
                       nodes_dev, node_area_dev);
    MCPT_TRY(hipGetLastError());
    return hipStreamSynchronize(stream); // scratch is released on return
}

} // namespace mcpt
