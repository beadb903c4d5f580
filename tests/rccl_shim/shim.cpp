// TEST INFRASTRUCTURE — a stand-in for librccl on a box with ONE GPU.
//
// The product's N-GPU host (csrc/capi.cpp, mcpt_tiled_renderer_*) binds seven RCCL entry points at run time.  Real RCCL
// refuses a communicator that lists one device twice, so the code path "N ranks, per-rank offsets, grouped send / recv,
// per-rank unpack" cannot execute with N > 1 on the 1-GPU test box.  This library implements exactly those seven entry
// points for LOGICAL ranks that share a device: a send and its matching receive (same group, sender's peer == receiver's
// rank and vice versa, matched in posting order like RCCL does) become one hipMemcpyAsync on the receiver's stream,
// ordered after the sender's stream by an event.  The product loads it only when MCPT_RCCL_LIBRARY names it
// (tests/test_gpu_parity.py); it is never part of a render without that variable.
#include <hip/hip_runtime_api.h>

#include <cstddef>
#include <cstdio>
#include <mutex>
#include <vector>

namespace
{

struct Comm
{
    int rank, size, device;
};

struct Posted
{
    bool send;
    const void *src;
    void *dst;
    size_t bytes;
    int rank, peer;
    hipStream_t stream;
    bool matched;
};

std::mutex g_mu;
int g_depth = 0;
std::vector<Posted> g_posted;
unsigned long long g_messages = 0, g_bytes = 0;

size_t TypeSize(int datatype)
{
    switch (datatype)
    {
    case 0: case 1: return 1;          // ncclInt8 / ncclUint8
    case 6: return 2;                  // ncclFloat16
    case 2: case 3: case 7: return 4;  // ncclInt32 / ncclUint32 / ncclFloat32
    case 4: case 5: case 8: return 8;  // ncclInt64 / ncclUint64 / ncclFloat64
    default: return 0;
    }
}

int Flush()
{
    int rc = 0;
    for (size_t i = 0; i < g_posted.size(); ++i)
    {
        Posted &s = g_posted[i];
        if (!s.send || s.matched)
            continue;
        for (size_t j = 0; j < g_posted.size(); ++j)
        {
            Posted &r = g_posted[j];
            if (r.send || r.matched || r.rank != s.peer || r.peer != s.rank)
                continue;
            if (r.bytes != s.bytes)
                return 3; // ncclInvalidArgument-like: sizes of a matched pair differ
            hipEvent_t ready;
            if (hipEventCreateWithFlags(&ready, hipEventDisableTiming) != hipSuccess)
                return 1;
            bool ok = hipEventRecord(ready, s.stream) == hipSuccess && hipStreamWaitEvent(r.stream, ready, 0) == hipSuccess &&
                      hipMemcpyAsync(r.dst, s.src, s.bytes, hipMemcpyDeviceToDevice, r.stream) == hipSuccess;
            // the sender's buffer must not be overwritten before the copy has read it
            hipEvent_t done;
            ok = ok && hipEventCreateWithFlags(&done, hipEventDisableTiming) == hipSuccess && hipEventRecord(done, r.stream) == hipSuccess &&
                 hipStreamWaitEvent(s.stream, done, 0) == hipSuccess;
            (void)hipEventDestroy(ready);
            if (ok)
                (void)hipEventDestroy(done);
            if (!ok)
                rc = 1;
            s.matched = r.matched = true;
            ++g_messages, g_bytes += s.bytes;
            break;
        }
    }
    for (const Posted &p : g_posted)
        if (!p.matched)
            rc = rc ? rc : 5; // an unmatched operation would hang real RCCL
    g_posted.clear();
    return rc;
}

} // namespace

extern "C"
{

int ncclCommInitAll(void **comms, int ndev, const int *devlist)
{
    if (!comms || ndev < 1)
        return 4;
    for (int k = 0; k < ndev; ++k)
        comms[k] = new Comm{k, ndev, devlist ? devlist[k] : k};
    return 0;
}

int ncclCommDestroy(void *comm)
{
    delete static_cast<Comm *>(comm);
    return 0;
}

int ncclGroupStart()
{
    std::lock_guard<std::mutex> lock(g_mu);
    ++g_depth;
    return 0;
}

int ncclGroupEnd()
{
    std::lock_guard<std::mutex> lock(g_mu);
    if (g_depth == 0)
        return 4;
    return --g_depth == 0 ? Flush() : 0;
}

int ncclSend(const void *sendbuff, size_t count, int datatype, int peer, void *comm, hipStream_t stream)
{
    const Comm *c = static_cast<const Comm *>(comm);
    const size_t size = TypeSize(datatype);
    if (!c || !size || peer < 0 || peer >= c->size)
        return 4;
    std::lock_guard<std::mutex> lock(g_mu);
    g_posted.push_back(Posted{true, sendbuff, nullptr, count * size, c->rank, peer, stream, false});
    return g_depth == 0 ? Flush() : 0;
}

int ncclRecv(void *recvbuff, size_t count, int datatype, int peer, void *comm, hipStream_t stream)
{
    const Comm *c = static_cast<const Comm *>(comm);
    const size_t size = TypeSize(datatype);
    if (!c || !size || peer < 0 || peer >= c->size)
        return 4;
    std::lock_guard<std::mutex> lock(g_mu);
    g_posted.push_back(Posted{false, nullptr, recvbuff, count * size, c->rank, peer, stream, false});
    return g_depth == 0 ? Flush() : 0;
}

const char *ncclGetErrorString(int rc)
{
    switch (rc)
    {
    case 0: return "no error";
    case 1: return "unhandled HIP error (shim)";
    case 3: return "matched send / recv of different sizes (shim)";
    case 4: return "invalid argument (shim)";
    case 5: return "unmatched send or recv in a group (shim): real RCCL would hang";
    default: return "unknown (shim)";
    }
}

// what went through the shim since it was loaded (the tests assert the gather really took this route)
void mcpt_rccl_shim_stats(unsigned long long *messages, unsigned long long *bytes)
{
    std::lock_guard<std::mutex> lock(g_mu);
    if (messages)
        *messages = g_messages;
    if (bytes)
        *bytes = g_bytes;
}

} // extern "C"
