// Collectives under the C ABI (include/csdr.h, "csdr_comm"): one process per GPU, RCCL over xGMI.
//
// What the reference does on one host thread -- `mix` = foldl1 (+) over the channel list (Trans.hs:119-122) behind
// `mux (replicate nch demod) . firpfbchChannelizer nc` (SoapySDR.hs:217-222) -- becomes, when the -c N channels are split over the
// GPUs of a node: every rank's chain folds its own channels (strict left fold, kernels_generic.hip k_mix) and ONE all-reduce(SUM) of
// nf output elements per chunk adds the partial mixes.  The hybrid partition needs one all-to-all of the channel-major
// CF32 plane instead (time stripes in, channel blocks out); channel shards on a common stream need the chunk broadcast.
//
// librccl.so is ~570 MB, so it is NOT a link-time dependency of libcsdr_hip.so: the first csdr_comm_* call dlopen()s "librccl.so.1"
// (the soname of both /opt/rocm/lib's and PyTorch's copy, so a process that already runs torch.distributed shares that instance).
// No fallback: without RCCL the calls fail with CSDR_ERR_INVALID and say so.  Product code: nothing here touches oracle/.
#include "../../include/csdr.h"
#include "csdr_internal.h"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstring>
#include <mutex>
#include <vector>

namespace {

struct Rccl {
    void *so = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*GetVersion)(int *) = nullptr;
    bool ok = false;
};

Rccl g_rccl;
std::once_flag g_once;
std::string g_load_error;

void load_rccl()
{
    const char *names[] = {getenv("CSDR_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char *n : names) {
        if (!n || !*n) continue;
        g_rccl.so = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (g_rccl.so) break;
        g_load_error = dlerror();
    }
    if (!g_rccl.so) return;
#define SYM(field, name)                                                                       \
    do {                                                                                       \
        g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(g_rccl.so, name));      \
        if (!g_rccl.field) { g_load_error = std::string("librccl: missing symbol ") + name; return; } \
    } while (0)
    SYM(GetUniqueId, "ncclGetUniqueId");
    SYM(CommInitRank, "ncclCommInitRank");
    SYM(CommDestroy, "ncclCommDestroy");
    SYM(AllReduce, "ncclAllReduce");
    SYM(Broadcast, "ncclBroadcast");
    SYM(Send, "ncclSend");
    SYM(Recv, "ncclRecv");
    SYM(GroupStart, "ncclGroupStart");
    SYM(GroupEnd, "ncclGroupEnd");
    SYM(GetErrorString, "ncclGetErrorString");
    SYM(GetVersion, "ncclGetVersion");
#undef SYM
    g_rccl.ok = true;
}

int need_rccl()
{
    std::call_once(g_once, load_rccl);
    if (!g_rccl.ok) {
        csdr::set_error("csdr_comm: RCCL is not available (%s); there is no fallback for the collectives", g_load_error.c_str());
        return CSDR_ERR_INVALID;
    }
    return CSDR_OK;
}

int nccl_fail(ncclResult_t r, const char *what)
{
    csdr::set_error("RCCL: %s failed: %s", what, g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?");
    return CSDR_ERR_HIP;
}
#define CSDR_NCCL(call, what)                                   \
    do {                                                        \
        ncclResult_t r__ = (call);                              \
        if (r__ != ncclSuccess) return nccl_fail(r__, what);    \
    } while (0)

}  // namespace

struct csdr_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1, device = 0;
    void *d_scratch = nullptr; size_t scratch_bytes = 0;     // host-buffer entry point: the partial mix on its way through the all-reduce
};

static_assert(CSDR_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "csdr_comm id = ncclUniqueId");

extern "C" {

int csdr_comm_unique_id(void *id_out)
{
    if (!id_out) { csdr::set_error("csdr_comm_unique_id: null buffer"); return CSDR_ERR_INVALID; }
    if (int rc = need_rccl()) return rc;
    ncclUniqueId id;
    CSDR_NCCL(g_rccl.GetUniqueId(&id), "ncclGetUniqueId");
    std::memcpy(id_out, id.internal, CSDR_COMM_ID_BYTES);
    return CSDR_OK;
}

int csdr_comm_create(int rank, int world, const void *id_in, int device, csdr_comm **out)
{
    if (!out || !id_in || world < 1 || rank < 0 || rank >= world) {
        csdr::set_error("csdr_comm_create: rank %d of world %d", rank, world);
        return CSDR_ERR_INVALID;
    }
    *out = nullptr;
    if (int rc = need_rccl()) return rc;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) { csdr::set_error("csdr_comm_create: no HIP device"); return CSDR_ERR_NODEV; }
    if (device < 0) CSDR_HIP(hipGetDevice(&device));
    if (device >= ndev) { csdr::set_error("csdr_comm_create: device %d of %d", device, ndev); return CSDR_ERR_INVALID; }
    CSDR_HIP(hipSetDevice(device));
    csdr_comm *c = new csdr_comm;
    c->rank = rank; c->world = world; c->device = device;
    ncclUniqueId id;
    std::memcpy(id.internal, id_in, CSDR_COMM_ID_BYTES);
    const ncclResult_t r = g_rccl.CommInitRank(&c->comm, world, id, rank);
    if (r != ncclSuccess) { delete c; return nccl_fail(r, "ncclCommInitRank"); }
    *out = c;
    return CSDR_OK;
}

int csdr_comm_rank(const csdr_comm *c) { return c ? c->rank : -1; }
int csdr_comm_world(const csdr_comm *c) { return c ? c->world : 0; }

int csdr_comm_destroy(csdr_comm *c)
{
    if (!c) return CSDR_OK;
    if (c->comm) {
        (void)hipSetDevice(c->device);
        (void)hipDeviceSynchronize();
        CSDR_NCCL(g_rccl.CommDestroy(c->comm), "ncclCommDestroy");
    }
    if (c->d_scratch) (void)hipFree(c->d_scratch);
    delete c;
    return CSDR_OK;
}

int csdr_comm_broadcast(csdr_comm *c, void *d_buf, size_t bytes, int root, void *stream)
{
    if (!c || (!d_buf && bytes) || root < 0 || root >= c->world) { csdr::set_error("csdr_comm_broadcast: bad argument"); return CSDR_ERR_INVALID; }
    if (bytes == 0) return CSDR_OK;
    CSDR_HIP(hipSetDevice(c->device));
    CSDR_NCCL(g_rccl.Broadcast(d_buf, d_buf, bytes, ncclUint8, root, c->comm, static_cast<hipStream_t>(stream)), "ncclBroadcast");
    return CSDR_OK;
}

int csdr_comm_allreduce_f32(csdr_comm *c, void *d_buf, size_t count, void *stream)
{
    if (!c || (!d_buf && count)) { csdr::set_error("csdr_comm_allreduce_f32: bad argument"); return CSDR_ERR_INVALID; }
    if (count == 0) return CSDR_OK;
    CSDR_HIP(hipSetDevice(c->device));
    CSDR_NCCL(g_rccl.AllReduce(d_buf, d_buf, count, ncclFloat32, ncclSum, c->comm, static_cast<hipStream_t>(stream)), "ncclAllReduce");
    return CSDR_OK;
}

int csdr_chain_process_device_mix(csdr_chain *h, csdr_comm *c, const void *d_in, uint32_t n_in, void *d_out, uint32_t *n_out, void *stream)
{
    if (!h || !c) { csdr::set_error("csdr_chain_process_device_mix: null handle"); return CSDR_ERR_INVALID; }
    csdr_chain_cfg cfg;
    if (int rc = csdr_chain_get_cfg(h, &cfg)) return rc;
    if (!cfg.mix) { csdr::set_error("csdr_chain_process_device_mix: the chain was not created with mix"); return CSDR_ERR_INVALID; }
    uint32_t n = 0;
    if (int rc = csdr_chain_process_device(h, d_in, n_in, d_out, &n, stream)) return rc;
    if (n_out) *n_out = n;
    // the rank's partial mix (strict left fold over its own channels) -> the sum over ranks, in place, on the caller's stream
    return csdr_comm_allreduce_f32(c, d_out, (size_t)n * (csdr_chain_out_elem_size(h) / 4u), stream);
}

int csdr_chain_process_mix(csdr_chain *h, csdr_comm *c, const float *in_cf32, uint32_t n_in, void *out, uint32_t *n_out)
{
    if (!h || !c) { csdr::set_error("csdr_chain_process_mix: null handle"); return CSDR_ERR_INVALID; }
    csdr_chain_cfg cfg;
    if (int rc = csdr_chain_get_cfg(h, &cfg)) return rc;
    if (!cfg.mix) { csdr::set_error("csdr_chain_process_mix: the chain was not created with mix"); return CSDR_ERR_INVALID; }
    uint32_t n = 0;
    if (int rc = csdr_chain_process(h, in_cf32, n_in, out, &n)) return rc;
    if (n_out) *n_out = n;
    const size_t bytes = (size_t)n * csdr_chain_out_elem_size(h);
    if (bytes == 0) return CSDR_OK;
    // nf output elements (16-32 KiB at the reference's chunk): up, summed over the ranks, down again
    CSDR_HIP(hipSetDevice(c->device));
    if (c->scratch_bytes < bytes) {
        if (c->d_scratch) { (void)hipFree(c->d_scratch); c->d_scratch = nullptr; c->scratch_bytes = 0; }
        CSDR_HIP(hipMalloc(&c->d_scratch, bytes));
        c->scratch_bytes = bytes;
    }
    CSDR_HIP(hipMemcpy(c->d_scratch, out, bytes, hipMemcpyHostToDevice));
    if (int rc = csdr_comm_allreduce_f32(c, c->d_scratch, bytes / 4, nullptr)) return rc;
    CSDR_HIP(hipStreamSynchronize(nullptr));
    CSDR_HIP(hipMemcpy(out, c->d_scratch, bytes, hipMemcpyDeviceToHost));
    return CSDR_OK;
}

int csdr_hybrid_exchange(csdr_comm *c, const void *d_plane, void *d_recv, uint32_t chan_per_rank, const uint32_t *stripe_frames,
                         uint32_t elem_bytes, void *stream)
{
    if (!c || !stripe_frames || chan_per_rank == 0 || (elem_bytes != 4 && elem_bytes != 8)) {
        csdr::set_error("csdr_hybrid_exchange: bad argument");
        return CSDR_ERR_INVALID;
    }
    const int G = c->world, g = c->rank;
    const size_t mine = stripe_frames[g];
    hipStream_t s = static_cast<hipStream_t>(stream);
    CSDR_HIP(hipSetDevice(c->device));
    // plane = [G][chan_per_rank][mine] (channel-major, destination p's rows are contiguous); recv = stretch p at sum_{q<p} cn * frames[q]
    std::vector<size_t> roff(G + 1, 0);
    for (int p = 0; p < G; p++) roff[p + 1] = roff[p] + (size_t)chan_per_rank * stripe_frames[p] * elem_bytes;
    const char *src = static_cast<const char *>(d_plane);
    char *dst = static_cast<char *>(d_recv);
    const size_t blk = (size_t)chan_per_rank * mine * elem_bytes;
    if (blk && (!d_plane || !d_recv)) { csdr::set_error("csdr_hybrid_exchange: null buffer"); return CSDR_ERR_INVALID; }
    if (blk) CSDR_HIP(hipMemcpyAsync(dst + roff[g], src + (size_t)g * blk, blk, hipMemcpyDeviceToDevice, s));   // my own block
    if (G == 1) return CSDR_OK;
    CSDR_NCCL(g_rccl.GroupStart(), "ncclGroupStart");
    // a failing send / receive must not leave the group open on this thread (every later collective of any communicator would be
    // queued into it instead of failing): remember the first error, always close the group, then report
    ncclResult_t first = ncclSuccess; const char *where = nullptr;
    for (int d = 1; d < G && first == ncclSuccess; d++) {
        // pair schedule: in step d rank g sends to g + d and receives from g - d (every xGMI link carries one block each way)
        const int to = (g + d) % G, from = (g - d + G) % G;
        if (blk) { first = g_rccl.Send(src + (size_t)to * blk, blk, ncclUint8, to, c->comm, s); where = "ncclSend"; }
        const size_t rb = roff[from + 1] - roff[from];
        if (rb && first == ncclSuccess) { first = g_rccl.Recv(dst + roff[from], rb, ncclUint8, from, c->comm, s); where = "ncclRecv"; }
    }
    const ncclResult_t closed = g_rccl.GroupEnd();
    if (first != ncclSuccess) return nccl_fail(first, where);
    if (closed != ncclSuccess) return nccl_fail(closed, "ncclGroupEnd");
    return CSDR_OK;
}

}  // extern "C"
