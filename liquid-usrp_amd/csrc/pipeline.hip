// pipeline.hip -- the multi-GPU receive pipeline behind the C-ABI (include/mcrx_hip.h: mcrx_hip_pipeline_*).
//
// One process per GPU.  Sub-slabs of the wideband stream go round robin to the ranks (sub-slab u -> rank u % G); a round is
//   A  channelize this rank's sub-slab into per-destination groups            out[g][tile][c][16], channel = g*Cg + c (MCRX_TILE = 16)
//   B  exchange: chunk g of rank r -> chunk r of rank g                         RCCL, grouped ncclSend / ncclRecv over xGMI
//   C  synchronizer bank of the rank's channel shard over the round             recv[s][tile][c][16] = [tile of the round][c][16]
// with `nbuf` rotating buffer sets, linked by events only -- channelize(c+1) || exchange(c) || sync(c-1) -- and nothing waits on
// the host.  Two HIP streams of the pipeline's own carry this: one for stage A, one for stage B and the launch of stage C (whose
// kernels run on the receiver handle's three internal streams anyway).  A third stream for C, as in rounds 2-3, put the process at
// eight streams and cost the one-GPU run 11 % (154-156 against 172.8 Gsample/s, direct path 178.9: scratch/r4an.sh).  This is liquid-usrp_amd/sharding.py's Pipeline (which stays as the mirror the gloo tests
// drive on CPU) without Python between the stages.  world == 1 runs the same code with no exchange: the channelizer writes
// straight into the synchronizers' buffer.
//
// RCCL is loaded at run time (dlopen "librccl.so.1": the copy a host process already holds -- e.g. PyTorch's -- is reused), so
// the library has no link-time dependency on it and single-GPU users never touch it.  The caller distributes the 128-byte
// ncclUniqueId of rank 0 (mcrx_hip_pipeline_unique_id) by whatever means it has (MPI, a file, torch.distributed).
#include "../../include/mcrx_hip.h"
#include "devel.h"
#include "devscope.hpp"
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <dlfcn.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

namespace {

thread_local std::string g_perr;
int pfail(int code, const std::string &msg) { g_perr = msg; return code; }
#define PCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return pfail(MCRX_EHIP, std::string(#x) + ": " + hipGetErrorString(e_)); } while (0)
#define PRC(x) do { int rc_ = (x); if (rc_ != MCRX_OK) return pfail(rc_, std::string(#x) + ": " + mcrx_hip_last_error()); } while (0)

struct Rccl {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;       // optional (reporting only)
    // every symbol is resolved into a candidate first and the table committed only when all of them were found (a partly filled
    // table behind a non-null `lib` would crash the next caller); once per process, thread safe
    std::once_flag once; bool ok = false;
    bool load()
    {
        std::call_once(once, [this] {
            void *h = nullptr;
            for (const char *name : { "librccl.so.1", "librccl.so" }) { h = dlopen(name, RTLD_NOW | RTLD_GLOBAL); if (h) break; }
            if (!h) return;
            Rccl t;
            bool all = true;
#define SYM(f, n) t.f = reinterpret_cast<decltype(t.f)>(dlsym(h, n)); all = all && t.f != nullptr
            SYM(GetUniqueId, "ncclGetUniqueId"); SYM(CommInitRank, "ncclCommInitRank"); SYM(CommDestroy, "ncclCommDestroy");
            SYM(GroupStart, "ncclGroupStart"); SYM(GroupEnd, "ncclGroupEnd"); SYM(Send, "ncclSend"); SYM(Recv, "ncclRecv");
            SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
            if (!all) { dlclose(h); return; }
            lib = h; GetUniqueId = t.GetUniqueId; CommInitRank = t.CommInitRank; CommDestroy = t.CommDestroy; GroupStart = t.GroupStart;
            GroupEnd = t.GroupEnd; Send = t.Send; Recv = t.Recv; GetErrorString = t.GetErrorString;
            CommCount = reinterpret_cast<decltype(CommCount)>(dlsym(h, "ncclCommCount"));
            ok = true;
        });
        return ok;
    }
};
Rccl g_rccl;
#define NCHK(x) do { ncclResult_t r_ = (x); if (r_ != ncclSuccess) return pfail(MCRX_EHIP, std::string(#x) + ": " + g_rccl.GetErrorString(r_)); } while (0)

constexpr unsigned kMaxBuf = 8;

}  // namespace

struct mcrx_hip_pipeline_s {
    int device = -1;            // the HIP device the handle was created on: every entry point runs with it current (devscope.hpp)
    mcrx_hip_t rx = nullptr;
    size_t halo = 13;                                           // blocks of filter history in front of a sub-slab: mcrx_hip_history_blocks (13; 27 with front_end = 1)
    int rank = 0, world = 1;
    unsigned N = 0, K = 0, cg = 0, hist = 0, nbuf = 3;
    size_t Tc = 0, tiles = 0, per = 0, hist_elems = 0;         // per: cf32 elements of one (source rank) chunk of a round
    float *out[kMaxBuf] = {}, *recv[kMaxBuf] = {};             // cf32 buffers as float pairs (world == 1: out[i] aliases recv[i] + history)
    hipStream_t sA = nullptr, sB = nullptr, sC = nullptr;
    hipEvent_t evA[kMaxBuf] = {}, evB[kMaxBuf] = {}, evC[kMaxBuf] = {}, ev_after = nullptr;
    uint64_t ticket[kMaxBuf] = {}; bool has_ticket[kMaxBuf] = {};
    uint64_t rounds = 0;
    uint64_t nco_base = 0;                                      // wideband samples in front of round 0 (the oscillator is never reset: lib/multichannelrx.cc:144)
    uint64_t chan_base = 0;                                     // channel-rate position of round 0's first block (moves at a reset)
    bool fresh_zeroed = true;                                   // recv[0]'s history tiles are zero already (creation); false after a reset: the first round zeroes them
    float *din[kMaxBuf] = {};                                   // push_host: this rank's sub-slab with its 13 halo blocks in front, rotating
    float *hin[kMaxBuf] = {};                                   // ... and the pinned host buffers they are copied from (mcrx_hip_pipeline_host_buffer)
    hipEvent_t evH[kMaxBuf] = {}; bool hin_busy[kMaxBuf] = {};  // recorded behind the copy out of hin[i]
    uint64_t host_pushes = 0;
    ncclComm_t comm = nullptr;
    bool timing = false; std::vector<std::pair<hipEvent_t, hipEvent_t>> xev; size_t xused = 0; double x_ms = 0; uint64_t x_n = 0;
};

extern "C" const char *mcrx_hip_pipeline_last_error(void) { return g_perr.c_str(); }

extern "C" int mcrx_hip_pipeline_unique_id(void *id128)
{
    if (!id128) return pfail(MCRX_EINVAL, "null argument");
    if (!g_rccl.load()) return pfail(MCRX_EUNSUPP, "librccl.so.1 not found");
    ncclUniqueId id;
    NCHK(g_rccl.GetUniqueId(&id));
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    memcpy(id128, &id, sizeof(id));
    return MCRX_OK;
}

extern "C" int mcrx_hip_pipeline_destroy(mcrx_hip_pipeline_t p)
{
    mcrx::DevScope dev_scope_(p ? p->device : -1);
    if (!p) return MCRX_OK;
    (void)hipDeviceSynchronize();
    if (p->comm) (void)g_rccl.CommDestroy(p->comm);
    for (unsigned i = 0; i < kMaxBuf; i++) {
        if (p->recv[i]) (void)hipFree(p->recv[i]);
        if (p->din[i]) (void)hipFree(p->din[i]);
        if (p->hin[i]) (void)hipHostFree(p->hin[i]);
        if (p->evH[i]) (void)hipEventDestroy(p->evH[i]);
        if (p->world > 1 && p->out[i]) (void)hipFree(p->out[i]);
        hipEvent_t ev[3] = { p->evA[i], p->evB[i], p->evC[i] };
        for (hipEvent_t e : ev) if (e) (void)hipEventDestroy(e);
    }
    for (auto &pr : p->xev) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
    if (p->ev_after) (void)hipEventDestroy(p->ev_after);
    if (p->sC && p->sC != p->sB && p->sC != p->sA) (void)hipStreamDestroy(p->sC);
    if (p->sB && p->sB != p->sA) (void)hipStreamDestroy(p->sB);
    if (p->sA) (void)hipStreamDestroy(p->sA);
    delete p;
    return MCRX_OK;
}

extern "C" int mcrx_hip_pipeline_create(mcrx_hip_pipeline_t *out, mcrx_hip_t rx, int rank, int world, const void *unique_id128,
                                        size_t sub_blocks, unsigned nbuf)
{
    if (!out || !rx) return pfail(MCRX_EINVAL, "null argument");
    *out = nullptr;
    const unsigned N = mcrx_hip_num_channels(rx);
    if (world < 1 || rank < 0 || rank >= world || N % (unsigned)world) return pfail(MCRX_EINVAL, "ranks must divide the channel count");
    if (sub_blocks == 0 || sub_blocks % MCRX_TILE) return pfail(MCRX_EINVAL, "sub-slabs are whole tiles of MCRX_TILE = 16 blocks");
    if (nbuf < 2 || nbuf > kMaxBuf) nbuf = 3;
    if (world > 1 && !unique_id128) return pfail(MCRX_EINVAL, "world > 1 needs rank 0's ncclUniqueId (mcrx_hip_pipeline_unique_id)");
    mcrx_hip_pipeline_t p = new mcrx_hip_pipeline_s();
    p->device = mcrx::current_device();
    auto bail = [&](int rc) { mcrx_hip_pipeline_destroy(p); return rc; };
    p->rx = rx; p->rank = rank; p->world = world; p->N = N; p->K = 2 * N; p->cg = N / (unsigned)world; p->nbuf = nbuf;
    p->Tc = sub_blocks; p->tiles = sub_blocks / MCRX_TILE; p->hist = mcrx_hip_history_tiles(rx); p->halo = mcrx_hip_history_blocks(rx);
    p->per = p->tiles * p->cg * MCRX_TILE; p->hist_elems = (size_t)p->hist * p->cg * MCRX_TILE;
    const size_t recv_elems = p->hist_elems + (size_t)world * p->per;
    for (unsigned i = 0; i < nbuf; i++) {
        if (hipMalloc((void **)&p->recv[i], recv_elems * 2 * sizeof(float)) != hipSuccess) return bail(pfail(MCRX_ENOMEM, "hipMalloc failed"));
        if (hipMemset(p->recv[i], 0, recv_elems * 2 * sizeof(float)) != hipSuccess) return bail(pfail(MCRX_EHIP, "hipMemset failed"));
        if (world > 1) {
            if (hipMalloc((void **)&p->out[i], (size_t)world * p->per * 2 * sizeof(float)) != hipSuccess) return bail(pfail(MCRX_ENOMEM, "hipMalloc failed"));
        } else p->out[i] = p->recv[i] + 2 * p->hist_elems;          // one group = the synchronizers' own layout: no exchange, no copy
        if (hipEventCreateWithFlags(&p->evA[i], hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&p->evB[i], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&p->evC[i], hipEventDisableTiming) != hipSuccess) return bail(pfail(MCRX_EHIP, "hipEventCreate failed"));
    }
    if (hipEventCreateWithFlags(&p->ev_after, hipEventDisableTiming) != hipSuccess) return bail(pfail(MCRX_EHIP, "hipEventCreate failed"));
    {   const char *ev = devel_env("MCRX_PIPE_STREAMS");           // experiments: 1 = one stream for all three stages, 3 = one each (rounds 2-3)
        const int ns = ev ? atoi(ev) : 2;
        if (hipStreamCreateWithFlags(&p->sA, hipStreamNonBlocking) != hipSuccess) return bail(pfail(MCRX_EHIP, "hipStreamCreate failed"));
        if (ns <= 1) p->sB = p->sA;
        else if (hipStreamCreateWithFlags(&p->sB, hipStreamNonBlocking) != hipSuccess) return bail(pfail(MCRX_EHIP, "hipStreamCreate failed"));
        if (ns <= 1) p->sC = p->sA;
        else if (ns == 2) p->sC = p->sB;
        else if (hipStreamCreateWithFlags(&p->sC, hipStreamNonBlocking) != hipSuccess) return bail(pfail(MCRX_EHIP, "hipStreamCreate failed"));
    }
    if (hipDeviceSynchronize() != hipSuccess) return bail(pfail(MCRX_EHIP, "device synchronize failed"));      // (the zeroed buffers, before the non-blocking streams use them)
    if (world > 1) {
        if (!g_rccl.load()) return bail(pfail(MCRX_EUNSUPP, "librccl.so.1 not found"));
        ncclUniqueId id; memcpy(&id, unique_id128, sizeof(id));
        ncclResult_t r = g_rccl.CommInitRank(&p->comm, world, id, rank);
        if (r != ncclSuccess) return bail(pfail(MCRX_EHIP, std::string("ncclCommInitRank: ") + g_rccl.GetErrorString(r)));
    }
    *out = p;
    return MCRX_OK;
}

extern "C" int mcrx_hip_pipeline_time_exchange(mcrx_hip_pipeline_t p, int on)
{
    if (!p) return pfail(MCRX_EINVAL, "null handle");
    p->timing = on != 0;
    return MCRX_OK;
}

extern "C" int mcrx_hip_pipeline_exchange_ms(mcrx_hip_pipeline_t p, double *total_ms, uint64_t *rounds, int reset)
{
    mcrx::DevScope dev_scope_(p ? p->device : -1);
    if (!p) return pfail(MCRX_EINVAL, "null handle");
    for (size_t i = 0; i < p->xused; i++) {
        float ms = 0;
        PCHK(hipEventSynchronize(p->xev[i].second));
        PCHK(hipEventElapsedTime(&ms, p->xev[i].first, p->xev[i].second));
        p->x_ms += ms; p->x_n++;
    }
    p->xused = 0;
    if (total_ms) *total_ms = p->x_ms;
    if (rounds) *rounds = p->x_n;
    if (reset) { p->x_ms = 0; p->x_n = 0; }
    return MCRX_OK;
}

extern "C" uint64_t mcrx_hip_pipeline_bytes_sent_per_round(mcrx_hip_pipeline_t p)
{
    return p ? (uint64_t)p->per * 8ull * (uint64_t)(p->world - 1) : 0;
}

// One round: this rank's sub-slab (sub_blocks blocks resident in HBM; d_halo = the 13 blocks in front of it in the stream, NULL
// = zeros), the exchange, the synchronizers of the rank's channel shard over the round.  `after_stream`: the stream that
// produced d_iq_sub (NULL: the legacy default stream; MCRX_STREAM_READY: nothing to wait for).  Returns after enqueuing.
extern "C" int mcrx_hip_pipeline_push(mcrx_hip_pipeline_t p, const void *d_iq_sub, const void *d_halo, void *after_stream)
{
    mcrx::DevScope dev_scope_(p ? p->device : -1);
    if (!p || !d_iq_sub) return pfail(MCRX_EINVAL, "null argument");
    const uint64_t c = p->rounds; const unsigned nb = p->nbuf, i = (unsigned)(c % nb);
    float *out = p->out[i], *recv = p->recv[i], *fresh = recv + 2 * p->hist_elems;
    // ---- A: channelize into per-destination groups
    // (the pipeline's streams are non-blocking: they do not order against the legacy default stream by themselves, so a NULL
    //  after_stream -- the default stream, e.g. torch's current one -- gets its event like any other; only MCRX_STREAM_READY skips it)
    if (after_stream != MCRX_STREAM_READY) { PCHK(hipEventRecord(p->ev_after, (hipStream_t)after_stream)); PCHK(hipStreamWaitEvent(p->sA, p->ev_after, 0)); }
    if (c >= nb) {
        PCHK(hipStreamWaitEvent(p->sA, p->evB[i], 0));                         // the exchange that last read out[i]
        if (p->world == 1) {                                                    // out[i] IS recv[i]: also its last readers
            if (p->has_ticket[i]) PRC(mcrx_hip_stream_wait_launch(p->rx, p->ticket[i], p->sA));
            PCHK(hipStreamWaitEvent(p->sA, p->evC[(i + 1) % nb], 0));
        }
    }
    const uint64_t first = p->nco_base + (c * (uint64_t)p->world + (uint64_t)p->rank) * (uint64_t)p->Tc * (uint64_t)p->K;
    PRC(mcrx_hip_channelize(p->rx, d_iq_sub, p->Tc, first, d_halo, out, (unsigned)p->world, p->sA));
    PCHK(hipEventRecord(p->evA[i], p->sA));
    // ---- B: time shards -> channel shards
    PCHK(hipStreamWaitEvent(p->sB, p->evA[i], 0));
    if (p->world > 1) {
        if (c >= nb) {
            if (p->has_ticket[i]) PRC(mcrx_hip_stream_wait_launch(p->rx, p->ticket[i], p->sB));     // the synchronizers that last read recv[i]
            PCHK(hipStreamWaitEvent(p->sB, p->evC[(i + 1) % nb], 0));                              // ... and the history copy that read its tail
        }
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if (p->timing) {
            if (p->xused == p->xev.size()) {
                hipEvent_t a, b; PCHK(hipEventCreate(&a)); PCHK(hipEventCreate(&b)); p->xev.emplace_back(a, b);
            }
            e0 = p->xev[p->xused].first; e1 = p->xev[p->xused].second; p->xused++;
            PCHK(hipEventRecord(e0, p->sB));
        }
        const size_t cnt = p->per * 2;                                          // floats per peer (RCCL has no complex type)
        NCHK(g_rccl.GroupStart());
        for (int g = 0; g < p->world; g++) {
            NCHK(g_rccl.Send(out + (size_t)g * cnt, cnt, ncclFloat, g, p->comm, p->sB));
            NCHK(g_rccl.Recv(fresh + (size_t)g * cnt, cnt, ncclFloat, g, p->comm, p->sB));
        }
        NCHK(g_rccl.GroupEnd());
        if (e1) PCHK(hipEventRecord(e1, p->sB));
    }
    PCHK(hipEventRecord(p->evB[i], p->sB));
    // ---- C: synchronizer history in front (tail of the previous round), then the bank over the round
    PCHK(hipStreamWaitEvent(p->sC, p->evB[i], 0));
    if (c > 0) {
        const float *prev = p->recv[(c - 1) % nb];
        const size_t total = p->hist_elems + (size_t)p->world * p->per;
        PCHK(hipMemcpyAsync(recv, prev + 2 * (total - p->hist_elems), p->hist_elems * 2 * sizeof(float), hipMemcpyDeviceToDevice, p->sC));
    } else if (!p->fresh_zeroed) {          // first round after a reset: what sits in front of it belongs to the stream before the reset
        PCHK(hipMemsetAsync(recv, 0, p->hist_elems * 2 * sizeof(float), p->sC));
    }
    p->fresh_zeroed = false;
    PCHK(hipEventRecord(p->evC[i], p->sC));
    const int64_t first_chan = (int64_t)(p->chan_base + c * (uint64_t)p->world * (uint64_t)p->Tc) - (int64_t)p->hist * MCRX_TILE;
    const size_t nsamp = (size_t)p->hist * MCRX_TILE + (size_t)p->world * p->Tc;
    PRC(mcrx_hip_sync(p->rx, recv, (uint64_t)first_chan, nsamp, p->sC));
    p->ticket[i] = mcrx_hip_launches(p->rx) - 1; p->has_ticket[i] = true;
    p->rounds++;
    return MCRX_OK;
}

// The same round fed from host memory (the reference's Execute takes host buffers: lib/multichannelrx.cc:155): `iq` holds the 13
// blocks in front of this rank's sub-slab followed by the sub-slab itself, (13 + sub_blocks) * 2N cf32, contiguous.  The samples are
// staged through `nbuf` rotating PINNED host buffers: a caller that fills the buffer mcrx_hip_pipeline_host_buffer() hands out pays no
// copy; any other pointer is copied into that buffer first, so the caller's memory is free again when the call returns either way
// (the call used to run hipMemcpyAsync from the caller's pageable buffer and say nothing about its lifetime).  The host
// waits only for the copy out of the pinned buffer it is about to refill -- the push of nbuf rounds ago -- never for a round's kernels.
static int host_slot(mcrx_hip_pipeline_t p, unsigned *slot)
{
    const unsigned i = (unsigned)(p->host_pushes % p->nbuf);
    const size_t bytes = (size_t)(p->halo + p->Tc) * p->K * 2 * sizeof(float);
    if (!p->hin[i]) {
        PCHK(hipHostMalloc((void **)&p->hin[i], bytes, hipHostMallocDefault));
        PCHK(hipMalloc((void **)&p->din[i], bytes));
        PCHK(hipEventCreateWithFlags(&p->evH[i], hipEventDisableTiming));
    }
    if (p->hin_busy[i]) { PCHK(hipEventSynchronize(p->evH[i])); p->hin_busy[i] = false; }
    *slot = i;
    return MCRX_OK;
}
extern "C" int mcrx_hip_pipeline_host_buffer(mcrx_hip_pipeline_t p, float **buf, size_t *nsamples)
{
    mcrx::DevScope dev_scope_(p ? p->device : -1);
    if (!p || !buf) return pfail(MCRX_EINVAL, "null argument");
    unsigned i = 0;
    int rc = host_slot(p, &i);
    if (rc != MCRX_OK) return rc;
    *buf = p->hin[i];
    if (nsamples) *nsamples = (size_t)(p->halo + p->Tc) * p->K;
    return MCRX_OK;
}
extern "C" int mcrx_hip_pipeline_push_host(mcrx_hip_pipeline_t p, const float *iq_with_halo)
{
    mcrx::DevScope dev_scope_(p ? p->device : -1);
    if (!p || !iq_with_halo) return pfail(MCRX_EINVAL, "null argument");
    unsigned i = 0;
    int rc = host_slot(p, &i);
    if (rc != MCRX_OK) return rc;
    const size_t n = (size_t)(p->halo + p->Tc) * p->K;
    if (iq_with_halo != p->hin[i]) memcpy(p->hin[i], iq_with_halo, n * 2 * sizeof(float));
    // din[i]'s last reader, the channelizer of nbuf host pushes ago, ran on sA: the copy is ordered behind it there
    PCHK(hipMemcpyAsync(p->din[i], p->hin[i], n * 2 * sizeof(float), hipMemcpyHostToDevice, p->sA));
    PCHK(hipEventRecord(p->evH[i], p->sA));
    p->hin_busy[i] = true; p->host_pushes++;
    return mcrx_hip_pipeline_push(p, p->din[i] + p->halo * p->K * 2, p->din[i], MCRX_STREAM_READY);
}

// multichannelrx::Reset() on a sharded receiver (lib/multichannelrx.cc:135-153 -- legal mid-stream, lib/multichanneltxrx.cc:333 calls it):
// synchronizers back to SEEK, channelizer windows empty (the caller's next halo is zeros), block alignment restarts with the next
// sample; the oscillator runs on (:144), so the `dropped_samples` of the unfinished round the caller discards still count for its
// phase: extra_samples = samples the caller was handed and did not push (> 0), or minus the samples it pushed that were never in the
// stream (< 0: zero padding behind the last real sample of a round it completed itself so that everything before the Reset is
// synchronized, as the reference has it).  Every rank calls this at the same point of the stream (they are all handed the same
// calls); no exchange is needed.  Frames decoded so far stay deliverable through the handle.
extern "C" int mcrx_hip_pipeline_reset(mcrx_hip_pipeline_t p, int64_t extra_samples)
{
    mcrx::DevScope dev_scope_(p ? p->device : -1);
    if (!p) return pfail(MCRX_EINVAL, "null handle");
    PCHK(hipStreamSynchronize(p->sA)); PCHK(hipStreamSynchronize(p->sB)); PCHK(hipStreamSynchronize(p->sC));
    const uint64_t blocks = p->rounds * (uint64_t)p->world * (uint64_t)p->Tc;
    p->nco_base += blocks * (uint64_t)p->K + (uint64_t)extra_samples;          // (two's complement: a negative adjustment subtracts)
    p->chan_base += blocks;
    p->rounds = 0; p->fresh_zeroed = false;
    for (unsigned i = 0; i < kMaxBuf; i++) p->has_ticket[i] = false;
    PRC(mcrx_hip_reset_at(p->rx, p->chan_base));          // (waits for the device: nothing of the old stream is in flight after this)
    return MCRX_OK;
}

// ranks of the RCCL communicator the exchange runs on, as RCCL itself counts them (1 without a communicator: world == 1); what a
// benchmark line quotes to prove that N ranks took part.  -1: this RCCL has no ncclCommCount.
extern "C" int mcrx_hip_pipeline_comm_count(mcrx_hip_pipeline_t p)
{
    mcrx::DevScope dev_scope_(p ? p->device : -1);
    if (!p) return -1;
    if (!p->comm) return 1;
    if (!g_rccl.CommCount) return -1;
    int n = -1;
    return g_rccl.CommCount(p->comm, &n) == ncclSuccess ? n : -1;
}

extern "C" int mcrx_hip_pipeline_wait(mcrx_hip_pipeline_t p)
{
    mcrx::DevScope dev_scope_(p ? p->device : -1);
    if (!p) return pfail(MCRX_EINVAL, "null handle");
    PCHK(hipStreamSynchronize(p->sA)); PCHK(hipStreamSynchronize(p->sB)); PCHK(hipStreamSynchronize(p->sC));
    return MCRX_OK;
}
