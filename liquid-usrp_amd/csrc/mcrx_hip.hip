// mcrx_hip.hip -- C-ABI implementation (include/mcrx_hip.h): handle management, HBM
// buffers, streaming glue between the channelizer and the synchronizer bank, frame
// harvesting.  Host code only; the kernels live in channelizer.hip / ofdmsync.hip.
#include "../../include/mcrx_hip.h"
#include "design.hpp"
#include "devel.h"
#include "devmath.h"
#include "devscope.hpp"
#include "kernels.h"
#include "txcode.hpp"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

using namespace mcrx;

#define MCRX_SLOTS 8        /* most per-launch buffer sets (channel tiles, job list, per-job scratch); launch k uses slot k % nslots */
#define MCRX_GENS 4         /* result generations (records + arenas), a ring: one fills while older ones are harvested or dropped */

static thread_local std::string g_err;
static void set_err(const char *what, hipError_t e, const char *file, int line)
{
    char buf[512];
    snprintf(buf, sizeof(buf), "%s: %s (%s:%d)", what, hipGetErrorString(e), file, line);
    g_err = buf;
}
static int fail(int code, const char *msg) { g_err = msg; return code; }
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { set_err(#x, e_, __FILE__, __LINE__); return MCRX_EHIP; } } while (0)
#define RC(x) do { int rc_ = (x); if (rc_ != MCRX_OK) return rc_; } while (0)

extern "C" const char *mcrx_hip_last_error(void) { return g_err.c_str(); }

// ---------------------------------------------------------------- process-wide coding tables
static std::mutex g_cod_mu;
static bool g_cod_ready = false;
static CodingDev g_cod;

template <class T> static int upload_raw(T **dst, const T *src, size_t n)
{
    HIPCHK(hipMalloc((void **)dst, std::max<size_t>(n, 1) * sizeof(T)));
    if (n) HIPCHK(hipMemcpy(*dst, src, n * sizeof(T), hipMemcpyHostToDevice));
    return MCRX_OK;
}
static int coding_tables(CodingDev *out)
{
    std::lock_guard<std::mutex> lk(g_cod_mu);
    if (!g_cod_ready) {
        CodingTables *t = new CodingTables();
        uint8_t *f, *g; uint32_t *d, *e;
        RC(upload_raw(&d, t->crc_byte, 256));
        RC(upload_raw(&e, &t->crc_zadv[0][0][0], 16 * 4 * 256));
        RC(upload_raw(&f, &t->qam16_nb[0][0], 64));
        RC(upload_raw(&g, &t->qam64_nb[0][0], 256));
        g_cod.crc_byte = d; g_cod.crc_zadv = e;
        g_cod.qam16_nb = f; g_cod.qam64_nb = g;
        delete t;
        g_cod_ready = true;
    }
    *out = g_cod;
    return MCRX_OK;
}

// ---------------------------------------------------------------- small kernels
__global__ void hist_update_kernel(const float2 *old_hist, const float2 *x, uint64_t nx, float2 *new_hist, uint64_t nh)
{
    // new_hist = last nh samples of concat(old_hist[nh], x[nx])
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nh) return;
    uint64_t pos = nx + i;
    new_hist[i] = (pos < nh) ? old_hist[pos] : x[pos - nh];
}

// oversampled front end, stage by stage (cfg.front_end = 2; front_end = 1 is the same chain folded into channelizer_kernel): the
// oscillator as its own pass (the oversampled bank takes plain samples) ...
__global__ void nco_mix_kernel(const float2 *x, float2 *y, uint64_t n, uint32_t first_lo, uint32_t dtheta)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    y[i] = mix_down_hw(x[i], (first_lo + (uint32_t)i) * dtheta);
}
// ... and the rate 2 -> 1 adapter behind it: per kept channel liquid's half-band decimator (resamp2_crcf, m = 7):
//     y[k] = 0.5 * ( Y[2k - 13] + sum_{i < 14} h1[i] Y[2 (k - 13 + i)] )
// over the bank's steps Y[s][0 .. M-1] (32 steps of history sit in front of s = 0), written as the synchronizers'
// (channel, tile) granules.  A thread takes one granule: channels are consecutive across a wave, so every read of a
// step row is one contiguous line and the 128-byte granules of a wave form one 8 KB store.
__global__ void halfband_adapter_kernel(const float2 *Y, uint32_t M, uint32_t N, uint32_t ntiles, const float *h1, float2 *out)
{
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x, tile = blockIdx.y;
    if (c >= N || tile >= ntiles) return;
    float h[14];
#pragma unroll
    for (int i = 0; i < 14; i++) h[i] = h1[i];
    constexpr int TS = MCRX_TILE_S;
    const long long k0 = (long long)tile * TS;
    float2 ev[13 + TS];                             // even steps 2 (k0 - 13) .. 2 (k0 + TS - 1)
#pragma unroll
    for (int i = 0; i < 13 + TS; i++) ev[i] = Y[(2 * (k0 - 13 + i)) * (long long)M + c];
    float4 *dst = reinterpret_cast<float4 *>(out + ((size_t)tile * N + c) * TS);
    float2 y[TS];
#pragma unroll
    for (int t = 0; t < TS; t++) {
        float2 acc = make_float2(0.f, 0.f);
#pragma unroll
        for (int i = 0; i < 14; i++) { acc.x += h[i] * ev[t + i].x; acc.y += h[i] * ev[t + i].y; }
        const float2 d = Y[(2 * (k0 + t) - 13) * (long long)M + c];
        y[t] = make_float2(0.5f * (d.x + acc.x), 0.5f * (d.y + acc.y));
    }
#pragma unroll
    for (int t = 0; t < TS; t += 2) dst[t / 2] = make_float4(y[t].x, y[t].y, y[t + 1].x, y[t + 1].y);
}

// ---------------------------------------------------------------- handle
// harvested payloads / equalised symbols on the host: pinned and never value-initialised (hundreds of MB per
// harvest at 512 channels), grown geometrically
struct HostArena {
    uint8_t *p = nullptr; size_t size = 0, cap = 0;
    int grow_to(size_t n)
    {
        if (n <= cap) { size = n; return MCRX_OK; }
        size_t ncap = std::max<size_t>(n, cap + cap / 2);
        uint8_t *np = nullptr;
        if (hipHostMalloc((void **)&np, ncap, hipHostMallocDefault) != hipSuccess) return MCRX_ENOMEM;
        if (size) memcpy(np, p, size);
        if (p) (void)hipHostFree(p);
        p = np; cap = ncap; size = n;
        return MCRX_OK;
    }
    void clear() { size = 0; }
    void release() { if (p) (void)hipHostFree(p); p = nullptr; size = cap = 0; }
};

struct mcrx_hip_s {
    int device = -1;            // the HIP device the handle was created on: every entry point runs with it current (devscope.hpp)
    unsigned N = 0, K = 0, M = 0, cp = 0, taper = 0;
    bool bypass = false;                    // one synchronizer fed channel-rate samples, no channelizer (ofdmtxrx)
    OfdmDesign od;
    mcrx_hip_config cfg{};
    std::vector<float> taps;
    uint32_t dtheta = 0, slab_blocks = 0, ncu = 256;
    const float *d_taps = nullptr;
    SyncConsts sc{};
    std::vector<void *> owned;              // device allocations freed at destroy
    // synchronizer bank
    unsigned ch_first = 0, nch = 0;
    uint32_t max_payload = 0, max_enc = 0, max_syms = 0, max_rec = 0;
    uint32_t max_jobs = 0;                  // job list entries per launch: the frames (max_rec) + the entries the segment waves fill for nothing (two per segment) + void reservations
    uint64_t arena_cap = 0;
    ChanState *d_st = nullptr; uint8_t *d_hbits = nullptr; float2 *d_R = nullptr;
    uint8_t *d_soft = nullptr, *d_tmpa = nullptr, *d_tmpb = nullptr; float2 *d_syms = nullptr;
    // results: two generations.  Launches write into generation `gen`; a harvest closes it (new launches go to
    // the other one) and copies it out once its last launch has finished -- the GPU keeps working meanwhile.
    FrameRec *d_rec[MCRX_GENS] = {}; uint8_t *d_arena[MCRX_GENS] = {}, *d_sarena[MCRX_GENS] = {};
    uint32_t *d_nrec[MCRX_GENS] = {}; unsigned long long *d_arena_used[MCRX_GENS] = {};
    int gen = 0; bool gen_used[MCRX_GENS] = {}, gen_closed[MCRX_GENS] = {}, gen_abandoned[MCRX_GENS] = {};
    uint32_t list_seen[3] = { 0, 0, 0 }, list_age[3] = { 64, 64, 64 };      // launch_sync: grids of the list-driven launches
    uint32_t round_lds_pad = 0;      // the same cap outside walk mode (MCRX_PAYLOAD_LDS_PAD)
    uint32_t walk_lds_pad = 13312;   // walk mode: unused LDS per payload worker = at most three of them per SIMD, the fourth slot is the walking scouts' (95 -> 110 Gsample/s on ragged traffic)
    int payload_fr = 1, payload_lean = 1, payload_xb = 63;      // MCRX_PAYLOAD_FR / _LEAN / _XB: which build of the M = 64 payload workers (tests)
    int debug = 0, no_fast = 0, seek_burst = 1, acq_mode = 0; bool free_run = false;      // MCRX_DEBUG (trace bits), MCRX_NO_FAST, MCRX_FREE_RUN: read once, at creation
    hipStream_t acq_stream = nullptr;        // the stream the last launch's acquisition / placement kernels ran on (they write the generation's counters)
    uint64_t gen_close_seq[MCRX_GENS] = {}, close_counter = 0;      // order in which generations were closed (= delivery order)
    hipEvent_t ev_gen[MCRX_GENS] = {};       // recorded behind the last launch that wrote into the generation
    hipEvent_t ev_clean[MCRX_GENS] = {};     // recorded behind the zeroing of a collected generation's counters: the next launch into it starts behind that
    bool clean_pending[MCRX_GENS] = {};
    uint64_t sarena_cap = 0;
    // per-launch slots
    PayloadJob *d_jobs[MCRX_SLOTS] = {}; uint32_t *d_njobs = nullptr; float2 *d_jR[MCRX_SLOTS] = {};
    uint32_t *d_gen[MCRX_SLOTS] = {}, *d_qam[MCRX_SLOTS] = {}, *d_live[MCRX_SLOTS] = {};
    uint2 *d_vit_scratch = nullptr; uint32_t vit_rows = 0, vit_waves = 0, vit_mode = 0; uint32_t *d_vit_passes = nullptr;      // the K = 7 decoder's decision rows (kernels.h: vit_scratch)
    uint64_t seq = 0;                       // synchronizer launches so far (slot = seq % nslots)
    unsigned nslots = 5;                    // buffer sets in use: the channelizer and the acquisition chain of up to nslots - 1 pushes run ahead of the payload workers
    uint32_t *d_stats = nullptr;            // speculation statistics (SyncArgs::stats)
    // internal streams: channelizer | acquisition (speculative waves, scouts, placement) | payload workers + decode.
    // Launch k+1's acquisition -- a chain of dependent events per channel, a few waves per CU -- runs under launch
    // k's payload/decode kernels, and launch k+1's channelizer as soon as CUs free up.
    bool pipelined = true;
    hipStream_t s_scout = nullptr, s_work = nullptr, s_copy = nullptr;
    // the launches that normally find nothing to do (kernels.h, split_rest), while that is what they have been finding
    hipStream_t s_side = nullptr; hipEvent_t ev_side[MCRX_SLOTS] = {}, ev_side_last = nullptr; bool side_last = false;

    hipEvent_t ev_ready[MCRX_SLOTS] = {}, ev_scout[MCRX_SLOTS] = {}, ev_done[MCRX_SLOTS] = {}, ev_in = nullptr, ev_consumed = nullptr, ev_tmp[4] = {};
    int last_slot = -1; size_t last_ntiles = 0;     // where the synchronizer history (tail of the previous launch) sits
    uint32_t spec_stride = MCRX_SPEC_MAX;           // slots per channel in d_spec (grows when a push holds more frames per channel: launch_sync)
    SpecSlot *d_spec = nullptr; float2 *d_spec_R = nullptr; int64_t *d_pred = nullptr, *d_anchor = nullptr, *d_seekst = nullptr; uint32_t *d_pred_n = nullptr;
    bool spec = false;
    uint32_t *h_hint = nullptr, *d_hint = nullptr;     // pinned, device-mapped: longest coded frame of the last launch
    uint8_t *d_jsoft[MCRX_SLOTS] = {}, *d_jtmp[MCRX_SLOTS] = {};
    bool scout = true, scout_tables = true, il_tried = false;
    int anchor_kind = 0; uint32_t anchor_retry = 0, anc_filled = 0, anc_adopted = 0;      // cadenced traffic: 0 / 1 = the lattice carried over from the previous push (absolute / from the push's beginning), 2 = an anchor phase (launch_sync)
    uint32_t nseg_fixed = 0, seg_frames = 4; float frames_per_push = 0.f; uint64_t last_nsamp = 0;      // segment-parallel acquisition: launch_sync
    bool cadenced = true; uint32_t cad_same = 0, cad_frames = 0; int cad_count = 0;                       // ... its anchor phase
    // scouts of the acquisition rounds: 1 = the lean scout's unbudgeted build (sync_walk_kernel: 256 + 40 registers, no spills), 0 = the
    // 168-register build (293 spilled) that round 2 made to fit beside four 80-register payload workers.  With eight 56-register
    // workers per SIMD neither fits beside them any more, and a scout that adopts 100 frames of a channel pays for every spill:
    // 8 channels 54.7 -> 64.8 Gsample/s, 512 channels 166.4 -> 169.9 (same box, same run; MCRX_LEAN_BUILD)
    int lean_build = 1;
    int seg_walker = 1;                     // 1: the segment waves are the general state machine's kernel (default); mcrx_hip_config::scout_build = 2: acq_lean.hpp's
    // streaming state
    uint64_t total_samples = 0;             // wideband samples accepted since creation (NCO phase)
    uint64_t stage_first = 0;               // absolute index of h_stage[0]
    int64_t chan_samples = 0;               // channel-rate samples produced since creation
    float2 *d_hist[2] = { nullptr, nullptr }; int hist_cur = 0;
    float2 *d_in = nullptr;                 // device staging (stage_cap samples)
    float2 *h_stage = nullptr; size_t stage_cap = 0, stage_fill = 0;   // pinned host staging (samples)
    float2 *d_chan[MCRX_SLOTS] = {}; size_t chan_cap_tiles = 0;
    unsigned hist_tiles = 0; uint64_t defer = 0;
    // taps per column of the bank channelizer_kernel runs and the blocks of FIR history it needs in front of every push: 14 / 13 = the
    // reference's firpfbch (m = 7), 28 / 27 = the oversampled front end folded into one bank (cfg.front_end = 1; channelizer.hip)
    unsigned chan_P = 14, hist_blocks = 13, col_shift = 0;
    // oversampled front end stage by stage (cfg.front_end = 2, the form the oracle runs; kept as the cross-check of front_end = 1): the bank, its input (28 N samples of history in front) and output (32
    // steps of history in front), both double buffered so that a push's history comes from the other buffer
    bool oversampled = false; mcrx_hip_pfb2_t pfb2 = nullptr; const float *d_h1 = nullptr;
    float2 *d_pfin[2] = { nullptr, nullptr }, *d_pfout[2] = { nullptr, nullptr }; size_t pf_cap_blocks = 0; int pf_cur = 0;
    uint64_t pf_in_valid = 0, pf_steps = 0; size_t pf_last_blocks = 0; bool pf_have_last = false;
    hipStream_t stream = nullptr;
    // large host buffers skip the staging copy: chunks go from the caller's memory to one of two device buffers
    float2 *d_direct[2] = { nullptr, nullptr }; size_t direct_cap = 0; int direct_idx = 0; bool direct_used[2] = { false, false };
    hipStream_t copy_stream = nullptr; hipEvent_t ev_dcopy[2] = { nullptr, nullptr }, ev_ddone[2] = { nullptr, nullptr };
    uint64_t min_frame = 1;                 // channel-rate samples of the shortest possible frame
    uint64_t pending_bound = 0;             // upper bound of frame records produced since the last harvest
    uint64_t drain_touch = 0;
    double t_copy = 0, t_harvest = 0, t_run = 0, t_wait = 0, t_d2h = 0, b_d2h = 0, t_grow = 0, t_cnt = 0, t_post = 0;   // MCRX_DEBUG=8: host seconds spent per phase of the bulk path
    // per-kernel HIP event rings (MCRX_NKERNELS of them, see mcrx_hip.h); pairs (start, stop)
    std::vector<hipEvent_t> evring[MCRX_NKERNELS];
    size_t ev_used[MCRX_NKERNELS] = {};
    double ev_ms_total[MCRX_NKERNELS] = {}; uint64_t ev_count[MCRX_NKERNELS] = {};
    float ev_last[MCRX_NKERNELS] = {};

    bool ev_on = false;                     // mcrx_hip_kernel_timing(): the pairs are recorded on request only -- ten more packets per push on the handle's
                                            // streams cost an 8-channel receiver of 0.25 ms pushes 9 % (scratch/r6/t23.sh: 267 -> 244 us per push)
    hipEvent_t ev_base = nullptr; bool ev_dump = false;      // (development builds, MCRX_EVT_DUMP: every pair's times from the first event on, on stderr -- scratch/r6/evt_timeline.py)
    int ev_begin(int which, hipStream_t st)
    {
        if (!ev_on) return MCRX_OK;
        if (ev_dump && !ev_base) { HIPCHK(hipEventCreate(&ev_base)); HIPCHK(hipEventRecord(ev_base, st)); }
        // ring full: fold the older half (long finished in a running stream, so the host does not stall on it)
        if (ev_used[which] + 2 > evring[which].size()) RC(ev_resolve(which, evring[which].size() / 2));
        HIPCHK(hipEventRecord(evring[which][ev_used[which]], st));
        return MCRX_OK;
    }
    int ev_end(int which, hipStream_t st)
    {
        if (!ev_on) return MCRX_OK;
        HIPCHK(hipEventRecord(evring[which][ev_used[which] + 1], st));
        ev_used[which] += 2;
        return MCRX_OK;
    }
    int ev_resolve(int which, size_t upto = ~(size_t)0)          // fold finished launches into the totals
    {
        const size_t n = std::min(upto & ~(size_t)1, ev_used[which]);
        for (size_t i = 0; i < n; i += 2) {
            float ms = 0;
            HIPCHK(hipEventSynchronize(evring[which][i + 1]));
            HIPCHK(hipEventElapsedTime(&ms, evring[which][i], evring[which][i + 1]));
            if (ev_dump && ev_base) {
                float t0 = 0;
                if (hipEventElapsedTime(&t0, ev_base, evring[which][i]) == hipSuccess) fprintf(stderr, "[evt] %d %.1f %.1f\n", which, t0 * 1e3f, (t0 + ms) * 1e3f);
                else (void)hipGetLastError();
            }
            ev_ms_total[which] += ms; ev_count[which]++; ev_last[which] = ms;
        }
        std::rotate(evring[which].begin(), evring[which].begin() + n, evring[which].end());
        ev_used[which] -= n;
        return MCRX_OK;
    }
    // harvested frames (host)
    std::vector<FrameRec> recs; HostArena arena_host, sarena_host; size_t next_frame = 0;
    uint64_t dropped = 0;

    template <class T> int upload(const T **dst, const T *src, size_t n)
    {
        T *p = nullptr;
        RC(upload_raw(&p, src, n));
        owned.push_back(p); *dst = p;
        return MCRX_OK;
    }
    template <class T> int alloc_raw(T **dst, size_t n)           // (scratch that is written before it is read: not cleared)
    {
        HIPCHK(hipMalloc((void **)dst, std::max<size_t>(n, 1) * sizeof(T)));
        owned.push_back(*dst);
        return MCRX_OK;
    }
    template <class T> int alloc(T **dst, size_t n)
    {
        HIPCHK(hipMalloc((void **)dst, std::max<size_t>(n, 1) * sizeof(T)));
        HIPCHK(hipMemset(*dst, 0, std::max<size_t>(n, 1) * sizeof(T)));
        owned.push_back(*dst);
        return MCRX_OK;
    }
};

static unsigned pow2ceil(unsigned v) { unsigned p = 1; while (p < v) p <<= 1; return p; }

// De-interleaver gather tables (SyncConsts::il_map) for every coded length the LDS decode path can meet: CRC-32 + outer
// Hamming(12,8), Golay(24,12) or the K = 7 convolutional code, no inner code, payloads up to the handle's limit, coded frames up to
// the 56 KiB of soft bits a workgroup stages.  16 bytes per coded byte and length (40 MB at 1200-byte payloads, 120 MB at the
// default 2048), built on the device by pushing indices through the inverse interleaver (ofdmsync.hip: ilmap_build_kernel).
// They depend on (payload limit, staging cap) only, so there is ONE copy per device and process, shared by every handle with those
// limits and reference counted (ADVICE r3 / VERDICT r4 #8: every handle -- single-channel ofdmtxrx ones included -- used to build
// its own at creation behind a hipDeviceSynchronize, and leaked the scratch on a failing call).
struct IlShared { int dev; uint32_t max_payload, cap; uint16_t *d_map; uint32_t *d_off; uint32_t n; int refs; };
static std::mutex g_il_mu;
static std::vector<IlShared> g_il;

static int il_tables_acquire(uint32_t max_payload, uint32_t max_enc, const uint16_t **map, const uint32_t **off, uint32_t *n, hipStream_t st)
{
    *map = nullptr; *off = nullptr; *n = 0;
    int dev = 0;
    HIPCHK(hipGetDevice(&dev));
    const uint32_t cap = std::min<uint32_t>(max_enc, 56u * 1024u / 8u);
    // (round 6, ADVICE r5: the table is built OUTSIDE the lock -- 40-120 MB of allocations and a stream synchronize used to stall the first
    //  push of every other handle and thread behind it -- and published under it with a second look: whoever lost a race frees its copy)
    auto find_shared = [&]() -> bool {
        for (auto &t : g_il)
            if (t.dev == dev && t.max_payload == max_payload && t.cap == cap) { t.refs++; *map = t.d_map; *off = t.d_off; *n = t.n; return true; }
        return false;
    };
    { std::lock_guard<std::mutex> lk(g_il_mu); if (find_shared()) return MCRX_OK; }
    std::vector<uint32_t> offv(cap + 1, ~0u), lens, offs;
    uint64_t total = 0;
    static const int outer[3] = { FEC_HAMMING128, FEC_GOLAY2412, FEC_CONV_V27 };
    for (int oc = 0; oc < 3; oc++)
        for (uint32_t np = 0; np <= max_payload; np++) {
            const uint32_t e = packet_enc_len(np, CRC_32, FEC_NONE, outer[oc]);
            if (e == 0 || e > cap || offv[e] != ~0u || e >= 65536u) continue;
            offv[e] = (uint32_t)total; lens.push_back(e); offs.push_back((uint32_t)total); total += e;
        }
    if (lens.empty() || total * 8 >= (1ull << 32)) return MCRX_OK;           // (no table: the decoder's in-place passes serve every length)
    uint16_t *d_map = nullptr; uint8_t *d_lo = nullptr, *d_hi = nullptr; uint32_t *d_lens = nullptr, *d_offs = nullptr, *d_off = nullptr;
    // (On the handle's own stream.  Round 5's first library created a stream for this and destroyed it again: streams are mapped onto the
    //  hardware queues in turn, and in a process with more streams than queues -- a handle behind a pipeline: nine -- the one that came
    //  and went moved two busy ones onto the same queue: `bench.py --pipeline` 177 -> 168 Gsample/s.)
    hipError_t e = hipSuccess;
    auto step = [&](hipError_t r) { if (e == hipSuccess) e = r; return e == hipSuccess; };
    if (step(hipMalloc((void **)&d_map, (size_t)total * 8 * sizeof(uint16_t))) &&
        step(hipMalloc((void **)&d_lo, (size_t)total * 8)) && step(hipMalloc((void **)&d_hi, (size_t)total * 8)) &&
        step(hipMalloc((void **)&d_lens, lens.size() * sizeof(uint32_t))) && step(hipMalloc((void **)&d_offs, offs.size() * sizeof(uint32_t))) &&
        step(hipMalloc((void **)&d_off, offv.size() * sizeof(uint32_t))) &&
        step(hipMemcpyAsync(d_lens, lens.data(), lens.size() * sizeof(uint32_t), hipMemcpyHostToDevice, st)) &&
        step(hipMemcpyAsync(d_offs, offs.data(), offs.size() * sizeof(uint32_t), hipMemcpyHostToDevice, st)) &&
        step(hipMemcpyAsync(d_off, offv.data(), offv.size() * sizeof(uint32_t), hipMemcpyHostToDevice, st)) &&
        step(ilmap_build_launch(d_lens, d_offs, (uint32_t)lens.size(), d_lo, d_hi, d_map, st)))
        step(hipStreamSynchronize(st));
    // scratch goes whatever happened; the tables stay only on success
    if (d_lo) (void)hipFree(d_lo);
    if (d_hi) (void)hipFree(d_hi);
    if (d_lens) (void)hipFree(d_lens);
    if (d_offs) (void)hipFree(d_offs);
    if (e != hipSuccess) {
        if (d_map) (void)hipFree(d_map);
        if (d_off) (void)hipFree(d_off);
        return fail(MCRX_EHIP, hipGetErrorString(e));
    }
    std::lock_guard<std::mutex> lk(g_il_mu);
    if (find_shared()) { (void)hipFree(d_map); (void)hipFree(d_off); return MCRX_OK; }       // another handle published the same table meanwhile
    g_il.push_back(IlShared{ dev, max_payload, cap, d_map, d_off, cap + 1, 1 });
    *map = d_map; *off = d_off; *n = cap + 1;
    return MCRX_OK;
}
static void il_tables_release(const uint16_t *map)
{
    if (!map) return;
    std::lock_guard<std::mutex> lk(g_il_mu);
    for (size_t i = 0; i < g_il.size(); i++)
        if (g_il[i].d_map == map) {
            if (--g_il[i].refs == 0) { (void)hipFree(g_il[i].d_map); (void)hipFree(g_il[i].d_off); g_il.erase(g_il.begin() + (long)i); }
            return;
        }
}

static int build_tables(mcrx_hip_t q)
{
    const OfdmDesign &od = q->od;
    SyncConsts &c = q->sc;
    c.M = (int)od.M; c.M2 = (int)od.M2; c.cp = (int)od.cp; c.L = (int)(od.M + od.cp); c.backoff = (int)od.backoff;
    c.M_pilot = (int)od.M_pilot; c.M_data = (int)od.M_data; c.Nen = (int)od.Nen; c.M_S0 = (int)od.M_S0; c.M_S1 = (int)od.M_S1;
    c.E = (int)pow2ceil((od.M + 63) / 64);
    c.log2M = 0;
    if ((od.M & (od.M - 1)) == 0) { unsigned l = 0; while ((1u << l) < od.M) l++; c.log2M = (int)l; }
    c.detect_thresh = od.detect_thresh; c.sync_thresh = od.sync_thresh;
    RC(q->upload(&c.sctype, od.p.data(), od.M));
    RC(q->upload(&c.S0, od.S0.data(), od.M));
    RC(q->upload(&c.S1, od.S1.data(), od.M));
    RC(q->upload(&c.s0t, reinterpret_cast<const float2 *>(od.s0.data()), od.M));
    RC(q->upload(&c.smk, od.smk.data(), od.smk.size()));
    RC(q->upload(&c.smn, od.smn.data(), od.smn.size()));
    { const double phi = (double)od.backoff * 6.283185307179586 / (double)od.M; c.backoff_rot = make_float2((float)cos(phi), (float)sin(phi)); }
    RC(q->upload(&c.Pfit, od.Pfit.data(), od.Pfit.size()));
    std::vector<int16_t> dr(od.M), pr(od.M), er(od.M);
    for (unsigned i = 0; i < od.M; i++) { dr[i] = (int16_t)od.data_rank[i]; pr[i] = (int16_t)od.pilot_rank[i]; er[i] = (int16_t)od.en_rank[i]; }
    RC(q->upload(&c.data_rank, dr.data(), od.M));
    RC(q->upload(&c.pilot_rank, pr.data(), od.M));
    RC(q->upload(&c.en_rank, er.data(), od.M));
    RC(q->upload(&c.pilot_seq, od.pilot_seq, 255));
    {   // Header bit map.  The header packet is always 36 bytes (14 + CRC-32 behind Golay(24,12),
        // ofdmflexframe header), so its 4-pass interleaver is one fixed permutation of 288 bits:
        // map[p] = position in the de-interleaved stream of received bit p (MSB first), bit 15 = the
        // scrambler's bit at p.  Found by pushing single bits through the transmit interleaver.
        std::vector<uint16_t> map(MCRX_HDR_SYMS, 0);
        static const uint8_t smask[4] = { 0xb4, 0x6a, 0x8b, 0xc5 };
        for (unsigned d = 0; d < MCRX_HDR_SYMS; d++) {
            std::vector<uint8_t> x(MCRX_HDR_ENC, 0);
            x[d >> 3] = (uint8_t)(0x80u >> (d & 7));
            interleave(x, 4);
            for (unsigned p = 0; p < MCRX_HDR_SYMS; p++)
                if ((x[p >> 3] >> (7 - (p & 7))) & 1) map[p] = (uint16_t)(d | (((smask[(p >> 3) & 3] >> (7 - (p & 7))) & 1) << 15));
        }
        RC(q->upload(&c.hdr_map, map.data(), map.size()));
    }
    std::vector<float2> tw(od.M);
    for (unsigned k = 0; k < od.M; k++) {
        double a = -2.0 * M_PI * (double)k / (double)od.M;
        tw[k] = make_float2((float)cos(a), (float)sin(a));
    }
    RC(q->upload(&c.dft_tw, tw.data(), od.M));
    RC(coding_tables(&c.cod));
    c.max_payload_len = q->max_payload; c.max_enc_len = q->max_enc; c.max_syms = q->max_syms;
    // CRC-32 is linear once the 0xFFFFFFFF preset is folded into the first four message bytes: crc = ~ XOR_i T[n-1-i][m_i],
    // T[0] = the byte table, T[d+1][b] = T[d][b] advanced through one zero byte.  With the table in HBM (1 KB per byte of
    // the longest payload; L2 resident) the decoder's check is one batch of independent loads per thread and an XOR
    // reduction instead of a chain of dependent operator look-ups down a tree.
    c.crc_pos = nullptr; c.crc_pos_n = 0;
    if (q->max_payload >= 4 && q->max_payload <= 8192) {
        CodingTables *t = new CodingTables();
        std::vector<uint32_t> pos((size_t)q->max_payload * 256);
        for (unsigned b = 0; b < 256; b++) pos[b] = t->crc_byte[b];
        for (size_t d = 1; d < q->max_payload; d++)
            for (unsigned b = 0; b < 256; b++) { const uint32_t v = pos[(d - 1) * 256 + b]; pos[d * 256 + b] = (v >> 8) ^ t->crc_byte[v & 0xff]; }
        delete t;
        RC(q->upload(&c.crc_pos, pos.data(), pos.size()));
        c.crc_pos_n = q->max_payload;
    }
    c.payload_soft = q->cfg.payload_soft ? 1 : 0;
    // De-interleaver gather tables (SyncConsts::il_map): shared per device, built when the handle first runs its synchronizers
    // (il_tables_acquire, launch_sync) -- a handle that is created and never fed pays nothing for them.
    c.il_off = nullptr; c.il_map = nullptr; c.il_n = 0;
    return MCRX_OK;
}

// every internal stream has finished what was enqueued so far before `st` goes on (no host wait)
static int join_into(mcrx_hip_t q, hipStream_t st)
{
    if (!q->pipelined) return MCRX_OK;
    hipStream_t in[4] = { q->stream, q->s_scout, q->s_work, q->s_side };
    for (int i = 0; i < 4; i++) {
        if (in[i] == st || !in[i]) continue;
        HIPCHK(hipEventRecord(q->ev_tmp[i], in[i]));
        HIPCHK(hipStreamWaitEvent(st, q->ev_tmp[i], 0));
    }
    return MCRX_OK;
}
// ... and the reverse: nothing enqueued on the internal streams from here on starts before `st` got here
static int fork_from(mcrx_hip_t q, hipStream_t st)
{
    if (!q->pipelined) return MCRX_OK;
    HIPCHK(hipEventRecord(q->ev_tmp[0], st));
    hipStream_t out[4] = { q->stream, q->s_scout, q->s_work, q->s_side };
    for (int i = 0; i < 4; i++) if (out[i] && out[i] != st) HIPCHK(hipStreamWaitEvent(out[i], q->ev_tmp[0], 0));
    return MCRX_OK;
}

// synchronizers back to SEEK, channelizer history cleared, undelivered device frames dropped
static int restart_async(mcrx_hip_t q, hipStream_t st, bool from_zero)
{
    if (from_zero) { q->total_samples = 0; q->chan_samples = 0; }
    q->stage_fill = 0; q->stage_first = q->total_samples;
    q->hist_cur = 0;
    q->last_slot = -1; q->last_ntiles = 0;
    q->pf_in_valid = 0; q->pf_steps = 0; q->pf_have_last = false;
    RC(join_into(q, st));
    // (both result generations: counters zeroed; the prediction lists only survive a Reset(), not a restart from zero)
    for (int g = 0; g < MCRX_GENS; g++) {
        HIPCHK(sync_reset_launch(q->d_st, q->nch, q->chan_samples, g == 0 ? q->d_hist[0] : nullptr, g == 0 ? q->d_hist[1] : nullptr,
                                 (size_t)q->hist_blocks * q->K, q->d_nrec[g], q->d_arena_used[g],
                                 (from_zero && g == 0) ? q->d_pred_n : nullptr, st));
        q->gen_used[g] = false; q->gen_closed[g] = false; q->gen_abandoned[g] = false;
    }
    if (q->d_seekst) HIPCHK(hipMemsetAsync(q->d_seekst, 0, (size_t)q->nch * 2 * sizeof(int64_t), st));     // (the synchronizers restart in SEEK: it is set before it is used)
    q->gen = 0;
    RC(fork_from(q, st));
    return MCRX_OK;
}

extern "C" int mcrx_hip_create(mcrx_hip_t *out, unsigned N, unsigned M, unsigned cp, unsigned taper,
                               const unsigned char *p, const mcrx_hip_config *cfg)
{
    if (!out) return fail(MCRX_EINVAL, "null output handle");
    *out = nullptr;
    // argument checks of multichannelrx::multichannelrx (lib/multichannelrx.cc:54-66)
    if (N < 1) return fail(MCRX_EINVAL, "error: multichannelrx, must have at least one channel");
    if (M < 8) return fail(MCRX_EINVAL, "error: multichannelrx, number of subcarriers must be at least 8");
    if (cp < 1) return fail(MCRX_EINVAL, "error: multichannelrx, cyclic prefix length must be at least 1");
    if (taper > cp) return fail(MCRX_EINVAL, "error: multichannelrx, taper length cannot exceed cyclic prefix length");
    const bool bypass = cfg && cfg->struct_size >= offsetof(mcrx_hip_config, single_channel) + sizeof(uint32_t) && cfg->single_channel;
    if (bypass && N != 1) return fail(MCRX_EINVAL, "single_channel needs num_channels == 1");
    if (!bypass && !channelizer_supported(2 * N)) return fail(MCRX_EUNSUPP, "at most 1024 channels");
    if (M > 1024) return fail(MCRX_EUNSUPP, "at most 1024 subcarriers");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(MCRX_EHIP, "no HIP device: the MI355X kernels are the only implementation (no CPU fallback)");

    mcrx_hip_t q = new mcrx_hip_s();
    q->device = current_device();
    q->N = N; q->K = bypass ? 1 : 2 * N; q->M = M; q->cp = cp; q->taper = taper; q->bypass = bypass;
    if (q->od.init(M, cp, taper, p) != 0) { delete q; return fail(MCRX_EINVAL, "invalid subcarrier allocation"); }
    if (cfg) memcpy(&q->cfg, cfg, std::min<size_t>(cfg->struct_size ? cfg->struct_size : sizeof(*cfg), sizeof(q->cfg)));
    else q->cfg.payload_soft = 1;
    q->max_payload = q->cfg.max_payload_len ? q->cfg.max_payload_len : 2048;
    q->max_enc = 4 * (q->max_payload + 4) + 16;
    // floor: the per-frame scratch rows (max_enc + 16 bytes) also hold the soft Viterbi decoder's checkpoints, 128 bytes
    // per 960 trellis steps = 128 * ceil((8 e0 + 6) / 960) <= 0.54 max_enc + 128 bytes for a frame that fits (e0 <= max_enc / 2
    // behind the rate-1/2 code): covered from max_enc = 256 on (smaller rows were overrun by 128 bytes)
    if (q->max_enc < 256) q->max_enc = 256;
    q->max_enc = (q->max_enc + 15) & ~15u;
    q->max_syms = 8 * q->max_enc;
    q->ch_first = q->cfg.channel_first;
    if (q->ch_first >= N || (uint64_t)q->ch_first + q->cfg.channel_count > N) { delete q; return fail(MCRX_EINVAL, "channel shard outside [0, N)"); }
    q->nch = q->cfg.channel_count ? q->cfg.channel_count : N - q->ch_first;
    q->max_rec = q->cfg.max_frames ? q->cfg.max_frames : 16 * q->nch + 64;
    q->min_frame = (uint64_t)(3 + (288 + q->od.M_data - 1) / q->od.M_data + 2) * (M + cp);
    if (!q->cfg.max_frames) {
        // ... and never fewer than one host batch (Execute() harvests per batch) can hold: the shortest frame is
        // S0a, S0b, S1, the header symbols, one payload symbol and the tail
        const uint64_t batch = q->cfg.batch_samples ? q->cfg.batch_samples : ((uint64_t)1 << 20);
        const uint64_t min_frame = q->min_frame;
        const uint64_t per_ch = (batch / q->K + 8) / min_frame + 2;
        q->max_rec = (uint32_t)std::max<uint64_t>(q->max_rec, std::min<uint64_t>(per_ch * q->nch, 1u << 22));
        // ... and enough for bulk Execute() calls to run in chunks of up to 16 Mi samples between harvests, within
        // about 2 GB of per-record buffers (arena share, soft bits, scratch, equaliser)
        const uint64_t per_rec = ((((uint64_t)q->max_payload + 15) & ~15ull) + 64ull * (2ull * (q->max_payload + 4) + 8)) +
                                 10ull * q->max_enc + 32 + 8ull * M;
        const uint64_t bulk = std::min<uint64_t>((((uint64_t)16 << 20) / q->K + 8) / min_frame + 3, ((uint64_t)2 << 30) / per_rec / q->nch);
        q->max_rec = (uint32_t)std::max<uint64_t>(q->max_rec, bulk * q->nch);
    }
    // frame arenas (per result generation): payload bytes | equalised symbols, the latter reserved for BPSK
    // behind one rate-1/2 code (longer frames still fit while the total stays below the cap; overflow is counted)
    q->arena_cap = (uint64_t)q->max_rec * (((uint64_t)q->max_payload + 15) & ~15ull);
    q->sarena_cap = (uint64_t)q->max_rec * (8ull * 8ull * (2ull * (q->max_payload + 4) + 8));
    if (q->sarena_cap > (8ull << 30)) q->sarena_cap = 8ull << 30;
    q->pipelined = !(q->cfg.struct_size >= offsetof(mcrx_hip_config, serial) + sizeof(uint32_t) && q->cfg.serial) && devel_env("MCRX_SERIAL") == nullptr;
    q->slab_blocks = q->cfg.slab_blocks ? ((q->cfg.slab_blocks + MCRX_TILE - 1) & ~(uint32_t)(MCRX_TILE - 1)) : 0;      // 0: sized per launch; slabs are whole tiles
    { int dev = 0, n = 0;
      if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0)
          q->ncu = (uint32_t)n; }
    const uint32_t front_end = (!bypass && q->cfg.struct_size >= offsetof(mcrx_hip_config, front_end) + sizeof(uint32_t)) ? q->cfg.front_end : 0;
    if (front_end > 2) { delete q; return fail(MCRX_EINVAL, "front_end must be 0, 1 or 2"); }
    q->oversampled = front_end == 2;
    if (front_end && ((q->K & (q->K - 1)) || q->K > 1024 || q->K < 2)) { delete q; return fail(MCRX_EUNSUPP, "the oversampled front end needs a power-of-two channel count <= 512"); }
    if (front_end == 1) { q->chan_P = 28; q->hist_blocks = 27; q->col_shift = q->K / 2 + 1; }
    q->taps = bypass ? std::vector<float>(14, 0.f) : pfb_prototype(q->K, 7, 60.0f);
    q->dtheta = bypass ? 0u : channel_center_step(N);
    // channel-rate history in front of every push's tiles: a symbol window, plus -- if frames straddling two pushes are
    // to be deferred rather than walked serially -- the longest frame start-to-push-end distance to be covered
    q->defer = (q->cfg.struct_size >= offsetof(mcrx_hip_config, defer_samples) + sizeof(uint32_t)) ? q->cfg.defer_samples : 0;
    q->hist_tiles = (unsigned)((q->defer + M + cp + 8 + MCRX_TILE - 1) / MCRX_TILE + 1);

    auto bail = [&](int rc) { mcrx_hip_destroy(q); return rc; };
    int rc; bool no_spec_cfg = false;
    {   // the channelizer's column tap table (kernels.h: ChanArgs::taps)
        std::vector<float> ct = q->chan_P == 28 ? pfb2_composite_taps(pfb2_prototype(q->K, 7, 60.0f), halfband_branch_taps(7, 60.0f), q->K)
                              : bypass ? q->taps : pfb_column_taps(q->taps, q->K);
        if ((rc = q->upload(&q->d_taps, ct.data(), ct.size()))) return bail(rc);
    }
    if ((rc = build_tables(q))) return bail(rc);
    if (q->oversampled) {
        if (mcrx_hip_pfb2_create(&q->pfb2, q->K, 7, 60.0f) != MCRX_OK) return bail(fail(MCRX_EHIP, mcrx_hip_pfb2_last_error()));
        std::vector<float> h1 = halfband_branch_taps(7, 60.0f);
        if ((rc = q->upload(&q->d_h1, h1.data(), h1.size()))) return bail(rc);
    }
    if ((rc = q->alloc(&q->d_st, q->nch))) return bail(rc);
    if ((rc = q->alloc(&q->d_seekst, (size_t)q->nch * 2))) return bail(rc);
    if ((rc = q->alloc(&q->d_hbits, (size_t)q->nch * MCRX_HDR_SYMS))) return bail(rc);
    if ((rc = q->alloc(&q->d_R, (size_t)q->nch * M))) return bail(rc);
    if ((rc = q->alloc(&q->d_soft, (size_t)q->nch * 8 * q->max_enc))) return bail(rc);
    if ((rc = q->alloc(&q->d_tmpa, (size_t)q->nch * (q->max_enc + 16)))) return bail(rc);
    if ((rc = q->alloc(&q->d_tmpb, (size_t)q->nch * (q->max_enc + 16)))) return bail(rc);
    if ((rc = q->alloc(&q->d_syms, (size_t)q->nch * q->max_syms))) return bail(rc);
    for (int g = 0; g < MCRX_GENS; g++) {
        if ((rc = q->alloc(&q->d_rec[g], q->max_rec))) return bail(rc);
        if ((rc = q->alloc(&q->d_arena[g], q->arena_cap))) return bail(rc);
        if ((rc = q->alloc(&q->d_sarena[g], q->sarena_cap))) return bail(rc);
        // (one 32-byte block per generation -- records, dropped | pad | payload bytes, symbol bytes -- so that a poll reads it with one
        //  small copy and clears it with one fill: each of those waits ~0.1 ms for a slot on the full chip)
        if ((rc = q->alloc(&q->d_nrec[g], 8))) return bail(rc);
        q->d_arena_used[g] = reinterpret_cast<unsigned long long *>(q->d_nrec[g] + 4);
        if (hipEventCreateWithFlags(&q->ev_gen[g], hipEventDisableTiming) != hipSuccess) return bail(fail(MCRX_EHIP, "hipEventCreate failed"));
        if (hipEventCreateWithFlags(&q->ev_clean[g], hipEventDisableTiming) != hipSuccess) return bail(fail(MCRX_EHIP, "hipEventCreate failed"));
    }
    q->scout = devel_env("MCRX_NO_SCOUT") == nullptr;
    q->debug = getenv("MCRX_DEBUG") ? atoi(getenv("MCRX_DEBUG")) : 0;
    q->no_fast = devel_env("MCRX_NO_FAST") ? atoi(devel_env("MCRX_NO_FAST")) : 0;
    q->free_run = devel_env("MCRX_FREE_RUN") != nullptr;
    {   // mcrx_hip_config::worker_build / acquisition / scout_build (fields past the caller's struct_size read as 0 = default)
        auto field = [&](size_t off) { return q->cfg.struct_size >= off + sizeof(uint32_t) ? *reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(&q->cfg) + off) : 0u; };
        const uint32_t wb = field(offsetof(mcrx_hip_config, worker_build)), aq = field(offsetof(mcrx_hip_config, acquisition));
        if (wb > 5 || aq > 5) return bail(fail(MCRX_EINVAL, "worker_build / acquisition out of range"));
        if (wb == 1) q->payload_xb = 0;
        if (wb >= 2) { q->payload_lean = 0; q->payload_fr = wb == 2 ? 1 : wb == 3 ? 2 : wb == 4 ? 4 : 0; }
        if ((aq >= 1 && aq <= 3) || aq == 5) q->acq_mode = (int)aq;
        no_spec_cfg = aq == 4;
        const uint32_t sb = field(offsetof(mcrx_hip_config, scout_build));
        if (sb > 2) return bail(fail(MCRX_EINVAL, "scout_build out of range"));
        if (sb == 1) q->lean_build = 0;
        if (sb == 2) q->seg_walker = 0;
    }
    if (devel_env("MCRX_PAYLOAD_FR")) q->payload_fr = atoi(devel_env("MCRX_PAYLOAD_FR"));
    if (devel_env("MCRX_PAYLOAD_LEAN")) q->payload_lean = atoi(devel_env("MCRX_PAYLOAD_LEAN")) != 0;
    if (devel_env("MCRX_PAYLOAD_XB")) q->payload_xb = atoi(devel_env("MCRX_PAYLOAD_XB"));
    if (devel_env("MCRX_SEEK_BURST")) q->seek_burst = atoi(devel_env("MCRX_SEEK_BURST"));
    if (devel_env("MCRX_ACQ_MODE")) q->acq_mode = atoi(devel_env("MCRX_ACQ_MODE"));
    if (devel_env("MCRX_WALK_LDS_PAD")) q->walk_lds_pad = (uint32_t)atoi(devel_env("MCRX_WALK_LDS_PAD"));
    if (devel_env("MCRX_PAYLOAD_LDS_PAD")) q->round_lds_pad = (uint32_t)atoi(devel_env("MCRX_PAYLOAD_LDS_PAD"));
    if (devel_env("MCRX_EVT")) q->ev_on = true;
    if (devel_env("MCRX_EVT_DUMP")) { q->ev_on = true; q->ev_dump = true; }
    if (devel_env("MCRX_SLOTS")) q->nslots = (unsigned)std::max(2, std::min(MCRX_SLOTS, atoi(devel_env("MCRX_SLOTS"))));
    if (!q->pipelined) q->nslots = 2;
    if ((rc = q->alloc(&q->d_njobs, MCRX_SLOTS))) return bail(rc);      // one counter per slot: a launch's placement kernel zeroes the next slot's
    if ((rc = q->alloc(&q->d_stats, 8))) return bail(rc);
    // (coherent: with hipHostMallocMapped alone the allocation is non-coherent and a free-running host -- one that never
    //  synchronizes with the device -- does not see the kernels' updates at all)
    if (hipHostMalloc((void **)&q->h_hint, 12 * sizeof(uint32_t), hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess) {
        for (int i = 0; i < 12; i++) q->h_hint[i] = 0;    // [0] longest coded frame, [1] widest prediction list, [2] walked, [3] adopted, [4] frames on a cadence, [5] frames seen, [11] a frame with the K = 7 code was decoded,
                                                          // [8] QAM hand-offs, [9] trellis blocks, [10] general-list frames of the most recent launch (kernels.h: list_hint)
        if (hipHostGetDevicePointer((void **)&q->d_hint, q->h_hint, 0) != hipSuccess) q->d_hint = nullptr;
    }
    if (q->scout) {
        q->max_jobs = 2 * q->max_rec + 4 * q->nch + 1024;
        for (unsigned sl = 0; sl < q->nslots; sl++) {
            if ((rc = q->alloc(&q->d_jobs[sl], q->max_jobs))) return bail(rc);
            if ((rc = q->alloc(&q->d_gen[sl], (size_t)q->max_jobs + 1))) return bail(rc);
            if ((rc = q->alloc(&q->d_qam[sl], (size_t)q->max_jobs + 1))) return bail(rc);
            if ((rc = q->alloc(&q->d_live[sl], (size_t)q->max_jobs + 1))) return bail(rc);
            if ((rc = q->alloc(&q->d_jR[sl], (size_t)q->max_jobs * M))) return bail(rc);
            if ((rc = q->alloc(&q->d_jsoft[sl], (size_t)q->max_jobs * 8 * q->max_enc))) return bail(rc);
            if ((rc = q->alloc(&q->d_jtmp[sl], (size_t)q->max_jobs * 2 * (q->max_enc + 16)))) return bail(rc);
        }
        if (q->sc.payload_soft) {
            // The K = 7 decoder (decode_general_kernel, viterbi_frames.hpp) takes a frame per wave and keeps 512 bytes of decisions per trellis
            // step of a block in HBM: one region per workgroup, as many workgroups as a launch can have frames (at most 2 per SIMD: it is
            // arithmetic).  Its launches follow each other on the work stream, so one set serves every slot.  ~200 KB per workgroup at 1200-byte
            // payloads = 0.4 GB: allocated when the first frame with the code has been seen (cfg.conv_scratch), launch_sync.
            q->vit_rows = vf::rows_for(4u * q->max_enc + 6u);
            q->vit_waves = (uint32_t)std::max<uint64_t>(64, std::min<uint64_t>(q->max_rec, 2048));
            q->vit_mode = q->cfg.struct_size >= offsetof(mcrx_hip_config, conv_scratch) + sizeof(uint32_t) ? q->cfg.conv_scratch : 0u;
            if (q->vit_mode > 2) return bail(fail(MCRX_EINVAL, "conv_scratch must be 0, 1 or 2"));
            if ((rc = q->alloc(&q->d_vit_passes, 8))) return bail(rc);
            if (q->vit_mode == 1 && (rc = q->alloc_raw(&q->d_vit_scratch, (size_t)q->vit_waves * q->vit_rows * 64))) return bail(rc);
        }
        // speculative acquisition (lean path only): slots, their equalisers, the prediction lists
        const bool lean = (q->sc.log2M >= 6 && q->sc.M == 64 * q->sc.E && q->sc.M_pilot <= 64) ||
                          (q->sc.M == 48 && q->sc.E == 1 && q->sc.M_pilot <= 64);        // (48 = 3 x 16: the reference applications' default, round 5)
        q->spec = lean && q->d_hint && !no_spec_cfg && devel_env("MCRX_NO_SPEC") == nullptr;
        if (devel_env("MCRX_NSEG")) q->nseg_fixed = (uint32_t)std::max(1, std::min(MCRX_SEG_MAX, atoi(devel_env("MCRX_NSEG"))));           // experiments: segments per channel, fixed
        if (devel_env("MCRX_SEG_FRAMES")) q->seg_frames = (uint32_t)std::max(1, std::min(64, atoi(devel_env("MCRX_SEG_FRAMES"))));            // ... or frames per segment aimed at
        if (devel_env("MCRX_LEAN_BUILD")) q->lean_build = atoi(devel_env("MCRX_LEAN_BUILD"));
        if (q->spec) {
            if ((rc = q->alloc(&q->d_spec, (size_t)q->nch * MCRX_SPEC_MAX))) return bail(rc);
            if ((rc = q->alloc(&q->d_spec_R, (size_t)q->nch * MCRX_SEG_MAX * M))) return bail(rc);
            if ((rc = q->alloc(&q->d_anchor, q->nch))) return bail(rc);
        }
    }
    if ((rc = q->alloc(&q->d_hist[0], (size_t)q->hist_blocks * q->K))) return bail(rc);
    if ((rc = q->alloc(&q->d_hist[1], (size_t)q->hist_blocks * q->K))) return bail(rc);
    // host staging for Execute(): whole tiles of MCRX_TILE blocks
    size_t tile_samples = (size_t)MCRX_TILE * q->K;
    size_t want = q->cfg.batch_samples ? q->cfg.batch_samples : ((size_t)1 << 20);
    q->stage_cap = std::max<size_t>(1, (want + tile_samples - 1) / tile_samples) * tile_samples;
    if (hipHostMalloc((void **)&q->h_stage, q->stage_cap * sizeof(float2), hipHostMallocDefault) != hipSuccess)
        return bail(fail(MCRX_ENOMEM, "pinned staging allocation failed"));
    if ((rc = q->alloc(&q->d_in, q->stage_cap))) return bail(rc);
    // (blocking streams: work a caller puts on the legacy default stream -- e.g. a torch copy of a result buffer --
    //  still orders against them, as it did when everything ran on one stream)
    if (hipStreamCreate(&q->stream) != hipSuccess) return bail(fail(MCRX_EHIP, "hipStreamCreate failed"));
    {   // the acquisition stream carries the serial per-channel chains every payload launch waits for: highest priority
        int least = 0, greatest = 0;
        const bool prio = devel_env("MCRX_NO_PRIO") == nullptr && hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && greatest != least;
        if ((prio ? hipStreamCreateWithPriority(&q->s_scout, hipStreamDefault, greatest) : hipStreamCreate(&q->s_scout)) != hipSuccess)
            return bail(fail(MCRX_EHIP, "hipStreamCreate failed"));
    }
    if (hipStreamCreate(&q->s_work) != hipSuccess) return bail(fail(MCRX_EHIP, "hipStreamCreate failed"));
    // (With the others: a stream created in mid-stream shifts the queue mapping.  Only for receivers of few channels: their pushes are
    //  chains of launch gaps, and three launches less in the work stream's chain are worth 15-18 % (8 channels: 65 -> 77 Gsample/s, with
    //  the K = 7 code 41 -> 47); a receiver that fills the chip gains nothing from it, and the stream's mere existence cost 1 % of the
    //  headline and 10 % of the pipeline leg in alternating runs -- scratch/r5/split_ab.sh.)
    if (q->pipelined && q->nch <= 32 && q->cfg.channel_count == 0 && hipStreamCreate(&q->s_side) != hipSuccess)      // (not a rank's shard behind a pipeline: streams enough there)
        return bail(fail(MCRX_EHIP, "hipStreamCreate failed"));
    if (hipStreamCreateWithFlags(&q->s_copy, hipStreamNonBlocking) != hipSuccess) return bail(fail(MCRX_EHIP, "hipStreamCreate failed"));
    {
        hipEvent_t *evs[] = { &q->ev_in, &q->ev_consumed, &q->ev_tmp[0], &q->ev_tmp[1], &q->ev_tmp[2], &q->ev_tmp[3], &q->ev_side_last };
        for (hipEvent_t *e : evs) if (hipEventCreateWithFlags(e, hipEventDisableTiming) != hipSuccess) return bail(fail(MCRX_EHIP, "hipEventCreate failed"));
        for (int sl = 0; sl < MCRX_SLOTS; sl++) {
            hipEvent_t *ev3[] = { &q->ev_ready[sl], &q->ev_scout[sl], &q->ev_done[sl], &q->ev_side[sl] };
            for (hipEvent_t *e : ev3) if (hipEventCreateWithFlags(e, hipEventDisableTiming) != hipSuccess) return bail(fail(MCRX_EHIP, "hipEventCreate failed"));
        }
    }
    for (int w = 0; w < MCRX_NKERNELS; w++) {
        q->evring[w].resize(2048, nullptr);
        for (auto &e : q->evring[w]) if (hipEventCreate(&e) != hipSuccess) return bail(fail(MCRX_EHIP, "hipEventCreate failed"));
    }
    if ((rc = restart_async(q, q->stream, true))) return bail(rc);
    if (hipStreamSynchronize(q->stream) != hipSuccess) return bail(fail(MCRX_EHIP, "stream sync failed"));
    *out = q;
    return MCRX_OK;
}

extern "C" int mcrx_hip_destroy(mcrx_hip_t q)
{
    DevScope dev_scope_(q ? q->device : -1);
    if (!q) return MCRX_OK;
    (void)hipDeviceSynchronize();
    if (q->debug & 8)
        fprintf(stderr, "mcrx bulk path: copy %.4f s, launch %.4f s, waiting for the GPU %.4f s, harvest (incl. that wait) %.4f s of which frame D2H %.4f s for %.1f MB, host arena growth %.4f s, counters D2H %.4f s, filter + sort %.4f s\n", q->t_copy, q->t_run, q->t_wait, q->t_harvest, q->t_d2h, q->b_d2h / 1e6, q->t_grow, q->t_cnt, q->t_post);
    if ((q->debug & 32) && q->d_stats) {
        uint32_t v[4] = {};
        (void)hipMemcpy(v, q->d_stats, sizeof(v), hipMemcpyDeviceToHost);
        fprintf(stderr, "mcrx stats: walked %u adopted %u last failed crc: key %08x computed %08x\n", v[0], v[1], v[2], v[3]);
    }
    for (void *p : q->owned) (void)hipFree(p);
    il_tables_release(q->sc.il_map);
    for (int i = 0; i < MCRX_SLOTS; i++) if (q->d_chan[i]) (void)hipFree(q->d_chan[i]);
    for (int i = 0; i < 2; i++) { if (q->d_pfin[i]) (void)hipFree(q->d_pfin[i]); if (q->d_pfout[i]) (void)hipFree(q->d_pfout[i]); }
    if (q->pfb2) (void)mcrx_hip_pfb2_destroy(q->pfb2);
    if (q->h_stage) (void)hipHostFree(q->h_stage);
    q->arena_host.release(); q->sarena_host.release();
    {
        hipEvent_t evs[] = { q->ev_in, q->ev_consumed, q->ev_tmp[0], q->ev_tmp[1], q->ev_tmp[2], q->ev_tmp[3], q->ev_side_last, q->ev_base };
        for (hipEvent_t e : evs) if (e) (void)hipEventDestroy(e);
        for (int g = 0; g < MCRX_GENS; g++) { if (q->ev_gen[g]) (void)hipEventDestroy(q->ev_gen[g]); if (q->ev_clean[g]) (void)hipEventDestroy(q->ev_clean[g]); }
        for (int sl = 0; sl < MCRX_SLOTS; sl++) {
            if (q->ev_ready[sl]) (void)hipEventDestroy(q->ev_ready[sl]);
            if (q->ev_scout[sl]) (void)hipEventDestroy(q->ev_scout[sl]);
            if (q->ev_done[sl]) (void)hipEventDestroy(q->ev_done[sl]);
            if (q->ev_side[sl]) (void)hipEventDestroy(q->ev_side[sl]);
        }
        hipStream_t sts[] = { q->s_scout, q->s_work, q->s_copy, q->s_side };
        for (hipStream_t t : sts) if (t) (void)hipStreamDestroy(t);
    }
    for (int i = 0; i < 2; i++) {
        if (q->d_direct[i]) (void)hipFree(q->d_direct[i]);
        if (q->ev_dcopy[i]) (void)hipEventDestroy(q->ev_dcopy[i]);
        if (q->ev_ddone[i]) (void)hipEventDestroy(q->ev_ddone[i]);
    }
    if (q->copy_stream) (void)hipStreamDestroy(q->copy_stream);
    if (q->h_hint) (void)hipHostFree(q->h_hint);
    for (int w = 0; w < MCRX_NKERNELS; w++) for (auto e : q->evring[w]) if (e) (void)hipEventDestroy(e);
    if (q->stream) (void)hipStreamDestroy(q->stream);
    delete q;
    return MCRX_OK;
}

extern "C" unsigned mcrx_hip_num_channels(mcrx_hip_t q) { return q ? q->N : 0; }
extern "C" uint32_t mcrx_hip_nco_step(mcrx_hip_t q) { return q ? q->dtheta : 0; }
extern "C" unsigned mcrx_hip_history_blocks(mcrx_hip_t q) { return q ? q->hist_blocks : 0; }
extern "C" int mcrx_hip_get_taps(mcrx_hip_t q, float *h, size_t n)
{
    if (!q || !h || n < q->taps.size()) return fail(MCRX_EINVAL, "taps buffer too small");
    memcpy(h, q->taps.data(), q->taps.size() * sizeof(float));
    return MCRX_OK;
}

// ---------------------------------------------------------------- stage level
static int launch_channelizer(mcrx_hip_t q, const float2 *x, size_t nblocks, uint64_t first_sample,
                              const float2 *halo, float2 *out, unsigned groups, size_t ntiles_stride, hipStream_t st)
{
    if (nblocks == 0) return MCRX_OK;
    if (nblocks % MCRX_TILE) return fail(MCRX_EINVAL, "nblocks must be a multiple of MCRX_TILE (16)");
    if (groups == 0 || q->N % groups) return fail(MCRX_EINVAL, "groups must divide the channel count");
    ChanArgs a;
    a.x = x; a.halo = halo; a.taps = q->d_taps; a.out = out;
    a.nblocks = (uint32_t)nblocks;
    a.slab_blocks = q->slab_blocks ? q->slab_blocks : channelizer_auto_slab(q->K, nblocks, q->ncu);
    a.first_sample_lo = (uint32_t)first_sample; a.dtheta = q->dtheta;
    a.ntiles = (uint32_t)ntiles_stride; a.cg = q->N / groups; a.col_shift = q->col_shift;
    RC(q->ev_begin(0, st));
    HIPCHK(channelizer_launch(q->K, q->chan_P, a, st));
    RC(q->ev_end(0, st));
    return MCRX_OK;
}
// One pass of the synchronizer bank over a channel-tile buffer that is ready in `st`'s order.
// Pipelined: the acquisition kernels go to s_scout, the payload workers and the decoder to s_work; `st` itself
// does not wait for them (mcrx_hip_stream_wait / flush / poll do).  Slot reuse: launch k waits for launch k-2's
// decode, which frees the job counter its placement kernel zeroes and (in order on s_work) its own slot.
static int launch_sync(mcrx_hip_t q, const float2 *chan, unsigned stride, unsigned off, int64_t buf_first, int64_t end, hipStream_t st)
{
    const unsigned slot = (unsigned)(q->seq % q->nslots), next = (unsigned)((q->seq + 1) % q->nslots);
    const int g = q->gen;
    if (!q->il_tried) {                    // first synchronizer launch of this handle: the device's shared de-interleaver tables
        q->il_tried = true;
        // (a table that cannot be built -- 40-120 MB plus twice that in scratch -- is no table, not an error: the decoder's in-place passes
        //  serve every length)
        if (q->scout_tables && devel_env("MCRX_NO_ILMAP") == nullptr &&
            il_tables_acquire(q->max_payload, q->max_enc, &q->sc.il_map, &q->sc.il_off, &q->sc.il_n, q->stream) != MCRX_OK) {
            (void)hipGetLastError();
            q->sc.il_map = nullptr; q->sc.il_off = nullptr; q->sc.il_n = 0;
            if (q->debug) fprintf(stderr, "[mcrx] de-interleaver gather tables not built (%s): decoding without them\n", mcrx_hip_last_error());
        }
    }
    SyncArgs a;
    a.c = q->sc; a.chan = chan; a.chan_stride = stride; a.chan_off = off;
    a.buf_first = buf_first; a.end = end; a.nch = q->nch; a.ch_first = q->ch_first;
    a.st = q->d_st; a.hbits = q->d_hbits; a.R = q->d_R; a.soft = q->d_soft; a.tmpa = q->d_tmpa; a.tmpb = q->d_tmpb;
    a.syms = q->d_syms; a.rec = q->d_rec[g]; a.arena = q->d_arena[g]; a.sarena = q->d_sarena[g];
    a.nrec = q->d_nrec[g]; a.arena_used = q->d_arena_used[g];
    a.arena_cap = q->arena_cap; a.sarena_cap = q->sarena_cap; a.max_rec = q->max_rec;
    a.debug = q->debug; a.no_fast = q->no_fast; a.seek_burst = q->seek_burst;
    a.payload_fr = q->payload_fr; a.payload_lean = q->payload_lean; a.payload_xb = q->payload_xb;
    a.payload_lds_pad = q->acq_mode == 2 ? q->walk_lds_pad : q->round_lds_pad;
    a.vit_off = 0;
    a.scout = q->scout ? 1 : 0;
    a.jobs = q->d_jobs[slot]; a.njobs = q->d_njobs + slot; a.njobs_next = q->d_njobs + next; a.max_jobs = q->max_jobs;
    a.gen_list = q->d_gen[slot]; a.dec_lds_soft = 0;
    if (!q->d_vit_scratch && q->vit_mode == 0 && q->vit_rows && q->h_hint && ((volatile uint32_t *)q->h_hint)[11]) {
        // a frame with the K = 7 code has been decoded: its decoder's scratch from here on (a failure is not an error: the block decoder stays)
        const size_t vit_bytes = (size_t)q->vit_waves * q->vit_rows * 64 * sizeof(uint2);       // [waves][rows][64 lanes] of 8 bytes
        if (hipMalloc((void **)&q->d_vit_scratch, vit_bytes) != hipSuccess) {
            (void)hipGetLastError(); q->d_vit_scratch = nullptr; q->vit_mode = 2;
            if (q->debug) fprintf(stderr, "[mcrx] the K = 7 decoder's scratch (%zu MB) could not be allocated: the block decoder stays\n", vit_bytes >> 20);
        } else q->owned.push_back(q->d_vit_scratch);
    }
    a.vit_scratch = q->d_vit_scratch; a.vit_rows = q->vit_rows; a.vit_waves = q->vit_waves; a.vit_passes = q->d_vit_passes;
    a.qam_list = q->d_qam[slot]; a.qam_next = q->d_qam[next]; a.list_hint = nullptr;
    a.live = q->d_live[slot]; a.live_next = q->d_live[next];
    a.live_off = 0; a.split_rest = 0; a.dec_phase = 0;
    a.frames_hint = (q->h_hint && q->d_hint) ? ((volatile uint32_t *)q->h_hint)[7] : ~0u;
    if (a.frames_hint == 0) a.frames_hint = ~0u;                 // (a launch without frames says nothing about the next)
    for (int i = 0; i < 3; i++) a.grid_hint[i] = ~0u;
    if (q->h_hint && q->d_hint) {                        // (sticky for a while: a list that was non-empty within the last 64 launches keeps its full grid)
        a.list_hint = q->d_hint + 8;
        for (int i = 0; i < 3; i++) {
            const uint32_t v = ((volatile uint32_t *)q->h_hint)[8 + i];
            if (v) { q->list_seen[i] = v; q->list_age[i] = 0; } else if (q->list_age[i] < 64) q->list_age[i]++; else q->list_seen[i] = 0;
            a.grid_hint[i] = q->list_seen[i];
        }
    }
    a.jR = q->d_jR[slot]; a.jsoft = q->d_jsoft[slot]; a.jtmp = q->d_jtmp[slot];
    a.stats = q->d_stats;
    a.hint = q->d_hint; a.enc_hint = (q->h_hint && q->d_hint) ? *(volatile uint32_t *)q->h_hint : 0u;
    a.spec = q->d_spec; a.spec_stride = q->spec_stride; a.spec_R = q->d_spec_R; a.pred = nullptr; a.pred_n = nullptr; a.spec_cap = 0; a.spec_hint = nullptr; a.walk_hint = nullptr;
    a.nseg = 0; a.seg_phase = 0; a.anchor = nullptr; a.seg_jobs = 1; a.seekst = q->d_seekst; a.seg_walker = q->seg_walker;
    a.no_syms = (q->cfg.struct_size >= offsetof(mcrx_hip_config, skip_framesyms) + sizeof(uint32_t) && q->cfg.skip_framesyms == 2) ? 1 : 0;
    hipStream_t sa = st, sw = st;
    // The host never waits for the device on this path (slots are handed over by stream waits), so a free-running caller
    // can be any number of launches ahead -- and the hints the kernels leave for the next launch (widest prediction list,
    // scouts that had to walk) would arrive after everything is enqueued.  Bound the lead to the slots: wait here for the
    // launch that last used this one (the device still has nslots - 1 launches queued behind it) -- once per turn of the
    // slots, i.e. a lead of nslots .. 2 nslots - 1 launches: a wait per launch costs a short-slab stream (8 channels,
    // 0.36 ms per push) 4 %.
    if (q->pipelined && q->scout && q->spec && q->seq >= q->nslots && slot == 0 && !q->free_run) HIPCHK(hipEventSynchronize(q->ev_done[slot]));
    if (q->pipelined && q->scout) {
        sa = q->s_scout; sw = q->s_work;
        HIPCHK(hipEventRecord(q->ev_ready[slot], st));
        HIPCHK(hipStreamWaitEvent(sa, q->ev_ready[slot], 0));
        if (q->seq + 1 >= q->nslots) HIPCHK(hipStreamWaitEvent(sa, q->ev_done[next], 0));     // launch seq + 1 - nslots
    }
    if (!q->scout) HIPCHK(hipMemsetAsync(a.njobs, 0, sizeof(uint32_t), sa));     // (with the scout, the previous launch's placement kernel zeroed it)
    q->acq_stream = sa;
    if (q->clean_pending[g]) { HIPCHK(hipStreamWaitEvent(sa, q->ev_clean[g], 0)); q->clean_pending[g] = false; }
    RC(q->ev_begin(1, sa));
    a.stop_after_walk = 0; a.tail_only = 0; a.defer_limit = (int64_t)q->defer; a.burst_limit = 0; a.round_idx = 0;
    if (q->spec) {
        // lean configurations: a payload that straddles two pushes is walked by the tail kernel (before the acquisition: the frame
        // the previous push left in progress; after it: the one this push ends in); everything else is acquired by the segment
        // waves (kernels.h, SpecSlot) and strung together by the per-channel scouts, which walk by themselves only where no
        // segment wave stood in their exact state.
        SyncArgs t = a;
        t.tail_only = 1; t.pred = nullptr; t.pred_n = nullptr; t.spec_cap = 0; t.nseg = 0; t.spec_hint = nullptr; t.stats = nullptr;
        // (Symbols wider than two samples per lane have no lean scout: theirs is the whole state machine, which carries a payload in progress
        //  across pushes by itself, and the two tail launches never find anything to do there.  Leaving them out was measured -- round 6, alternating
        //  runs of a development build, configs[2]: 105.2-106.8 without against 106.7-109.4 Gsample/s with them -- and is not done.)
        HIPCHK(sync_launch_tail(t, sa));
        // How many segments: enough waves to fill the chip and short chains (a wave's frames are acquired one after the other),
        // but every segment costs two acquisitions that produce nothing (its first frame, taken from an arbitrary state, and the
        // frame that links it to the next segment), and its share of the channel's MCRX_SPEC_MAX slots must hold its frames.
        // F = frames per channel and push, from the hand-offs the most recent finished launch counted (host-mapped, read without
        // a sync, a launch or two late); before the first report: one segment per 16 Ki samples.  Only speed depends on it.
        const uint64_t nsamp = (uint64_t)(end - buf_first) > (uint64_t)q->hist_tiles * MCRX_TILE ? (uint64_t)(end - buf_first) - (uint64_t)q->hist_tiles * MCRX_TILE : 1;
        uint32_t nseg = q->nseg_fixed;
        if (!nseg) {
            const uint32_t nj = q->h_hint ? ((volatile uint32_t *)q->h_hint)[7] : 0u;
            if (nj) q->frames_per_push = 0.5f * q->frames_per_push + 0.5f * ((float)nj / (float)q->nch) * ((float)nsamp / (float)(q->last_nsamp ? q->last_nsamp : nsamp));
            const float F = q->frames_per_push > 0.f ? q->frames_per_push : (float)nsamp / 16384.0f;
            float want = F / (float)q->seg_frames;                                   // chains of seg_frames (+ 2) frames ...
            const float fill = 1536.0f / (float)q->nch;                              // ... shorter while the chip is not full (1.5 waves per SIMD)
            if (want < fill) want = fill < F / 2.0f ? fill : F / 2.0f;
            // on a cadence the lattice starts make segments free (no wasted acquisitions), so chains go down to two frames while
            // all the waves still run at once (2 per SIMD): what a few-channel receiver's push is made of is this chain's latency
            // (the lean segment waves of the 64-subcarrier designs run acq_lean_waves() to a SIMD, and a frame there is a chain of ~15 us:
            //  chains of ONE frame while every wave is resident at once)
            if (q->cadenced) {
                const int lw = q->seg_walker ? 0 : acq_lean_waves(q->sc);
                const float conc = (lw ? 1024.0f * (float)lw : 2048.0f) / (float)q->nch, shortest = lw ? F : F / 2.0f;
                const float c2 = conc < shortest ? conc : shortest;
                if (want < c2) want = c2;
            }
            // (a wave's slots: F / nseg * 1.25 + 4, at most half a window of the scouts' slot headers -- see spw below)
            const float least = 1.25f * F / (float)(MCRX_SPEC_MAX / 2 - 4);
            if (want < least) want = least;
            const float most = q->nch * 128u <= 4096u ? 128.0f : 64.0f;               // (few channels: even 128 waves each leave the chip mostly empty)
            nseg = want < 1.0f ? 1u : (want > most ? (uint32_t)most : (uint32_t)(want + 0.999f));
        }
        if (nseg > MCRX_SEG_MAX) nseg = MCRX_SEG_MAX;
        // The anchor phase (kernels.h, SyncArgs::seg_phase) is one more launch and one frame's latency in front of everything else: worth it
        // while most frames follow their predecessor at the distance of the pair before (the scouts count both, place_jobs_kernel
        // copies the totals to host-mapped words), useless on traffic without a cadence.  Windows of 8 launches; MCRX_ACQ_MODE=3 / 1 pins it.
        if (q->h_hint && ++q->cad_count >= 8) {
            volatile uint32_t *h = (volatile uint32_t *)q->h_hint;
            const uint32_t sm = h[4], fr = h[5];
            const uint32_t ds = sm >= q->cad_same ? sm - q->cad_same : 0u, df = fr >= q->cad_frames ? fr - q->cad_frames : 0u;     // (mcrx_hip_spec_stats may have reset them)
            if (df > q->nch) q->cadenced = 2ull * ds > df;
            q->cad_same = sm; q->cad_frames = fr; q->cad_count = 0;
            // ... and which anchor.  The lattice carried over from the previous push needs no launch in front of the segment waves: as it
            // stood in the stream (a continuous stream), or as it stood from the push's beginning (pushes that are bursts of their own, a
            // replayed slab with a gap at its end).  A wrong one shows: every segment hands off two frames nobody adopts.  So: one
            // after the other while more than an eighth of the slots filled go to waste, the anchor phase (which finds the lattice inside
            // every push) when neither holds, and from the start again after 256 launches.
            const uint32_t sf = h[6], ad = h[3];
            const uint32_t dsf = sf >= q->anc_filled ? sf - q->anc_filled : 0u, dad = ad >= q->anc_adopted ? ad - q->anc_adopted : 0u;
            if (q->anchor_kind < 2 && dad > q->nch && dsf > dad + dad / 8) { if (++q->anchor_kind == 2) q->anchor_retry = 256; }
            q->anc_filled = sf; q->anc_adopted = ad;
        }
        if (q->anchor_kind == 2 && q->anchor_retry && --q->anchor_retry == 0) q->anchor_kind = 0;
        if (q->acq_mode == 3 || q->acq_mode == 5) q->cadenced = true;
        q->last_nsamp = nsamp;
        // Slots per wave: its share of MCRX_SPEC_MAX while the push's frames fit there (the scouts then hold every slot header in
        // registers), else what its frames need -- the channel's slots then exceed the window and the scouts move it along
        // (ofdmsync.hip: adopt_lookup), which takes two neighbouring waves' slots to fit in one window.
        uint32_t spw = MCRX_SPEC_MAX / nseg;
        { const float Fs = q->frames_per_push > 0.f ? q->frames_per_push : (float)nsamp / 16384.0f;
          const uint32_t need = (uint32_t)(1.25f * Fs / (float)nseg) + 4u;
          if (need > spw) spw = need > MCRX_SPEC_MAX / 2 ? MCRX_SPEC_MAX / 2 : need; }
        if ((uint64_t)nseg * spw > q->spec_stride) {
            // (rare: the first long push of a few-channel stream.  Launches in flight use the old slots: wait for them.)
            uint32_t ns = q->spec_stride; while (ns < nseg * spw) ns *= 2;
            SpecSlot *nb = nullptr;
            HIPCHK(hipDeviceSynchronize());
            RC(q->alloc(&nb, (size_t)q->nch * ns));
            q->d_spec = nb; q->spec_stride = ns;            // (the old array stays with the handle until it is destroyed)
            a.spec = q->d_spec; a.spec_stride = q->spec_stride;
        }
        a.nseg = nseg; a.spec_cap = nseg * spw;
        { const float Fq = q->frames_per_push > 0.f ? q->frames_per_push : (float)nsamp / 16384.0f;
          const float per = Fq / (float)nseg + 2.0f;                     // its share of the channel's frames + the two at the segment's ends
          a.seg_jobs = per < 2.0f ? 2u : (per > 32.0f ? 32u : (uint32_t)(per + 0.999f));
          // ... and all the first blocks together leave the scouts' own hand-offs and second blocks half of the list
          const uint32_t room = q->max_jobs / 2 / (q->nch * nseg);
          if (a.seg_jobs > room) a.seg_jobs = room ? room : 1u; }
        a.walk_hint = q->d_hint ? q->d_hint + 2 : nullptr;
        if (a.debug & 4) fprintf(stderr, "[host] launch %llu: %u segments per channel, %.1f frames per channel and push expected\n", (unsigned long long)q->seq, nseg, q->frames_per_push);
        a.anchor = q->d_anchor;
        if (q->acq_mode == 2) a.spec_cap = 0;                        // (MCRX_ACQ_MODE=2: no segment waves, the scouts walk everything)
        else if (q->acq_mode == 1 || nseg == 1 || (q->acq_mode == 0 && !q->cadenced)) { a.seg_phase = 0; HIPCHK(sync_launch_spec(a, sa)); }     // one launch, coarse starts
        else if (q->acq_mode != 3 && q->anchor_kind < 2) { a.seg_phase = 3 + q->anchor_kind; HIPCHK(sync_launch_spec(a, sa)); }       // one launch, anchored on the entry state (a channel that does not stand behind a frame: coarse starts)
        else {
            const uint32_t sj = a.seg_jobs;
            a.seg_phase = 1; a.seg_jobs = 1; HIPCHK(sync_launch_spec(a, sa));        // the first frame of every channel, from its real state: the cadence's anchor
            a.seg_phase = 2; a.seg_jobs = sj; HIPCHK(sync_launch_spec(a, sa));       // everything behind it, segment-parallel
        }
        a.seg_phase = 0;
        if (q->lean_build == 1) HIPCHK(sync_launch_walk(a, sa)); else HIPCHK(sync_launch_lean(a, sa));
        // a frame the lean scout could neither hand off nor defer runs past the end of this buffer: walked up to there
        HIPCHK(sync_launch_tail(t, sa));
    } else {
        HIPCHK(sync_launch(a, sa));           // general configurations: one wave per channel walks everything
    }
    RC(q->ev_end(1, sa));
    if (q->scout) {
        RC(q->ev_begin(2, sa));               // record placement
        HIPCHK(sync_launch_payload(a, 0, sa));
        RC(q->ev_end(2, sa));
        if (sw != sa) {
            HIPCHK(hipEventRecord(q->ev_scout[slot], sa));
            HIPCHK(hipStreamWaitEvent(sw, q->ev_scout[slot], 0));
        }
        // (Round 5: the K = 7 decoder's kernel + the general decoder on a stream of their own while convolutional frames arrive, so that the
        //  next push's workers need not queue behind 0.3 ms of trellis: 8 channels x 100 frames 41.4 -> 35.9 Gsample/s, x 50 frames 25.9 -> 19.8.
        //  Not kept: streams are not free, scratch/README.md.)
        // payload workers; the LDS-path packet decoder; the general decoder (nearly always an empty launch: ~12 us.
        // Giving it a stream of its own was tried: a fifth stream makes the harvest's copy stream share a hardware
        // queue with a busy one, and every poll then waits a slab's time for its 16-byte copies -- harvest 123 -> 98 Gsample/s)
        // While the lists of the launch behind the workers (QAM payloads, frames beyond the grid) and of the general decoder have been
        // empty for 64 launches, those two go to a stream of their own behind this push's decoder, with a decoder launch for their frames
        // only -- whatever they find is still done, in this push, it just no longer stands between this push's decoder and the next push's
        // workers (kernels.h, split_rest).  Everything of the push is finished when THAT stream is; a change of mode orders the two.
        const bool split = sw == q->s_work && q->s_side && q->h_hint && q->d_hint && q->list_seen[0] == 0 && q->list_seen[2] == 0 &&
                           q->seq >= 8 && a.frames_hint != ~0u && a.frames_hint < 2048u && sync_payload_splits(a);
        // Round 6: while frames arrive on the general decoder's list (the K = 7 code: 0.25-0.3 ms of trellis per 800 frames, one wave per
        // frame) THAT launch goes to the fourth stream behind this push's decoder, so that the next push's workers do not queue behind it:
        // the work stream carried workers + decoder + trellis = 0.47 ms per push of an 8-channel receiver, now 0.2 | 0.3 side by side.  All
        // general-decoder launches of such a phase follow each other on that one stream (they share the K = 7 scratch).
        const bool gen_side = !split && sw == q->s_work && q->s_side && q->h_hint && q->d_hint && q->list_seen[2] != 0 && q->seq >= 8 &&
                              !devel_env("MCRX_NO_GEN_SIDE");
        hipStream_t sd = sw;
        if (split) { a.split_rest = 1; a.dec_phase = 1; sd = q->s_side; }
        else if (gen_side) sd = q->s_side;
        else if (q->side_last) HIPCHK(hipStreamWaitEvent(sw, q->ev_side_last, 0));       // (back in one line: behind what the other stream still holds)
        RC(q->ev_begin(3, sw));
        HIPCHK(sync_launch_payload(a, 1, sw));
        RC(q->ev_end(3, sw));
        RC(q->ev_begin(4, sw));
        HIPCHK(sync_launch_payload(a, 2, sw));
        if (split) {
            HIPCHK(hipEventRecord(q->ev_side[slot], sw));
            HIPCHK(hipStreamWaitEvent(sd, q->ev_side[slot], 0));
            HIPCHK(sync_launch_payload(a, 4, sd));
            a.dec_phase = 2;
            HIPCHK(sync_launch_payload(a, 2, sd));
        } else if (gen_side) {
            HIPCHK(hipEventRecord(q->ev_side[slot], sw));
            HIPCHK(hipStreamWaitEvent(sd, q->ev_side[slot], 0));
        }
        HIPCHK(sync_launch_payload(a, 3, sd));
        RC(q->ev_end(4, sd));
        if (split || gen_side) HIPCHK(hipEventRecord(q->ev_side_last, sd));
        q->side_last = split || gen_side;
        sw = sd;
    }
    HIPCHK(hipEventRecord(q->ev_done[slot], sw));
    HIPCHK(hipEventRecord(q->ev_gen[g], sw));
    q->gen_used[g] = true;
    q->seq++;
    return MCRX_OK;
}

extern "C" int mcrx_hip_channelize(mcrx_hip_t q, const void *d_iq, size_t nblocks, uint64_t first_sample,
                                   const void *d_halo, void *d_out, unsigned groups, void *stream)
{
    DevScope dev_scope_(q ? q->device : -1);
    if (q && q->bypass) return fail(MCRX_EUNSUPP, "single_channel handle has no channelizer");
    if (q && q->oversampled) return fail(MCRX_EUNSUPP, "front_end = 2 (the oversampled front end stage by stage) runs inside execute_host / execute_device only");
    if (!q || !d_iq || !d_out) return fail(MCRX_EINVAL, "null argument");
    hipStream_t st = stream ? (hipStream_t)stream : q->stream;
    return launch_channelizer(q, (const float2 *)d_iq, nblocks, first_sample, (const float2 *)d_halo,
                              (float2 *)d_out, groups, nblocks / MCRX_TILE, st);
}
extern "C" int mcrx_hip_sync(mcrx_hip_t q, const void *d_chan, uint64_t first_sample, size_t nsamples, void *stream)
{
    DevScope dev_scope_(q ? q->device : -1);
    if (!q || !d_chan) return fail(MCRX_EINVAL, "null argument");
    hipStream_t st = stream ? (hipStream_t)stream : q->stream;
    return launch_sync(q, (const float2 *)d_chan, q->nch, 0, (int64_t)first_sample, (int64_t)(first_sample + nsamples), st);
}
extern "C" int mcrx_hip_restart(mcrx_hip_t q, void *stream)
{
    DevScope dev_scope_(q ? q->device : -1);
    if (!q) return fail(MCRX_EINVAL, "null handle");
    return restart_async(q, stream ? (hipStream_t)stream : q->stream, true);
}
extern "C" int mcrx_hip_kernel_time_ms(mcrx_hip_t q, float *ch_ms, float *sy_ms)
{
    DevScope dev_scope_(q ? q->device : -1);
    if (!q) return fail(MCRX_EINVAL, "null handle");
    for (int w = 0; w < MCRX_NKERNELS; w++) RC(q->ev_resolve(w));
    if (ch_ms) *ch_ms = q->ev_last[0];
    if (sy_ms) *sy_ms = q->ev_last[1] + q->ev_last[2] + q->ev_last[3] + q->ev_last[4];      // scout .. decode
    return MCRX_OK;
}
extern "C" int mcrx_hip_kernel_stats(mcrx_hip_t q, double ms_total[MCRX_NKERNELS], uint64_t launches[MCRX_NKERNELS], int reset)
{
    DevScope dev_scope_(q ? q->device : -1);
    // HIP-event durations of every launch since the last reset, recorded on the stream the
    // kernels were launched on (order: mcrx_hip.h)
    if (!q) return fail(MCRX_EINVAL, "null handle");
    for (int w = 0; w < MCRX_NKERNELS; w++) {
        RC(q->ev_resolve(w));
        if (ms_total) ms_total[w] = q->ev_ms_total[w];
        if (launches) launches[w] = q->ev_count[w];
        if (reset) { q->ev_ms_total[w] = 0; q->ev_count[w] = 0; }
    }
    return MCRX_OK;
}

// per-kernel HIP-event timing on / off (off when the handle is made): returns the previous setting, -1 for a null handle.
// Launches enqueued while it is off are not in mcrx_hip_kernel_stats / mcrx_hip_kernel_time_ms.
extern "C" int mcrx_hip_kernel_timing(mcrx_hip_t q, int on)
{
    if (!q) return -1;
    const int was = q->ev_on ? 1 : 0;
    q->ev_on = on != 0;
    return was;
}

// ---------------------------------------------------------------- streaming Execute()
static int ensure_chan(mcrx_hip_t q, size_t tiles)
{
    if (tiles <= q->chan_cap_tiles) return MCRX_OK;
    HIPCHK(hipDeviceSynchronize());             // every stream that may still read the old buffers
    float2 *nb[MCRX_SLOTS] = {};
    const size_t n = tiles * (size_t)q->N * MCRX_TILE, tile_elems = (size_t)q->N * MCRX_TILE;
    for (unsigned i = 0; i < q->nslots; i++) {
        HIPCHK(hipMalloc((void **)&nb[i], n * sizeof(float2)));
        HIPCHK(hipMemset(nb[i], 0, n * sizeof(float2)));
    }
    if (q->last_slot >= 0)                       // keep the synchronizer history: the tail of the last launch's tiles
        HIPCHK(hipMemcpy(nb[q->last_slot] + q->last_ntiles * tile_elems, q->d_chan[q->last_slot] + q->last_ntiles * tile_elems,
                         (size_t)q->hist_tiles * tile_elems * sizeof(float2), hipMemcpyDeviceToDevice));
    for (int i = 0; i < MCRX_SLOTS; i++) { if (q->d_chan[i]) (void)hipFree(q->d_chan[i]); q->d_chan[i] = nb[i]; }
    q->chan_cap_tiles = tiles;
    return MCRX_OK;
}

// cfg.front_end = 2: oscillator -> 2N-channel oversampled bank (two steps per block) -> half-band decimator per kept
// channel -> the synchronizers' tiles.  Input and bank output are double buffered with their filter history in front.
static int run_oversampled(mcrx_hip_t q, const float2 *x, size_t nblocks, uint64_t first_abs, float2 *tiles, hipStream_t sc)
{
    const size_t M = q->K, N = q->N, lead = (size_t)28 * N, hsteps = 32, nsteps = 2 * nblocks, nin = nblocks * q->K;
    if (nblocks > q->pf_cap_blocks) {
        HIPCHK(hipDeviceSynchronize());
        float2 *ni[2] = { nullptr, nullptr }, *no[2] = { nullptr, nullptr };
        for (int i = 0; i < 2; i++) {
            HIPCHK(hipMalloc((void **)&ni[i], (lead + nblocks * q->K) * sizeof(float2)));
            HIPCHK(hipMalloc((void **)&no[i], (hsteps + 2 * nblocks) * M * sizeof(float2)));
            HIPCHK(hipMemset(ni[i], 0, lead * sizeof(float2)));
            HIPCHK(hipMemset(no[i], 0, hsteps * M * sizeof(float2)));
        }
        if (q->pf_have_last) {      // carry the filter histories over: the tails of the last push
            const int c = q->pf_cur;
            HIPCHK(hipMemcpy(ni[c] + q->pf_last_blocks * q->K, q->d_pfin[c] + q->pf_last_blocks * q->K, lead * sizeof(float2), hipMemcpyDeviceToDevice));
            HIPCHK(hipMemcpy(no[c] + 2 * q->pf_last_blocks * M, q->d_pfout[c] + 2 * q->pf_last_blocks * M, hsteps * M * sizeof(float2), hipMemcpyDeviceToDevice));
        }
        for (int i = 0; i < 2; i++) {
            if (q->d_pfin[i]) (void)hipFree(q->d_pfin[i]);
            if (q->d_pfout[i]) (void)hipFree(q->d_pfout[i]);
            q->d_pfin[i] = ni[i]; q->d_pfout[i] = no[i];
        }
        q->pf_cap_blocks = nblocks;
    }
    const int prev = q->pf_cur, cur = prev ^ 1;
    float2 *in = q->d_pfin[cur], *out = q->d_pfout[cur];
    if (q->pf_have_last) {
        HIPCHK(hipMemcpyAsync(in, q->d_pfin[prev] + q->pf_last_blocks * q->K, lead * sizeof(float2), hipMemcpyDeviceToDevice, sc));
        HIPCHK(hipMemcpyAsync(out, q->d_pfout[prev] + 2 * q->pf_last_blocks * M, hsteps * M * sizeof(float2), hipMemcpyDeviceToDevice, sc));
    } else {
        HIPCHK(hipMemsetAsync(in, 0, lead * sizeof(float2), sc));
        HIPCHK(hipMemsetAsync(out, 0, hsteps * M * sizeof(float2), sc));
    }
    hipLaunchKernelGGL(nco_mix_kernel, dim3((unsigned)((nin + 255) / 256)), dim3(256), 0, sc, x, in + lead, (uint64_t)nin, (uint32_t)first_abs, q->dtheta);
    HIPCHK(hipGetLastError());
    RC(q->ev_begin(0, sc));
    if (mcrx_hip_pfb2_analyze(q->pfb2, in + lead, (size_t)std::min<uint64_t>(q->pf_in_valid, lead), nsteps, q->pf_steps, out + hsteps * M, sc) != MCRX_OK)
        return fail(MCRX_EHIP, mcrx_hip_pfb2_last_error());
    hipLaunchKernelGGL(halfband_adapter_kernel, dim3((unsigned)((N + 63) / 64), (unsigned)(nblocks / MCRX_TILE)), dim3(64), 0, sc,
                       out + hsteps * M, (uint32_t)M, (uint32_t)N, (uint32_t)(nblocks / MCRX_TILE), q->d_h1, tiles);
    HIPCHK(hipGetLastError());
    RC(q->ev_end(0, sc));
    q->pf_cur = cur; q->pf_last_blocks = nblocks; q->pf_have_last = true;
    q->pf_in_valid += nin; q->pf_steps += nsteps;
    return MCRX_OK;
}

// channelize + synchronize `nblocks` (multiple of MCRX_TILE) blocks sitting in device memory, readable in `st`'s order.
// Launch k writes the channel tiles of slot k % 3: [history: the last hist_tiles tiles of launch k-1][ntiles new].
static int run_blocks(mcrx_hip_t q, const float2 *x, size_t nblocks, uint64_t first_abs, hipStream_t st)
{
    if (nblocks == 0) return MCRX_OK;
    const size_t ntiles = nblocks / MCRX_TILE;
    RC(ensure_chan(q, q->hist_tiles + ntiles));
    const int slot = (int)(q->seq % q->nslots);
    float2 *buf = q->d_chan[slot];
    const size_t tile_elems = (size_t)q->N * MCRX_TILE;
    hipStream_t sc = st;
    if (q->pipelined) {
        // the channelizer runs on the handle's own stream once the input is there and launch k-3's workers have
        // finished with this slot's tiles
        sc = q->stream;
        if (st != sc) { HIPCHK(hipEventRecord(q->ev_in, st)); HIPCHK(hipStreamWaitEvent(sc, q->ev_in, 0)); }
        if (q->seq >= q->nslots) HIPCHK(hipStreamWaitEvent(sc, q->ev_done[slot], 0));
    }
    if (q->last_slot >= 0)
        HIPCHK(hipMemcpyAsync(buf, q->d_chan[q->last_slot] + q->last_ntiles * tile_elems,
                              (size_t)q->hist_tiles * tile_elems * sizeof(float2), hipMemcpyDeviceToDevice, sc));
    if (q->bypass)      // the input already is the channel's sample stream (one channel: its tiles are contiguous)
        HIPCHK(hipMemcpyAsync(buf + q->hist_tiles * tile_elems, x, nblocks * sizeof(float2), hipMemcpyDeviceToDevice, sc));
    else if (q->oversampled)
        RC(run_oversampled(q, x, nblocks, first_abs, buf + q->hist_tiles * tile_elems, sc));
    else
        RC(launch_channelizer(q, x, nblocks, first_abs, q->d_hist[q->hist_cur], buf + q->hist_tiles * tile_elems, 1, ntiles, sc));
    // FIR history: the last 13 (27) blocks of (history, x)
    if (!q->bypass && !q->oversampled) {
        const uint64_t nh = (uint64_t)q->hist_blocks * q->K;
        hipLaunchKernelGGL(hist_update_kernel, dim3((unsigned)((nh + 255) / 256)), dim3(256), 0, sc,
                           q->d_hist[q->hist_cur], x, (uint64_t)nblocks * q->K, q->d_hist[1 - q->hist_cur], nh);
        HIPCHK(hipGetLastError());
        q->hist_cur ^= 1;
    }
    if (sc != st) {     // the caller's stream may reuse x once the channelizer has read it -- not once the frames are decoded
        HIPCHK(hipEventRecord(q->ev_consumed, sc));
        HIPCHK(hipStreamWaitEvent(st, q->ev_consumed, 0));
    }
    const int64_t buf_first = q->chan_samples - (int64_t)q->hist_tiles * MCRX_TILE;
    q->last_slot = slot; q->last_ntiles = ntiles;
    RC(launch_sync(q, buf, q->N, q->ch_first, buf_first, q->chan_samples + (int64_t)nblocks, sc));
    q->chan_samples += (int64_t)nblocks;
    return MCRX_OK;
}

static int process_staged(mcrx_hip_t q)
{
    const size_t tile_samples = (size_t)MCRX_TILE * q->K;
    const size_t n = (q->stage_fill / tile_samples) * tile_samples;
    if (n == 0) return MCRX_OK;
    HIPCHK(hipMemcpyAsync(q->d_in, q->h_stage, n * sizeof(float2), hipMemcpyHostToDevice, q->stream));
    RC(run_blocks(q, q->d_in, n / q->K, q->stage_first, q->stream));
    HIPCHK(hipStreamSynchronize(q->stream));        // staging buffers are reused (the channelizer has read them; decoding goes on)
    const size_t rest = q->stage_fill - n;
    if (rest) memmove(q->h_stage, q->h_stage + n, rest * sizeof(float2));
    q->stage_fill = rest; q->stage_first += n;
    return MCRX_OK;
}

static int harvest(mcrx_hip_t q);
static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// frame records `blocks` channelizer blocks can produce at most (every channel sending shortest frames back to back)
static uint64_t record_bound(mcrx_hip_t q, uint64_t blocks) { return ((blocks + MCRX_TILE) / q->min_frame + 2) * q->nch; }

// Bulk Execute(buf, n): whole tiles go straight from the caller's (pageable) memory into one of two device
// buffers on a copy stream while the previous chunk is still being processed; frames are harvested only when
// the record pool could otherwise fill up.  What does not fill a tile is left for the staging path.
static int execute_direct(mcrx_hip_t q, const float2 *&src, size_t &nsamples, bool &overflow)
{
    const size_t tile_samples = (size_t)MCRX_TILE * q->K;
    if (q->stage_fill || nsamples < 64 * tile_samples) return MCRX_OK;
    // largest chunk whose frames are sure to fit the record pool, at most 16 Mi samples
    uint64_t blocks_cap = q->max_rec / q->nch > 3 ? (uint64_t)(q->max_rec / q->nch - 3) * q->min_frame : 0;
    size_t chunk = (size_t)std::min<uint64_t>(blocks_cap * q->K, (uint64_t)16 << 20) / tile_samples * tile_samples;
    if (chunk < 64 * tile_samples) return MCRX_OK;
    if (!q->copy_stream) {
        HIPCHK(hipStreamCreateWithFlags(&q->copy_stream, hipStreamNonBlocking));
        for (int i = 0; i < 2; i++) {
            HIPCHK(hipEventCreateWithFlags(&q->ev_dcopy[i], hipEventDisableTiming));
            HIPCHK(hipEventCreateWithFlags(&q->ev_ddone[i], hipEventDisableTiming));
        }
    }
    while (nsamples >= 64 * tile_samples) {
        const size_t take = std::min(chunk, nsamples / tile_samples * tile_samples);
        if (take > q->direct_cap) {
            HIPCHK(hipDeviceSynchronize());
            for (int i = 0; i < 2; i++) {
                if (q->d_direct[i]) (void)hipFree(q->d_direct[i]);
                q->d_direct[i] = nullptr; q->direct_used[i] = false;
                HIPCHK(hipMalloc((void **)&q->d_direct[i], take * sizeof(float2)));
            }
            q->direct_cap = take;
        }
        const uint64_t bound = record_bound(q, take / q->K);
        if (q->pending_bound && q->pending_bound + bound > q->max_rec) {
            int rc = harvest(q);
            if (rc != MCRX_OK && rc != MCRX_EOVERFLOW) return rc;
            overflow |= rc == MCRX_EOVERFLOW;
        }
        const int b = q->direct_idx ^= 1;
        double t0 = now_s();
        if (q->direct_used[b]) HIPCHK(hipEventSynchronize(q->ev_ddone[b]));
        q->t_wait += now_s() - t0; t0 = now_s();
        HIPCHK(hipMemcpyAsync(q->d_direct[b], src, take * sizeof(float2), hipMemcpyHostToDevice, q->copy_stream));
        HIPCHK(hipEventRecord(q->ev_dcopy[b], q->copy_stream));
        HIPCHK(hipStreamWaitEvent(q->stream, q->ev_dcopy[b], 0));
        q->t_copy += now_s() - t0; t0 = now_s();
        RC(run_blocks(q, q->d_direct[b], take / q->K, q->total_samples, q->stream));
        q->t_run += now_s() - t0; t0 = now_s();
        HIPCHK(hipEventRecord(q->ev_ddone[b], q->stream));
        q->direct_used[b] = true;
        HIPCHK(hipStreamSynchronize(q->copy_stream));          // the caller's buffer has been read
        q->t_copy += now_s() - t0;
        q->pending_bound += bound;
        q->total_samples += take; q->stage_first = q->total_samples;
        src += take; nsamples -= take;
    }
    return MCRX_OK;
}

extern "C" int mcrx_hip_execute_host(mcrx_hip_t q, const float *iq, size_t nsamples)
{
    DevScope dev_scope_(q ? q->device : -1);
    if (!q || (!iq && nsamples)) return fail(MCRX_EINVAL, "null argument");
    const float2 *src = reinterpret_cast<const float2 *>(iq);
    bool overflow = false;
    const size_t tile_samples = (size_t)MCRX_TILE * q->K;
    if (q->stage_fill && nsamples >= 64 * tile_samples) {
        // a partial tile is waiting: complete it from this buffer so that the bulk path can take over
        const size_t need = std::min(nsamples, (tile_samples - q->stage_fill % tile_samples) % tile_samples);
        memcpy(q->h_stage + q->stage_fill, src, need * sizeof(float2));
        q->stage_fill += need; q->total_samples += need; src += need; nsamples -= need;
        RC(process_staged(q));
        q->pending_bound += record_bound(q, q->stage_cap / q->K);
    }
    RC(execute_direct(q, src, nsamples, overflow));
    while (nsamples) {
        const size_t take = std::min(nsamples, q->stage_cap - q->stage_fill);
        memcpy(q->h_stage + q->stage_fill, src, take * sizeof(float2));
        q->stage_fill += take; q->total_samples += take; src += take; nsamples -= take;
        if (q->stage_fill == q->stage_cap) {
            RC(process_staged(q));
            int rc = harvest(q);                // frames become deliverable as soon as a batch is done
            if (rc != MCRX_OK && rc != MCRX_EOVERFLOW) return rc;
            overflow |= rc == MCRX_EOVERFLOW;
        }
    }
    return overflow ? fail(MCRX_EOVERFLOW, "frame pool exhausted: frames were dropped (raise max_frames)") : MCRX_OK;
}

extern "C" int mcrx_hip_execute_device(mcrx_hip_t q, const void *d_iq, size_t nsamples, void *stream)
{
    DevScope dev_scope_(q ? q->device : -1);
    if (!q || (!d_iq && nsamples)) return fail(MCRX_EINVAL, "null argument");
    if (q->stage_fill) return fail(MCRX_EINVAL, "host samples are still staged: flush before pushing device buffers");
    if (nsamples % ((size_t)MCRX_TILE * q->K)) return fail(MCRX_EINVAL, "device pushes must be whole tiles of MCRX_TILE = 16 blocks (32*N samples)");
    hipStream_t st = stream ? (hipStream_t)stream : q->stream;
    // optional split into sub-slabs: within one call, sub-slab i+1's channelizer and acquisition overlap sub-slab i's
    // payload workers (consecutive calls overlap in the same way without it)
    const size_t nblocks = nsamples / q->K;
    size_t chunk = q->cfg.struct_size >= offsetof(mcrx_hip_config, chunk_blocks) + sizeof(uint32_t) && q->cfg.chunk_blocks
                       ? ((size_t)q->cfg.chunk_blocks + MCRX_TILE - 1) / MCRX_TILE * MCRX_TILE : nblocks;
    const float2 *x = (const float2 *)d_iq;
    for (size_t b = 0; b < nblocks; b += chunk) {
        const size_t nb = std::min(chunk, nblocks - b);
        RC(run_blocks(q, x + b * q->K, nb, q->total_samples, st));
        q->total_samples += nb * q->K;
    }
    q->stage_first = q->total_samples;
    q->pending_bound += record_bound(q, nblocks);
    return MCRX_OK;
}

extern "C" int mcrx_hip_stream_wait(mcrx_hip_t q, void *stream)
{
    DevScope dev_scope_(q ? q->device : -1);
    if (!q) return fail(MCRX_EINVAL, "null handle");
    return join_into(q, stream ? (hipStream_t)stream : q->stream);
}
extern "C" uint64_t mcrx_hip_launches(mcrx_hip_t q) { return q ? q->seq : 0; }
extern "C" unsigned mcrx_hip_history_tiles(mcrx_hip_t q) { return q ? q->hist_tiles : 0; }
extern "C" int mcrx_hip_stream_wait_launch(mcrx_hip_t q, uint64_t launch, void *stream)
{
    DevScope dev_scope_(q ? q->device : -1);
    // the event ring holds the last MCRX_SLOTS launches; an older launch shares its slot with a later one that
    // finishes after it (payload workers and decoders run in launch order), so waiting for the slot is enough
    if (!q) return fail(MCRX_EINVAL, "null handle");
    if (launch >= q->seq) return fail(MCRX_EINVAL, "no such launch");
    HIPCHK(hipStreamWaitEvent(stream ? (hipStream_t)stream : q->stream, q->ev_done[launch % q->nslots], 0));
    return MCRX_OK;
}

// ---------------------------------------------------------------- frames
// Copy one closed result generation to the host once its last launch has finished (only that is waited for:
// launches enqueued later write the other generation and keep the GPU busy during the copy).
static int collect(mcrx_hip_t q, int g)
{
    if (!q->gen_closed[g]) return MCRX_OK;
    const double t0 = now_s();
    HIPCHK(hipEventSynchronize(q->ev_gen[g]));
    q->t_wait += now_s() - t0;
    uint32_t cnt[2] = { 0, 0 }; unsigned long long used[2] = { 0, 0 };
    if (q->gen_abandoned[g]) {              // dropped by mcrx_hip_discard: never delivered, only cleaned
        HIPCHK(hipMemsetAsync(q->d_nrec[g], 0, 32, q->s_copy));
        HIPCHK(hipEventRecord(q->ev_clean[g], q->s_copy)); q->clean_pending[g] = true;
        q->gen_closed[g] = false; q->gen_used[g] = false; q->gen_abandoned[g] = false;
        return MCRX_OK;
    }
    { const double t1 = now_s();
    { unsigned long long blk[4] = { 0, 0, 0, 0 };
      HIPCHK(hipMemcpyAsync(blk, q->d_nrec[g], sizeof(blk), hipMemcpyDeviceToHost, q->s_copy));
      HIPCHK(hipStreamSynchronize(q->s_copy));
      cnt[0] = (uint32_t)blk[0]; cnt[1] = (uint32_t)(blk[0] >> 32); used[0] = blk[2]; used[1] = blk[3]; }
    q->t_cnt += now_s() - t1; }
    q->dropped += cnt[1];
    const uint32_t n = std::min(cnt[0], q->max_rec);
    if (used[0] > q->arena_cap) used[0] = q->arena_cap;
    if (used[1] > q->sarena_cap) used[1] = q->sarena_cap;
    const bool want_syms = !(q->cfg.struct_size >= offsetof(mcrx_hip_config, skip_framesyms) + sizeof(uint32_t) && q->cfg.skip_framesyms);
    if (n) {
        // drop frames already delivered, then append
        if (q->next_frame == q->recs.size()) { q->recs.clear(); q->arena_host.clear(); q->sarena_host.clear(); q->next_frame = 0; }
        const size_t base = q->arena_host.size, sbase = q->sarena_host.size, r0 = q->recs.size();
        q->recs.resize(r0 + n);
        { const double t1 = now_s();
          if (q->arena_host.grow_to(base + (size_t)used[0]) != MCRX_OK) return fail(MCRX_ENOMEM, "host frame arena allocation failed");
          if (want_syms && q->sarena_host.grow_to(sbase + (size_t)used[1]) != MCRX_OK) return fail(MCRX_ENOMEM, "host frame arena allocation failed");
          q->t_grow += now_s() - t1; }
        { const double t1 = now_s();
          HIPCHK(hipMemcpyAsync(q->recs.data() + r0, q->d_rec[g], (size_t)n * sizeof(FrameRec), hipMemcpyDeviceToHost, q->s_copy));
          if (used[0]) HIPCHK(hipMemcpyAsync(q->arena_host.p + base, q->d_arena[g], (size_t)used[0], hipMemcpyDeviceToHost, q->s_copy));
          if (want_syms && used[1]) HIPCHK(hipMemcpyAsync(q->sarena_host.p + sbase, q->d_sarena[g], (size_t)used[1], hipMemcpyDeviceToHost, q->s_copy));
          HIPCHK(hipStreamSynchronize(q->s_copy));
          q->t_d2h += now_s() - t1; q->b_d2h += (double)used[0] + (want_syms ? (double)used[1] : 0.0); }
        const double t2 = now_s();
        // (record slots of frames that found no room in an arena stay unfilled: Walker::place_owned marks them)
        q->recs.erase(std::remove_if(q->recs.begin() + r0, q->recs.end(), [](const FrameRec &r) { return r.channel == 0xFFFFFFFFu; }), q->recs.end());
        for (size_t i = r0; i < q->recs.size(); i++) {
            q->recs[i].payload_off += base;
            if (want_syms) q->recs[i].syms_off += sbase; else q->recs[i].num_framesyms = 0;
        }
        // reference order: by end time, then channel index (lib/multichannelrx.cc:193-194)
        std::stable_sort(q->recs.begin() + r0, q->recs.end(), [](const FrameRec &a, const FrameRec &b) {
            return a.end_sample != b.end_sample ? a.end_sample < b.end_sample : a.channel < b.channel; });
        q->t_post += now_s() - t2;
    }
    // (the counters are zeroed on the copy stream without waiting for it: a fill is a kernel, and on a full chip a kernel waits
    //  100-200 us for a wave slot -- with the host waiting for it, every poll.  The next launch into this generation waits instead.)
    HIPCHK(hipMemsetAsync(q->d_nrec[g], 0, 32, q->s_copy));
    HIPCHK(hipEventRecord(q->ev_clean[g], q->s_copy)); q->clean_pending[g] = true;
    q->gen_closed[g] = false; q->gen_used[g] = false;
    return cnt[1] ? MCRX_EOVERFLOW : MCRX_OK;
}
// close the generation launches are writing into; later launches write the next one of the ring (which must be clean)
static void close_generation(mcrx_hip_t q)
{
    const int g = q->gen, next = (g + 1) % MCRX_GENS;
    if (!q->gen_used[g] || q->gen_used[next] || q->gen_closed[next]) return;
    q->gen_closed[g] = true; q->gen_close_seq[g] = ++q->close_counter;
    q->gen = next;
}
// collect every closed generation, oldest first (keep_newest: all but the one closed last -- mcrx_hip_poll)
static int collect_closed(mcrx_hip_t q, int keep_newest = 0)
{
    int rc = MCRX_OK;
    while (true) {
        int best = -1, newest = -1, nclosed = 0;
        for (int g = 0; g < MCRX_GENS; g++) {
            if (!q->gen_closed[g]) continue;
            nclosed++;
            if (best < 0 || q->gen_close_seq[g] < q->gen_close_seq[best]) best = g;
            if (newest < 0 || q->gen_close_seq[g] > q->gen_close_seq[newest]) newest = g;
        }
        if (best < 0 || nclosed <= keep_newest) break;
        const int r = collect(q, best);
        if (r != MCRX_OK && r != MCRX_EOVERFLOW) return r;
        if (r == MCRX_EOVERFLOW) rc = r;
    }
    return rc;
}

// Everything decoded so far becomes deliverable: all generations, blocking.
static int harvest(mcrx_hip_t q)
{
    const double t0 = now_s();
    int rc = collect_closed(q);
    if (rc == MCRX_OK || rc == MCRX_EOVERFLOW) {
        close_generation(q);
        const int rc2 = collect_closed(q);
        rc = (rc2 == MCRX_OK) ? rc : rc2;
    }
    q->pending_bound = 0;
    q->t_harvest += now_s() - t0;
    return rc;
}

// Overlapped harvest for a stream that keeps coming: deliver the frames of the launches enqueued before the
// PREVIOUS poll (waiting only for those), and close the current generation for the next poll.  With one
// execute_device + one poll per slab, slab k-1's frames cross the host link while slab k is being processed.
extern "C" int mcrx_hip_poll(mcrx_hip_t q)
{
    DevScope dev_scope_(q ? q->device : -1);
    if (!q) return fail(MCRX_EINVAL, "null handle");
    // Two pushes stay in flight: what is collected here was closed by the poll before the previous one, so the wait is for the
    // launches of two pushes ago while the device has the last two queued.  (Collecting the previous poll's generation left the
    // device with ONE push in flight while the host waited, copied and sorted: consecutive pushes no longer overlapped, 0.82 x value.)
    const double t0 = now_s();
    static const int keep = devel_env("MCRX_POLL_KEEP") ? std::max(0, std::min(MCRX_GENS - 2, atoi(devel_env("MCRX_POLL_KEEP")))) : 1;
    const int rc = collect_closed(q, keep);
    if (rc == MCRX_OK || rc == MCRX_EOVERFLOW) close_generation(q);
    q->pending_bound = record_bound(q, 0);
    q->t_harvest += now_s() - t0;
    return rc;
}

// Benchmark helper: the frames decoded since the last poll/discard stay in HBM and are dropped.  Launches move on
// to the next generation of the ring; if that one still holds (abandoned) frames, its counters are zeroed on the
// device once its launches -- MCRX_GENS - 1 discards ago -- have finished.  Nothing is waited for on the host.
extern "C" int mcrx_hip_discard(mcrx_hip_t q)
{
    DevScope dev_scope_(q ? q->device : -1);
    if (!q) return fail(MCRX_EINVAL, "null handle");
    const int g = q->gen, next = (g + 1) % MCRX_GENS;
    if (!q->gen_used[g]) return MCRX_OK;
    if (q->gen_used[next] || q->gen_closed[next]) {
        HIPCHK(hipStreamWaitEvent(q->s_copy, q->ev_gen[next], 0));
        HIPCHK(hipMemsetAsync(q->d_nrec[next], 0, 32, q->s_copy));
        HIPCHK(hipEventRecord(q->ev_gen[next], q->s_copy));
        // the next launches into that generation start behind the zeroing: on the stream the acquisition kernels (which
        // place records and advance these counters) were last launched on -- the caller's own stream on a serial handle
        // driven through execute_device(stream), the handle's otherwise -- and on the handle's default ones as well
        hipStream_t sa = (q->pipelined && q->scout) ? q->s_scout : q->stream;
        HIPCHK(hipStreamWaitEvent(sa, q->ev_gen[next], 0));
        if (q->acq_stream && q->acq_stream != sa) HIPCHK(hipStreamWaitEvent(q->acq_stream, q->ev_gen[next], 0));
        q->gen_closed[next] = false; q->gen_used[next] = false; q->gen_abandoned[next] = false;
    }
    // abandoned: a later poll / flush cleans it without delivering anything
    q->gen_closed[g] = true; q->gen_abandoned[g] = true; q->gen_close_seq[g] = ++q->close_counter;
    q->gen = next;
    q->pending_bound = 0;
    return MCRX_OK;
}

extern "C" int mcrx_hip_spec_stats(mcrx_hip_t q, uint64_t *walked, uint64_t *adopted, int reset)
{
    DevScope dev_scope_(q ? q->device : -1);
    if (!q) return fail(MCRX_EINVAL, "null handle");
    HIPCHK(hipDeviceSynchronize());
    uint32_t v[4] = { 0, 0, 0, 0 };
    HIPCHK(hipMemcpy(v, q->d_stats, sizeof(v), hipMemcpyDeviceToHost));
    if (walked) *walked = v[0];
    if (adopted) *adopted = v[1];
    if (reset) HIPCHK(hipMemset(q->d_stats, 0, 8 * sizeof(uint32_t)));
    return MCRX_OK;
}

extern "C" int mcrx_hip_viterbi_stats(mcrx_hip_t q, uint64_t *frames, uint64_t *forward_repeats, uint64_t *traceback_repeats, int reset)
{
    DevScope dev_scope_(q ? q->device : -1);
    if (!q) return fail(MCRX_EINVAL, "null handle");
    uint32_t v[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
    if (q->d_vit_passes) {
        HIPCHK(hipDeviceSynchronize());
        HIPCHK(hipMemcpy(v, q->d_vit_passes, sizeof(v), hipMemcpyDeviceToHost));
        if (reset) HIPCHK(hipMemset(q->d_vit_passes, 0, sizeof(v)));
    }
    if (frames) *frames = v[2];
#ifdef VF_PROF                  // (make S1FLAGS=-DVF_PROF: the kernel then leaves its clock readings in words 4..7)
    if (q->debug && v[2]) fprintf(stderr, "[viterbi] frames %u, per frame: forward %.1f us, traceback %.1f us ; first to last wave start %.1f us (builds with -DVF_PROF)\n", v[2], v[4] / 100.0 / v[2], v[5] / 100.0 / v[2], (v[7] - ~v[6]) / 100.0);
#endif
    if (forward_repeats) *forward_repeats = v[0];
    if (traceback_repeats) *traceback_repeats = v[1];
    return MCRX_OK;
}

extern "C" int mcrx_hip_flush(mcrx_hip_t q)
{
    DevScope dev_scope_(q ? q->device : -1);
    if (!q) return fail(MCRX_EINVAL, "null handle");
    RC(process_staged(q));
    return harvest(q);
}
extern "C" size_t mcrx_hip_frames_pending(mcrx_hip_t q) { return q ? q->recs.size() - q->next_frame : 0; }
extern "C" uint64_t mcrx_hip_frames_dropped(mcrx_hip_t q) { return q ? q->dropped : 0; }
extern "C" int mcrx_hip_next_frame(mcrx_hip_t q, mcrx_frame *out)
{
    if (!q || !out) return fail(MCRX_EINVAL, "null argument");
    if (q->next_frame >= q->recs.size()) return 0;
    const FrameRec &r = q->recs[q->next_frame++];
    out->channel = r.channel; out->header_valid = r.header_valid; out->payload_valid = r.payload_valid;
    out->payload_len = r.payload_len; memcpy(out->header, r.header, 8);
    out->evm = r.evm; out->rssi = r.rssi; out->cfo = r.cfo;
    out->mod_scheme = r.mod_scheme; out->mod_bps = r.mod_bps; out->check = r.check; out->fec0 = r.fec0; out->fec1 = r.fec1;
    out->num_framesyms = r.num_framesyms; out->end_sample = (uint64_t)r.end_sample;
    out->payload = r.payload_len ? q->arena_host.p + r.payload_off : nullptr;
    out->framesyms = r.num_framesyms ? reinterpret_cast<const float *>(q->sarena_host.p + r.syms_off) : nullptr;
    return 1;
}

extern "C" int mcrx_hip_drain_count(mcrx_hip_t q, uint64_t *frames, uint64_t *valid, uint64_t *bytes)
{
    if (!q) return fail(MCRX_EINVAL, "null handle");
    uint64_t n = 0, ok = 0, nb = 0, touch = 0;
    mcrx_frame f;
    while (mcrx_hip_next_frame(q, &f) == 1) {
        n++; ok += (f.header_valid && f.payload_valid) ? 1 : 0; nb += f.payload_len;
        if (f.payload_len) touch += f.payload[0] + f.payload[f.payload_len - 1];       // the payload really is in host memory
    }
    q->drain_touch += touch;
    if (frames) *frames = n;
    if (valid) *valid = ok;
    if (bytes) *bytes = nb;
    return MCRX_OK;
}

extern "C" int mcrx_hip_reset(mcrx_hip_t q)
{
    DevScope dev_scope_(q ? q->device : -1);
    // multichannelrx::Reset (lib/multichannelrx.cc:135-153): synchronizers and channelizer windows
    // reset, partial block dropped, NCO keeps running.  Everything pushed before the Reset has been
    // synchronized by the reference at this point, so the whole tiles still in the staging buffer are processed
    // first and only the sub-tile tail (< 32 N samples; the reference drops < 2 N) is discarded.  Frames decoded
    // so far stay deliverable.
    if (!q) return fail(MCRX_EINVAL, "null handle");
    RC(process_staged(q));
    int rc = harvest(q);
    if (rc != MCRX_OK && rc != MCRX_EOVERFLOW) return rc;
    RC(restart_async(q, q->stream, false));
    HIPCHK(hipDeviceSynchronize());
    return MCRX_OK;
}
extern "C" int mcrx_hip_reset_at(mcrx_hip_t q, uint64_t chan_position)
{
    DevScope dev_scope_(q ? q->device : -1);
    if (!q) return fail(MCRX_EINVAL, "null handle");
    HIPCHK(hipDeviceSynchronize());                      // stage-level launches run on the caller's and the handle's streams
    int rc = harvest(q);
    if (rc != MCRX_OK && rc != MCRX_EOVERFLOW) return rc;
    q->chan_samples = (int64_t)chan_position;
    RC(restart_async(q, q->stream, false));
    HIPCHK(hipDeviceSynchronize());
    return MCRX_OK;
}

// ---------------------------------------------------------------- device bookkeeping, testable without a second GPU
extern "C" int mcrx_hip_device(mcrx_hip_t q) { return q ? q->device : -1; }
// the per-device "done once" table of devscope.hpp driven with device ids that need not exist: 0 = every check passed
extern "C" int mcrx_hip_selftest_device_table(void)
{
    PerDeviceOnce t;
    int calls = 0;
    auto work = [&]() { calls++; return hipSuccess; };
    auto fail_once = [&]() { calls++; return hipErrorInvalidValue; };
    for (int dev : { 0, 1, 7, 63, 64, 255 }) {
        if (t.is_done(dev)) return 1;
        if (t.run(dev, work) != hipSuccess || !t.is_done(dev)) return 2;
        const int before = calls;
        if (t.run(dev, work) != hipSuccess || calls != before) return 3;          // second time on the same device: not done again
    }
    if (t.is_done(2) || t.is_done(65)) return 4;                                   // other devices are untouched
    if (t.run(3, fail_once) == hipSuccess || t.is_done(3)) return 5;               // a failure is not remembered as done
    if (t.run(3, work) != hipSuccess || !t.is_done(3)) return 6;
    const int before = calls;
    if (t.run(-1, work) != hipSuccess || t.run(300, work) != hipSuccess || calls != before + 2) return 7;     // outside the table: every time
    return 0;
}
