// multichannelrx.cc -- host class over the C-ABI of libmcrx_hip.so (include/mcrx_hip.h).
// Mirrors liquid-usrp's lib/multichannelrx.cc: ctor :45-104, dtor :107-132, Reset :135-153,
// Execute :155-182; the DSP itself runs in the gfx950 kernels.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <mutex>
#include <string>
#include <vector>
#include <unistd.h>

#include "multichannelrx.h"
#include "mcrx_hip.h"

struct multichannelrx::impl {
    mcrx_hip_t h;
    std::recursive_mutex mu;                // multichanneltxrx resets the receiver while its worker is in Execute()
    std::vector<void *> userdata;
    std::vector<framesync_callback> callback;
    std::vector<unsigned char> payload;     // callbacks get mutable buffers, like liquid's
    const char *debug_dir;                                          // $MCRX_DEBUG_DIR (NULL: no dump at destruction)
    std::vector<std::vector<std::complex<float> > > debug_syms;    // [channel]: equalised symbols of its last frame
    std::vector<unsigned long> debug_frames;
    // Sharded over the GPUs of a node (one process per GPU), switched on from outside the unchanged application:
    //   MCRX_WORLD / MCRX_RANK   ranks of the job and this process's number (a launcher's WORLD_SIZE / RANK are taken as well)
    //   MCRX_UID_FILE            where rank 0 leaves the 128-byte ncclUniqueId for the others (world > 1)
    //   MCRX_SUB_BLOCKS          blocks of 2N samples per rank and round (default 8192)
    // Every rank is handed the whole wideband stream, like every process behind a shared radio would be; of each round of
    // world x sub_blocks blocks it channelizes sub-slab number `rank`, one exchange turns the time shards into channel shards, and
    // this object's callbacks fire for its shard of num_channels / world channels only (mcrx_hip_pipeline_*, csrc/pipeline.hip).
    mcrx_hip_pipeline_t pipe;
    int rank, world;
    size_t sub_blocks, K, fill;                                     // fill: samples of the current round in `round` behind the halo
    std::vector<std::complex<float> > round;                       // [13 halo blocks][world * sub_blocks blocks]
};

static int env_int(const char *a, const char *b, int dflt)
{
    const char *v = getenv(a);
    if (!v && b) v = getenv(b);
    return v ? atoi(v) : dflt;
}

multichannelrx::multichannelrx(unsigned int _num_channels, unsigned int _M, unsigned int _cp_len,
                               unsigned int _taper_len, unsigned char *_p, void **_userdata,
                               framesync_callback *_callback)
    : num_channels(_num_channels), pimpl(new impl)
{
    pimpl->h = NULL; pimpl->pipe = NULL;
    pimpl->world = getenv("MCRX_WORLD") ? env_int("MCRX_WORLD", NULL, 1) : 0;       // 0: not sharded (the plain receiver, no pipeline in between)
    pimpl->rank = env_int("MCRX_RANK", "RANK", 0);
    pimpl->sub_blocks = (size_t)env_int("MCRX_SUB_BLOCKS", NULL, 8192) / MCRX_TILE * MCRX_TILE;
    pimpl->K = 2 * (size_t)_num_channels; pimpl->fill = 0;
    pimpl->debug_dir = getenv("MCRX_DEBUG_DIR");
    pimpl->debug_syms.resize(_num_channels); pimpl->debug_frames.assign(_num_channels, 0);
    mcrx_hip_config cfg;
    memset(&cfg, 0, sizeof(cfg));
    cfg.struct_size = sizeof(cfg);
    cfg.payload_soft = 1;
    const int W = pimpl->world;
    if (W > 0) {
        if (pimpl->rank < 0 || pimpl->rank >= W || _num_channels % (unsigned)W || pimpl->sub_blocks == 0) {
            fprintf(stderr, "error: multichannelrx, MCRX_WORLD = %d must divide the %u channels, MCRX_RANK = %d lie in [0, MCRX_WORLD), MCRX_SUB_BLOCKS be at least %d\n",
                    W, _num_channels, pimpl->rank, MCRX_TILE);
            delete pimpl;
            throw 0;
        }
        cfg.channel_count = _num_channels / (unsigned)W; cfg.channel_first = (unsigned)pimpl->rank * cfg.channel_count;
        cfg.defer_samples = 16384;              // a frame cut by a round boundary is acquired again, whole, by the next round
    }
    int rc = mcrx_hip_create(&pimpl->h, _num_channels, _M, _cp_len, _taper_len, _p, &cfg);
    if (rc != MCRX_OK) {
        fprintf(stderr, "%s\n", mcrx_hip_last_error());
        delete pimpl;
        throw 0;
    }
    if (W > 0) {
        unsigned char uid[128];
        memset(uid, 0, sizeof(uid));
        const char *uf = getenv("MCRX_UID_FILE");
        bool ok = true;
        if (W > 1) {
            ok = uf != NULL;
            if (ok && pimpl->rank == 0) {           // rank 0 makes the id and leaves it where the others look (written aside, then renamed)
                ok = mcrx_hip_pipeline_unique_id(uid) == MCRX_OK;
                std::string tmp = std::string(uf) + ".tmp";
                FILE *f = ok ? fopen(tmp.c_str(), "wb") : NULL;
                ok = f && fwrite(uid, 1, 128, f) == 128;
                if (f) fclose(f);
                ok = ok && rename(tmp.c_str(), uf) == 0;
            } else if (ok) {
                ok = false;
                for (int t = 0; t < 600 && !ok; t++) {      // up to a minute
                    FILE *f = fopen(uf, "rb");
                    if (f) { ok = fread(uid, 1, 128, f) == 128; fclose(f); }
                    if (!ok) usleep(100000);
                }
            }
        }
        if (!ok || mcrx_hip_pipeline_create(&pimpl->pipe, pimpl->h, pimpl->rank, W, W > 1 ? uid : NULL, pimpl->sub_blocks, 0) != MCRX_OK) {
            fprintf(stderr, "error: multichannelrx, sharded receiver (MCRX_WORLD = %d): %s\n", W,
                    ok ? mcrx_hip_pipeline_last_error() : "no ncclUniqueId (MCRX_UID_FILE: rank 0 writes it, the others read it)");
            mcrx_hip_destroy(pimpl->h);
            delete pimpl;
            throw 0;
        }
        pimpl->round.assign((13 + (size_t)W * pimpl->sub_blocks) * pimpl->K, std::complex<float>(0.f, 0.f));
    }
    for (unsigned int i = 0; i < _num_channels; i++) {
        pimpl->userdata.push_back(_userdata ? _userdata[i] : NULL);
        pimpl->callback.push_back(_callback ? _callback[i] : NULL);
    }
}

multichannelrx::~multichannelrx()
{
    if (pimpl->h) {
        if (pimpl->pipe) mcrx_hip_pipeline_wait(pimpl->pipe);
        mcrx_hip_flush(pimpl->h);
        Deliver();
        if (pimpl->pipe) mcrx_hip_pipeline_destroy(pimpl->pipe);
        mcrx_hip_destroy(pimpl->h);
    }
    if (pimpl->debug_dir) {
        // the reference, built with BST_DEBUG, leaves liquid's internal dump of every synchronizer behind
        // (ofdmflexframesync_debug_print, lib/multichannelrx.cc:118-122: "framesync_channel%u.m"); here, with MCRX_DEBUG_DIR
        // set, the same file names hold what this receiver can show: the equalised symbols of each channel's last frame
        for (unsigned int i = 0; i < num_channels; i++) {
            char fn[1024];
            snprintf(fn, sizeof(fn), "%s/framesync_channel%u.m", pimpl->debug_dir, i);
            FILE *fid = fopen(fn, "w");
            if (!fid) continue;
            fprintf(fid, "%% channel %u: %lu frames received; equalised payload symbols of the last one\nclear all; close all;\n", i, pimpl->debug_frames[i]);
            fprintf(fid, "framesyms = [");
            for (const auto &v : pimpl->debug_syms[i]) fprintf(fid, " %.6e%+.6ej", v.real(), v.imag());
            fprintf(fid, " ];\nfigure; plot(real(framesyms), imag(framesyms), 'x'); axis square; grid on;\n");
            fclose(fid);
        }
    }
    delete pimpl;
}

void multichannelrx::Deliver()
{
    mcrx_frame f;
    while (mcrx_hip_next_frame(pimpl->h, &f) == 1) {
        if (pimpl->debug_dir && f.channel < num_channels && f.num_framesyms) {      // keep the last frame of every channel for the dump
            const std::complex<float> *p = reinterpret_cast<const std::complex<float> *>(f.framesyms);
            pimpl->debug_syms[f.channel].assign(p, p + f.num_framesyms);
            pimpl->debug_frames[f.channel]++;
        }
        if (f.channel >= num_channels || !pimpl->callback[f.channel]) continue;
        framesyncstats_s st;
        st.evm = f.evm; st.rssi = f.rssi; st.cfo = f.cfo;
        st.framesyms = reinterpret_cast<liquid_float_complex *>(const_cast<float *>(f.framesyms));
        st.num_framesyms = f.num_framesyms;
        st.mod_scheme = f.mod_scheme; st.mod_bps = f.mod_bps; st.check = f.check; st.fec0 = f.fec0; st.fec1 = f.fec1;
        unsigned char header[8];
        memcpy(header, f.header, 8);
        pimpl->payload.assign(f.payload, f.payload + f.payload_len);
        pimpl->callback[f.channel](header, f.header_valid, f.payload_len ? pimpl->payload.data() : NULL,
                                   f.payload_len, f.payload_valid, st, pimpl->userdata[f.channel]);
    }
}

void multichannelrx::Reset()
{
    std::lock_guard<std::recursive_mutex> lk(pimpl->mu);
    mcrx_hip_reset(pimpl->h);
    Deliver();
}

void multichannelrx::Execute(std::complex<float> *_x, unsigned int _num_samples)
{
    std::lock_guard<std::recursive_mutex> lk(pimpl->mu);
    if (pimpl->pipe) {
        // sharded: the stream is collected a round at a time; of every full round this rank's sub-slab (the 13 blocks in front of
        // it included: they sit right there in the stream) goes to the GPU, and the frames of the rounds before come back
        const size_t K = pimpl->K, halo = 13 * K, cap = (size_t)pimpl->world * pimpl->sub_blocks * K;
        size_t done = 0;
        while (done < _num_samples) {
            const size_t take = std::min<size_t>(_num_samples - done, cap - pimpl->fill);
            memcpy(pimpl->round.data() + halo + pimpl->fill, _x + done, take * sizeof(std::complex<float>));
            pimpl->fill += take; done += take;
            if (pimpl->fill < cap) break;
            const std::complex<float> *mine = pimpl->round.data() + (size_t)pimpl->rank * pimpl->sub_blocks * K;      // = halo of sub-slab `rank`
            if (mcrx_hip_pipeline_push_host(pimpl->pipe, reinterpret_cast<const float *>(mine)) != MCRX_OK) {
                fprintf(stderr, "error: multichannelrx::Execute(), %s\n", mcrx_hip_pipeline_last_error());
                throw 0;
            }
            mcrx_hip_pipeline_wait(pimpl->pipe);                    // (the round buffer is about to be overwritten)
            memmove(pimpl->round.data(), pimpl->round.data() + cap, halo * sizeof(std::complex<float>));             // the next round's first halo
            pimpl->fill = 0;
            mcrx_hip_poll(pimpl->h);
            if (mcrx_hip_frames_pending(pimpl->h)) Deliver();
        }
        return;
    }
    int rc = mcrx_hip_execute_host(pimpl->h, reinterpret_cast<const float *>(_x), _num_samples);
    if (rc != MCRX_OK && rc != MCRX_EOVERFLOW) { fprintf(stderr, "error: multichannelrx::Execute(), %s\n", mcrx_hip_last_error()); throw 0; }
    if (rc == MCRX_EOVERFLOW)
        fprintf(stderr, "warning: multichannelrx::Execute(), frame pool exhausted, %llu frames dropped so far\n",
                (unsigned long long)mcrx_hip_frames_dropped(pimpl->h));
    if (mcrx_hip_frames_pending(pimpl->h)) Deliver();
}

void multichannelrx::ExecuteDevice(const void *_d_x, unsigned int _num_samples)
{
    std::lock_guard<std::recursive_mutex> lk(pimpl->mu);
    int rc = mcrx_hip_execute_device(pimpl->h, _d_x, _num_samples, NULL);
    if (rc != MCRX_OK) { fprintf(stderr, "error: multichannelrx::ExecuteDevice(), %s\n", mcrx_hip_last_error()); throw 0; }
}

void multichannelrx::Flush()
{
    std::lock_guard<std::recursive_mutex> lk(pimpl->mu);
    if (pimpl->pipe) mcrx_hip_pipeline_wait(pimpl->pipe);       // (samples of an unfinished round stay where they are: every rank needs the whole round)
    mcrx_hip_flush(pimpl->h);
    Deliver();
}
