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
#include <sys/stat.h>
#include <time.h>

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
    //   MCRX_UID_FILE            where rank 0 leaves the 128-byte ncclUniqueId for the others (world > 1); rank 0 removes what it finds
    //                            there first and removes its own file again when the object goes
    //   MCRX_JOB_ID              (optional) a string unique to this job, the same on every rank: it is appended to the file name and
    //                            stored in the file, so that a rank can never pick up the id an earlier job left behind
    //   MCRX_SUB_BLOCKS          blocks of 2N samples per rank and round (default 8192)
    // Every rank is handed the whole wideband stream, like every process behind a shared radio would be; of each round of
    // world x sub_blocks blocks it channelizes sub-slab number `rank`, one exchange turns the time shards into channel shards, and
    // this object's callbacks fire for its shard of num_channels / world channels only (mcrx_hip_pipeline_*, csrc/pipeline.hip).
    mcrx_hip_pipeline_t pipe;
    int rank, world;
    size_t sub_blocks, K, fill;                                     // fill: samples of the current round seen so far
    // Of every round only this rank's part is kept: the 13 blocks in front of its sub-slab and the sub-slab, copied as they arrive
    // into the pinned staging buffer the pipeline hands out (mcrx_hip_pipeline_host_buffer) -- (13 + sub_blocks) x 2N samples, not
    // the round's world x sub_blocks x 2N (0.5 GB per process at 512 channels and 8 ranks until round 5).  Rank 0's halo is the tail
    // of the previous round: the last 13 blocks of every round are remembered in `tail`.
    float *mine;                                                    // pinned buffer of the round being collected (NULL: not asked for yet)
    std::vector<std::complex<float> > tail;                         // [13 blocks]: the end of the round being collected = rank 0's next halo
    std::string uid_path;                                           // rank 0: the file it wrote (removed at destruction)
    // Where the stream stops (Reset, destruction) the unfinished round is completed with zeros: a frame whose last sample would lie
    // in that padding never ended in the reference's stream (ofdmflexframesync_reset drops it, lib/multichannelrx.cc:139-140; its
    // destructor synchronizes nothing further) and is not delivered here either.  chan_pos: channel-rate samples
    // (= blocks) pushed since the object was made -- what mcrx_frame::end_sample counts; drop_from: the first padded one.
    uint64_t chan_pos = 0, drop_from = ~0ull;
    void take(const std::complex<float> *x, size_t n);              // sharded Execute: n samples at position `fill` of the round
    void push_round(multichannelrx *self);
    size_t finish_round(multichannelrx *self);                      // zero the trailing partial block, pad the round with zeros, push it
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
    pimpl->h = NULL; pimpl->pipe = NULL; pimpl->mine = NULL;
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
    // MCRX_FRONT_END=1 (round 6): the oversampled analysis bank BASELINE.json's north_star names (liquid's firpfbch2, 2N channels, + a half-band
    // decimator per kept channel, folded into one kernel: include/mcrx_hip.h, front_end) in place of the reference's critically sampled
    // firpfbch (lib/multichannelrx.cc:89-91) -- the class interface has no argument for it, so the unchanged applications reach it
    // through the environment; not with MCRX_WORLD (the sharded class keeps 13 blocks of history per round)
    cfg.front_end = (uint32_t)env_int("MCRX_FRONT_END", NULL, 0);
    const int W = pimpl->world;
    if (W > 0 && cfg.front_end) { fprintf(stderr, "error: multichannelrx, MCRX_FRONT_END and MCRX_WORLD cannot be combined\n"); delete pimpl; throw 0; }
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
        const char *uf = getenv("MCRX_UID_FILE"), *job = getenv("MCRX_JOB_ID");
        bool ok = true;
        if (W > 1) {
            // file = 128 bytes of ncclUniqueId + the job string (may be empty).  ADVICE r4: a second job on the same path used to read the
            // first job's id before rank 0 had replaced it.  Now rank 0 removes whatever is there BEFORE it makes its id and removes its own
            // file when the object goes; with MCRX_JOB_ID the name and the content are per job, which closes the window in which a reader
            // that starts before rank 0 could still see the old file.
            ok = uf != NULL;
            std::string path = ok ? std::string(uf) : std::string(), tag = job ? std::string(job) : std::string();
            if (job) path += "." + tag;
            if (ok && pimpl->rank == 0) {           // rank 0 makes the id and leaves it where the others look (written aside, then renamed)
                unlink(path.c_str());
                ok = mcrx_hip_pipeline_unique_id(uid) == MCRX_OK;
                std::string tmp = path + ".tmp";
                FILE *f = ok ? fopen(tmp.c_str(), "wb") : NULL;
                ok = f && fwrite(uid, 1, 128, f) == 128 && fwrite(tag.data(), 1, tag.size(), f) == tag.size();
                if (f) fclose(f);
                ok = ok && rename(tmp.c_str(), path.c_str()) == 0;
                if (ok) pimpl->uid_path = path;
            } else if (ok) {
                ok = false;
                // (without MCRX_JOB_ID the tag comparison below is '' == '', and a rank that starts before rank 0 could still
                //  read a dead job's id before rank 0 unlinks it -- and hang in ncclCommInitRank.  A file written more than two minutes
                //  before this rank started is not this job's: ranks of one job start together; jobs that cannot promise that set MCRX_JOB_ID.)
                const time_t t_start = time(NULL);
                for (int t = 0; t < 600 && !ok; t++) {      // up to a minute
                    struct stat sb;
                    const bool fresh = job != NULL || (stat(path.c_str(), &sb) == 0 && sb.st_mtime + 120 >= t_start);
                    FILE *f = fresh ? fopen(path.c_str(), "rb") : NULL;
                    if (f) {
                        char got[256]; memset(got, 0, sizeof(got));
                        ok = fread(uid, 1, 128, f) == 128;
                        const size_t nt = ok ? fread(got, 1, sizeof(got) - 1, f) : 0;
                        ok = ok && std::string(got, nt) == tag;
                        fclose(f);
                    }
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
        pimpl->tail.assign(13 * pimpl->K, std::complex<float>(0.f, 0.f));
    }
    for (unsigned int i = 0; i < _num_channels; i++) {
        pimpl->userdata.push_back(_userdata ? _userdata[i] : NULL);
        pimpl->callback.push_back(_callback ? _callback[i] : NULL);
    }
}

multichannelrx::~multichannelrx()
{
    if (pimpl->h) {
        if (pimpl->pipe) {
            // The stream ends here.  The reference has synchronized every sample it was given (lib/multichannelrx.cc:185-195); a sharded
            // round needs every rank's part, so the unfinished round is padded with zeros and pushed -- by every rank: they all reach
            // their destructor at the same point of the stream -- and what it held is delivered below.  (Flush() cannot do this: padding
            // in the middle of a stream would move the block alignment of everything behind it.)
            pimpl->drop_from = pimpl->chan_pos + pimpl->fill / pimpl->K;
            try { pimpl->finish_round(this); } catch (...) { }
            mcrx_hip_pipeline_wait(pimpl->pipe);
        }
        mcrx_hip_flush(pimpl->h);
        Deliver();
        pimpl->drop_from = ~0ull;
        if (pimpl->pipe) mcrx_hip_pipeline_destroy(pimpl->pipe);
        mcrx_hip_destroy(pimpl->h);
    }
    if (!pimpl->uid_path.empty()) unlink(pimpl->uid_path.c_str());
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
        if (f.end_sample >= pimpl->drop_from) continue;            // completed on the zeros behind the end of the stream
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
    if (pimpl->pipe) {
        // sharded: the pipeline's round counter and the halo are part of the state a Reset clears.  The reference has
        // synchronized every whole block pushed before the Reset and drops the partial one (lib/multichannelrx.cc:152,167-174); here the
        // unfinished round is completed with zeros and pushed first -- block alignment restarts behind a Reset anyway, so the padding is
        // invisible -- and the oscillator is told that the padding was never in the stream (it is not reset, :144, and the samples of the
        // dropped partial block still count for it).  Every rank gets the same call at the same point of the stream.
        pimpl->drop_from = pimpl->chan_pos + pimpl->fill / pimpl->K;                 // (frames that end in the padding are not delivered)
        const size_t real = pimpl->fill, pushed = pimpl->finish_round(this);         // pushed: 0 (nothing but a partial block) or a whole round
        if (mcrx_hip_pipeline_reset(pimpl->pipe, (long long)real - (long long)pushed) != MCRX_OK) {
            fprintf(stderr, "error: multichannelrx::Reset(), %s\n", mcrx_hip_pipeline_last_error());
            throw 0;
        }
        pimpl->fill = 0;
        std::fill(pimpl->tail.begin(), pimpl->tail.end(), std::complex<float>(0.f, 0.f));
        if (pimpl->mine && pimpl->rank == 0) memset(pimpl->mine, 0, 13 * pimpl->K * sizeof(std::complex<float>));      // (a buffer taken but not pushed: its halo is the old tail)
        Deliver();
        pimpl->drop_from = ~0ull;
        return;
    }
    mcrx_hip_reset(pimpl->h);
    Deliver();
}

// n samples at position `fill` of the round: keep what falls into this rank's part, [rank * T - 13 blocks, (rank + 1) * T) with
// T = sub_blocks * 2N (rank 0: the 13 blocks come from the previous round's tail), and into the round's last 13 blocks
void multichannelrx::impl::take(const std::complex<float> *x, size_t n)
{
    const size_t halo = 13 * K, T = sub_blocks * K, cap = (size_t)world * T;
    if (!mine) {
        size_t ns = 0;
        if (mcrx_hip_pipeline_host_buffer(pipe, &mine, &ns) != MCRX_OK || ns != halo + T) {
            fprintf(stderr, "error: multichannelrx::Execute(), %s\n", mcrx_hip_pipeline_last_error());
            throw 0;
        }
        if (rank == 0) memcpy(mine, tail.data(), halo * sizeof(std::complex<float>));
    }
    std::complex<float> *dst = reinterpret_cast<std::complex<float> *>(mine);
    const size_t a = fill, b = fill + n;                                    // [a, b) of the round
    {   // this rank's part: round positions [lo, hi) -> dst[pos - lo + off]
        const size_t lo = rank == 0 ? 0 : (size_t)rank * T - halo, hi = ((size_t)rank + 1) * T, off = rank == 0 ? halo : 0;
        const size_t s = std::max(a, lo), e = std::min(b, hi);
        if (s < e) memcpy(dst + (s - lo) + off, x + (s - a), (e - s) * sizeof(std::complex<float>));
    }
    {   // the round's tail
        const size_t lo = cap - halo, s = std::max(a, lo), e = std::min(b, cap);
        if (s < e) memcpy(tail.data() + (s - lo), x + (s - a), (e - s) * sizeof(std::complex<float>));
    }
    fill = b;
}

// The stream stops here (Reset, destruction): whole blocks collected so far are synchronized like the reference's, the partial block
// behind them is dropped (zeroed), the rest of the round is zeros.  Returns the samples pushed (0: the round was empty).
size_t multichannelrx::impl::finish_round(multichannelrx *self)
{
    if (!fill) return 0;
    const size_t cap = (size_t)world * sub_blocks * K;
    fill -= fill % K;                                                       // re-taking from here overwrites the partial block
    if (!fill) return 0;
    std::vector<std::complex<float> > zeros(std::min<size_t>(cap - fill, (size_t)1 << 20), std::complex<float>(0.f, 0.f));
    while (fill < cap) take(zeros.data(), std::min(zeros.size(), cap - fill));
    push_round(self);
    return cap;
}

void multichannelrx::impl::push_round(multichannelrx *self)
{
    if (mcrx_hip_pipeline_push_host(pipe, mine) != MCRX_OK) {               // returns once the round is enqueued: nothing waits for its kernels
        fprintf(stderr, "error: multichannelrx::Execute(), %s\n", mcrx_hip_pipeline_last_error());
        throw 0;
    }
    mine = NULL; fill = 0;
    chan_pos += (uint64_t)world * sub_blocks;
    const int rc = mcrx_hip_poll(h);                                        // the frames of the rounds before
    if (rc == MCRX_EOVERFLOW)
        fprintf(stderr, "warning: multichannelrx::Execute(), frame pool exhausted, %llu frames dropped so far\n",
                (unsigned long long)mcrx_hip_frames_dropped(h));
    if (mcrx_hip_frames_pending(h)) self->Deliver();
}

void multichannelrx::Execute(std::complex<float> *_x, unsigned int _num_samples)
{
    std::lock_guard<std::recursive_mutex> lk(pimpl->mu);
    if (pimpl->pipe) {
        // sharded: of every round of world x sub_blocks blocks this rank keeps its own sub-slab and the 13 blocks in front of it, in a
        // pinned buffer of the pipeline; a full round is pushed without waiting for the one before (the buffers rotate) and the frames
        // of the rounds before come back through the poll
        const size_t cap = (size_t)pimpl->world * pimpl->sub_blocks * pimpl->K;
        size_t done = 0;
        while (done < _num_samples) {
            const size_t n = std::min<size_t>(_num_samples - done, cap - pimpl->fill);
            pimpl->take(_x + done, n);
            done += n;
            if (pimpl->fill == cap) pimpl->push_round(this);
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
