// multichannelrx.cc -- host class over the C-ABI of libmcrx_hip.so (include/mcrx_hip.h).
// Mirrors liquid-usrp's lib/multichannelrx.cc: ctor :45-104, dtor :107-132, Reset :135-153,
// Execute :155-182; the DSP itself runs in the gfx950 kernels.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

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
};

multichannelrx::multichannelrx(unsigned int _num_channels, unsigned int _M, unsigned int _cp_len,
                               unsigned int _taper_len, unsigned char *_p, void **_userdata,
                               framesync_callback *_callback)
    : num_channels(_num_channels), pimpl(new impl)
{
    pimpl->h = NULL;
    pimpl->debug_dir = getenv("MCRX_DEBUG_DIR");
    pimpl->debug_syms.resize(_num_channels); pimpl->debug_frames.assign(_num_channels, 0);
    mcrx_hip_config cfg;
    memset(&cfg, 0, sizeof(cfg));
    cfg.struct_size = sizeof(cfg);
    cfg.payload_soft = 1;
    int rc = mcrx_hip_create(&pimpl->h, _num_channels, _M, _cp_len, _taper_len, _p, &cfg);
    if (rc != MCRX_OK) {
        fprintf(stderr, "%s\n", mcrx_hip_last_error());
        delete pimpl;
        throw 0;
    }
    for (unsigned int i = 0; i < _num_channels; i++) {
        pimpl->userdata.push_back(_userdata ? _userdata[i] : NULL);
        pimpl->callback.push_back(_callback ? _callback[i] : NULL);
    }
}

multichannelrx::~multichannelrx()
{
    if (pimpl->h) {
        mcrx_hip_flush(pimpl->h);
        Deliver();
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
    mcrx_hip_flush(pimpl->h);
    Deliver();
}
