// ofdmtxrx.cc -- host class over the C-ABI of libmcrx_hip.so.  Mirrors liquid-usrp's lib/ofdmtxrx.cc:
// ctor :52-130 (defaults: tx/rx 462 MHz, 500 kHz, soft gain -12 dB, uhd gains 40/20 dB), dtor :208-252,
// transmitter methods :259-478, receiver methods :485-535, receiver worker :553-639.
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>

#include "ofdmtxrx.h"
#include "mcrx_hip.h"

struct ofdmtxrx::impl {
    unsigned int M, cp_len, taper_len;
    mctx_hip_t fg;                                  // frame generator
    mcrx_hip_t fs;                                  // frame synchronizer (single_channel handle)
    std::recursive_mutex fs_mu;                               // the C-ABI allows one caller per handle: worker (execute/deliver/flush) vs reset_rx()
    framesync_callback callback; void *userdata;
    int mod, fec0, fec1;
    float tx_gain;
    std::vector<std::complex<float> > frame;        // samples of the assembled frame
    size_t frame_pos; bool assembled;
    uhd::usrp::multi_usrp::sptr usrp_tx, usrp_rx;
    uhd::tx_metadata_t metadata_tx;
    // receiver thread: sleeps until start_rx(), runs until stop_rx(), exits at destruction
    std::thread rx_thread; std::mutex rx_mutex; std::condition_variable rx_cond;
    std::atomic<bool> rx_running, rx_thread_running;
    bool rx_idle;
    bool debug_enabled;
    bool blocking;
    ofdmtxrx *self;
    std::vector<std::vector<std::complex<float> > > debug_syms;      // equalised symbols of the frames seen while debug_enable()d

    void deliver()
    {
        mcrx_frame f;
        while (mcrx_hip_next_frame(fs, &f) == 1) {
            if (!callback) continue;
            framesyncstats_s st;
            st.evm = f.evm; st.rssi = f.rssi; st.cfo = f.cfo;
            st.framesyms = reinterpret_cast<liquid_float_complex *>(const_cast<float *>(f.framesyms));
            st.num_framesyms = f.num_framesyms;
            st.mod_scheme = f.mod_scheme; st.mod_bps = f.mod_bps; st.check = f.check; st.fec0 = f.fec0; st.fec1 = f.fec1;
            unsigned char header[8];
            memcpy(header, f.header, 8);
            if (debug_enabled && f.num_framesyms && debug_syms.size() < 4096)
                debug_syms.emplace_back(reinterpret_cast<const std::complex<float> *>(f.framesyms),
                                        reinterpret_cast<const std::complex<float> *>(f.framesyms) + f.num_framesyms);
            std::vector<unsigned char> payload(f.payload, f.payload + f.payload_len);
            callback(header, f.header_valid, payload.empty() ? NULL : &payload[0], f.payload_len, f.payload_valid, st, userdata);
        }
    }

    void rx_worker()
    {
        const size_t max_samps = usrp_rx->get_device()->get_max_recv_samps_per_packet();
        std::vector<std::complex<float> > buffer(max_samps);
        uhd::rx_metadata_t md;
        while (rx_thread_running) {
            {
                std::unique_lock<std::mutex> lk(rx_mutex);
                rx_idle = true; rx_cond.notify_all();
                rx_cond.wait(lk, [this] { return rx_running.load() || !rx_thread_running.load(); });
                rx_idle = false;
            }
            while (rx_running) {
                size_t n;
                if (blocking) {
                    // lib/ofdmtxrx.cc:686-722: fill the public buffer under its mutex, tell the editor, wait until it is done
                    pthread_mutex_lock(&self->rx_buffer_mutex);
                    self->rx_buffer = &buffer;
                    n = usrp_rx->get_device()->recv(&buffer.front(), buffer.size(), md,
                                                    uhd::io_type_t::COMPLEX_FLOAT32, uhd::device::RECV_MODE_ONE_PACKET);
                    pthread_cond_signal(&self->rx_buffer_filled_cond);
                    if (rx_running) pthread_cond_wait(&self->rx_buffer_modified_cond, &self->rx_buffer_mutex);
                } else
                n = usrp_rx->get_device()->recv(&buffer.front(), buffer.size(), md,
                                                       uhd::io_type_t::COMPLEX_FLOAT32, uhd::device::RECV_MODE_ONE_PACKET);
                // the synchronizer sees every sample in order (lib/ofdmtxrx.cc:620-626); frames surface per batch
                std::lock_guard<std::recursive_mutex> fl(fs_mu);
                int rc = mcrx_hip_execute_host(fs, reinterpret_cast<const float *>(&buffer.front()), n);
                if (rc != MCRX_OK && rc != MCRX_EOVERFLOW) { fprintf(stderr, "ofdmtxrx rx worker: %s\n", mcrx_hip_last_error()); rx_running = false; }
                deliver();
                if (blocking) pthread_mutex_unlock(&self->rx_buffer_mutex);
            }
            std::lock_guard<std::recursive_mutex> fl(fs_mu);
            mcrx_hip_flush(fs);
            deliver();
        }
    }
};

void ofdmtxrx::init(unsigned int _M, unsigned int _cp_len, unsigned int _taper_len, unsigned char *_p,
                    framesync_callback _callback, void *_userdata, bool _blocking)
{
    if (_M < 8) { fprintf(stderr, "error: ofdmtxrx::ofdmtxrx(), number of subcarriers must be at least 8\n"); throw 0; }
    if (_cp_len < 1) { fprintf(stderr, "error: ofdmtxrx::ofdmtxrx(), cyclic prefix length must be at least 1\n"); throw 0; }
    if (_taper_len > _cp_len) { fprintf(stderr, "error: ofdmtxrx::ofdmtxrx(), taper length cannot exceed cyclic prefix length\n"); throw 0; }
    pimpl = new impl;
    pimpl->M = _M; pimpl->cp_len = _cp_len; pimpl->taper_len = _taper_len;
    pimpl->fg = NULL; pimpl->fs = NULL; pimpl->blocking = _blocking;
    pimpl->callback = _callback; pimpl->userdata = _userdata;
    pimpl->mod = LIQUID_MODEM_QPSK; pimpl->fec0 = LIQUID_FEC_NONE; pimpl->fec1 = LIQUID_FEC_HAMMING128;    // :80-83
    pimpl->frame_pos = 0; pimpl->assembled = false; pimpl->debug_enabled = false; pimpl->rx_idle = false;
    pimpl->self = this; rx_buffer = NULL;
    pthread_mutex_init(&rx_buffer_mutex, NULL); pthread_cond_init(&rx_buffer_filled_cond, NULL); pthread_cond_init(&rx_buffer_modified_cond, NULL);
    fgbuffer_len = _M + _cp_len;
    fgbuffer = new std::complex<float>[fgbuffer_len]();
    // like the reference (:78), both objects use the default subcarrier allocation whatever _p says
    (void)_p;
    mcrx_hip_config cfg;
    memset(&cfg, 0, sizeof(cfg));
    cfg.struct_size = sizeof(cfg); cfg.payload_soft = 1; cfg.single_channel = 1; cfg.batch_samples = 1u << 16;
    if (mctx_hip_create(&pimpl->fg, 1, _M, _cp_len, _taper_len, NULL) != MCRX_OK) {
        fprintf(stderr, "%s\n", mctx_hip_last_error());
        delete[] fgbuffer; delete pimpl; throw 0;
    }
    if (mcrx_hip_create(&pimpl->fs, 1, _M, _cp_len, _taper_len, NULL, &cfg) != MCRX_OK) {
        fprintf(stderr, "%s\n", mcrx_hip_last_error());
        mctx_hip_destroy(pimpl->fg); delete[] fgbuffer; delete pimpl; throw 0;
    }
    uhd::device_addr_t dev_addr;
    pimpl->usrp_tx = uhd::usrp::multi_usrp::make(dev_addr);
    pimpl->usrp_rx = uhd::usrp::multi_usrp::make(dev_addr);
    set_tx_freq(462.0e6f); set_tx_rate(500e3); set_tx_gain_soft(-12.0f); set_tx_gain_uhd(40.0f);
    set_rx_freq(462.0e6f); set_rx_rate(500e3); set_rx_gain_uhd(20.0f);
    reset_tx();
    reset_rx();
    pimpl->rx_running = false;
    pimpl->rx_thread_running = true;
    pimpl->rx_thread = std::thread(&impl::rx_worker, pimpl);
}

ofdmtxrx::ofdmtxrx(unsigned int _M, unsigned int _cp_len, unsigned int _taper_len, unsigned char *_p,
                   framesync_callback _callback, void *_userdata)
{ init(_M, _cp_len, _taper_len, _p, _callback, _userdata, false); }

// second constructor: chooses between ofdmtxrx_rx_worker() and ofdmtxrx_rx_worker_blocking() (lib/ofdmtxrx.cc:133-206)
ofdmtxrx::ofdmtxrx(unsigned int _M, unsigned int _cp_len, unsigned int _taper_len, unsigned char *_p,
                   framesync_callback _callback, void *_userdata, bool _blocking_rx_worker)
{ init(_M, _cp_len, _taper_len, _p, _callback, _userdata, _blocking_rx_worker); }

ofdmtxrx::~ofdmtxrx()
{
    if (pimpl->rx_running) stop_rx();
    {
        std::lock_guard<std::mutex> lk(pimpl->rx_mutex);
        pimpl->rx_thread_running = false;
    }
    pimpl->rx_cond.notify_all();
    pimpl->rx_thread.join();
    if (const char *dir = getenv("MCRX_DEBUG_DIR")) {
        // stand-in for ofdmflexframesync_debug_print (lib/ofdmtxrx.cc:241-242): what this receiver can show of its inside
        if (!pimpl->debug_syms.empty()) {
            std::string fn = std::string(dir) + "/ofdmtxrx_framesyms.m";
            if (FILE *f = fopen(fn.c_str(), "w")) {
                fprintf(f, "%% equalised payload symbols of %zu frames (ofdmtxrx, debug_enable())\nclear all; close all;\n", pimpl->debug_syms.size());
                for (size_t k = 0; k < pimpl->debug_syms.size(); k++) {
                    fprintf(f, "framesyms{%zu} = [", k + 1);
                    for (const auto &v : pimpl->debug_syms[k]) fprintf(f, " %.6e%+.6ej", v.real(), v.imag());
                    fprintf(f, " ];\n");
                }
                fprintf(f, "figure; plot(real([framesyms{:}]), imag([framesyms{:}]), 'x'); axis square; grid on;\n");
                fclose(f);
            }
        }
    }
    mctx_hip_destroy(pimpl->fg);
    mcrx_hip_destroy(pimpl->fs);
    pthread_mutex_destroy(&rx_buffer_mutex); pthread_cond_destroy(&rx_buffer_filled_cond); pthread_cond_destroy(&rx_buffer_modified_cond);
    delete[] fgbuffer;
    delete pimpl;
}

// ---- transmitter
void ofdmtxrx::set_tx_freq(float _tx_freq) { pimpl->usrp_tx->set_tx_freq(_tx_freq); }
void ofdmtxrx::set_tx_rate(float _tx_rate) { pimpl->usrp_tx->set_tx_rate(_tx_rate); }
void ofdmtxrx::set_tx_gain_soft(float _tx_gain_soft) { pimpl->tx_gain = powf(10.0f, _tx_gain_soft / 20.0f); }
void ofdmtxrx::set_tx_gain_uhd(float _tx_gain_uhd) { pimpl->usrp_tx->set_tx_gain(_tx_gain_uhd); }
void ofdmtxrx::set_tx_antenna(char *_tx_antenna) { pimpl->usrp_tx->set_tx_antenna(_tx_antenna); }

void ofdmtxrx::reset_tx()
{
    pimpl->assembled = false; pimpl->frame_pos = 0; pimpl->frame.clear();
}

void ofdmtxrx::assemble_frame(unsigned char *_header, unsigned char *_payload, unsigned int _payload_len,
                              int _mod, int _fec0, int _fec1)
{
    pimpl->mod = _mod; pimpl->fec0 = _fec0; pimpl->fec1 = _fec1;
    size_t n = mctx_hip_frame_len(pimpl->fg, _payload_len, _mod, _fec0, _fec1);
    if (n == 0) { fprintf(stderr, "error: ofdmtxrx::assemble_frame(), unsupported frame properties\n"); throw 0; }
    pimpl->frame.resize(n);
    if (mctx_hip_frame(pimpl->fg, _header, _payload, _payload_len, _mod, _fec0, _fec1, 1.0f,
                       reinterpret_cast<float *>(&pimpl->frame[0]), n) != MCRX_OK) {
        fprintf(stderr, "ofdmtxrx::assemble_frame: %s\n", mctx_hip_last_error());
        throw 0;
    }
    pimpl->frame_pos = 0; pimpl->assembled = true;
}

bool ofdmtxrx::write_symbol()
{
    if (!pimpl->assembled) {                        // an idle generator writes zeros
        memset((void *)fgbuffer, 0, fgbuffer_len * sizeof(std::complex<float>));
        return false;
    }
    memcpy((void *)fgbuffer, &pimpl->frame[pimpl->frame_pos], fgbuffer_len * sizeof(std::complex<float>));
    pimpl->frame_pos += fgbuffer_len;
    if (pimpl->frame_pos >= pimpl->frame.size()) { pimpl->assembled = false; return true; }
    return false;
}

void ofdmtxrx::send_buffer()
{
    std::vector<std::complex<float> > usrp_buffer(fgbuffer_len);
    for (unsigned int i = 0; i < fgbuffer_len; i++) usrp_buffer[i] = fgbuffer[i] * pimpl->tx_gain;
    pimpl->usrp_tx->get_device()->send(&usrp_buffer.front(), usrp_buffer.size(), pimpl->metadata_tx,
                                       uhd::io_type_t::COMPLEX_FLOAT32, uhd::device::SEND_MODE_FULL_BUFF);
}

void ofdmtxrx::transmit_symbol() { send_buffer(); }

void ofdmtxrx::end_transmit_frame()
{
    send_buffer();                                  // "a few extra samples" (:462-470): the last symbol buffer again
    pimpl->metadata_tx.start_of_burst = false;
    pimpl->metadata_tx.end_of_burst = true;
    pimpl->usrp_tx->get_device()->send("", 0, pimpl->metadata_tx, uhd::io_type_t::COMPLEX_FLOAT32, uhd::device::SEND_MODE_FULL_BUFF);
}

void ofdmtxrx::transmit_packet(unsigned char *_header, unsigned char *_payload, unsigned int _payload_len,
                               int _mod, int _fec0, int _fec1)
{
    pimpl->metadata_tx.start_of_burst = false;
    pimpl->metadata_tx.end_of_burst = false;
    pimpl->metadata_tx.has_time_spec = false;
    assemble_frame(_header, _payload, _payload_len, _mod, _fec0, _fec1);
    bool last_symbol = false;
    while (!last_symbol) {
        last_symbol = write_symbol();
        send_buffer();
    }
    end_transmit_frame();
}

// ---- receiver
void ofdmtxrx::set_rx_freq(float _rx_freq) { pimpl->usrp_rx->set_rx_freq(_rx_freq); }
void ofdmtxrx::set_rx_rate(float _rx_rate) { pimpl->usrp_rx->set_rx_rate(_rx_rate); }
void ofdmtxrx::set_rx_gain_uhd(float _rx_gain_uhd) { pimpl->usrp_rx->set_rx_gain(_rx_gain_uhd); }
void ofdmtxrx::set_rx_antenna(char *_rx_antenna) { pimpl->usrp_rx->set_rx_antenna(_rx_antenna); }

void ofdmtxrx::reset_rx()
{
    // the worker may be inside execute/deliver on the same handle (the reference's race only resets state; here
    // a harvest reallocates host buffers): serialise the two
    std::lock_guard<std::recursive_mutex> fl(pimpl->fs_mu);
    if (mcrx_hip_reset(pimpl->fs) != MCRX_OK) { fprintf(stderr, "ofdmtxrx::reset_rx: %s\n", mcrx_hip_last_error()); throw 0; }
}

void ofdmtxrx::start_rx()
{
    {
        std::unique_lock<std::mutex> lk(pimpl->rx_mutex);
        pimpl->rx_cond.wait(lk, [this] { return pimpl->rx_idle; });     // the worker is parked
        pimpl->rx_running = true;
    }
    pimpl->usrp_rx->issue_stream_cmd(uhd::stream_cmd_t::STREAM_MODE_START_CONTINUOUS);
    pimpl->rx_cond.notify_all();
}

void ofdmtxrx::stop_rx()
{
    pimpl->rx_running = false;
    if (pimpl->blocking) {                          // a worker waiting for the editor must not wait for ever
        pthread_mutex_lock(&rx_buffer_mutex); pthread_cond_broadcast(&rx_buffer_modified_cond); pthread_mutex_unlock(&rx_buffer_mutex);
    }
    pimpl->usrp_rx->issue_stream_cmd(uhd::stream_cmd_t::STREAM_MODE_STOP_CONTINUOUS);
    std::unique_lock<std::mutex> lk(pimpl->rx_mutex);                   // returns once the worker has flushed and parked
    pimpl->rx_cond.wait(lk, [this] { return pimpl->rx_idle; });
}

void ofdmtxrx::debug_enable() { pimpl->debug_enabled = true; }
void ofdmtxrx::debug_disable() { pimpl->debug_enabled = false; }
