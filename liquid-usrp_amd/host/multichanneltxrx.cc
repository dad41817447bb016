// multichanneltxrx.cc -- worker threads around the GPU multichanneltx / multichannelrx classes.  Mirrors
// liquid-usrp's lib/multichanneltxrx.cc: ctor :53-121, transmitter methods :158-299, receiver methods
// :306-366, transmit worker :403-501, receive worker :541-624.
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <mutex>
#include <thread>
#include <vector>
#include <unistd.h>

#include "multichanneltxrx.h"

namespace {
// a worker that sleeps until started, runs its body until stopped, and can be waited on
struct worker {
    std::thread th; std::mutex mu; std::condition_variable cv;
    std::atomic<bool> running, alive; bool idle;
    worker() : running(false), alive(true), idle(false) {}
    template <class F> void launch(F body)
    {
        th = std::thread([this, body] {
            while (alive) {
                {
                    std::unique_lock<std::mutex> lk(mu);
                    idle = true; cv.notify_all();
                    cv.wait(lk, [this] { return running.load() || !alive.load(); });
                    idle = false;
                }
                if (running) body();
            }
        });
    }
    void start()
    {
        std::unique_lock<std::mutex> lk(mu);
        if (running) { cv.notify_all(); return; }   // already started: the reference only re-signals its condition (:320-333)
        cv.wait(lk, [this] { return idle; });
        running = true;
        cv.notify_all();
    }
    void stop()                                     // returns once the body has wound down
    {
        running = false;
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [this] { return idle; });
    }
    void join()
    {
        { std::lock_guard<std::mutex> lk(mu); alive = false; running = false; }
        cv.notify_all();
        th.join();
    }
};
}  // namespace

struct multichanneltxrx::impl {
    float tx_gain;
    bool debug_enabled;
    uhd::usrp::multi_usrp::sptr usrp_tx, usrp_rx;
    worker tx, rx;
};

multichanneltxrx::multichanneltxrx(unsigned int _num_channels, unsigned int _M, unsigned int _cp_len,
                                   unsigned int _taper_len, unsigned char *_p, framesync_callback *_callback,
                                   void **_userdata)
    : num_channels(_num_channels),
      mctx(_num_channels, _M, _cp_len, _taper_len, _p),
      mcrx(_num_channels, _M, _cp_len, _taper_len, _p, _userdata, _callback),
      pimpl(new impl)
{
    // (the member constructors above have already rejected bad arguments with their own messages, :64-77)
    pimpl->debug_enabled = false;
    uhd::device_addr_t dev_addr;
    pimpl->usrp_tx = uhd::usrp::multi_usrp::make(dev_addr);
    pimpl->usrp_rx = uhd::usrp::multi_usrp::make(dev_addr);
    set_tx_freq(462.0e6f); set_tx_rate(500e3); set_tx_gain_soft(-12.0f); set_tx_gain_uhd(40.0f);
    set_rx_freq(462.0e6f); set_rx_rate(500e3); set_rx_gain_uhd(20.0f);
    reset_tx();
    reset_rx();

    pimpl->rx.launch([this] {
        const size_t max_samps = pimpl->usrp_rx->get_device()->get_max_recv_samps_per_packet();
        std::vector<std::complex<float> > buffer(max_samps);
        uhd::rx_metadata_t md;
        while (pimpl->rx.running) {
            size_t n = pimpl->usrp_rx->get_device()->recv(&buffer.front(), buffer.size(), md,
                                                          uhd::io_type_t::COMPLEX_FLOAT32, uhd::device::RECV_MODE_ONE_PACKET);
            mcrx.Execute(&buffer.front(), (unsigned int)n);     // every sample, in order (:613)
        }
        mcrx.Flush();                                           // frames still inside the GPU batch
    });

    pimpl->tx.launch([this] {
        const unsigned int tx_buffer_len = 2 * num_channels;
        std::vector<std::complex<float> > tx_buffer(tx_buffer_len), usrp_buffer(256);
        unsigned int usrp_sample_counter = 0;
        uhd::tx_metadata_t md;
        md.start_of_burst = false; md.end_of_burst = false; md.has_time_spec = false;
        mctx.Reset();                                           // (:449)
        while (pimpl->tx.running) {
            mctx.GenerateSamples(&tx_buffer.front());
            for (unsigned int i = 0; i < tx_buffer_len; i++) {
                usrp_buffer[usrp_sample_counter++] = tx_buffer[i] * pimpl->tx_gain;
                if (usrp_sample_counter == 256) {
                    usrp_sample_counter = 0;
                    pimpl->usrp_tx->get_device()->send(&usrp_buffer.front(), usrp_buffer.size(), md,
                                                       uhd::io_type_t::COMPLEX_FLOAT32, uhd::device::SEND_MODE_FULL_BUFF);
                }
            }
        }
        // a few extra samples, then an end-of-burst packet (:472-487)
        pimpl->usrp_tx->get_device()->send(&usrp_buffer.front(), usrp_buffer.size(), md,
                                           uhd::io_type_t::COMPLEX_FLOAT32, uhd::device::SEND_MODE_FULL_BUFF);
        md.end_of_burst = true;
        pimpl->usrp_tx->get_device()->send("", 0, md, uhd::io_type_t::COMPLEX_FLOAT32, uhd::device::SEND_MODE_FULL_BUFF);
    });
}

multichanneltxrx::~multichanneltxrx()
{
    pimpl->rx.join();
    pimpl->tx.join();
    delete pimpl;
}

// ---- transmitter
void multichanneltxrx::set_tx_freq(float _tx_freq) { pimpl->usrp_tx->set_tx_freq(_tx_freq); }
void multichanneltxrx::set_tx_rate(float _tx_rate) { pimpl->usrp_tx->set_tx_rate(_tx_rate); }
void multichanneltxrx::set_tx_gain_soft(float _tx_gain_soft) { pimpl->tx_gain = powf(10.0f, _tx_gain_soft / 20.0f); }
void multichanneltxrx::set_tx_gain_uhd(float _tx_gain_uhd) { pimpl->usrp_tx->set_tx_gain(_tx_gain_uhd); }
void multichanneltxrx::set_tx_antenna(char *_tx_antenna) { pimpl->usrp_tx->set_tx_antenna(_tx_antenna); }
void multichanneltxrx::reset_tx() { mctx.Reset(); }
void multichanneltxrx::start_tx() { pimpl->tx.start(); }
void multichanneltxrx::stop_tx() { pimpl->tx.stop(); }

int multichanneltxrx::transmit_packet(unsigned int _channel, unsigned char *_header, unsigned char *_payload,
                                      unsigned int _payload_len, int _mod, int _fec0, int _fec1)
{
    if (!pimpl->tx.running) {
        fprintf(stderr, "error: multichanneltxrx:transmit_packet(), transmitter not yet running\n");
        throw 0;
    } else if (_channel >= num_channels) {
        fprintf(stderr, "error: multichanneltxrx:transmit_packet(), invalid channel %u\n", _channel);
        throw 0;
    } else if (!mctx.IsChannelReadyForData(_channel)) {
        fprintf(stderr, "warning: multichanneltxrx:transmit_packet(), channel %u not ready for data\n", _channel);
        return -1;
    }
    mctx.UpdateData(_channel, _header, _payload, _payload_len, _mod, _fec0, _fec1);
    return 0;
}

bool multichanneltxrx::is_channel_available(unsigned int _channel) { return mctx.IsChannelReadyForData(_channel) != 0; }

unsigned int multichanneltxrx::get_available_channel()
{
    while (true) {                                  // poll, like the reference (:249-267)
        for (unsigned int i = 0; i < num_channels; i++)
            if (mctx.IsChannelReadyForData(i)) return i;
        usleep(500);
    }
}

void multichanneltxrx::wait_for_channel(unsigned int _channel)
{
    while (!mctx.IsChannelReadyForData(_channel)) usleep(100);
}

void multichanneltxrx::wait_for_tx_to_complete()
{
    while (true) {
        bool all_available = true;
        for (unsigned int i = 0; i < num_channels; i++)
            if (!mctx.IsChannelReadyForData(i)) all_available = false;
        if (all_available) return;
        usleep(100);
    }
}

// ---- receiver
void multichanneltxrx::set_rx_freq(float _rx_freq) { pimpl->usrp_rx->set_rx_freq(_rx_freq); }
void multichanneltxrx::set_rx_rate(float _rx_rate) { pimpl->usrp_rx->set_rx_rate(_rx_rate); }
void multichanneltxrx::set_rx_gain_uhd(float _rx_gain_uhd) { pimpl->usrp_rx->set_rx_gain(_rx_gain_uhd); }
void multichanneltxrx::set_rx_antenna(char *_rx_antenna) { pimpl->usrp_rx->set_rx_antenna(_rx_antenna); }
void multichanneltxrx::reset_rx() { mcrx.Reset(); }

void multichanneltxrx::start_rx()
{
    pimpl->usrp_rx->issue_stream_cmd(uhd::stream_cmd_t::STREAM_MODE_START_CONTINUOUS);
    pimpl->rx.start();
}

void multichanneltxrx::stop_rx()
{
    pimpl->rx.stop();
    pimpl->usrp_rx->issue_stream_cmd(uhd::stream_cmd_t::STREAM_MODE_STOP_CONTINUOUS);
}

void multichanneltxrx::debug_enable() { pimpl->debug_enabled = true; }
void multichanneltxrx::debug_disable() { pimpl->debug_enabled = false; }
