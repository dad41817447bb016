// multichanneltxrx.h -- MI355X-native multichannel OFDM transceiver, source compatible with liquid-usrp's
// class of the same name (reference: include/multichanneltxrx.h:42-112 for the public interface;
// lib/multichanneltxrx.cc for the behaviour): one multichanneltx and one multichannelrx (both on the GPU),
// a transmit worker that streams GenerateSamples() x soft gain to the device in 256-sample packets while
// the transmitter is running, and a receive worker that pushes device packets through Execute().
//
// Differences visible to callers: receiver callbacks fire on the receive worker at flush points (buffer full
// and stop_rx()), not per sample.
#ifndef LIQUID_USRP_AMD_MULTICHANNELTXRX_H
#define LIQUID_USRP_AMD_MULTICHANNELTXRX_H

#include <complex>
#include <liquid/liquid.h>
#include <uhd/usrp/multi_usrp.hpp>

#include "multichanneltx.h"
#include "multichannelrx.h"

class multichanneltxrx {
public:
    // num_channels > 0, M >= 8, cp_len >= 1, taper_len <= cp_len (stderr + `throw 0` otherwise,
    // lib/multichanneltxrx.cc:64-77); per-channel callbacks / userdata as for multichannelrx
    multichanneltxrx(unsigned int _num_channels, unsigned int _M, unsigned int _cp_len, unsigned int _taper_len,
                     unsigned char *_p, framesync_callback *_callback, void **_userdata);
    ~multichanneltxrx();

    // transmitter methods
    void set_tx_freq(float _tx_freq);
    void set_tx_rate(float _tx_rate);
    void set_tx_gain_soft(float _tx_gain_soft);     // [dB]
    void set_tx_gain_uhd(float _tx_gain_uhd);
    void set_tx_antenna(char *_tx_antenna);
    void reset_tx();
    void start_tx();
    void stop_tx();
    // non-blocking; 0 on success, -1 (with a warning) when the channel still has a frame going out;
    // `throw 0` when the transmitter is not running or the channel does not exist
    int transmit_packet(unsigned int _channel, unsigned char *_header, unsigned char *_payload,
                        unsigned int _payload_len, int _mod, int _fec0, int _fec1);
    bool is_channel_available(unsigned int _channel);
    unsigned int get_available_channel();           // blocking
    void wait_for_channel(unsigned int _channel);   // blocking
    void wait_for_tx_to_complete();                 // blocking

    // receiver methods
    void set_rx_freq(float _rx_freq);
    void set_rx_rate(float _rx_rate);
    void set_rx_gain_uhd(float _rx_gain_uhd);
    void set_rx_antenna(char *_rx_antenna);
    void reset_rx();
    void start_rx();
    void stop_rx();

    void debug_enable();
    void debug_disable();

private:
    multichanneltxrx(const multichanneltxrx &);
    multichanneltxrx &operator=(const multichanneltxrx &);
    unsigned int num_channels;
    multichanneltx mctx;
    multichannelrx mcrx;
    struct impl;
    impl *pimpl;
};

#endif
