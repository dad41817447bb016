// ofdmtxrx.h -- MI355X-native single-channel OFDM transceiver, source compatible with liquid-usrp's class
// of the same name (reference: include/ofdmtxrx.h:43-121 for the public interface; lib/ofdmtxrx.cc for the
// behaviour).  The frame generator and the frame synchronizer run on the GPU (mctx_hip_frame and a
// single_channel mcrx handle, include/mcrx_hip.h); the radio is whatever <uhd/usrp/multi_usrp.hpp> resolves to
// (in this repository: the synthetic-IQ stand-in).
//
// Differences visible to callers, forced by batching on the GPU:
//   * callbacks fire on the receiver thread at flush points (every rx batch and in stop_rx()), not per sample;
//   * a frame's samples are produced by one GPU call in assemble_frame()/transmit_packet(); write_symbol()
//     hands them out one M+cp symbol at a time exactly as ofdmflexframegen_writesymbol would;
//   * debug_enable(): no liquid-internal dump; with $MCRX_DEBUG_DIR set, the destructor writes the equalised symbols of
//     the frames received while debugging was on to $MCRX_DEBUG_DIR/ofdmtxrx_framesyms.m (a stand-in for
//     ofdmflexframesync_debug_print, lib/ofdmtxrx.cc:241-242).
// The "blocking" receiver worker (lib/ofdmtxrx.cc:642-739; second constructor, _blocking_rx_worker = true) is provided with
// the reference's public handshake: the worker fills *rx_buffer under rx_buffer_mutex, signals rx_buffer_filled_cond and
// waits on rx_buffer_modified_cond; another thread edits the samples and signals that; then they are synchronized.
#ifndef LIQUID_USRP_AMD_OFDMTXRX_H
#define LIQUID_USRP_AMD_OFDMTXRX_H

#include <complex>
#include <vector>
#include <pthread.h>
#include <liquid/liquid.h>
#include <uhd/usrp/multi_usrp.hpp>

class ofdmtxrx {
public:
    // M >= 8 subcarriers, cp_len >= 1, taper_len <= cp_len (message on stderr and `throw 0` otherwise,
    // lib/ofdmtxrx.cc:59-69); p = subcarrier allocation or NULL; callback/userdata of the synchronizer
    ofdmtxrx(unsigned int _M, unsigned int _cp_len, unsigned int _taper_len, unsigned char *_p,
             framesync_callback _callback, void *_userdata);
    ofdmtxrx(unsigned int _M, unsigned int _cp_len, unsigned int _taper_len, unsigned char *_p,
             framesync_callback _callback, void *_userdata, bool _blocking_rx_worker);
    ~ofdmtxrx();

    // transmitter methods
    void set_tx_freq(float _tx_freq);
    void set_tx_rate(float _tx_rate);
    void set_tx_gain_soft(float _tx_gain_soft);     // [dB]
    void set_tx_gain_uhd(float _tx_gain_uhd);
    void set_tx_antenna(char *_tx_antenna);
    void reset_tx();
    // whole frame: every symbol x soft gain to the device, the last symbol buffer once more, then an
    // end-of-burst packet (lib/ofdmtxrx.cc:297-363)
    void transmit_packet(unsigned char *_header, unsigned char *_payload, unsigned int _payload_len,
                         int _mod, int _fec0, int _fec1);
    // the same in steps, so the caller can edit fgbuffer between write_symbol() and transmit_symbol()
    void assemble_frame(unsigned char *_header, unsigned char *_payload, unsigned int _payload_len,
                        int _mod, int _fec0, int _fec1);
    bool write_symbol();                            // next symbol into fgbuffer; true on the last one
    void transmit_symbol();
    void end_transmit_frame();

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

    // frame generator output buffer, one OFDM symbol (public in the reference too)
    unsigned int fgbuffer_len;                      // M + cp_len
    std::complex<float> *fgbuffer;

    // receiver objects of the blocking worker (public in the reference: include/ofdmtxrx.h:134-137)
    std::vector<std::complex<float> > *rx_buffer;   // the packet the worker has just received (valid between the two conditions)
    pthread_mutex_t rx_buffer_mutex;
    pthread_cond_t  rx_buffer_filled_cond;          // worker -> editor: *rx_buffer holds a packet
    pthread_cond_t  rx_buffer_modified_cond;        // editor -> worker: go on, synchronize it

private:
    ofdmtxrx(const ofdmtxrx &);
    ofdmtxrx &operator=(const ofdmtxrx &);
    void init(unsigned int _M, unsigned int _cp_len, unsigned int _taper_len, unsigned char *_p,
              framesync_callback _callback, void *_userdata, bool _blocking);
    void send_buffer();
    struct impl;
    impl *pimpl;
};

#endif
