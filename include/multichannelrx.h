// multichannelrx.h -- MI355X-native multichannel OFDM receiver, source compatible with
// liquid-usrp's class of the same name (reference: include/multichannelrx.h:29-58 for the
// public interface; lib/multichannelrx.cc for the behaviour).
//
// Differences visible to callers, all forced by batching on the GPU:
//   * callbacks fire on the calling thread at flush points -- when the internal staging buffer
//     fills inside Execute(), in Reset(), in Flush() and in the destructor -- in the reference's
//     order (frame end time, then channel index), not synchronously per sample;
//   * samples are consumed in whole tiles of 16 channelizer blocks (MCRX_TILE: 32*N samples); a shorter
//     tail waits for more input;
//   * MCRX_WORLD / MCRX_RANK / MCRX_UID_FILE / MCRX_SUB_BLOCKS in the environment make the object one rank of a receiver
//     sharded over the GPUs of a node (host/multichannelrx.cc): callbacks fire for its channel shard;
//   * the reference's BST_DEBUG file dump at destruction (lib/multichannelrx.cc:118-122) is liquid's internal state; with
//     $MCRX_DEBUG_DIR set the destructor writes the same file names (framesync_channel%u.m) with each channel's frame count
//     and the equalised symbols of its last frame.
#ifndef LIQUID_USRP_AMD_MULTICHANNELRX_H
#define LIQUID_USRP_AMD_MULTICHANNELRX_H

#include <complex>
#include <liquid/liquid.h>

class multichannelrx {
public:
    // num_channels >= 1 (any count up to 1024; 2*num_channels a power of two <= 1024 takes the fast channelizer), M >= 8
    // subcarriers (<= 1024), cp_len >= 1,
    // taper_len <= cp_len, p = subcarrier allocation or NULL, per-channel userdata / callbacks
    // (both arrays are copied).  Invalid arguments: message on stderr and `throw 0`, like the
    // reference (lib/multichannelrx.cc:54-66).
    multichannelrx(unsigned int _num_channels, unsigned int _M, unsigned int _cp_len,
                   unsigned int _taper_len, unsigned char *_p, void **_userdata,
                   framesync_callback *_callback);
    ~multichannelrx();

    void Reset();
    unsigned int GetNumChannels() { return num_channels; }
    void Execute(std::complex<float> *_x, unsigned int _num_samples);

    // additions
    void Flush();                               // process what is buffered, deliver callbacks
    void ExecuteDevice(const void *_d_x, unsigned int _num_samples);   // samples already in HBM

private:
    multichannelrx(const multichannelrx &);
    multichannelrx &operator=(const multichannelrx &);
    void Deliver();
    unsigned int num_channels;
    struct impl;
    impl *pimpl;
};

#endif
