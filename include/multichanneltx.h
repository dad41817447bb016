// multichanneltx.h -- MI355X-native multichannel OFDM transmitter, source compatible with
// liquid-usrp's class of the same name (reference: include/multichanneltx.h:29-90 for the public
// interface; lib/multichanneltx.cc for the behaviour).
//
// The frame generators, the 2N-channel synthesis bank and the oscillator run on the GPU one OFDM
// symbol period (M + cp_len calls of GenerateSamples) at a time -- the granularity at which the
// reference steps its own frame generators (lib/multichanneltx.cc:230-242) -- so what a caller
// observes through IsChannelReadyForData / UpdateData / GenerateSamples is call-for-call the same.
#ifndef LIQUID_USRP_AMD_MULTICHANNELTX_H
#define LIQUID_USRP_AMD_MULTICHANNELTX_H

#include <complex>
#include <liquid/liquid.h>

class multichanneltx {
public:
    // num_channels >= 1 (2*num_channels a power of two <= 1024), M >= 8 subcarriers (a power of two
    // <= 1024 on the GPU), cp_len >= 1, taper_len <= cp_len, p = subcarrier allocation or NULL.
    // Invalid arguments: message on stderr and `throw 0` (lib/multichanneltx.cc:48-60).
    multichanneltx(unsigned int _num_channels, unsigned int _M, unsigned int _cp_len,
                   unsigned int _taper_len, unsigned char *_p);
    ~multichanneltx();

    void Reset();
    unsigned int GetNumChannels() { return num_channels; }

    // 1: the channel takes a new frame; 0: its frame is still going out.  Bad id: `throw 0`.
    int IsChannelReadyForData(unsigned int _channel);

    // assemble a frame (crc32, the given modulation / inner / outer code) on one channel; on a busy
    // channel prints the reference's warning and returns
    void UpdateData(unsigned int _channel, unsigned char *_header, unsigned char *_payload,
                    unsigned int _payload_len, int _mod, int _fec0, int _fec1);

    // the next 2*num_channels wideband samples
    void GenerateSamples(std::complex<float> *_buffer);

private:
    multichanneltx(const multichanneltx &);
    multichanneltx &operator=(const multichanneltx &);
    unsigned int num_channels;
    struct impl;
    impl *pimpl;
};

#endif
