// multichanneltx.cc -- host class over the streaming C-ABI of libmcrx_hip.so (include/mcrx_hip.h,
// mctx_hip_stream_*).  Mirrors liquid-usrp's lib/multichanneltx.cc: ctor :41-100, dtor :103-122,
// Reset :126-149, IsChannelReadyForData :151-162, UpdateData :165-189, GenerateSamples :192-227.
#include <cstdio>
#include <mutex>

#include "multichanneltx.h"
#include "mcrx_hip.h"

// one lock around every call: multichanneltxrx polls / updates from the application thread while its
// transmit worker pulls samples (lib/multichanneltxrx.cc:217-299,451-470)
struct multichanneltx::impl { mctx_hip_t h; std::mutex mu; };
#define TX_LOCK std::lock_guard<std::mutex> lk_(pimpl->mu)

static void tx_fail(const char *what)
{
    fprintf(stderr, "%s: %s\n", what, mctx_hip_last_error());
    throw 0;
}

multichanneltx::multichanneltx(unsigned int _num_channels, unsigned int _M, unsigned int _cp_len,
                               unsigned int _taper_len, unsigned char *_p)
    : num_channels(_num_channels), pimpl(new impl)
{
    pimpl->h = NULL;
    if (mctx_hip_create(&pimpl->h, _num_channels, _M, _cp_len, _taper_len, _p) != MCRX_OK ||
        mctx_hip_stream_begin(pimpl->h, 2048) != MCRX_OK) {
        fprintf(stderr, "%s\n", mctx_hip_last_error());
        if (pimpl->h) mctx_hip_destroy(pimpl->h);
        delete pimpl;
        throw 0;
    }
}

multichanneltx::~multichanneltx()
{
    mctx_hip_destroy(pimpl->h);
    delete pimpl;
}

void multichanneltx::Reset()
{
    TX_LOCK;
    if (mctx_hip_stream_reset(pimpl->h) != MCRX_OK) tx_fail("multichanneltx::Reset");
}

int multichanneltx::IsChannelReadyForData(unsigned int _channel)
{
    if (_channel >= num_channels) {
        fprintf(stderr, "error: multichanneltx:IsChannelReadyForData(%u), invalid channel id\n", _channel);
        throw 0;
    }
    TX_LOCK;
    return mctx_hip_stream_ready(pimpl->h, _channel);
}

void multichanneltx::UpdateData(unsigned int _channel, unsigned char *_header, unsigned char *_payload,
                                unsigned int _payload_len, int _mod, int _fec0, int _fec1)
{
    if (_channel >= num_channels) {
        fprintf(stderr, "error: multichanneltx:UpdateData(%u), invalid channel id\n", _channel);
        throw 0;
    }
    TX_LOCK;
    int rc = mctx_hip_stream_update(pimpl->h, _channel, _header, _payload, _payload_len, _mod, _fec0, _fec1);
    if (rc == MCRX_EBUSY) {
        fprintf(stderr, "warning: multichanneltx:UpdateData(%u), channel not ready yet\n", _channel);
        return;
    }
    if (rc != MCRX_OK) tx_fail("multichanneltx::UpdateData");
}

void multichanneltx::GenerateSamples(std::complex<float> *_buffer)
{
    TX_LOCK;
    if (mctx_hip_stream_generate(pimpl->h, reinterpret_cast<float *>(_buffer)) != MCRX_OK)
        tx_fail("multichanneltx::GenerateSamples");
}
