// shard_test.cc -- drives the multichannelrx class through a call sequence the unchanged applications do not make
// (bulk Execute() in uneven pieces, a Reset() in mid-stream, destruction right behind the last frame) so that the sharded form of the
// class (MCRX_WORLD set) can be held against the plain one on the same calls: tests/test_gpu_refapp.py.
//   shard_test <iq file (cf32)> <channels> <M> <cp> <taper> <piece samples> <reset after this many samples, 0 = never>
// Prints one line per callback: channel, packet id, validity flags, payload length, a checksum of the payload.
#include <cstdio>
#include <cstdlib>
#include <complex>
#include <vector>
#include "multichannelrx.h"

static int callback(unsigned char *_header, int _header_valid, unsigned char *_payload, unsigned int _payload_len,
                    int _payload_valid, framesyncstats_s _stats, void *_userdata)
{
    unsigned long sum = 5381;
    for (unsigned int i = 0; i < _payload_len; i++) sum = sum * 33 + _payload[i];
    printf("frame ch %u pid %u hv %d pv %d len %u sum %lu\n", *(unsigned int *)_userdata, (_header[0] << 8) | _header[1], _header_valid, _payload_valid,
           _payload_len, sum & 0xffffffffUL);
    (void)_stats;
    return 0;
}

int main(int argc, char **argv)
{
    if (argc < 8) { fprintf(stderr, "usage: %s iqfile channels M cp taper piece reset_at\n", argv[0]); return 2; }
    const unsigned N = (unsigned)atoi(argv[2]), M = (unsigned)atoi(argv[3]), cp = (unsigned)atoi(argv[4]), taper = (unsigned)atoi(argv[5]);
    const size_t piece = (size_t)atol(argv[6]), reset_at = (size_t)atol(argv[7]);
    FILE *f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 2; }
    std::vector<std::complex<float> > x;
    std::complex<float> buf[4096];
    size_t n;
    while ((n = fread(buf, sizeof(buf[0]), 4096, f)) > 0) x.insert(x.end(), buf, buf + n);
    fclose(f);
    std::vector<unsigned int> ids(N);
    std::vector<void *> ud(N);
    std::vector<framesync_callback> cb(N, callback);
    for (unsigned i = 0; i < N; i++) { ids[i] = i; ud[i] = &ids[i]; }
    {
        multichannelrx rx(N, M, cp, taper, NULL, ud.data(), cb.data());
        size_t pos = 0; bool did = reset_at == 0;
        while (pos < x.size()) {
            size_t take = std::min(piece, x.size() - pos);
            if (!did && pos + take >= reset_at) take = reset_at - pos;
            if (take) rx.Execute(x.data() + pos, (unsigned int)take);
            pos += take;
            if (!did && pos >= reset_at) { rx.Reset(); did = true; printf("reset at %zu\n", pos); }
        }
    }       // (destruction: what the stream held to its end is delivered)
    printf("done\n");
    return 0;
}
