// ref_harness.cc -- driver of the pin-if-present harness (VERDICT r5 "next" #3).
//
// NOT part of the product and NOT a restatement: this file is the only new code in a build whose other translation units are the
// reference's own lib/multichanneltx.cc and lib/multichannelrx.cc, compiled UNCHANGED from /root/reference against a real liquid-dsp
// (tests/golden/make_ref_golden.sh).  It plays the reference applications' traffic recipe (src/multichannel_tx.cc:163-213: header =
// packet id, channel, five more bytes; payload bytes; one UpdateData per ready channel; GenerateSamples; software gain 1/N) with a
// seeded generator in place of rand(), pushes the samples through multichannelrx::Execute in one piece, and writes what the callbacks
// received (include/multichannelrx.h:45): the wideband IQ, one text line per frame, the equalised payload symbols.
//   ref_harness <out_prefix> <num_channels> <M> <cp_len> <taper_len> <mod> <fec0> <fec1> <payload_len> <frames_per_channel> <seed>
// mod / fec0 / fec1 are liquid's scheme names ("qpsk", "none", "h128", "g2412", "v27" ...): liquid_getopt_str2mod / _str2fec, as the
// reference's command lines do (src/multichannel_tx.cc:92-94).
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include <liquid/liquid.h>

#include "multichanneltx.h"
#include "multichannelrx.h"

struct Sink { FILE *frames; FILE *syms; unsigned count; };
struct Chan { Sink *sink; unsigned channel; };

static int on_frame(unsigned char *_header, int _header_valid, unsigned char *_payload, unsigned int _payload_len, int _payload_valid,
                    framesyncstats_s _stats, void *_userdata)
{
    Chan *c = (Chan *)_userdata;
    FILE *f = c->sink->frames;
    fprintf(f, "%u %d %d %u ", c->channel, _header_valid, _payload_valid, _payload_len);
    for (int i = 0; i < 8; i++) fprintf(f, "%02x", _header[i]);
    fprintf(f, " ");
    if (_payload_len == 0) fprintf(f, "-");
    for (unsigned i = 0; i < _payload_len; i++) fprintf(f, "%02x", _payload[i]);
    fprintf(f, " %.9g %.9g %.9g %u %u %u %u %u %u\n", _stats.evm, _stats.rssi, _stats.cfo, (unsigned)_stats.mod_scheme, _stats.mod_bps,
            (unsigned)_stats.check, (unsigned)_stats.fec0, (unsigned)_stats.fec1, _stats.num_framesyms);
    if (_stats.num_framesyms && _stats.framesyms) fwrite(_stats.framesyms, sizeof(std::complex<float>), _stats.num_framesyms, c->sink->syms);
    c->sink->count++;
    return 0;
}

static unsigned lcg(unsigned &s) { s = s * 1664525u + 1013904223u; return s >> 24; }

int main(int argc, char **argv)
{
    if (argc != 12) { fprintf(stderr, "usage: %s out_prefix N M cp taper mod fec0 fec1 payload_len frames seed\n", argv[0]); return 2; }
    const char *prefix = argv[1];
    const unsigned N = atoi(argv[2]), M = atoi(argv[3]), cp = atoi(argv[4]), taper = atoi(argv[5]);
    const int ms = liquid_getopt_str2mod(argv[6]), fec0 = liquid_getopt_str2fec(argv[7]), fec1 = liquid_getopt_str2fec(argv[8]);
    const unsigned plen = atoi(argv[9]), frames = atoi(argv[10]);
    unsigned seed = (unsigned)strtoul(argv[11], NULL, 0);
    char path[1024];
    snprintf(path, sizeof(path), "%s.meta", prefix);
    FILE *fm = fopen(path, "w");
    if (!fm) { perror(path); return 1; }
    fprintf(fm, "liquid_libversion %s\nN %u\nM %u\ncp %u\ntaper %u\nmod %s %d\nfec0 %s %d\nfec1 %s %d\npayload_len %u\nframes %u\n",
            liquid_libversion(), N, M, cp, taper, argv[6], ms, argv[7], fec0, argv[8], fec1, plen, frames);
    printf("liquid_libversion %s\n", liquid_libversion());

    // ---- transmit side: the reference's class, the reference applications' loop
    multichanneltx mctx(N, M, cp, taper, NULL);
    std::vector<std::complex<float> > iq;
    std::vector<unsigned> pid(N, 0), sent(N, 0);
    std::vector<unsigned char> payload(plen ? plen : 1);
    unsigned char header[8];
    std::vector<std::complex<float> > buf(2 * N);
    const float g = 1.0f / (float)N;                          // src/multichannel_tx.cc:134-135 with txgain_dB = 0
    snprintf(path, sizeof(path), "%s.sent", prefix);
    FILE *fs = fopen(path, "w");
    if (!fs) { perror(path); return 1; }
    unsigned done = 0, idle = 0;
    while (idle < 64) {                                       // every channel's frames, then 64 more blocks of tail
        for (unsigned c = 0; c < N; c++) {
            if (sent[c] < frames && mctx.IsChannelReadyForData(c)) {
                pid[c]++;
                header[0] = (pid[c] >> 8) & 0xff; header[1] = pid[c] & 0xff; header[2] = c & 0xff;
                for (int i = 3; i < 8; i++) header[i] = (unsigned char)lcg(seed);
                for (unsigned i = 0; i < plen; i++) payload[i] = (unsigned char)lcg(seed);
                mctx.UpdateData(c, header, &payload[0], plen, ms, fec0, fec1);
                fprintf(fs, "%u ", c);
                for (int i = 0; i < 8; i++) fprintf(fs, "%02x", header[i]);
                fprintf(fs, " ");
                if (plen == 0) fprintf(fs, "-");
                for (unsigned i = 0; i < plen; i++) fprintf(fs, "%02x", payload[i]);
                fprintf(fs, "\n");
                sent[c]++; done++;
            }
        }
        mctx.GenerateSamples(&buf[0]);
        for (unsigned i = 0; i < 2 * N; i++) iq.push_back(g * buf[i]);
        bool all = true;
        for (unsigned c = 0; c < N; c++) all = all && sent[c] >= frames && mctx.IsChannelReadyForData(c);
        if (all) idle++;
        if (iq.size() > (size_t)400000000) { fprintf(stderr, "runaway generator\n"); return 1; }
    }
    fclose(fs);
    // whole tiles of 32 N samples (the GPU library consumes 16 blocks of 2N at a time)
    while (iq.size() % (32 * N)) iq.push_back(std::complex<float>(0.f, 0.f));
    snprintf(path, sizeof(path), "%s.iq", prefix);
    FILE *fi = fopen(path, "wb");
    if (!fi) { perror(path); return 1; }
    fwrite(&iq[0], sizeof(std::complex<float>), iq.size(), fi);
    fclose(fi);
    fprintf(fm, "samples %zu\nframes_sent %u\n", iq.size(), done);

    // ---- receive side: the reference's class
    Sink sink;
    snprintf(path, sizeof(path), "%s.frames", prefix);
    sink.frames = fopen(path, "w");
    snprintf(path, sizeof(path), "%s.syms", prefix);
    sink.syms = fopen(path, "wb");
    sink.count = 0;
    if (!sink.frames || !sink.syms) { perror(path); return 1; }
    std::vector<Chan> chans(N);
    std::vector<void *> userdata(N);
    std::vector<framesync_callback> callbacks(N);
    for (unsigned c = 0; c < N; c++) { chans[c].sink = &sink; chans[c].channel = c; userdata[c] = &chans[c]; callbacks[c] = on_frame; }
    {
        multichannelrx mcrx(N, M, cp, taper, NULL, &userdata[0], &callbacks[0]);
        mcrx.Execute(&iq[0], (unsigned)iq.size());
    }
    fclose(sink.frames); fclose(sink.syms);
    fprintf(fm, "frames_received %u\n", sink.count);
    fclose(fm);
    printf("%s: %zu samples, %u frames sent, %u callbacks\n", prefix, iq.size(), done, sink.count);
    return 0;
}
