/*
 * liquid/liquid.h -- minimal interface shim (own declarations, not liquid-dsp's header).
 *
 * liquid-usrp's sources include <liquid/liquid.h> for the framing types that cross the
 * multichannelrx / multichanneltx API (include/multichannelrx.h:27,45; src/multichannel_rx.cc:28,
 * 37-47; lib/multichanneltx.cc:70-75,184).  liquid-dsp itself is not part of this build; this
 * header declares exactly those names so the reference's application sources compile
 * unchanged against the MI355X-native library.
 */
#ifndef LIQUID_USRP_AMD_LIQUID_SHIM_H
#define LIQUID_USRP_AMD_LIQUID_SHIM_H

#ifdef __cplusplus
#include <complex>
typedef std::complex<float> liquid_float_complex;
extern "C" {
#else
#include <complex.h>
typedef float complex liquid_float_complex;
#endif

/* data validity checks, forward error correction, modulation (numeric values as in liquid-dsp) */
typedef enum { LIQUID_CRC_UNKNOWN = 0, LIQUID_CRC_NONE, LIQUID_CRC_CHECKSUM, LIQUID_CRC_8,
               LIQUID_CRC_16, LIQUID_CRC_24, LIQUID_CRC_32 } crc_scheme;
typedef enum { LIQUID_FEC_UNKNOWN = 0, LIQUID_FEC_NONE, LIQUID_FEC_REP3, LIQUID_FEC_REP5,
               LIQUID_FEC_HAMMING74, LIQUID_FEC_HAMMING84, LIQUID_FEC_HAMMING128,
               LIQUID_FEC_GOLAY2412, LIQUID_FEC_SECDED2216, LIQUID_FEC_SECDED3932, LIQUID_FEC_SECDED7264,
               LIQUID_FEC_CONV_V27 /* = 11: r = 1/2, K = 7 */ } fec_scheme;
typedef enum { LIQUID_MODEM_UNKNOWN = 0, LIQUID_MODEM_QAM16 = 27, LIQUID_MODEM_QAM64 = 29,
               LIQUID_MODEM_BPSK = 39, LIQUID_MODEM_QPSK = 40 } modulation_scheme;
enum { LIQUID_ANALYZER = 0, LIQUID_SYNTHESIZER = 1 };

/* frame synchronizer statistics handed to the callback by value */
typedef struct {
    float evm;                              /* error vector magnitude [dB] */
    float rssi;                             /* received signal strength indication [dB] */
    float cfo;                              /* carrier frequency offset [f/Fs] */
    liquid_float_complex *framesyms;        /* equalised payload symbols */
    unsigned int num_framesyms;
    unsigned int mod_scheme, mod_bps, check, fec0, fec1;
} framesyncstats_s;

typedef int (*framesync_callback)(unsigned char *_header, int _header_valid,
                                  unsigned char *_payload, unsigned int _payload_len,
                                  int _payload_valid, framesyncstats_s _stats, void *_userdata);

/* option parsing used by the applications (src/multichannel_tx.cc:49,52,93-95) */
modulation_scheme liquid_getopt_str2mod(const char *_str);
fec_scheme        liquid_getopt_str2fec(const char *_str);
void              liquid_print_modulation_schemes(void);
void              liquid_print_fec_schemes(void);
/* (liquid-dsp's own reports its version here; this shim says what it is, so that a log or a fixture cannot be mistaken for liquid-dsp's:
 *  tests/golden/ref_harness.cc writes it into every fixture's .meta) */
const char *      liquid_libversion(void);

/* frame generator properties (lib/multichanneltx.cc:70-75,184) */
typedef struct { unsigned int check, fec0, fec1, mod_scheme; } ofdmflexframegenprops_s;

#ifdef __cplusplus
}
#endif
#endif
