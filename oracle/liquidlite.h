/*
 * liquidlite.h -- CPU ORACLE (test infrastructure, NOT product code)
 *
 * Scalar, single-thread C restatement of the liquid-dsp objects that the
 * liquid-usrp multichannel receive path calls, plus the
 * multichannelrx / multichanneltx sample flows of the reference itself
 * (/root/reference/lib/multichannelrx.cc:155-195, lib/multichanneltx.cc:192-242).
 *
 * PARITY UNPINNED: liquid-dsp is an un-vendored, un-pinned dependency of the
 * reference ("HEAD revision", /root/reference/README.md:21; link check only,
 * configure.ac:56) and is absent from the build image, and the reference holds
 * no tests, fixtures or golden vectors.  This restatement follows liquid-dsp's
 * published algorithms (source file named at every function) and is pinned only
 * by standards-level known-answer tests and independent float64 models
 * (tests/test_oracle_*.py).  Deliberate deviations are listed in DESIGN.md.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * link or call anything in this directory.
 */
#ifndef LIQUIDLITE_H
#define LIQUIDLITE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* interleaved complex float, layout-compatible with std::complex<float> */
typedef struct { float re, im; } ll_cf;

/* ---- enums (numeric values follow liquid.h of the v1.2/v1.3 era) ---------- */
enum { LL_CRC_UNKNOWN=0, LL_CRC_NONE, LL_CRC_CHECKSUM, LL_CRC_8, LL_CRC_16, LL_CRC_24, LL_CRC_32 };
enum { LL_FEC_UNKNOWN=0, LL_FEC_NONE, LL_FEC_REP3, LL_FEC_REP5, LL_FEC_HAMMING74,
       LL_FEC_HAMMING84, LL_FEC_HAMMING128, LL_FEC_GOLAY2412,
       LL_FEC_CONV_V27 = 11 };     /* liquid's numbering: SECDED 8..10, then the convolutional codes (r = 1/2, K = 7 first) */
enum { LL_MODEM_UNKNOWN=0, LL_MODEM_QAM16=27, LL_MODEM_QAM64=29, LL_MODEM_BPSK=39, LL_MODEM_QPSK=40,
       LL_MODEM_NUM_SCHEMES=52 };
enum { LL_ANALYZER=0, LL_SYNTHESIZER=1 };
enum { LL_SCTYPE_NULL=0, LL_SCTYPE_PILOT=1, LL_SCTYPE_DATA=2 };

/* ---- math / filter design (liquid: src/math, src/filter/src/firdes.c) ----- */
float  ll_besseli0(float z);
float  ll_kaiser_beta_As(float As);
void   ll_firdes_kaiser(unsigned n, float fc, float As, float mu, float *h);
void   ll_fft(unsigned n, const ll_cf *x, ll_cf *y, int backward);   /* unnormalised, any n */

/* ---- NCO with 32-bit phase accumulator (liquid: src/nco/src/nco.proto.c) -- */
typedef struct { uint32_t theta, d_theta; } ll_nco;
uint32_t ll_nco_rad2u32(float rad);
float    ll_nco_u32rad(uint32_t u);             /* signed (-pi,pi] */
void     ll_nco_reset(ll_nco *q);
void     ll_nco_set_frequency(ll_nco *q, float dtheta);
void     ll_nco_adjust_frequency(ll_nco *q, float df);
float    ll_nco_get_frequency(const ll_nco *q);
void     ll_nco_step(ll_nco *q);
void     ll_nco_sincos_u32(uint32_t theta, float *s, float *c);
ll_cf    ll_nco_mix_down(const ll_nco *q, ll_cf x);
ll_cf    ll_nco_mix_up(const ll_nco *q, ll_cf x);

/* ---- m-sequence (liquid: src/sequence/src/msequence.c) -------------------- */
typedef struct { unsigned m, g, a, n, v, b; } ll_msequence;
void     ll_msequence_init_default(ll_msequence *ms, unsigned m);
void     ll_msequence_reset(ll_msequence *ms);
unsigned ll_msequence_advance(ll_msequence *ms);
unsigned ll_msequence_generate_symbol(ll_msequence *ms, unsigned bps);

/* ---- polyphase channelizer (liquid: src/multichannel/src/firpfbch.c) ------ */
typedef struct ll_firpfbch_s *ll_firpfbch;
ll_firpfbch ll_firpfbch_create_kaiser(int type, unsigned K, unsigned m, float As);
void        ll_firpfbch_destroy(ll_firpfbch q);
void        ll_firpfbch_reset(ll_firpfbch q);
void        ll_firpfbch_analyzer_execute(ll_firpfbch q, const ll_cf *x, ll_cf *y);
void        ll_firpfbch_synthesizer_execute(ll_firpfbch q, const ll_cf *X, ll_cf *y);
unsigned    ll_firpfbch_get_taps(ll_firpfbch q, float *h);   /* returns p*K, copies prototype */
void        ll_firpfbch_copy_state(ll_firpfbch dst, const struct ll_firpfbch_s *src);

/* 2x-oversampled analysis bank (liquid firpfbch2_crcf, analyzer): M channels, M/2 samples in, M out per call */
typedef struct ll_firpfbch2_s *ll_firpfbch2;
ll_firpfbch2 ll_firpfbch2_create_kaiser(unsigned M, unsigned m, float As);
void         ll_firpfbch2_destroy(ll_firpfbch2 q);
void         ll_firpfbch2_reset(ll_firpfbch2 q);
void         ll_firpfbch2_analyzer_execute(ll_firpfbch2 q, const ll_cf *x, ll_cf *y);
unsigned     ll_firpfbch2_get_taps(ll_firpfbch2 q, float *h);  /* returns 2m*M, copies prototype */
void         ll_firpfbch2_analyze(ll_firpfbch2 q, const ll_cf *x, unsigned nsteps, ll_cf *y);

/* ---- modem (liquid: src/modem/src/modem_{bpsk,qpsk,qam}.c, modem_common.c) - */
typedef struct ll_modem_s *ll_modem;
ll_modem ll_modem_create(int scheme);
void     ll_modem_destroy(ll_modem q);
unsigned ll_modem_bps(ll_modem q);
ll_cf    ll_modem_modulate(ll_modem q, unsigned sym);
unsigned ll_modem_demodulate(ll_modem q, ll_cf r);               /* hard; stores x_hat */
unsigned ll_modem_demodulate_soft(ll_modem q, ll_cf r, unsigned char *soft_bits);
float    ll_modem_get_evm(ll_modem q);
const unsigned char *ll_modem_soft_neighbors(ll_modem q, unsigned *p);

/* ---- bit-level coding (liquid: src/fec/src, src/utility) -------------- */
unsigned ll_crc_length(int scheme);
unsigned ll_crc_generate_key(int scheme, const unsigned char *msg, unsigned n);
unsigned ll_fec_enc_len(int scheme, unsigned dec_len);
int      ll_fec_supported(int scheme);      /* the schemes restated here: none, rep3, rep5, h74, h84, h128, g2412, v27 */
void     ll_fec_encode(int scheme, unsigned dec_len, const unsigned char *dec, unsigned char *enc);
void     ll_fec_decode(int scheme, unsigned dec_len, const unsigned char *enc, unsigned char *dec);
void     ll_fec_decode_soft(int scheme, unsigned dec_len, const unsigned char *enc_soft, unsigned char *dec);
unsigned ll_hamming128_encode_symbol(unsigned s);
unsigned ll_hamming74_encode_symbol(unsigned s);
unsigned ll_hamming84_encode_symbol(unsigned s);
unsigned ll_hamming128_decode_symbol(unsigned c);
unsigned ll_golay2412_encode_symbol(unsigned s);
unsigned ll_golay2412_decode_symbol(unsigned r);
void     ll_interleaver_dims(unsigned n, unsigned *M, unsigned *N);
void     ll_interleaver_encode(unsigned n, unsigned depth, const unsigned char *in, unsigned char *out);
void     ll_interleaver_decode(unsigned n, unsigned depth, const unsigned char *in, unsigned char *out);
void     ll_interleaver_decode_soft(unsigned n, unsigned depth, const unsigned char *in, unsigned char *out);
void     ll_scramble(unsigned char *x, unsigned n);
void     ll_repack_bytes(const unsigned char *in, unsigned in_bps, unsigned in_len,
                         unsigned char *out, unsigned out_bps, unsigned out_len, unsigned *written);

typedef struct ll_packetizer_s *ll_packetizer;
ll_packetizer ll_packetizer_create(unsigned n, int crc, int fec0, int fec1);
void     ll_packetizer_destroy(ll_packetizer p);
unsigned ll_packetizer_enc_len(ll_packetizer p);
unsigned ll_packetizer_compute_enc_len(unsigned n, int crc, int fec0, int fec1);
void     ll_packetizer_encode(ll_packetizer p, const unsigned char *msg, unsigned char *pkt);
int      ll_packetizer_decode(ll_packetizer p, const unsigned char *pkt, unsigned char *msg);
int      ll_packetizer_decode_soft(ll_packetizer p, const unsigned char *pkt_soft, unsigned char *msg);

/* ---- OFDM framing (liquid: src/multichannel/src/ofdmframe*.c) ------------- */
void ll_ofdmframe_init_default_sctype(unsigned M, unsigned char *p);
int  ll_ofdmframe_validate_sctype(const unsigned char *p, unsigned M,
                                  unsigned *M_null, unsigned *M_pilot, unsigned *M_data);
void ll_ofdmframe_init_S0(const unsigned char *p, unsigned M, ll_cf *S0, ll_cf *s0, unsigned *M_S0);
void ll_ofdmframe_init_S1(const unsigned char *p, unsigned M, ll_cf *S1, ll_cf *s1, unsigned *M_S1);
/* least-squares projection matrices shared by oracle and product parity tests */
void ll_ofdmframe_eq_smoother(const unsigned char *p, unsigned M, unsigned order, float *S /* [M][Nen] */);
void ll_ofdmframe_pilot_fit(const unsigned char *p, unsigned M, float *P /* [2][M_pilot] */);

typedef struct ll_ofdmframegen_s *ll_ofdmframegen;
ll_ofdmframegen ll_ofdmframegen_create(unsigned M, unsigned cp, unsigned taper, const unsigned char *p);
void ll_ofdmframegen_destroy(ll_ofdmframegen q);
void ll_ofdmframegen_reset(ll_ofdmframegen q);
void ll_ofdmframegen_write_S0a(ll_ofdmframegen q, ll_cf *y);
void ll_ofdmframegen_write_S0b(ll_ofdmframegen q, ll_cf *y);
void ll_ofdmframegen_write_S1(ll_ofdmframegen q, ll_cf *y);
void ll_ofdmframegen_writesymbol(ll_ofdmframegen q, const ll_cf *X, ll_cf *y);
void ll_ofdmframegen_writetail(ll_ofdmframegen q, ll_cf *y);   /* taper_len samples */

typedef int (*ll_ofdmframesync_callback)(ll_cf *X, const unsigned char *p, unsigned M, void *ud);
typedef struct ll_ofdmframesync_s *ll_ofdmframesync;
ll_ofdmframesync ll_ofdmframesync_create(unsigned M, unsigned cp, unsigned taper, const unsigned char *p,
                                         ll_ofdmframesync_callback cb, void *ud);
void  ll_ofdmframesync_destroy(ll_ofdmframesync q);
void  ll_ofdmframesync_reset(ll_ofdmframesync q);
void  ll_ofdmframesync_execute(ll_ofdmframesync q, const ll_cf *x, unsigned n);
float ll_ofdmframesync_get_rssi(ll_ofdmframesync q);
float ll_ofdmframesync_get_cfo(ll_ofdmframesync q);
int   ll_ofdmframesync_get_state(ll_ofdmframesync q);

/* ---- flexible OFDM frames (liquid: src/framing/src/ofdmflexframe{gen,sync}.c) */
typedef struct {
    float evm, rssi, cfo;
    ll_cf *framesyms;
    unsigned num_framesyms;
    unsigned mod_scheme, mod_bps, check, fec0, fec1;
} ll_framesyncstats;
typedef int (*ll_framesync_callback)(unsigned char *header, int header_valid,
                                     unsigned char *payload, unsigned payload_len, int payload_valid,
                                     ll_framesyncstats stats, void *userdata);
/* benchmark helper: a callback that only counts, and its userdata */
typedef struct { unsigned long long frames, headers_valid, payloads_valid, bytes; } ll_frame_counter;
int ll_counting_callback(unsigned char *header, int header_valid, unsigned char *payload, unsigned payload_len,
                         int payload_valid, ll_framesyncstats stats, void *userdata);
typedef struct { unsigned check, fec0, fec1, mod_scheme; } ll_ofdmflexframegenprops;

typedef struct ll_ofdmflexframegen_s *ll_ofdmflexframegen;
ll_ofdmflexframegen ll_ofdmflexframegen_create(unsigned M, unsigned cp, unsigned taper,
                                               const unsigned char *p, const ll_ofdmflexframegenprops *props);
void ll_ofdmflexframegen_destroy(ll_ofdmflexframegen q);
void ll_ofdmflexframegen_reset(ll_ofdmflexframegen q);
int  ll_ofdmflexframegen_is_assembled(ll_ofdmflexframegen q);
void ll_ofdmflexframegen_setprops(ll_ofdmflexframegen q, const ll_ofdmflexframegenprops *props);
unsigned ll_ofdmflexframegen_getframelen(ll_ofdmflexframegen q);      /* OFDM symbols incl. tail */
void ll_ofdmflexframegen_assemble(ll_ofdmflexframegen q, const unsigned char *header,
                                  const unsigned char *payload, unsigned payload_len);
int  ll_ofdmflexframegen_writesymbol(ll_ofdmflexframegen q, ll_cf *buf);      /* M+cp samples; 1 = last */
int  ll_ofdmflexframegen_write(ll_ofdmflexframegen q, ll_cf *buf, unsigned len); /* 1 = frame complete */

typedef struct ll_ofdmflexframesync_s *ll_ofdmflexframesync;
ll_ofdmflexframesync ll_ofdmflexframesync_create(unsigned M, unsigned cp, unsigned taper,
                                                 const unsigned char *p, ll_framesync_callback cb, void *ud);
void ll_ofdmflexframesync_destroy(ll_ofdmflexframesync q);
void ll_ofdmflexframesync_reset(ll_ofdmflexframesync q);
void ll_ofdmflexframesync_set_soft(ll_ofdmflexframesync q, int payload_soft);
void ll_ofdmflexframesync_execute(ll_ofdmflexframesync q, const ll_cf *x, unsigned n);

/* ---- named deviations from liquid-dsp's ofdmframesync (liquid-dsp is not available to confirm
 * either way, so both are switches: flip them the day a libliquid can be put beside this file.  The GPU kernels carry
 * the same two switches, MCRX_S1_BACKOFF_CORRECTION / MCRX_S1_METRIC_G0_NORMALISED in csrc/kernels.h -- keep them equal) */
#ifndef LL_S1_BACKOFF_CORRECTION
#define LL_S1_BACKOFF_CORRECTION 0      /* D6: 1 = apply liquid's "timing backoff correction" G[k] *= e^{j 2 pi k backoff / M}
                                           after the S1 gain estimate.  0 here: the S1 window and every data window sit the
                                           same `backoff` samples inside the cyclic prefix, so 1/G already removes that ramp */
#endif
#ifndef LL_S1_METRIC_G0_NORMALISED
#define LL_S1_METRIC_G0_NORMALISED 1    /* D7: 1 = the S1 detection metric is scaled by the S0-stage gain g0 (level independent,
                                           like the S0 metric); 0 = raw */
#endif

/* ---- multi-stage arbitrary resampler (liquid: src/filter/src/msresamp*.c) -- */
typedef struct ll_msresamp_s *ll_msresamp;
ll_msresamp ll_msresamp_create(float rate, float As);
void  ll_msresamp_destroy(ll_msresamp q);
void  ll_msresamp_reset(ll_msresamp q);
float ll_msresamp_get_delay(ll_msresamp q);
void  ll_msresamp_execute(ll_msresamp q, const ll_cf *x, unsigned nx, ll_cf *y, unsigned *ny);

/* ---- reference flows: /root/reference/lib/multichannel{rx,tx}.cc ---------- */
typedef struct ll_mcrx_s *ll_mcrx;
ll_mcrx  ll_mcrx_create(unsigned N, unsigned M, unsigned cp, unsigned taper, const unsigned char *p,
                        void **userdata, ll_framesync_callback *cb);
void     ll_mcrx_destroy(ll_mcrx q);
void     ll_mcrx_reset(ll_mcrx q);
void     ll_mcrx_set_soft(ll_mcrx q, int payload_soft);
void     ll_mcrx_execute(ll_mcrx q, const ll_cf *x, unsigned n);
void     ll_mcrx_execute_parallel(ll_mcrx q, const ll_cf *x, unsigned n, int nthreads);   /* OpenMP: banks over time, synchronizers over channels */
/* stage tap for parity tests: NCO + analyzer only, keeps bins [0,N): out[nblocks][N] */
void     ll_mcrx_channelize(ll_mcrx q, const ll_cf *x, unsigned nblocks, ll_cf *out);
/* alternate front end: firpfbch2 (2x oversampled, 2N channels) + half-band decimator per channel, see ll_multichannel.c */
void     ll_mcrx_set_front_end(ll_mcrx q, int oversampled);
void     ll_mcrx_channelize_oversampled(ll_mcrx q, const ll_cf *x, unsigned nblocks, ll_cf *out);

typedef struct ll_mctx_s *ll_mctx;
ll_mctx  ll_mctx_create(unsigned N, unsigned M, unsigned cp, unsigned taper, const unsigned char *p);
void     ll_mctx_destroy(ll_mctx q);
void     ll_mctx_reset(ll_mctx q);
int      ll_mctx_is_channel_ready(ll_mctx q, unsigned ch);
int      ll_mctx_update_data(ll_mctx q, unsigned ch, const unsigned char *header,
                             const unsigned char *payload, unsigned payload_len,
                             int mod, int fec0, int fec1);
void     ll_mctx_generate_samples(ll_mctx q, ll_cf *buf /* 2N */);

#ifdef __cplusplus
}
#endif
#endif /* LIQUIDLITE_H */
