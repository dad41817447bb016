/*
 * ll_ofdmflexframe.c -- CPU ORACLE (test infrastructure): flexible OFDM frame generator
 * and synchronizer (header + variable payload on top of ofdmframegen / ofdmframesync).
 *
 * Restates liquid-dsp src/framing/src/ofdmflexframegen.c and ofdmflexframesync.c, the
 * objects the reference creates at /root/reference/lib/multichanneltx.cc:70-81 and
 * lib/multichannelrx.cc:82 and drives at lib/multichanneltx.cc:184-188,234-236 and
 * lib/multichannelrx.cc:194.
 *
 * Frame: S0a S0b S1 | header symbols | payload symbols | tail.
 * Header: 8 user bytes + [protocol=104, len_hi, len_lo, mod_scheme, (check&7)<<5 | fec0,
 * fec1] -> CRC-32 -> Golay(24,12) -> interleave -> scramble = 36 bytes = 288 BPSK
 * symbols on the data subcarriers in ascending bin order.
 * Payload: packetizer(len, check, fec0, fec1) bytes repacked MSB-first into bps-bit
 * modem symbols.  The synchronizer demodulates the header hard and the payload soft
 * (8-bit soft bits -> packetizer_decode_soft) unless set_soft(0).
 *
 * Deviation: unused data subcarriers of the last header / payload symbol
 * are filled from a fixed LCG instead of rand().
 */
#include "liquidlite.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define FLEX_PROTOCOL 104
#define FLEX_H_USER 8
#define FLEX_H_DEC 14
#define FLEX_H_ENC 36
#define FLEX_H_SYM 288

static inline ll_cf cf(float re, float im) { ll_cf r = { re, im }; return r; }

/* ================================================================== generator */
enum { FG_S0A = 0, FG_S0B, FG_S1, FG_HEADER, FG_PAYLOAD, FG_TAIL, FG_ZEROS };

struct ll_ofdmflexframegen_s {
    unsigned M, cp, taper;
    unsigned char *p;
    unsigned M_null, M_pilot, M_data;
    ll_ofdmframegen fg;
    ll_cf *X;
    ll_ofdmflexframegenprops props;
    /* header */
    ll_modem mod_header; ll_packetizer p_header;
    unsigned char header[FLEX_H_DEC], header_enc[FLEX_H_ENC], header_mod[FLEX_H_SYM];
    unsigned num_symbols_header;
    /* payload */
    ll_packetizer p_payload; ll_modem mod_payload;
    unsigned payload_dec_len, payload_enc_len, payload_mod_len, num_symbols_payload;
    unsigned char *payload_enc, *payload_mod;
    /* state */
    int state, frame_assembled, frame_complete;
    unsigned symbol_number, header_symbol_index, payload_symbol_index;
    uint32_t lcg;
    /* buffered write() */
    ll_cf *buf; unsigned buf_index;
};

static void fg_reconfigure(ll_ofdmflexframegen q)
{
    if (q->p_payload) ll_packetizer_destroy(q->p_payload);
    q->p_payload = ll_packetizer_create(q->payload_dec_len, (int)q->props.check, (int)q->props.fec0, (int)q->props.fec1);
    q->payload_enc_len = ll_packetizer_enc_len(q->p_payload);
    q->payload_enc = (unsigned char *)realloc(q->payload_enc, q->payload_enc_len + 8);
    if (q->mod_payload) ll_modem_destroy(q->mod_payload);
    q->mod_payload = ll_modem_create((int)q->props.mod_scheme);
    unsigned bps = ll_modem_bps(q->mod_payload);
    unsigned nb = 8 * q->payload_enc_len;
    q->payload_mod_len = nb / bps + ((nb % bps) ? 1 : 0);
    q->payload_mod = (unsigned char *)realloc(q->payload_mod, q->payload_mod_len + 8);
    q->num_symbols_payload = q->payload_mod_len / q->M_data + ((q->payload_mod_len % q->M_data) ? 1 : 0);
}

ll_ofdmflexframegen ll_ofdmflexframegen_create(unsigned M, unsigned cp, unsigned taper,
                                               const unsigned char *p, const ll_ofdmflexframegenprops *props)
{
    ll_ofdmflexframegen q = (ll_ofdmflexframegen)calloc(1, sizeof(*q));
    q->M = M; q->cp = cp; q->taper = taper;
    q->p = (unsigned char *)malloc(M);
    if (p) memcpy(q->p, p, M); else ll_ofdmframe_init_default_sctype(M, q->p);
    ll_ofdmframe_validate_sctype(q->p, M, &q->M_null, &q->M_pilot, &q->M_data);
    q->fg = ll_ofdmframegen_create(M, cp, taper, q->p);
    q->X = (ll_cf *)calloc(M, sizeof(ll_cf));
    q->buf = (ll_cf *)calloc(M + cp, sizeof(ll_cf));
    q->mod_header = ll_modem_create(LL_MODEM_BPSK);
    q->p_header = ll_packetizer_create(FLEX_H_DEC, LL_CRC_32, LL_FEC_GOLAY2412, LL_FEC_NONE);
    q->num_symbols_header = FLEX_H_SYM / q->M_data + ((FLEX_H_SYM % q->M_data) ? 1 : 0);
    q->payload_dec_len = 1;
    if (props) q->props = *props;
    else { q->props.check = LL_CRC_32; q->props.fec0 = LL_FEC_NONE; q->props.fec1 = LL_FEC_NONE; q->props.mod_scheme = LL_MODEM_QPSK; }
    fg_reconfigure(q);
    ll_ofdmflexframegen_reset(q);
    return q;
}
void ll_ofdmflexframegen_destroy(ll_ofdmflexframegen q)
{
    if (!q) return;
    ll_ofdmframegen_destroy(q->fg); ll_modem_destroy(q->mod_header); ll_modem_destroy(q->mod_payload);
    ll_packetizer_destroy(q->p_header); ll_packetizer_destroy(q->p_payload);
    free(q->p); free(q->X); free(q->buf); free(q->payload_enc); free(q->payload_mod); free(q);
}
void ll_ofdmflexframegen_reset(ll_ofdmflexframegen q)
{
    q->symbol_number = 0; q->state = FG_S0A;
    q->frame_assembled = 0; q->frame_complete = 0;
    q->header_symbol_index = 0; q->payload_symbol_index = 0;
    q->buf_index = q->M + q->cp;
    q->lcg = 0x1234567u;
    ll_ofdmframegen_reset(q->fg);
}
int ll_ofdmflexframegen_is_assembled(ll_ofdmflexframegen q) { return q->frame_assembled; }
void ll_ofdmflexframegen_setprops(ll_ofdmflexframegen q, const ll_ofdmflexframegenprops *props)
{
    if (!props) return;
    if (memcmp(&q->props, props, sizeof(*props)) == 0) return;
    q->props = *props;
    fg_reconfigure(q);
}
unsigned ll_ofdmflexframegen_getframelen(ll_ofdmflexframegen q)
{ return 3 + q->num_symbols_header + q->num_symbols_payload + 1; }

void ll_ofdmflexframegen_assemble(ll_ofdmflexframegen q, const unsigned char *header,
                                  const unsigned char *payload, unsigned payload_len)
{
    if (payload_len != q->payload_dec_len) { q->payload_dec_len = payload_len; fg_reconfigure(q); }
    q->frame_assembled = 1;
    memmove(q->header, header, FLEX_H_USER);
    unsigned n = FLEX_H_USER;
    q->header[n + 0] = FLEX_PROTOCOL;
    q->header[n + 1] = (unsigned char)((q->payload_dec_len >> 8) & 0xff);
    q->header[n + 2] = (unsigned char)(q->payload_dec_len & 0xff);
    q->header[n + 3] = (unsigned char)q->props.mod_scheme;
    q->header[n + 4] = (unsigned char)(((q->props.check & 0x07) << 5) | (q->props.fec0 & 0x1f));
    q->header[n + 5] = (unsigned char)(q->props.fec1 & 0x1f);
    ll_packetizer_encode(q->p_header, q->header, q->header_enc);
    ll_scramble(q->header_enc, FLEX_H_ENC);
    unsigned nw;
    ll_repack_bytes(q->header_enc, 8, FLEX_H_ENC, q->header_mod, 1, FLEX_H_SYM, &nw);
    ll_packetizer_encode(q->p_payload, payload, q->payload_enc);
    memset(q->payload_mod, 0, q->payload_mod_len);
    ll_repack_bytes(q->payload_enc, 8, q->payload_enc_len, q->payload_mod,
                    ll_modem_bps(q->mod_payload), q->payload_mod_len, &nw);
}

static unsigned fg_fill(ll_ofdmflexframegen q, unsigned bps)
{ q->lcg = q->lcg * 1664525u + 1013904223u; return (q->lcg >> 16) & ((1u << bps) - 1); }

int ll_ofdmflexframegen_writesymbol(ll_ofdmflexframegen q, ll_cf *buf)
{
    unsigned L = q->M + q->cp;
    q->symbol_number++;
    switch (q->state) {
    case FG_S0A: ll_ofdmframegen_write_S0a(q->fg, buf); q->state = FG_S0B; break;
    case FG_S0B: ll_ofdmframegen_write_S0b(q->fg, buf); q->state = FG_S1; break;
    case FG_S1:  ll_ofdmframegen_write_S1(q->fg, buf); q->symbol_number = 0; q->state = FG_HEADER; break;
    case FG_HEADER:
        for (unsigned i = 0; i < q->M; i++) {
            if (q->p[i] != LL_SCTYPE_DATA) { q->X[i] = cf(0, 0); continue; }
            unsigned s = (q->header_symbol_index < FLEX_H_SYM) ? q->header_mod[q->header_symbol_index++]
                                                               : fg_fill(q, 1);
            q->X[i] = ll_modem_modulate(q->mod_header, s);
        }
        ll_ofdmframegen_writesymbol(q->fg, q->X, buf);
        if (q->symbol_number == q->num_symbols_header) { q->symbol_number = 0; q->state = FG_PAYLOAD; }
        break;
    case FG_PAYLOAD: {
        unsigned bps = ll_modem_bps(q->mod_payload);
        for (unsigned i = 0; i < q->M; i++) {
            if (q->p[i] != LL_SCTYPE_DATA) { q->X[i] = cf(0, 0); continue; }
            unsigned s = (q->payload_symbol_index < q->payload_mod_len) ? q->payload_mod[q->payload_symbol_index++]
                                                                        : fg_fill(q, bps);
            q->X[i] = ll_modem_modulate(q->mod_payload, s);
        }
        ll_ofdmframegen_writesymbol(q->fg, q->X, buf);
        if (q->symbol_number == q->num_symbols_payload) q->state = FG_TAIL;
    } break;
    case FG_TAIL:
        memset(buf, 0, sizeof(ll_cf) * L);
        ll_ofdmframegen_writetail(q->fg, buf);
        q->frame_complete = 1;
        break;
    default:
        memset(buf, 0, sizeof(ll_cf) * L);
    }
    if (q->frame_complete) { ll_ofdmflexframegen_reset(q); return 1; }
    return 0;
}

int ll_ofdmflexframegen_write(ll_ofdmflexframegen q, ll_cf *out, unsigned len)
{
    /* buffered variant: zeros when nothing is assembled; returns 1 once the frame's
     * last symbol has been produced into the internal buffer */
    unsigned L = q->M + q->cp;
    int complete = 0;
    for (unsigned i = 0; i < len; i++) {
        if (q->buf_index >= L) {
            if (q->frame_assembled) complete |= ll_ofdmflexframegen_writesymbol(q, q->buf);
            else memset(q->buf, 0, sizeof(ll_cf) * L);
            q->buf_index = 0;
        }
        out[i] = q->buf[q->buf_index++];
    }
    return complete;
}

/* ================================================================== synchronizer */
enum { FS_HEADER = 0, FS_PAYLOAD };

struct ll_ofdmflexframesync_s {
    unsigned M, cp, taper;
    unsigned char *p;
    unsigned M_null, M_pilot, M_data;
    ll_modem mod_header; ll_packetizer p_header;
    unsigned char header[FLEX_H_DEC], header_enc[FLEX_H_ENC], header_mod[FLEX_H_SYM];
    int header_valid;
    unsigned ms_payload, bps_payload, payload_len, check, fec0, fec1;
    ll_packetizer p_payload; ll_modem mod_payload;
    unsigned char *payload_enc, *payload_dec;     /* payload_enc: bytes (hard) or 8x soft bits */
    ll_cf *payload_syms;
    unsigned payload_enc_len, payload_mod_len;
    int payload_valid, payload_soft;
    ll_framesync_callback cb; void *ud;
    ll_framesyncstats stats;
    float evm_hat;
    ll_ofdmframesync fs;
    int state;
    unsigned header_symbol_index, payload_symbol_index, payload_buffer_index;
};

static int fs_internal_callback(ll_cf *X, const unsigned char *p, unsigned M, void *ud);

ll_ofdmflexframesync ll_ofdmflexframesync_create(unsigned M, unsigned cp, unsigned taper,
                                                 const unsigned char *p, ll_framesync_callback cb, void *ud)
{
    ll_ofdmflexframesync q = (ll_ofdmflexframesync)calloc(1, sizeof(*q));
    q->M = M; q->cp = cp; q->taper = taper; q->cb = cb; q->ud = ud;
    q->p = (unsigned char *)malloc(M);
    if (p) memcpy(q->p, p, M); else ll_ofdmframe_init_default_sctype(M, q->p);
    ll_ofdmframe_validate_sctype(q->p, M, &q->M_null, &q->M_pilot, &q->M_data);
    q->fs = ll_ofdmframesync_create(M, cp, taper, q->p, fs_internal_callback, q);
    q->mod_header = ll_modem_create(LL_MODEM_BPSK);
    q->p_header = ll_packetizer_create(FLEX_H_DEC, LL_CRC_32, LL_FEC_GOLAY2412, LL_FEC_NONE);
    q->payload_soft = 1;
    q->ms_payload = LL_MODEM_QPSK; q->bps_payload = 2; q->payload_len = 1;
    q->check = LL_CRC_32; q->fec0 = LL_FEC_NONE; q->fec1 = LL_FEC_NONE;
    q->mod_payload = ll_modem_create((int)q->ms_payload);
    q->p_payload = ll_packetizer_create(q->payload_len, (int)q->check, (int)q->fec0, (int)q->fec1);
    q->payload_enc_len = ll_packetizer_enc_len(q->p_payload);
    q->payload_mod_len = 0;
    ll_ofdmflexframesync_reset(q);
    return q;
}
void ll_ofdmflexframesync_destroy(ll_ofdmflexframesync q)
{
    if (!q) return;
    ll_ofdmframesync_destroy(q->fs); ll_modem_destroy(q->mod_header); ll_modem_destroy(q->mod_payload);
    ll_packetizer_destroy(q->p_header); ll_packetizer_destroy(q->p_payload);
    free(q->p); free(q->payload_enc); free(q->payload_dec); free(q->payload_syms); free(q);
}
void ll_ofdmflexframesync_reset(ll_ofdmflexframesync q)
{
    q->state = FS_HEADER;
    q->header_symbol_index = 0; q->payload_symbol_index = 0; q->payload_buffer_index = 0;
    q->evm_hat = 0.0f;
    ll_ofdmframesync_reset(q->fs);
}
void ll_ofdmflexframesync_set_soft(ll_ofdmflexframesync q, int s) { q->payload_soft = s ? 1 : 0; }
void ll_ofdmflexframesync_execute(ll_ofdmflexframesync q, const ll_cf *x, unsigned n)
{ ll_ofdmframesync_execute(q->fs, x, n); }

static void fs_decode_header(ll_ofdmflexframesync q)
{
    unsigned nw;
    ll_repack_bytes(q->header_mod, 1, FLEX_H_SYM, q->header_enc, 8, FLEX_H_ENC, &nw);
    ll_scramble(q->header_enc, FLEX_H_ENC);
    q->header_valid = ll_packetizer_decode(q->p_header, q->header_enc, q->header);
    if (!q->header_valid) return;
    unsigned n = FLEX_H_USER;
    if (q->header[n + 0] != FLEX_PROTOCOL) { q->header_valid = 0; return; }
    unsigned payload_len = ((unsigned)q->header[n + 1] << 8) | q->header[n + 2];
    unsigned mod_scheme = q->header[n + 3];
    unsigned check = (q->header[n + 4] >> 5) & 0x07;
    unsigned fec0 = q->header[n + 4] & 0x1f, fec1 = q->header[n + 5] & 0x1f;
    ll_modem m = ll_modem_create((int)mod_scheme);
    if (!m) { q->header_valid = 0; return; }
    if (check == LL_CRC_UNKNOWN || check > LL_CRC_32 ||
        !ll_fec_supported((int)fec0) || !ll_fec_supported((int)fec1)) {
        ll_modem_destroy(m); q->header_valid = 0; return;
    }
    ll_modem_destroy(q->mod_payload); q->mod_payload = m;
    q->ms_payload = mod_scheme; q->bps_payload = ll_modem_bps(m);
    q->payload_len = payload_len; q->check = check; q->fec0 = fec0; q->fec1 = fec1;
    ll_packetizer_destroy(q->p_payload);
    q->p_payload = ll_packetizer_create(payload_len, (int)check, (int)fec0, (int)fec1);
    q->payload_enc_len = ll_packetizer_enc_len(q->p_payload);
    unsigned nb = 8 * q->payload_enc_len;
    q->payload_mod_len = nb / q->bps_payload + ((nb % q->bps_payload) ? 1 : 0);
    q->payload_enc = (unsigned char *)realloc(q->payload_enc, 8 * (size_t)q->payload_enc_len + 64);
    memset(q->payload_enc, 0, 8 * (size_t)q->payload_enc_len + 64);
    q->payload_dec = (unsigned char *)realloc(q->payload_dec, payload_len + 8);
    q->payload_syms = (ll_cf *)realloc(q->payload_syms, sizeof(ll_cf) * (q->payload_mod_len + 1));
}

static void fs_fill_stats(ll_ofdmflexframesync q, int with_payload)
{
    q->stats.rssi = ll_ofdmframesync_get_rssi(q->fs);
    q->stats.cfo = ll_ofdmframesync_get_cfo(q->fs);
    if (with_payload) {
        q->stats.framesyms = q->payload_syms; q->stats.num_framesyms = q->payload_mod_len;
        q->stats.mod_scheme = q->ms_payload; q->stats.mod_bps = q->bps_payload;
        q->stats.check = q->check; q->stats.fec0 = q->fec0; q->stats.fec1 = q->fec1;
    } else {
        q->stats.framesyms = NULL; q->stats.num_framesyms = 0;
        q->stats.mod_scheme = LL_MODEM_UNKNOWN; q->stats.mod_bps = 0;
        q->stats.check = LL_CRC_UNKNOWN; q->stats.fec0 = LL_FEC_UNKNOWN; q->stats.fec1 = LL_FEC_UNKNOWN;
    }
}

static void fs_rxheader(ll_ofdmflexframesync q, ll_cf *X)
{
    for (unsigned i = 0; i < q->M; i++) {
        if (q->p[i] != LL_SCTYPE_DATA) continue;
        unsigned sym = ll_modem_demodulate(q->mod_header, X[i]);
        q->header_mod[q->header_symbol_index++] = (unsigned char)sym;
        float evm = ll_modem_get_evm(q->mod_header);
        q->evm_hat += evm * evm;
        if (q->header_symbol_index == FLEX_H_SYM) {
            fs_decode_header(q);
            q->stats.evm = 10.0f * log10f(q->evm_hat / (float)FLEX_H_SYM);
            if (q->header_valid) q->state = FS_PAYLOAD;
            else {
                fs_fill_stats(q, 0);
                if (q->cb) q->cb(q->header, q->header_valid, NULL, 0, 0, q->stats, q->ud);
                ll_ofdmflexframesync_reset(q);
            }
            break;
        }
    }
}

/* write the low `b` bits of sym at bit offset k of an MSB-first byte array of n bytes */
static void pack_array(unsigned char *dst, unsigned n, unsigned k, unsigned b, unsigned sym)
{
    for (unsigned i = 0; i < b; i++) {
        unsigned bit = (sym >> (b - 1 - i)) & 1, pos = k + i;
        if (pos / 8 >= n) return;
        if (bit) dst[pos / 8] |= (unsigned char)(0x80u >> (pos % 8));
        else     dst[pos / 8] &= (unsigned char)~(0x80u >> (pos % 8));
    }
}

static void fs_rxpayload(ll_ofdmflexframesync q, ll_cf *X)
{
    for (unsigned i = 0; i < q->M; i++) {
        if (q->p[i] != LL_SCTYPE_DATA) continue;
        q->payload_syms[q->payload_symbol_index] = X[i];
        if (q->payload_soft) {
            unsigned char soft[8];
            ll_modem_demodulate_soft(q->mod_payload, X[i], soft);
            for (unsigned k = 0; k < q->bps_payload; k++) {
                unsigned pos = q->payload_buffer_index + k;
                if (pos < 8 * q->payload_enc_len) q->payload_enc[pos] = soft[k];
            }
        } else {
            unsigned sym = ll_modem_demodulate(q->mod_payload, X[i]);
            pack_array(q->payload_enc, q->payload_enc_len, q->payload_buffer_index, q->bps_payload, sym);
        }
        q->payload_buffer_index += q->bps_payload;
        q->payload_symbol_index++;
        if (q->payload_symbol_index == q->payload_mod_len) {
            q->payload_valid = q->payload_soft
                ? ll_packetizer_decode_soft(q->p_payload, q->payload_enc, q->payload_dec)
                : ll_packetizer_decode(q->p_payload, q->payload_enc, q->payload_dec);
            fs_fill_stats(q, 1);
            if (q->cb) q->cb(q->header, q->header_valid, q->payload_dec, q->payload_len, q->payload_valid, q->stats, q->ud);
            ll_ofdmflexframesync_reset(q);
            break;
        }
    }
}

static int fs_internal_callback(ll_cf *X, const unsigned char *p, unsigned M, void *ud)
{
    (void)p; (void)M;
    ll_ofdmflexframesync q = (ll_ofdmflexframesync)ud;
    if (q->state == FS_HEADER) fs_rxheader(q, X);
    else fs_rxpayload(q, X);
    return 0;
}
