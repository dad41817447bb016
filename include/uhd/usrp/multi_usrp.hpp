// uhd/usrp/multi_usrp.hpp -- synthetic-IQ stand-in for the slice of the UHD API that
// liquid-usrp's src/multichannel_rx.cc uses (:121-141,161-162,176,186-199,220).  No radio:
// recv() replays a raw cf32 file named by $MCRX_IQ_FILE in a loop (zeros if unset), one
// "packet" of $MCRX_IQ_PACKET (default 4096) samples per call.  The transmit side used by
// src/multichannel_tx.cc (:102-121,154-158,202-207,213-218) appends what send() is given to the raw
// cf32 file named by $MCTX_IQ_FILE and ends the process (exit status 0) once $MCTX_IQ_SAMPLES
// (default 2^20) samples have gone out -- the reference's transmit loop has no exit of its own.
// With $MCTX_LOOPBACK=1 send() feeds an in-process FIFO instead and recv() drains it (idle: 1 ms nap, then a
// short packet of zeros), so a transceiver object hears its own transmissions -- the stand-in for the second
// radio of src/multichannel_txrx.cc ($MCTX_TEE_FILE additionally records what recv() handed out).
// Own code, header only.
#ifndef LIQUID_USRP_AMD_UHD_SHIM_HPP
#define LIQUID_USRP_AMD_UHD_SHIM_HPP

#include <complex>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <vector>
#include <unistd.h>

namespace uhd {

struct stream_cmd_t {
    enum stream_mode_t { STREAM_MODE_START_CONTINUOUS = 'a', STREAM_MODE_STOP_CONTINUOUS = 'o' };
    stream_mode_t stream_mode;
    bool stream_now;
    stream_cmd_t(stream_mode_t m) : stream_mode(m), stream_now(true) {}
};

struct device_addr_t { std::string args; };

struct rx_metadata_t {
    enum error_code_t { ERROR_CODE_NONE = 0x0, ERROR_CODE_TIMEOUT = 0x1, ERROR_CODE_OVERFLOW = 0x8 };
    error_code_t error_code;
    rx_metadata_t() : error_code(ERROR_CODE_NONE) {}
};

struct tx_metadata_t {
    bool start_of_burst, end_of_burst, has_time_spec;
    tx_metadata_t() : start_of_burst(false), end_of_burst(false), has_time_spec(false) {}
};

struct io_type_t { enum tid_t { COMPLEX_FLOAT32 = 'f' }; };

// process-wide loopback FIFO ($MCTX_LOOPBACK=1)
struct loop_fifo {
    std::mutex mu; std::vector<std::complex<float> > q; size_t rd; FILE *tee; bool tee_checked, in_burst;
    loop_fifo() : rd(0), tee(NULL), tee_checked(false), in_burst(false) {}
    ~loop_fifo() { if (tee) fclose(tee); }
    void end_burst() { std::lock_guard<std::mutex> lk(mu); in_burst = false; }
    bool bursting() { std::lock_guard<std::mutex> lk(mu); return in_burst; }
    void push(const std::complex<float> *x, size_t n)
    {
        std::lock_guard<std::mutex> lk(mu);
        if (n) in_burst = true;                                 // until a packet flagged end_of_burst
        if (q.size() - rd > ((size_t)1 << 27)) return;          // nobody is listening: drop
        if (rd > ((size_t)1 << 22)) { q.erase(q.begin(), q.begin() + rd); rd = 0; }
        q.insert(q.end(), x, x + n);
    }
    size_t pop(std::complex<float> *x, size_t n)
    {
        std::lock_guard<std::mutex> lk(mu);
        size_t have = q.size() - rd; if (n > have) n = have;
        if (n) memcpy((void *)x, &q[0] + rd, n * sizeof(*x));
        rd += n;
        return n;
    }
    void record(const std::complex<float> *x, size_t n)         // optional capture of what the receiver was handed
    {
        std::lock_guard<std::mutex> lk(mu);
        if (!tee_checked) { tee_checked = true; if (const char *f = getenv("MCTX_TEE_FILE")) tee = fopen(f, "wb"); }
        if (tee) fwrite(x, sizeof(*x), n, tee);
    }
};
inline loop_fifo &loopback() { static loop_fifo f; return f; }
inline bool loopback_enabled() { const char *e = getenv("MCTX_LOOPBACK"); return e && *e && *e != '0'; }

class device {
public:
    enum recv_mode_t { RECV_MODE_FULL_BUFF = 0, RECV_MODE_ONE_PACKET = 1 };
    enum send_mode_t { SEND_MODE_FULL_BUFF = 0, SEND_MODE_ONE_PACKET = 1 };
    typedef std::shared_ptr<device> sptr;
    device() : pos(0), packet(4096), txfp(NULL), txsent(0), txlimit((size_t)1 << 20)
    {
        if (const char *e = getenv("MCTX_IQ_SAMPLES")) txlimit = (size_t)atol(e);
        if (const char *e = getenv("MCRX_IQ_PACKET")) packet = (size_t)atol(e);
        if (const char *f = getenv("MCRX_IQ_FILE")) {
            if (FILE *fp = fopen(f, "rb")) {
                fseek(fp, 0, SEEK_END); long n = ftell(fp); fseek(fp, 0, SEEK_SET);
                iq.resize((size_t)n / sizeof(std::complex<float>));
                if (fread(iq.data(), sizeof(std::complex<float>), iq.size(), fp) != iq.size()) iq.clear();
                fclose(fp);
            } else fprintf(stderr, "uhd shim: cannot open %s\n", f);
        }
    }
    ~device() { if (txfp) fclose(txfp); }
    size_t get_max_recv_samps_per_packet() const { return packet; }
    size_t recv(void *buff, size_t n, rx_metadata_t &md, io_type_t::tid_t, recv_mode_t)
    {
        std::complex<float> *out = static_cast<std::complex<float> *>(buff);
        if (n > packet) n = packet;
        md.error_code = rx_metadata_t::ERROR_CODE_NONE;
        if (loopback_enabled()) {
            size_t got = loopback().pop(out, n);
            // mid-burst the transmitter is merely a little behind: the air is continuous, wait for it (at most 2 s)
            for (int tries = 0; !got && tries < 10000 && loopback().bursting(); tries++) { usleep(200); got = loopback().pop(out, n); }
            if (!got) {                                         // idle air: a short packet of silence per millisecond
                usleep(1000);
                got = n < 64 ? n : 64;
                memset((void *)out, 0, got * sizeof(*out));
            }
            loopback().record(out, got);
            return got;
        }
        if (iq.empty()) { memset((void *)out, 0, n * sizeof(*out)); return n; }
        for (size_t i = 0; i < n; i++) { out[i] = iq[pos]; if (++pos == iq.size()) pos = 0; }
        return n;
    }
    size_t send(const void *buff, size_t n, const tx_metadata_t &md, io_type_t::tid_t, send_mode_t)
    {
        if (loopback_enabled()) {
            loopback().push(static_cast<const std::complex<float> *>(buff), n);
            if (md.end_of_burst) loopback().end_burst();
            return n;
        }
        if (!txfp) { const char *f = getenv("MCTX_IQ_FILE"); txfp = fopen(f ? f : "/dev/null", "wb"); }
        if (n > txlimit - txsent) n = txlimit - txsent;
        if (txfp && n) fwrite(buff, sizeof(std::complex<float>), n, txfp);
        txsent += n;
        if (txsent >= txlimit) { if (txfp) fclose(txfp); printf("uhd shim: %zu samples sent\n", txsent); exit(0); }
        return n;
    }
private:
    std::vector<std::complex<float> > iq;
    size_t pos, packet;
    FILE *txfp; size_t txsent, txlimit;
};

namespace usrp {
class multi_usrp {
public:
    typedef std::shared_ptr<multi_usrp> sptr;
    static sptr make(const device_addr_t &) { return sptr(new multi_usrp()); }
    multi_usrp() : dev(new device()), rate(0), txrate(0) {}
    void set_rx_rate(double r) { rate = r; }
    double get_rx_rate() const { return rate; }
    void set_rx_freq(double) {}
    void set_rx_gain(double) {}
    void set_tx_rate(double r) { txrate = r; }
    double get_tx_rate() const { return txrate; }
    void set_tx_freq(double) {}
    void set_tx_gain(double) {}
    void set_tx_antenna(const std::string &) {}
    void set_rx_antenna(const std::string &) {}
    device::sptr get_device() { return dev; }
    void issue_stream_cmd(const stream_cmd_t &) {}
private:
    device::sptr dev;
    double rate, txrate;
};
}  // namespace usrp
}  // namespace uhd
#endif
