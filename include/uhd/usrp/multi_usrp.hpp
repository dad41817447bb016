// uhd/usrp/multi_usrp.hpp -- synthetic-IQ stand-in for the slice of the UHD API that
// liquid-usrp's src/multichannel_rx.cc uses (:121-141,161-162,176,186-199,220).  No radio:
// recv() replays a raw cf32 file named by $MCRX_IQ_FILE in a loop (zeros if unset), one
// "packet" of $MCRX_IQ_PACKET (default 4096) samples per call.  The transmit side used by
// src/multichannel_tx.cc (:102-121,154-158,202-207,213-218) appends what send() is given to the raw
// cf32 file named by $MCTX_IQ_FILE and ends the process (exit status 0) once $MCTX_IQ_SAMPLES
// (default 2^20) samples have gone out -- the reference's transmit loop has no exit of its own.
// Own code, header only.
#ifndef LIQUID_USRP_AMD_UHD_SHIM_HPP
#define LIQUID_USRP_AMD_UHD_SHIM_HPP

#include <complex>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
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
        if (iq.empty()) { memset((void *)out, 0, n * sizeof(*out)); return n; }
        for (size_t i = 0; i < n; i++) { out[i] = iq[pos]; if (++pos == iq.size()) pos = 0; }
        return n;
    }
    size_t send(const void *buff, size_t n, const tx_metadata_t &, io_type_t::tid_t, send_mode_t)
    {
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
