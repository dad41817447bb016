// blocking_test.cc -- exercises ofdmtxrx's "blocking" receiver worker (lib/ofdmtxrx.cc:642-739; second constructor with
// _blocking_rx_worker = true) through its public handshake: the worker fills *rx_buffer under rx_buffer_mutex, signals
// rx_buffer_filled_cond and waits on rx_buffer_modified_cond; an editor thread changes the samples before they reach the
// synchronizer.  Run with MCTX_LOOPBACK=1 (transmitter looped into the receiver by the UHD stand-in).
//   phase 1: the editor turns every sample by 180 degrees -- invisible to a receiver that estimates the channel: all frames arrive
//   phase 2: the editor zeroes every sample -- nothing arrives although the transmitter keeps sending
// Prints "phase1 <sent> <received valid>  phase2 <sent> <received>".  tests/test_gpu_refapp.py runs it.
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <time.h>
#include <unistd.h>
#include "ofdmtxrx.h"

static std::atomic<int> g_valid(0), g_mode(0), g_stop(0);
static long g_edited = 0;

static int callback(unsigned char *, int header_valid, unsigned char *, unsigned int, int payload_valid, framesyncstats_s, void *)
{
    if (header_valid && payload_valid) g_valid++;
    return 0;
}

static void editor(ofdmtxrx *t)
{
    while (!g_stop) {
        pthread_mutex_lock(&t->rx_buffer_mutex);
        struct timespec ts; clock_gettime(CLOCK_REALTIME, &ts);
        ts.tv_nsec += 2000000; if (ts.tv_nsec >= 1000000000) { ts.tv_nsec -= 1000000000; ts.tv_sec++; }
        // (the reference's protocol has no predicate: a signal sent while nobody waits is lost, so do not wait for ever)
        pthread_cond_timedwait(&t->rx_buffer_filled_cond, &t->rx_buffer_mutex, &ts);
        if (t->rx_buffer) {
            const float g = g_mode == 0 ? -1.0f : 0.0f;
            for (auto &v : *t->rx_buffer) v *= g;
            g_edited++;
        }
        pthread_cond_signal(&t->rx_buffer_modified_cond);
        pthread_mutex_unlock(&t->rx_buffer_mutex);
    }
}

int main()
{
    ofdmtxrx txcvr(64, 8, 4, NULL, callback, NULL, true);
    txcvr.set_tx_gain_soft(-6.0f);
    txcvr.debug_enable();
    std::thread ed(editor, &txcvr);
    usleep(20000);
    txcvr.start_rx();
    unsigned char header[8] = {0}, payload[200];
    for (unsigned i = 0; i < sizeof(payload); i++) payload[i] = (unsigned char)(i * 7 + 1);
    const int nsend = 12;
    for (int i = 0; i < nsend; i++) { header[1] = (unsigned char)i; txcvr.transmit_packet(header, payload, sizeof(payload), LIQUID_MODEM_QPSK, LIQUID_FEC_NONE, LIQUID_FEC_HAMMING128); usleep(2000); }
    usleep(300000);
    txcvr.stop_rx();                                // (frames surface per receive batch and when the receiver stops)
    const int got1 = g_valid;
    g_mode = 1;
    txcvr.start_rx();
    for (int i = 0; i < nsend; i++) { header[1] = (unsigned char)(100 + i); txcvr.transmit_packet(header, payload, sizeof(payload), LIQUID_MODEM_QPSK, LIQUID_FEC_NONE, LIQUID_FEC_HAMMING128); usleep(2000); }
    usleep(300000);
    const int got2 = g_valid - got1;
    txcvr.stop_rx();
    g_stop = 1;
    ed.join();
    printf("phase1 %d %d  phase2 %d %d  packets_edited %ld\n", nsend, got1, nsend, got2, g_edited);
    return 0;
}
