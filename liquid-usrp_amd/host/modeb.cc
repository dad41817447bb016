// modeb.cc -- timing mode B of the measurement plan: the drop-in call itself, multichannelrx::Execute(buf, n)
// on a buffer in ordinary host memory, host-to-device copies and callback delivery included.  Not the headline
// number (bench.py times the path with the samples resident in HBM); reported in DESIGN.md next to it.
//   modeb [num_channels=512] [frames_per_channel=8] [reps=3]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include <hip/hip_runtime_api.h>

#include "multichannelrx.h"
#include "mcrx_hip.h"

static unsigned long long g_frames, g_valid, g_bytes;
static int on_frame(unsigned char *, int, unsigned char *, unsigned int _payload_len, int _payload_valid, framesyncstats_s, void *)
{
    g_frames++;
    if (_payload_valid) { g_valid++; g_bytes += _payload_len; }
    return 0;
}

int main(int argc, char **argv)
{
    const unsigned N = argc > 1 ? atoi(argv[1]) : 512, frames = argc > 2 ? atoi(argv[2]) : 8, reps = argc > 3 ? atoi(argv[3]) : 3;
    const unsigned M = 64, cp = 8, taper = 4, plen = 1200;
    mctx_hip_t tx;
    if (mctx_hip_create(&tx, N, M, cp, taper, NULL) != MCRX_OK) { fprintf(stderr, "%s\n", mctx_hip_last_error()); return 1; }
    size_t nb = mctx_hip_blocks_for(tx, frames, plen, LIQUID_MODEM_QPSK, LIQUID_FEC_NONE, LIQUID_FEC_HAMMING128);
    size_t n = nb * 2 * N;
    void *d_iq = NULL;
    if (hipMalloc(&d_iq, n * 8) != hipSuccess) return 1;
    if (mctx_hip_generate(tx, d_iq, nb, frames, plen, LIQUID_MODEM_QPSK, LIQUID_FEC_NONE, LIQUID_FEC_HAMMING128, 1.0f / N, 0xC0FFEE,
                          NULL, NULL, NULL) != MCRX_OK) { fprintf(stderr, "%s\n", mctx_hip_last_error()); return 1; }
    std::vector<std::complex<float> > x(n);                    // ordinary pageable memory, like an application's
    if (hipMemcpy(x.data(), d_iq, n * 8, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    (void)hipFree(d_iq); mctx_hip_destroy(tx);

    std::vector<void *> ud(N, (void *)NULL);
    std::vector<framesync_callback> cb(N, on_frame);
    double dt;
    {
        multichannelrx rx(N, M, cp, taper, NULL, ud.data(), cb.data());
        rx.Execute(x.data(), (unsigned int)n);                  // warm-up: allocations
        rx.Flush();
        g_frames = g_valid = g_bytes = 0;
        auto t0 = std::chrono::steady_clock::now();
        for (unsigned r = 0; r < reps; r++) rx.Execute(x.data(), (unsigned int)n);
        rx.Flush();
        dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    printf("{\"mode\": \"B: multichannelrx::Execute(host buffer), H2D and callbacks included\", \"channels\": %u, \"samples\": %zu, "
           "\"seconds\": %.4f, \"Msamples_per_s\": %.1f, \"GB_per_s_host_to_device\": %.2f, \"frames\": %llu, \"frames_valid\": %llu, "
           "\"frames_expected\": %u}\n", N, n * reps, dt, n * reps / dt / 1e6, n * reps * 8.0 / dt / 1e9, g_frames, g_valid, N * frames * reps);
    return g_valid == (unsigned long long)N * frames * reps ? 0 : 2;
}
