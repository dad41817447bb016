/*
 * mcrx_hip.h -- C-ABI of the MI355X-native multichannel OFDM receiver (libmcrx_hip.so).
 *
 * Drop-in boundary for liquid-usrp's `multichannelrx` hot path.  Each entry point names
 * the reference interface it replaces (paths relative to the liquid-usrp tree):
 *
 *   mcrx_hip_create          multichannelrx::multichannelrx        lib/multichannelrx.cc:45-104
 *                            (= N x ofdmflexframesync_create :82, firpfbch_crcf_create_kaiser :91,
 *                             nco_crcf_create/set_frequency :99-100)
 *   mcrx_hip_destroy         multichannelrx::~multichannelrx       lib/multichannelrx.cc:107-132
 *   mcrx_hip_reset           multichannelrx::Reset                 lib/multichannelrx.cc:135-153
 *   mcrx_hip_execute_host    multichannelrx::Execute               lib/multichannelrx.cc:155-182
 *   mcrx_hip_execute_device  same, IQ already resident in HBM (synthetic source; replaces the
 *                            UHD recv loop of src/multichannel_rx.cc:184-212)
 *   mcrx_hip_flush / _next_frame
 *                            framesync_callback delivery           include/multichannelrx.h:45,
 *                                                                  src/multichannel_rx.cc:37-66
 *   mcrx_hip_channelize      nco_crcf_mix_down + firpfbch_crcf_analyzer_execute
 *                                                                  lib/multichannelrx.cc:163-164,188
 *   mcrx_hip_sync            N x ofdmflexframesync_execute         lib/multichannelrx.cc:193-194
 *
 * Plain C: pointers and sizes only.  All functions return 0 on success or a negative
 * MCRX_E* code; nothing throws across this boundary.  One caller thread per handle
 * (the reference classes are not thread-safe either).
 */
#ifndef MCRX_HIP_H
#define MCRX_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MCRX_OK          0
#define MCRX_EINVAL     -1      /* bad argument (reference: fprintf + throw 0) */
#define MCRX_ENOMEM     -2
#define MCRX_EHIP       -3      /* HIP runtime error; see mcrx_hip_last_error() */
#define MCRX_EUNSUPP    -4      /* configuration outside the kernels' supported set */
#define MCRX_EOVERFLOW  -5      /* frame pool exhausted: frames were dropped */
#define MCRX_EBUSY      -6      /* multichanneltx::UpdateData on a channel whose frame is still going out */

#define MCRX_TILE 16            /* time samples per (channel, tile) granule = 128 bytes: one channel per cache line, so a
                                   synchronizer that reads one channel's time series never drags a neighbour channel's samples in */

typedef struct mcrx_hip_s *mcrx_hip_t;

typedef struct {
    uint32_t struct_size;        /* sizeof(mcrx_hip_config) */
    uint32_t max_payload_len;    /* largest decodable payload [bytes]; 0 -> 2048.  The per-frame buffers hold coded frames of up to
                                    4 (max_payload_len + 4) + 16 bytes (two rate-1/2 codes); a frame that codes to more -- rep5 expands
                                    five-fold -- or carries a longer payload is reported with payload_valid = 0 */
    uint32_t max_frames;         /* frame records kept between flushes; 0 -> auto (covers one host batch of
                                    the shortest possible frames on every channel) */
    uint32_t payload_soft;       /* 1 = soft-decision payload decoding (default), 0 = hard.  (Hard decisions of the Hamming(8,4) code: a double
                                    error -- a word at distance 2 from several codewords -- decodes to the lowest symbol at that distance; liquid-dsp's
                                    own table for that case is not visible from the reference, so such frames, which fail their CRC either way, may
                                    carry other bytes than upstream's: DESIGN.md section 2, D8) */
    uint32_t slab_blocks;        /* channelizer blocks per workgroup slab; 0 -> auto */
    uint32_t channel_first;      /* synchronizer shard: first channel ...          */
    uint32_t channel_count;      /* ... and count handled by this handle; 0 -> all  */
    uint32_t batch_samples;      /* execute_host staging size [wideband samples]; 0 -> auto */
    uint32_t single_channel;     /* 1 = no channelizer: num_channels must be 1 and the samples pushed are that
                                    channel's own stream, i.e. one ofdmflexframesync (lib/ofdmtxrx.cc:91,620-626);
                                    samples are consumed MCRX_TILE at a time */
    uint32_t serial;             /* 1 = every kernel of a push runs in order on the caller's stream (profiling, debugging).
                                    Default 0: the handle overlaps its own stages -- channelizer, acquisition and payload/decode
                                    kernels of consecutive pushes run on three internal streams with three sets of buffers */
    uint32_t chunk_blocks;       /* execute_device: split a push into sub-slabs of this many blocks (rounded up to whole tiles) so that
                                    the stages of ONE call overlap too; 0 = one launch sequence per call */
    uint32_t defer_samples;      /* > 0: every push keeps this many channel-rate samples of history (plus a symbol) in front of its
                                    channel tiles, and a frame that begins less than that before the end of a push and does not
                                    end in it is acquired again by the next push -- whole, by the parallel path -- instead of
                                    being walked symbol by symbol across the boundary by one wave.  0 = off.  Stage-level callers
                                    (mcrx_hip_sync) must supply mcrx_hip_history_tiles() tiles of history themselves */
    uint32_t front_end;          /* 0 = the reference's analysis bank: critically sampled firpfbch, 2N channels, lower N kept
                                    (lib/multichannelrx.cc:89-91).  1 = the oversampled bank BASELINE.json names: firpfbch2 with 2N
                                    channels (twice the channel rate, prototype cut off at the neighbour's centre) followed per kept
                                    channel by a half-band decimator back to the channel rate -- less aliasing at the channel edges.
                                    Oscillator, bank and decimator run as ONE kernel (the chain is a critically sampled bank with a
                                    28-tap composite prototype per column: csrc/channelizer.hip): 12 algorithmic bytes per sample like
                                    front_end = 0, 27 blocks of filter history instead of 13 (mcrx_hip_history_blocks).
                                    2 = the same chain stage by stage (oscillator pass, bank at twice the rate, adapter: three kernels,
                                    52 bytes per sample) -- the form the oracle computes, kept as the cross-check of 1; execute_host /
                                    execute_device only.  Power-of-two channel counts for 1 and 2 */
    uint32_t skip_framesyms;     /* 1 = harvests leave the equalised symbols in HBM (frames report num_framesyms = 0): 1.2 KB
                                    instead of 59 KB per frame over the host link at the benchmark's frame size; 2 = the payload workers
                                    of 64-subcarrier symbols do not even store them (a third of the bytes that stage moves): for callers
                                    whose callback never reads stats.framesyms (the reference's src/multichannel_rx.cc:37-66 does not) */
    /* Alternate builds of the same stages.  Every one decodes the same frames (the -m gpu tests hold each to the oracle); they
     * differ in speed only and exist so that a regression in the default can be told from one in the algorithm.  0 = default. */
    uint32_t worker_build;       /* M = 64 payload workers: 0 = lean workers, butterfly exchanges through the LDS crossbar;
                                    1 = lean workers with the exchanges on the VALU (DPP / permlane swaps); 2 = the round-2 worker, one
                                    frame per wave; 3 / 4 = that worker with two / four frames per wave; 5 = the width-generic kernel.
                                    M = 128 / 256: 0 = the lean one-frame-per-wave workers of csrc/payload_wide.hpp (round 6), 2 .. 5 = the
                                    width-generic kernel they replace */
    uint32_t acquisition;        /* 0 = segment-parallel acquisition, anchored on the state a push begins in while the traffic has a cadence;
                                    1 = one launch of segment waves, no anchor; 2 = no segment waves (the scouts walk every frame); 3 = an
                                    anchor phase in front of the segment waves, always (rounds 4-5: a launch that acquires every channel's first
                                    frame); 4 = no speculation at all (the round-1 scout: one kernel per push does everything); 5 = as 0 with
                                    the cadence taken for granted */
    uint32_t scout_build;        /* 0 = default: the general state machine's segment waves, the scouts' unbudgeted build; 1 = the scouts'
                                    168-register build of rounds 2-3; 2 = the lean segment waves of 48- / 64-subcarrier symbols
                                    (csrc/acq_lean.hpp: half the instructions, the same time on periodic traffic, 10 % behind on ragged) */
    uint32_t conv_scratch;       /* the K = 7 rate-1/2 decoder's scratch in HBM (512 bytes per trellis step and decoder wave: min(max_frames, 2048) waves x
                                    (32 max_payload_len + ~100) steps = 0.4 GB at 1200-byte payloads, 0.65 GB at the default 2048).  0 = allocated when
                                    the first frame with that code has been seen (the launch after the device reports it; until then such frames go
                                    through the block decoder of rounds 3-4: the same bytes on every frame that decodes, a fixed 192-step overlap
                                    instead of an exact one on frames that do not); 1 = with the handle (callers that know they will receive the code);
                                    2 = never.  An allocation that fails is not an error: the block decoder stays */
} mcrx_hip_config;

/* One decoded frame = the arguments of the reference's framesync_callback
 * (include/multichannelrx.h:45) plus where it ended. */
typedef struct {
    uint32_t channel;
    int32_t  header_valid;
    int32_t  payload_valid;
    uint32_t payload_len;
    uint8_t  header[8];
    float    evm, rssi, cfo;                 /* framesyncstats_s */
    uint32_t mod_scheme, mod_bps, check, fec0, fec1;
    uint32_t num_framesyms;
    uint64_t end_sample;                     /* channel-rate sample index of the last symbol */
    const uint8_t *payload;                  /* payload_len bytes   (host memory owned by the handle, */
    const float   *framesyms;                /* 2*num_framesyms floats   valid until the next flush)  */
} mcrx_frame;

/* ---- receiver object ---------------------------------------------------------------- */
int  mcrx_hip_create(mcrx_hip_t *out, unsigned num_channels, unsigned M, unsigned cp_len,
                     unsigned taper_len, const unsigned char *p, const mcrx_hip_config *cfg);
int  mcrx_hip_destroy(mcrx_hip_t q);
int  mcrx_hip_reset(mcrx_hip_t q);
/* the same for a handle driven through the stage-level calls (mcrx_hip_sync: it never learns the stream position by itself):
 * the synchronizers restart in SEEK at channel-rate sample `chan_position`.  What mcrx_hip_pipeline_reset calls. */
int  mcrx_hip_reset_at(mcrx_hip_t q, uint64_t chan_position);
unsigned mcrx_hip_num_channels(mcrx_hip_t q);

/* push wideband cf32 samples (interleaved re,im).  Any n; partial blocks are buffered.  MCRX_EOVERFLOW: the
 * samples were all processed but the frame pool filled up and frames were dropped (mcrx_hip_frames_dropped). */
int  mcrx_hip_execute_host(mcrx_hip_t q, const float *iq, size_t nsamples);
/* same with the samples already in device memory; `stream` is a hipStream_t (NULL = default).
 * nsamples must be a multiple of 2*num_channels.  Asynchronous, with a bounded lead: the call returns when the work is
 * enqueued, but waits first (once per turn of the handle's buffer sets) for the launch one turn back, so the host stays
 * within 2 * MCRX_SLOTS launches of the device and the acquisition's feedback words reach the next launches. */
int  mcrx_hip_execute_device(mcrx_hip_t q, const void *d_iq, size_t nsamples, void *stream);

/* wait for all pushed samples, gather decoded frames (ordered by end time, then channel). */
int  mcrx_hip_flush(mcrx_hip_t q);
/* Overlapped harvest for a stream that keeps coming (callbacks inside Execute, lib/multichannelrx.cc:193-194, at
 * slab granularity): gathers the frames of everything pushed before the poll BEFORE THE PREVIOUS one -- waiting only for those
 * launches -- and marks what was pushed since.  With one push + one poll per slab, slab k-2's frames cross the host link while
 * the GPU has slabs k-1 and k in flight (two, so that consecutive slabs keep overlapping on the device); flush delivers the rest. */
int  mcrx_hip_poll(mcrx_hip_t q);
/* as poll, but the frames are dropped on the device (no host wait, no copy): steady-state benchmarking.  Dropped frames are
 * never delivered by a later poll / flush. */
int  mcrx_hip_discard(mcrx_hip_t q);
/* make `stream` wait (on the device) for everything pushed so far, e.g. before a stage-level buffer is reused */
int  mcrx_hip_stream_wait(mcrx_hip_t q, void *stream);
/* ... or only for synchronizer launch number `launch` (0-based; mcrx_hip_launches() - 1 right after a
 * mcrx_hip_sync / execute_device call): what a rotating stage-level buffer needs before it is overwritten */
uint64_t mcrx_hip_launches(mcrx_hip_t q);
unsigned mcrx_hip_history_tiles(mcrx_hip_t q);      /* tiles of channel-rate history a mcrx_hip_sync buffer must start with */
int  mcrx_hip_stream_wait_launch(mcrx_hip_t q, uint64_t launch, void *stream);
/* frames the per-channel scouts acquired themselves / took over from speculative waves since the last reset of
 * the statistics (synchronises the device) */
int  mcrx_hip_spec_stats(mcrx_hip_t q, uint64_t *walked, uint64_t *adopted, int reset);
/* the K = 7 convolutional decoder (csrc/viterbi_frames.hpp: a frame per wave, a trellis block per lane, each block run from
 * an overlap and CHECKED against its neighbours): frames it decoded, forward passes and traceback passes it had to repeat because a
 * block's survivors had not merged inside the overlap (0 on any decodable signal; noise costs passes, never exactness).  Synchronises
 * the device. */
int  mcrx_hip_viterbi_stats(mcrx_hip_t q, uint64_t *frames, uint64_t *forward_repeats, uint64_t *traceback_repeats, int reset);
size_t mcrx_hip_frames_pending(mcrx_hip_t q);
int  mcrx_hip_next_frame(mcrx_hip_t q, mcrx_frame *out);      /* 1 = frame written, 0 = none */
/* a caller's delivery loop without a per-frame FFI crossing: walks every pending frame through mcrx_hip_next_frame,
 * touches its payload, and reports how many there were, how many with valid header and payload, and their payload bytes */
int  mcrx_hip_drain_count(mcrx_hip_t q, uint64_t *frames, uint64_t *valid, uint64_t *payload_bytes);
uint64_t mcrx_hip_frames_dropped(mcrx_hip_t q);

/* ---- stage level (multi-GPU split, parity tests, benchmarks) ------------------------- */
/* NCO + analysis bank on `nblocks` blocks of 2N samples.  `first_sample` is the absolute
 * index of d_iq[0] (NCO phase); d_halo holds the mcrx_hip_history_blocks() blocks preceding d_iq -- 13 = 2m - 1 for the
 * reference's bank, 27 for front_end = 1 -- (NULL = zeros).
 * Output layout: out[g][tile][c][MCRX_TILE] cf32 with channel = g*(N/groups)+c,
 * tile = block / MCRX_TILE; nblocks must be a multiple of MCRX_TILE. */
int  mcrx_hip_channelize(mcrx_hip_t q, const void *d_iq, size_t nblocks, uint64_t first_sample,
                         const void *d_halo, void *d_out, unsigned groups, void *stream);
/* run the synchronizer bank of this handle's channel shard over d_chan[tile][c][MCRX_TILE],
 * holding channel-rate samples [first_sample, first_sample + nsamples).  The buffer must
 * still contain the M+cp samples preceding the first unconsumed one. */
int  mcrx_hip_sync(mcrx_hip_t q, const void *d_chan, uint64_t first_sample, size_t nsamples, void *stream);

/* benchmark replay: discard undelivered frames and return the object to its
 * post-construction state (sample counters and NCO phase zero), asynchronously on `stream`. */
int  mcrx_hip_restart(mcrx_hip_t q, void *stream);

/* design data, for parity tests against the oracle */
int  mcrx_hip_get_taps(mcrx_hip_t q, float *h, size_t n);             /* p*K prototype taps */
uint32_t mcrx_hip_nco_step(mcrx_hip_t q);                             /* 32-bit phase increment */
unsigned mcrx_hip_history_blocks(mcrx_hip_t q);                       /* blocks of 2N samples of filter history in front of a push: 13 (27: front_end = 1) */
int  mcrx_hip_kernel_time_ms(mcrx_hip_t q, float *channelizer_ms, float *sync_ms); /* last launches */
/* summed HIP-event durations [ms] and launch counts since the last reset of the statistics, per
 * kernel: [0] channelizer_kernel, [1] sync_kernel (per-channel scout), [2] place_jobs_kernel,
 * [3] payload_kernel (per-frame symbol loop), [4] decode_kernel (per-frame packet decode).
 * Events are recorded on the launch stream; no host sync per launch -- and only while timing is switched on
 * (mcrx_hip_kernel_timing; off in a new handle: the five event pairs per push are ten more packets on the handle's
 * streams, 9 % of an 8-channel receiver's 0.25 ms push). */
#define MCRX_NKERNELS 5
int  mcrx_hip_kernel_stats(mcrx_hip_t q, double ms_total[MCRX_NKERNELS], uint64_t launches[MCRX_NKERNELS], int reset);
int  mcrx_hip_kernel_timing(mcrx_hip_t q, int on);                    /* returns the previous setting (-1: null handle) */

const char *mcrx_hip_last_error(void);

/* Devices.  A handle (receiver, resampler, bank, generator, pipeline) belongs to the HIP device that was current when it was created:
 * every entry point that touches the device makes that device current for the call and restores the caller's, and per-kernel device
 * settings (dynamic LDS limits) are made once per device, not once per process -- a host may open handles on several GPUs of one
 * process (the design and the benchmark use one process per GPU).  Device buffers and streams passed in must belong to the handle's device. */
int  mcrx_hip_device(mcrx_hip_t q);                    /* the handle's device index, -1 for a null handle */
int  mcrx_hip_selftest_device_table(void);             /* the once-per-device bookkeeping driven with made-up device ids: 0 = ok (no GPU needed) */

/* ---- multi-stage resampler: decimating front end (0 < rate <= 1) and the transmit side's interpolator
 *      (rate > 1: msresamp_crcf_create(2.0, 60), src/flexframe_tx.cc:170) -------------
 * Replaces msresamp_crcf_create(rate, As) / _execute / _destroy as the reference applications
 * call it in front of a synchronizer (src/flexframe_rx.cc:179,240,275; rate computed as in
 * src/multichannel_rx.cc:129-138).  Buffers are device pointers (cf32).  `stream` is a hipStream_t; NULL = the legacy
 * default stream (so a receiver handle fed next, which orders against that stream, sees the samples written). */
typedef struct msresamp_hip_s *msresamp_hip_t;
int    msresamp_hip_create(msresamp_hip_t *out, float rate, float As);
int    msresamp_hip_destroy(msresamp_hip_t q);
int    msresamp_hip_reset(msresamp_hip_t q);
float  msresamp_hip_get_delay(msresamp_hip_t q);
size_t msresamp_hip_max_output(msresamp_hip_t q, size_t nin);
int    msresamp_hip_execute_device(msresamp_hip_t q, const void *d_in, size_t nin, void *d_out,
                                   size_t out_cap, size_t *nout, void *stream);
const char *msresamp_hip_last_error(void);

/* ---- alternate front end: 2x-oversampled analysis bank --------------------------------
 * Replaces firpfbch2_crcf_create_kaiser(LIQUID_ANALYZER, M, m, As) / _execute / _destroy (liquid-dsp; the
 * channelizer BASELINE.json's north_star names -- liquid-usrp's own receiver uses the critically sampled bank,
 * lib/multichannelrx.cc:89-91).  M channels (a power of two here), every step consumes M/2 samples and
 * produces one sample on each of the M channels.  Stage-level operator on device buffers:
 * d_x[0] is absolute sample first_step * M/2 of the stream and is preceded in memory by `lead_samples`
 * valid samples of history (0 = cold start; 2*m*M covers the whole filter); d_out is [nsteps][M] cf32. */
typedef struct mcrx_hip_pfb2_s *mcrx_hip_pfb2_t;
int    mcrx_hip_pfb2_create(mcrx_hip_pfb2_t *out, unsigned num_channels, unsigned m, float As);
int    mcrx_hip_pfb2_destroy(mcrx_hip_pfb2_t q);
int    mcrx_hip_pfb2_get_taps(mcrx_hip_pfb2_t q, float *h, size_t n);        /* 2*m*M prototype taps */
int    mcrx_hip_pfb2_analyze(mcrx_hip_pfb2_t q, const void *d_x, size_t lead_samples, size_t nsteps,
                             uint64_t first_step, void *d_out, void *stream);
const char *mcrx_hip_pfb2_last_error(void);

/* ---- multi-GPU receive pipeline (one process per GPU) ----------------------------------------
 * The reference runs one multichannelrx on one host thread (lib/multichannelrx.cc:155-195).  Sharded: sub-slabs of
 * `sub_blocks` blocks of the wideband stream go round robin to the ranks (sub-slab u -> rank u % world); per round every
 * rank channelizes its sub-slab into per-destination groups, one grouped ncclSend / ncclRecv exchange over xGMI turns the
 * time shards into channel shards, and the rank's handle -- created with channel_first / channel_count = its shard of
 * N / world channels and defer_samples covering a frame -- synchronizes them; rounds overlap
 * (channelize(c+1) || exchange(c) || synchronizers(c-1)) on two streams of the pipeline's own plus the handle's, events only,
 * nothing waits on the host.  world == 1 is the same
 * code without the exchange.  RCCL is loaded at run time (librccl.so.1); the caller hands rank 0's 128-byte ncclUniqueId to
 * every rank (MPI_Bcast, a file, torch.distributed).  Frames surface through the handle (poll / flush / next_frame) on the
 * rank that owns their channel.  liquid-usrp_amd/sharding.py is the Python mirror of the same schedule. */
typedef struct mcrx_hip_pipeline_s *mcrx_hip_pipeline_t;
int      mcrx_hip_pipeline_unique_id(void *id128);                                   /* rank 0 */
int      mcrx_hip_pipeline_create(mcrx_hip_pipeline_t *out, mcrx_hip_t rx, int rank, int world, const void *unique_id128,
                                  size_t sub_blocks, unsigned nbuf /* rotating buffer sets, 0 = 3 */);
/* one round: d_iq_sub = this rank's sub-slab (sub_blocks * 2N cf32 in HBM), d_halo = the 13 blocks in front of it in the
 * stream (NULL = zeros: the stream's first sub-slab), after_stream = the hipStream_t that produced them: the round starts behind
 * what is enqueued there now.  NULL is the legacy default stream (it is waited for like any other: the pipeline's own streams are
 * non-blocking); MCRX_STREAM_READY = the buffers are complete, wait for nothing */
#define MCRX_STREAM_READY ((void *)(intptr_t)-1)
int      mcrx_hip_pipeline_push(mcrx_hip_pipeline_t p, const void *d_iq_sub, const void *d_halo, void *after_stream);
/* the same from host memory: iq_with_halo = the 13 blocks in front of this rank's sub-slab, then the sub-slab ((13 + sub_blocks) * 2N
 * cf32, contiguous; zeros in front of the stream's first sub-slab).  What the sharded multichannelrx class calls */
int      mcrx_hip_pipeline_push_host(mcrx_hip_pipeline_t p, const float *iq_with_halo);
/* Lifetime: iq_with_halo may be reused as soon as push_host returns -- it is copied to a pinned staging buffer before the call comes
 * back.  To skip that host copy, fill the buffer this call hands out ((13 + sub_blocks) * 2N cf32 = *nsamples; pinned; `nbuf` of them
 * rotate) and pass it to the next push_host.  Neither call waits for a round's kernels, only for the host-to-device copy of the push
 * `nbuf` rounds back. */
int      mcrx_hip_pipeline_host_buffer(mcrx_hip_pipeline_t p, float **buf, size_t *nsamples);
/* multichannelrx::Reset() for the sharded receiver (lib/multichannelrx.cc:135-153): called by every rank at the same point of the
 * stream.  The oscillator is not reset (:144) and runs on the samples of the stream, so the caller says how the samples it pushed
 * differ from the samples it was handed: extra_samples > 0 = handed but not pushed (a discarded partial round), < 0 = pushed but
 * never in the stream (zero padding that completed the last round so that everything before the Reset got synchronized).
 * Waits for everything pushed; decoded frames stay deliverable (poll / flush on the handle). */
int      mcrx_hip_pipeline_reset(mcrx_hip_pipeline_t p, int64_t extra_samples);
/* ranks of the RCCL communicator behind the exchange as RCCL counts them (ncclCommCount); 1 when world == 1 (no communicator);
 * -1 when the loaded RCCL does not export the call */
int      mcrx_hip_pipeline_comm_count(mcrx_hip_pipeline_t p);
int      mcrx_hip_pipeline_wait(mcrx_hip_pipeline_t p);                              /* host wait for everything pushed */
int      mcrx_hip_pipeline_time_exchange(mcrx_hip_pipeline_t p, int on);            /* HIP events around every exchange */
int      mcrx_hip_pipeline_exchange_ms(mcrx_hip_pipeline_t p, double *total_ms, uint64_t *rounds, int reset);
uint64_t mcrx_hip_pipeline_bytes_sent_per_round(mcrx_hip_pipeline_t p);             /* to OTHER ranks */
int      mcrx_hip_pipeline_destroy(mcrx_hip_pipeline_t p);
const char *mcrx_hip_pipeline_last_error(void);

/* ---- synthetic IQ source: multichanneltx on the GPU ------------------------------------
 * Replaces multichanneltx (lib/multichanneltx.cc:41-242: N x ofdmflexframegen -> 2N-channel
 * synthesis bank, m = 13 -> NCO mix-up) driven by the traffic loop of src/multichannel_tx.cc:
 * 163-213: frames back to back on every channel, header = [pid_hi, pid_lo, channel, 5 seeded
 * bytes], seeded payloads, soft gain.  The stream is written to device memory. */
typedef struct mctx_hip_s *mctx_hip_t;
int    mctx_hip_create(mctx_hip_t *out, unsigned num_channels, unsigned M, unsigned cp_len,
                       unsigned taper_len, const unsigned char *p);
int    mctx_hip_destroy(mctx_hip_t q);
/* blocks of 2N samples needed for `frames_per_channel` frames plus the filter tail (multiple of MCRX_TILE) */
size_t mctx_hip_blocks_for(mctx_hip_t q, unsigned frames_per_channel, unsigned payload_len,
                           int mod, int fec0, int fec1);
/* writes nblocks*2N cf32 samples to d_iq; the headers / payloads that were sent are returned in
 * hdr[ch][frame][8] and pay[ch][frame][payload_len] (host buffers, may be NULL) */
int    mctx_hip_generate(mctx_hip_t q, void *d_iq, size_t nblocks, unsigned frames_per_channel,
                         unsigned payload_len, int mod, int fec0, int fec1, float gain, uint32_t seed,
                         uint8_t *hdr, uint8_t *pay, void *stream);
/* Ragged traffic, the kind src/multichannel_txrx.cc:227-267 sends (every packet `rand() % payload_len` bytes, handed to
 * whichever channel is free): per channel, seeded, frames of len_lo .. len_hi payload bytes, 0 .. gap_max idle OFDM symbols
 * before each, with probability 1 / long_every (0: never) a silence of 16 .. 16 + long_max symbols instead; placed until the
 * stream is full.  Host outputs (may be NULL), sized for max_frames per channel: count[ch], hdr[ch][f][8], len[ch][f],
 * pay[ch][f][len_hi], start[ch][f] = block index of the frame's first sample. */
int    mctx_hip_generate_ragged(mctx_hip_t q, void *d_iq, size_t nblocks, unsigned max_frames, unsigned len_lo, unsigned len_hi,
                                unsigned gap_max, unsigned long_every, unsigned long_max, int mod, int fec0, int fec1,
                                float gain, uint32_t seed, uint32_t *count, uint8_t *hdr, uint32_t *len, uint8_t *pay,
                                uint64_t *start, void *stream);
/* Sharded form (the transmit side of src/multichannel_txrx.cc over several GPUs; mirror image of the receiver's
 * stage interface): frame generators are channel-sharded (multichanneltx.cc:230-242 steps N independent
 * ofdmflexframegen objects), the synthesis bank + oscillator (multichanneltx.cc:192-227) time-sharded.
 *   traffic_create    frames of channels [ch_first, ch_first+ch_count), the traffic recipe and seeds of
 *                     mctx_hip_generate; hdr[c][frame][8] / pay[c][frame][payload_len] may be NULL
 *   traffic_tiles     their channel-rate samples for blocks [first_block, first_block+nblocks) as granules
 *                     d_tiles[tile][c][8] (cf32; zeros before block 0 and after the last frame); nblocks % 8 == 0
 *   synthesize_tiles  d_tiles[groups][(lead+nblocks)/8][N/groups][8] (granules as received from `groups` channel
 *                     shards, covering blocks [first_block-lead, first_block+nblocks)) -> wideband samples of blocks
 *                     [first_block-keep, first_block+nblocks) into d_iq; lead >= 25 + keep blocks of filter history,
 *                     lead % 8 == 0.  Equals the same blocks of mctx_hip_generate's stream bit for bit. */
typedef struct mctx_hip_traffic_s *mctx_hip_traffic_t;
int    mctx_hip_traffic_create(mctx_hip_t q, mctx_hip_traffic_t *out, unsigned ch_first, unsigned ch_count,
                               unsigned frames_per_channel, unsigned payload_len, int mod, int fec0, int fec1,
                               uint32_t seed, uint8_t *hdr, uint8_t *pay, void *stream);
int    mctx_hip_traffic_destroy(mctx_hip_traffic_t t);
int    mctx_hip_traffic_tiles(mctx_hip_traffic_t t, long long first_block, size_t nblocks, void *d_tiles, void *stream);
int    mctx_hip_synthesize_tiles(mctx_hip_t q, const void *d_tiles, unsigned groups, long long first_block,
                                 size_t nblocks, size_t lead_blocks, size_t keep_blocks, float gain, void *d_iq,
                                 void *stream);
/* Streaming form = the class interface of lib/multichanneltx.cc, one call per reference method:
 *   stream_begin    starts streaming with frame slots for payloads up to max_payload_len (slots grow on demand)
 *   stream_ready    IsChannelReadyForData (:147-162): 1 ready, 0 frame still going out, <0 error
 *   stream_update   UpdateData (:165-189): assemble a frame for one channel (MCRX_EBUSY when not ready)
 *   stream_generate GenerateSamples (:192-227): the next 2N wideband samples into a host buffer
 *   stream_reset    Reset (:126-149): frames and filter state dropped, the oscillator keeps its phase
 * The GPU works one OFDM symbol period (M + cp calls of stream_generate) ahead, which is the
 * granularity at which the reference's own frame generators are stepped (:230-242). */
/* One frame of one frame generator at the channel rate = ofdmflexframegen_assemble + _writesymbol until the
 * last symbol, as ofdmtxrx::transmit_packet / assemble_frame + write_symbol drive it (lib/ofdmtxrx.cc:297-342,
 * 366-388): frame_len() samples (a whole number of M+cp symbols, tail symbol included), scaled by `gain`,
 * into a host buffer.  Works on any mctx handle (the channel count does not matter). */
size_t mctx_hip_frame_len(mctx_hip_t q, unsigned payload_len, int mod, int fec0, int fec1);
int    mctx_hip_frame(mctx_hip_t q, const uint8_t *header8, const uint8_t *payload, unsigned payload_len,
                      int mod, int fec0, int fec1, float gain, float *out, size_t out_cap_samples);
int    mctx_hip_stream_begin(mctx_hip_t q, unsigned max_payload_len);
int    mctx_hip_stream_ready(mctx_hip_t q, unsigned channel);
int    mctx_hip_stream_update(mctx_hip_t q, unsigned channel, const uint8_t *header8, const uint8_t *payload,
                              unsigned payload_len, int mod, int fec0, int fec1);
int    mctx_hip_stream_generate(mctx_hip_t q, float *buffer);
int    mctx_hip_stream_reset(mctx_hip_t q);
const char *mctx_hip_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* MCRX_HIP_H */
