// kernels.h -- argument blocks and launch entry points of the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mcrx {

#define MCRX_TILE_S 16           // channel-rate samples per (channel, tile) granule = 128 bytes = one cache line (== MCRX_TILE, include/mcrx_hip.h)
#define MCRX_TILE_SH 4           // log2 of it
#define MCRX_HDR_SYMS 288        // BPSK header symbols
#define MCRX_HDR_ENC 36
#define MCRX_HDR_DEC 14

// ---------------------------------------------------------------- channelizer.hip
struct ChanArgs {
    const float2 *x;            // new wideband samples, nblocks * K
    const float2 *halo;         // the P - 1 blocks preceding x (NULL = zeros: cold start)
    const float *taps;          // column tap table tap[j][n]: column n's tap on the block j back (j = 0: the newest), P * K floats
    float2 *out;                // out[g][tile][c][16]
    uint32_t nblocks;           // blocks in x (multiple of 16)
    uint32_t slab_blocks;       // blocks per slab (multiple of 16)
    uint32_t first_sample_lo;   // absolute sample index of x[0], low 32 bits (NCO phase)
    uint32_t dtheta;            // NCO phase increment per sample
    uint32_t ntiles;            // tiles per group in `out`
    uint32_t cg;                // channels per group
    uint32_t col_shift;         // P = 28 (oversampled front end): column n's FIR output is transform input (n + col_shift) mod K
};
int channelizer_supported(unsigned K);
// blocks per workgroup slab such that the grid is a whole number of waves over `ncu` compute units
uint32_t channelizer_auto_slab(unsigned K, size_t nblocks, unsigned ncu);
// P = taps per column: 14 = the reference's bank (any even K <= 2048), 28 = the oversampled front end's composite bank (power-of-two K <= 1024)
hipError_t channelizer_launch(unsigned K, unsigned P, const ChanArgs &a, hipStream_t st);

// ---------------------------------------------------------------- ofdmsync.hip
// The two named deviations from liquid-dsp's ofdmframesync in the S1 stage -- the same
// switches as LL_S1_BACKOFF_CORRECTION / LL_S1_METRIC_G0_NORMALISED in oracle/liquidlite.h; keep them equal.
#ifndef MCRX_S1_BACKOFF_CORRECTION
#define MCRX_S1_BACKOFF_CORRECTION 0        /* D6: 1 = G[k] *= e^{j 2 pi k backoff / M} after the S1 gain estimate */
#endif
#ifndef MCRX_S1_METRIC_G0_NORMALISED
#define MCRX_S1_METRIC_G0_NORMALISED 1      /* D7: 1 = S1 metric scaled by the S0-stage gain g0 */
#endif
enum { SY_SEEK = 0, SY_S0A, SY_S0B, SY_S1, SY_RX };
enum { FX_HEADER = 0, FX_PAYLOAD };

// coding tables in device memory (one copy per process)
struct CodingDev {
    const uint32_t *crc_byte;       // [256]
    const uint32_t *crc_zadv;       // [16][4][256]
    const uint8_t *qam16_nb;        // [16][4]
    const uint8_t *qam64_nb;        // [64][4]
};

// design tables of one (M, cp, allocation)
struct SyncConsts {
    int M, M2, cp, L, backoff, M_pilot, M_data, Nen, M_S0, M_S1;
    int E;                      // elements per lane = ceil(M / 64)
    int log2M;                  // > 0: power-of-two shuffle FFT; 0: direct DFT
    float detect_thresh, sync_thresh;
    const uint8_t *sctype;      // [M]
    const float *S0, *S1;       // [M]  +-1 / 0
    const float2 *s0t;          // [M]  time-domain S0
    const float *smk, *smn;     // [M][5], [Nen][5]: equaliser smoother (order-4 LSQ projection) = smk smn^T, orthonormal basis
    float2 backoff_rot;         // e^{+j 2 pi backoff / M}: S1 gain estimate de-rotation
    const float *Pfit;          // [2][M_pilot]
    const int16_t *data_rank, *pilot_rank, *en_rank;    // [M]
    const uint8_t *pilot_seq;   // [255]
    const uint16_t *hdr_map;    // [288] received header bit -> de-interleaved position | scrambler bit << 15
    const float2 *dft_tw;       // [M]  W_M^k
    CodingDev cod;
    uint32_t max_payload_len, max_enc_len, max_syms;
    const uint32_t *crc_pos;    // [crc_pos_n][256]: CRC-32 contribution of byte value b at distance d from the end of the message (NULL: none)
    uint32_t crc_pos_n;
    // de-interleaver as one gather (decode_kernel): the four passes of liquid's packet interleaver are a fixed permutation of
    // the soft bits for a given coded length e -- soft bit kb of coded byte i comes from bit kb of coded byte il_map[8 (il_off[e] + i) + kb];
    // il_off[e] = ~0: no map for that length (the passes run in LDS as before).  One copy per device, shared (mcrx_hip.hip: il_tables_acquire).
    const uint32_t *il_off;     // [il_n]
    const uint16_t *il_map;
    uint32_t il_n;
    int payload_soft;
};

// per-channel synchronizer state, persistent across launches
struct ChanState {
    int32_t state; uint32_t timer;
    int64_t cur;                // next channel-rate sample to consume (absolute)
    uint32_t nco_theta_ref, nco_dtheta; int64_t nco_t_ref;
    float g0; float2 s_hat_0;
    uint32_t num_symbols, pilot_count;
    float phi_prime, p1_prime;
    int32_t fstate; uint32_t header_symbol_index, payload_symbol_index;
    float evm_hat, evm;
    uint32_t payload_len, mod_scheme, bps, check, fec0, fec1, enc_len, mod_len;
    int32_t header_valid;
    uint32_t hw[4];             // decoded header bytes 0..13, little-endian words
    // frame cadence seen so far (speculation only; never part of the synchronizer's decisions)
    uint32_t period_hint;       // distance between the last two consecutive post-frame states (0: none yet): where segment waves look for their first frame
    int64_t last_fresh;         // the most recent post-frame state's position
    uint32_t burst_hint;        // frames a round adopted in a row before it met a state nobody predicted (0: not learnt): how far a
                                // stopped scout predicts when pushes hold several bursts (SyncArgs::burst_limit)
    uint32_t burst_pad;
};

// one decoded frame (device side); host copies payload / framesyms from the slot arrays
struct FrameRec {
    uint32_t channel; int32_t header_valid, payload_valid; uint32_t payload_len;
    uint8_t header[8];
    float evm, rssi, cfo;
    uint32_t mod_scheme, mod_bps, check, fec0, fec1, num_framesyms;
    int64_t end_sample;
    uint64_t payload_off, syms_off;     // byte offsets into the payload arena / the symbol arena
};

// payload of one frame handed from the per-channel scout to a payload worker wave
struct PayloadJob {
    ChanState s;                // synchronizer state right after the last header symbol
    uint32_t ch;                // channel within the shard
    uint32_t pad;               // record slot (set by place_jobs_kernel)
    uint64_t arena_off;         // pre-allocated record space: payload bytes in the payload arena ...
    uint64_t syms_off;          // ... and framesyms in the symbol arena (~0: no room, frame dropped)
};

// Segment-parallel acquisition (round 4; replaces the cadence predictions of rounds 1-3).  A channel's synchronizer is a
// chain by information -- where frame k+1 is looked for depends on where frame k ended, which is only known once its header
// is decoded -- so one wave per channel is a serial walk over every frame of the push, whatever the traffic.  But after a
// frame's last symbol liquid leaves the synchronizer in one fixed state (SEEK, timer = M+cp, everything else reset), so the
// walk can be cut: the channel's stretch of the push is divided into `nseg` segments and each gets a wave (sync_seg_kernel)
// that runs the state machine's own code through its segment, frame after frame, parking every frame's hand-off in a
// SpecSlot KEYED BY THE EXACT STATE THE FRAME WAS ACQUIRED FROM.  Segment 0 starts from the channel's real state; the others
// start in a fresh SEEK state just in front of the first preamble a cheap half-symbol autocorrelation finds in their segment
// (their first frame is acquired from a state the real chain never stands in -- its slot has no key -- but it ends where the
// real chain's does whenever both lock to the same symbol timing, and from there on the wave IS the real chain); every wave
// runs on until it has acquired the first frame that begins in the NEXT segment, which links it to that segment's wave.
// The per-channel scout then only hops from slot to slot: a slot is used iff its key equals the scout's state bit for bit, and
// then the segment wave has executed exactly the events the scout would have, so the result is the sequential one; where no
// slot matches (a timing that locked differently, an invalid header, an idle stretch) the scout walks on by itself --
// slower, never different.  Nothing depends on the traffic having a cadence.
#define MCRX_SPEC_MAX 256
// A segment wave writes its hand-offs straight into the launch's job list -- synchronizer state, equaliser, everything a payload
// worker needs -- but with the owner field void (PayloadJob::ch = ~0: workers, placement and decoder skip such entries, as they skip
// a reservation that came to nothing); the scout makes a frame real by storing the channel number there.  Nothing is copied.
struct SpecSlot {
    int64_t start;              // the state it was acquired from: next sample | timer << 48 | state << 61 (spec_key); -1: none
    int64_t t_last;             // status 1, 3: event index of the frame's last payload symbol / of its failed header; status 2: the SEEK position the frame was detected from
    int32_t status;             // 1: frame acquired and handed off (its job list entry: pad); 2: frame acquired, but its payload runs past the end of
                                //    the buffer and the next push still holds its beginning: the scout goes back to (t_last, pad) -- deferred;
                                //    3: header decoded, check failed: the scout writes the record (state in job list entry pad); 0: nothing usable
    uint32_t pad;               // status 1, 3: job list entry; status 2: the timer of that SEEK state
};

struct SyncArgs {
    SyncConsts c;
    const float2 *chan;         // [tile][chan_stride][16]; my channel c sits at chan_off + c
    uint32_t chan_stride, chan_off;
    int64_t buf_first;          // absolute index of chan sample (tile 0, slot 0)
    int64_t end;                // absolute end (exclusive) of valid samples
    uint32_t nch;               // channels in this shard
    uint32_t ch_first;          // global index of shard channel 0 (reported in FrameRec)
    ChanState *st;              // [nch]
    uint8_t *hbits;             // [nch][MCRX_HDR_SYMS] hard header bits
    float2 *R;                  // [nch][M] equaliser
    uint8_t *soft;              // [nch][8*max_enc_len]
    uint8_t *tmpa, *tmpb;       // [nch][max_enc_len + 16]
    float2 *syms;               // [nch][max_syms]
    // output pool
    // (payloads and equalised symbols live in separate arenas: a harvest that only wants the decoded bytes
    //  moves 1.2 KB per frame over the host link instead of 59 KB)
    FrameRec *rec; uint8_t *arena; uint8_t *sarena;
    uint32_t *nrec;             // [0] allocated count, [1] dropped count
    unsigned long long *arena_used;     // [0] payload arena bytes, [1] symbol arena bytes
    uint64_t arena_cap, sarena_cap;
    uint32_t max_rec;
    // scout -> payload worker hand-off (frame-parallel payload processing)
    int scout;                  // 1: scouts hand complete in-buffer frames to payload workers
    PayloadJob *jobs; uint32_t *njobs; uint32_t max_jobs;
    uint32_t *gen_list;         // [0] = number of hand-offs the LDS decode path does not take, [1 ..] their job indices (place_jobs_kernel)
    uint32_t dec_lds_soft;      // bytes of LDS the decode launch of this push gives a frame's soft bits (decides which path a frame takes)
    uint32_t *njobs_next;       // the next launch's job counter: zeroed by this launch's placement kernel
    float2 *jR;                 // [max_jobs][M]
    uint8_t *jsoft;             // [max_jobs][8*max_enc_len]
    uint8_t *jtmp;              // [max_jobs][2*(max_enc_len+16)]
    int debug;                 // MCRX_DEBUG=1: trace state-machine events of channel 0
    uint2 *vit_scratch;         // the K = 7 decoder's decision rows: [vit_waves][vit_rows][64] (one region per workgroup of decode_general_kernel; csrc/viterbi_frames.hpp)
    uint32_t vit_rows, vit_waves;
    uint32_t *vit_passes;       // [0] forward passes repeated, [1] traceback passes repeated (blocks whose survivors had not merged inside the overlap), [2] frames; NULL: not counted
    // list-driven launches that are nearly always empty (QAM payload workers, trellis blocks, general decoder): a launch of
    // 8192 do-nothing workgroups still has to find 8192 wave slots, behind the channelizer's whole-CU workgroups -- measured
    // 0.15-0.27 ms on the work stream.  The kernels walk their lists with a grid stride, so any grid is correct; the host
    // sizes it from what the most recent launch published here (list_hint, host-mapped) and keeps a floor for the first
    // frames of a kind it has not seen yet.
    uint32_t *qam_list;         // [0] = hand-offs with a 16- / 64-QAM payload, [1 ..] their job indices (filled by the scouts: Walker::place_owned)
    uint32_t *qam_next;         // the next launch's list: its count is zeroed by this launch's housekeeping kernel
    // The job list has holes (entries of frames the segment waves acquired for nothing, blocks not used up, void reservations): workers and
    // decoders go by the list of entries that got an owner, live[0] = count, live[1 ..] = job list entries (filled by the scouts a
    // block per channel: Walker::place_owned), with a grid stride -- the host sizes their grids from the frames of the most recent
    // launch (frames_hint) instead of the list's capacity
    uint32_t *live, *live_next;
    uint32_t frames_hint;       // ~0: unknown
    uint32_t live_off;          // lean workers: the main launch's grid (payload_lean.hpp, REST)
    uint32_t *list_hint;        // [0] QAM hand-offs, [1] unused, [2] frames on the general list, of the most recent launch
    uint32_t grid_hint[3];      // what the host last read there (~0: no hint, full grids)
    uint32_t payload_lds_pad;  // bytes of unused dynamic LDS per payload worker: caps the workers' occupancy (walk mode, launch_sync)
    int seek_burst;            // idle stretches: SEEK events four at a time, their windows requested together (Walker::seek_burst)
    // speculation (see SpecSlot)
    SpecSlot *spec; float2 *spec_R;      // [nch][spec_stride]; [nch][MCRX_SEG_MAX][M]: a segment wave's equaliser between the S1 fit and the hand-off
#define MCRX_SEG_MAX 128
    int64_t *pred; uint32_t *pred_n;     // predicted fresh-state positions for the next launch: [nch][MCRX_SPEC_MAX], [nch]
    uint32_t spec_cap;                   // slots per channel the segment waves fill in this launch = nseg * (slots per wave) (0: off)
    uint32_t spec_stride;                // slots between two channels in `spec` (>= spec_cap; MCRX_SPEC_MAX unless a push holds more frames per
                                         // channel than that: the scouts then read the slot headers a window of MCRX_SPEC_MAX at a time)
    uint32_t nseg;                       // segment waves per channel
    uint32_t seg_jobs;                   // job list entries a segment wave reserves at a time (the frames it is expected to hand off)
    // The two wasted acquisitions of a segment (its first frame from an arbitrary state, the frame that links it to the next
    // segment) disappear where the traffic HAS a cadence: phase 1 (one wave per channel) acquires the first frame of the push from
    // the channel's real state and leaves the position it ends at in anchor[ch]; phase 2 starts segment g in the exact fresh
    // state at the first point of anchor + n * period_hint inside it -- if the half-symbol autocorrelation finds a preamble right
    // behind that point -- and a wave whose chain arrives exactly at the next segment's validated start stops there.  A wrong or
    // missing prediction costs nothing but the two acquisitions again (the coarse start); the results never depend on it.
    int no_syms;                         // 1 (mcrx_hip_config::skip_framesyms = 2): the lean payload workers do not store the equalised symbols at all
    // The launch behind the lean workers (QAM payloads, frames beyond their grid) and the general decoder normally find nothing to do,
    // but an empty launch still waits for a wave slot on a chip full of channelizer and worker waves -- 0.15 + 0.08 ms of the work
    // stream's 0.98 ms per push (profiles/r5_t3_kernel_stats.csv).  While their lists have been empty the host puts them on a stream
    // of their own, with a decoder launch for their frames only, so that the next push's workers do not queue behind them:
    int split_rest;                      // 1: stage 1 launches the main workers only, stage 4 the launch behind them
    int dec_phase;                       // decode_kernel: 0 = every live frame, 1 = the main workers' frames only, 2 = the others
    int seg_walker;                      // 1 (default): the segment waves are the Walker's kernel (sync_spec_kernel); 0 (scout_build = 2): acq_lean.hpp's where the design allows
    int seg_phase;                       // 1: first frame from the entry state only (grid = channels); 2: the rest; 0: one launch, no cadence;
                                         // 3 / 4 (round 5): one launch, the lattice carried over from the previous push -- absolute / relative to the push's beginning
    int64_t *anchor;                     // [nch] where phase 1's frame ended + 1 (-1: it did not hand a frame off)
    int64_t *seekst;                     // [nch][2] the SEEK state (cur, timer) the acquisition a channel's push ENDED in was detected from: where that
                                         // frame is re-acquired from if the next push defers it (a frame detected at the end of one push and deferred
                                         // in the next used to go back to position 0 -- pushes shorter than a frame re-delivered the whole history)
    uint32_t *spec_hint;                 // host-mapped word: largest prediction count, sizes the next launch's grid
    uint32_t *walk_hint;                 // host-mapped word: frames the scouts had to acquire themselves so far (the host adds a full-width round while it moves)
    uint32_t *hint;             // host-mapped word: longest coded frame (bytes) among this launch's jobs
    uint32_t enc_hint;          // the value the host last saw there (0: none yet)
    int round_idx;              // which acquisition round of the launch this is (a scout that stops in round >= 1 reports it: stats[6], the host
                                // sizes the number of rounds by the last one that was needed)
    int burst_limit;            // 1: a stopped scout predicts one burst ahead (ChanState::burst_hint), not to the end of the buffer: the host
                                // sets it while it runs several extra rounds per launch -- every wrong slot is an acquisition attempt per round
    int stop_after_walk;        // 1: a scout standing in a post-frame (or the entry) state nobody predicted stops there and predicts
                                //    the frames that follow from it (cadence re-anchored inside the launch); the next round continues
    int64_t defer_limit;        // > 0: the next push carries this many channel-rate samples of history, so a frame whose payload
                                //    does not fit this buffer and that began less than that before its end is not walked serially
                                //    by the tail kernel: the lean scout rewinds to the SEEK state it detected the frame from
                                //    and the next push acquires it again, whole (0: never)
    int tail_only;              // sync_kernel as the lean configurations' tail kernel: only channels with a payload in progress.
                                //    1 (before the acquisition rounds): the frame a previous push left unfinished, to its end;
                                //    2 (after them): a frame the lean scout could not hand off -- it straddles the end of the buffer,
                                //    is oversize, or the job list is full -- and then on to the end of the buffer
    uint32_t *stats;            // [0] frames the scouts acquired themselves, [1] frames adopted from speculative waves (NULL: not counted);
                                // [2] slots the segment waves filled (the frames among them that nobody adopts are the host's measure of a wrong anchor)
    int no_fast;               // MCRX_NO_FAST=1: payload workers use the general symbol path (A/B experiments)
    int payload_fr, payload_lean, payload_xb;      // which build of the M = 64 workers (MCRX_PAYLOAD_FR / _LEAN / _XB at creation; defaults 1, 1, 63): the parity tests compare them
    uint32_t vit_off;          // byte offset of the convolutional decoder's 8 KB block scratch in the launch's dynamic LDS (0: none; set by the launchers)
};
// the K = 7 decoder's geometry (csrc/viterbi_frames.hpp): T trellis steps in at most 64 blocks of a multiple of 24 steps, each run W steps
// early and W steps long; a wave keeps B + 2 W decision rows of 512 bytes
namespace vf {
constexpr unsigned W = 48;
__host__ __device__ constexpr unsigned block_steps(unsigned T) { return 24u * ((T + 1535u) / 1536u); }
__host__ __device__ constexpr unsigned rows_for(unsigned T) { return block_steps(T) + 2u * W; }
}
hipError_t sync_launch(const SyncArgs &a, hipStream_t st);           // full state machine, one wave per channel (general configurations; tail kernel)
hipError_t sync_launch_tail(const SyncArgs &a, hipStream_t st);      // lean configurations: payloads in progress, to the frame's end (a.tail_only = 1)
// fill `map` for the `nlen` coded lengths lens[k] at coded-byte offsets offs[k]; lo / hi: scratch of the map's size in bytes / 2 each
hipError_t ilmap_build_launch(const uint32_t *d_lens, const uint32_t *d_offs, uint32_t nlen, uint8_t *d_lo, uint8_t *d_hi, uint16_t *d_map, hipStream_t st);
hipError_t sync_launch_walk(const SyncArgs &a, hipStream_t st);      // the lean scout built without a register budget: for streams it has to walk by itself
hipError_t sync_launch_lean(const SyncArgs &a, hipStream_t st);      // lean scout: acquisition + header + hand-off, one wave per channel
int acq_lean_waves(const SyncConsts &c);
hipError_t sync_launch_spec(const SyncArgs &a, hipStream_t st);      // segment-parallel acquisition: one wave per (channel, segment)
// stage 0: record placement (one workgroup), 1: payload workers (one wave per handed-off frame),
// 2: packet decode (one workgroup per frame; only after the lean workers -- the general ones decode in place)
hipError_t sync_launch_payload(const SyncArgs &a, int stage, hipStream_t st);
bool sync_payload_splits(const SyncArgs &a);                         // the payload stage has a launch behind the main workers (SyncArgs::split_rest)
// synchronizers back to SEEK at sample `cur`; optionally also zero two history buffers of hist_n cf32 each (hist_n even)
// and the result counters -- a whole restart in one launch
hipError_t sync_reset_launch(ChanState *st, uint32_t nch, int64_t cur, float2 *hist0, float2 *hist1, size_t hist_n,
                             uint32_t *nrec, unsigned long long *arena_used, uint32_t *pred_n, hipStream_t stream);

}  // namespace mcrx
