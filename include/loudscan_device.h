/*
 * loudscan_device.h -- device-level C ABI of the MI355X EBU R128 scanner.
 *
 * Plain C, plain pointers and sizes: this is the inner boundary that the
 * scan.h drop-in (include/loudscan.h) and every FFI binding (ctypes, cgo, JNI)
 * call.  It replaces what /root/reference/src/scan.c obtains from libebur128:
 *
 *   lgd_plan            <- ebur128_init per file            scan.c:203-207
 *   lgd_execute         <- ebur128_add_frames_short loop    scan.c:225-250,448
 *   lgd_fetch (track)   <- ebur128_loudness_global          scan.c:294
 *                          ebur128_loudness_range           scan.c:297
 *                          ebur128_true_peak per channel    scan.c:300-307
 *   lgd_fetch (album)   <- ebur128_loudness_global_multiple scan.c:383
 *                          ebur128_loudness_range_multiple  scan.c:388
 *                          album peak loop                  scan.c:359-378
 *
 * PCM is interleaved and resident in HBM: f32 (scale 1.0 == ebur128_add_frames_float; values on the
 * S16 grid k/32768 reproduce the reference's S16 feed, scan.c:414) or that S16 feed itself
 * (lgd_plan_formats, LGD_PCM_S16 == ebur128_add_frames_short: same results bit for bit, half the bytes).
 * Every call returns 0 on success or a negative LGD_E* code; lgd_last_error()
 * gives the text.  No call falls back to a CPU implementation.
 */
#ifndef LOUDSCAN_DEVICE_H
#define LOUDSCAN_DEVICE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LGD_MAX_CHANNELS 64 /* ebur128_init rejects > 64 */

enum {
  LGD_OK = 0,
  LGD_EINVAL = -1,  /* bad argument (channels 0 / > 64, rate < 16 / > 2822400, ...) */
  LGD_ENOMEM = -2,  /* hipMalloc failed */
  LGD_EHIP = -3,    /* HIP runtime / launch error */
  LGD_ESTATE = -4,  /* call order violated (execute before plan, ...) */
  LGD_EUNSUP = -5   /* configuration the kernels do not cover */
};

enum {
  LGD_FLAG_TRUE_PEAK = 1u << 0, /* oversampled true peak (always on in the reference) */
  LGD_FLAG_ALBUM = 1u << 1,     /* also run the album reductions (scan.c:380-405) */
  LGD_FLAG_ALBUM_PART1 = 1u << 2 /* multi-GPU album: stop after this rank's part1; the
                                    caller exchanges partials and drives stages 2 and 3 */
};

typedef struct lgd_ctx lgd_ctx;

/* PCM element formats (lgd_plan_formats).  LGD_PCM_S16 is the reference's own feed: scan.c:442 converts every
 * decoded frame to interleaved int16 and hands it to ebur128_add_frames_short (scan.c:448), which scales by
 * 1/32768.  The S16 kernel variants read those 2 bytes per sample from HBM as they are: half the memory and half
 * the traffic of the f32 form, results bit-identical to the f32 form fed with k/32768. */
enum { LGD_PCM_F32 = 0, LGD_PCM_S16 = 1 };

/* one input file == one ebur128_state in the reference (scan.c:126) */
typedef struct {
  const float *pcm;  /* device pointer, interleaved f32 (or int16 behind the same pointer, see lgd_plan_formats),
                        16-byte aligned */
  uint64_t frames;   /* frames (samples per channel) */
  uint32_t channels; /* 1..64; default libebur128 channel map by index */
  uint32_t rate;     /* Hz */
} lgd_track;

typedef struct {
  double loudness;       /* LUFS; -HUGE_VAL when no block passes the gates */
  double lra;            /* LU; 0.0 when no short-term block is listed */
  double peak;           /* max over channels of max(true peak, sample peak) */
  double sample_peak;    /* max over channels */
  double true_peak;      /* max over channels of ebur128_true_peak = max(interpolated, sample);
                            0 when the plan had no LGD_FLAG_TRUE_PEAK */
  double rel_threshold;  /* relative gate, energy units */
  double sum_abs;        /* sum of block energies >= absolute gate */
  double sum_rel;        /* sum of block energies >= relative gate */
  uint64_t n_blocks;     /* 400 ms blocks evaluated */
  uint64_t n_abs;        /* blocks >= absolute gate (the reference's list length) */
  uint64_t n_rel;        /* blocks >= relative gate */
  uint64_t n_st_blocks;  /* 3 s blocks evaluated */
  uint64_t n_st;         /* 3 s blocks >= absolute gate */
  double max_momentary;  /* LUFS of the loudest 400 ms window on the 100 ms grid (-HUGE_VAL: none) */
  double max_shortterm;  /* LUFS of the loudest 3 s window on the 100 ms grid (-HUGE_VAL: none) */
} lgd_track_result;

typedef struct {
  double loudness, lra, peak, rel_threshold, sum_abs, sum_rel;
  uint64_t n_abs, n_rel, n_st;
  /* evidence of the exchange, read back from the gathered records on the device: how many ranks' record-1
   * heads stage 2 folded (1 on a single GPU), how many of them held blocks or a peak of their own, and how
   * many records 2 stage 3 folded */
  uint32_t ranks_stage2, ranks_with_content, ranks_stage3, reserved;
} lgd_album_result;

/* What ranks exchange for an album (one all-gather each, see below).
 * Record 1 is `n_doubles` doubles: this head, then the rank's listed 3 s energies
 * (0.0 = no entry), then 0.0 padding up to the "album_slots" parameter. */
typedef struct {
  double sum_abs; /* sum of the 400 ms block energies above the absolute gate */
  double n_abs;   /* their count (integer-valued, exact below 2^53) */
  double peak;    /* max over this rank's tracks */
  double n_st;    /* listed short-term blocks */
} lgd_album_part1;
typedef struct {
  double sum_rel; /* sum / count above the album's relative gate */
  double n_rel;
} lgd_album_part2;

lgd_ctx *lgd_create(int device);
void lgd_destroy(lgd_ctx *ctx);
const char *lgd_last_error(void);

/* tuning knobs: "chunk" (frames per lane, 0 = auto), "seg_subblocks" (100 ms
 * sub-blocks per wave segment, 0 = auto), "warm_subblocks" (K-filter warm-up
 * before a segment, default 2 = 200 ms: measured 2e-16 relative to any longer warm-up), "waves_per_cu" (auto segmentation target), "timing" (1 = bracket the kernels that
 * read PCM and the whole scan with hipEvents for lgd_kernel_ms_stats, default; 2 = one more marker between the scan
 * kernels and the true-peak kernels, for lgd_scan_only_ms_stats -- it costs ~5 us of queue time inside the bracket;
 * 0 = no event packets), "overlap"
 * (0 = every scan runs on the caller's stream, in stream order like any kernel, default;
 * 1 = consecutive lgd_execute calls alternate between the caller's stream and an internal
 * one so that independent scans pipeline -- then a scan may still be READING its PCM after
 * the caller's stream has drained: do not overwrite or free a track buffer before
 * lgd_fetch, or order the writer behind the scans with lgd_join), "album_slots"
 * (short-term slots in album record 1, see the multi-GPU album below; 0 = this plan's own),
 * "tp_prune" (1 = the true-peak interpolator is evaluated only where it can exceed the track's
 * sample peak -- exact, default; 0 = everywhere: the reference mode of the pruning tests),
 * "tp_dense_min" (rows of a tile with at least this many of their 64 chunks flagged are walked as a whole by
 * lgd_tp_kernel instead of chunk by chunk; default 32, 65 = never),
 * "album_world" (ranks the scratch of the multi-GPU album's loudness range is sized for, default 8),
 * "strided" (3+ channel streams as one workgroup per channel pair or triple of every segment: 0 never,
 * 1 where measured faster = triples for 5 / 5.1 channels (and for 3 channels with true peak), quads for 7 / 7.1 / 9 and for 16+ channels, default; 2 pairs always;
 * 4 quads for every layout from 5 channels up;
 * 3 triples wherever the channel count divides by three), "merge_launches" (1, default: the (rate, channels) groups of a plan that run the same kernel
 * instance -- e.g. its 48, 96 and 192 kHz stereo tracks -- are scanned by one launch, sized to fill the GPU
 * together; 0: one launch per group), "group_streams" (1 = the groups of a
 * mixed-rate plan are launched on several streams at once; measured slower, default 0). */
int lgd_set_param(lgd_ctx *ctx, const char *name, long value);

/* Build the segment table + workspace for a batch of tracks (host work and
 * hipMalloc happen here, never in lgd_execute). */
int lgd_plan(lgd_ctx *ctx, const lgd_track *tracks, uint32_t n_tracks, uint32_t flags);
/* Element format of every track of the NEXT lgd_plan / lgd_plan_albums call (consumed by it): formats[t] is
 * LGD_PCM_F32 or LGD_PCM_S16, n_tracks must match that plan's.  Without this call (or with a null array) every
 * track is f32.  A track announced as S16 carries `const int16_t *` behind lgd_track::pcm (frames x channels
 * interleaved, 16-byte aligned); every layout and rate the f32 form covers (ebur128_add_frames_short, scan.c:448). */
int lgd_plan_formats(lgd_ctx *ctx, const uint8_t *formats, uint32_t n_tracks);
/* The same for a batch of several albums on one GPU (LGD_FLAG_ALBUM): track t belongs to
 * album album_of_track[t] (< n_albums, non-decreasing: the tracks of an album are
 * consecutive; an album may be empty).  One launch scans every track and reduces every
 * album -- the shape of a library scan (one loudgain call per folder in the reference's
 * bin/rgbpm2:120-175).  lgd_fetch then fills n_albums album records. */
int lgd_plan_albums(lgd_ctx *ctx, const lgd_track *tracks, uint32_t n_tracks,
                    const uint32_t *album_of_track, uint32_t n_albums, uint32_t flags);
/* Enqueue the whole scan (hipStream_t as void*; NULL = default stream): K-weight +
 * block-energy + peak kernel(s), gating / LRA epilogue and, with LGD_FLAG_ALBUM, the
 * album stages.  Asynchronous, no allocation.  Work enqueued on `hip_stream` before
 * the call is ordered before the scan; with "overlap" 1 the scan itself may run on an
 * internal stream, so its results are defined (and its PCM may be reused) after lgd_fetch
 * or behind lgd_join, not after a synchronisation of `hip_stream` alone. */
int lgd_execute(lgd_ctx *ctx, void *hip_stream);
/* "overlap" 1 only: make `hip_stream` wait for every scan enqueued so far (they may run on
 * an internal stream), e.g. before the stream re-uploads PCM into a buffer being scanned. */
int lgd_join(lgd_ctx *ctx, void *hip_stream);
/* Synchronise the streams used by the last lgd_execute calls and copy results out.
 * `album` may be NULL; otherwise it has room for one record per album of the plan. */
int lgd_fetch(lgd_ctx *ctx, lgd_track_result *tracks_out, lgd_album_result *album);

/* Multi-GPU album (plan with LGD_FLAG_ALBUM_PART1).  Two all-gathers per album:
 *   lgd_execute                      this rank's record 1 (lgd_album_record1)
 *   all-gather record 1           -> all_rec1[world][n_doubles]
 *   lgd_album_stage2(all_rec1)       relative gate from the heads summed in rank order
 *                                    (bit-identical on every rank), second pass over this
 *                                    rank's blocks -> record 2; CLEARS the heads in
 *                                    all_rec1, which thereby becomes the album's
 *                                    short-term list and must stay valid until stage 3 ran
 *   all-gather record 2           -> all_rec2[world]
 *   lgd_album_stage3(all_rec2)       album loudness, range (exact order statistics over
 *                                    the gathered list), peak -> lgd_fetch
 * Every rank must use the same n_doubles: set "album_slots" to the largest short-term
 * slot count of any rank (with "album_slots" 0, n_doubles - 4 is this rank's) and plan again.
 * The getters and the stage calls refer to the workspace of the MOST RECENT lgd_execute
 * (workspaces are used in turn, so the pointers change: query them after each
 * lgd_execute).  The exchange may run on any stream R of the caller:
 * lgd_album_join(ctx, R) orders R behind that scan; lgd_album_stage3 marks the end of
 * the reduction and a later scan into the same workspace waits for it -- the exchange
 * of scan k overlaps the kernels of scans k+1.., and lgd_fetch joins everything.
 * all_rec1 / all_rec2 = NULL: this rank alone (what LGD_FLAG_ALBUM does internally). */
int lgd_album_join(lgd_ctx *ctx, void *hip_stream);
int lgd_album_record1(lgd_ctx *ctx, double **dev_ptr, uint64_t *n_doubles);
int lgd_album_stage2(lgd_ctx *ctx, double *all_rec1, uint32_t world, void *hip_stream);
int lgd_album_record2(lgd_ctx *ctx, double **dev_ptr); /* 2 doubles, lgd_album_part2 */
int lgd_album_stage3(lgd_ctx *ctx, const double *all_rec2, uint32_t world, void *hip_stream);

/* ingest helper: interleaved S16 in HBM -> f32 / 32768 (exact; the scaling of
 * ebur128_add_frames_short), so that PCM crosses PCIe at 2 bytes per sample */
int lgd_convert_s16(const short *dev_in, float *dev_out, uint64_t n_samples, void *hip_stream);

/* per-track block energies for parity tests: 100 ms sub-block energies
 * (sum_c w_c sum y^2, not yet divided by the block length) */
int lgd_copy_subblock_energies(lgd_ctx *ctx, uint32_t track, double *host_out, uint64_t cap,
                               uint64_t *n_out);
/* per-channel peaks of one track: ebur128_sample_peak and ebur128_true_peak =
 * max(interpolated, sample) (scan.c:303,371 take them channel by channel); true_peak is 0
 * when the plan had no LGD_FLAG_TRUE_PEAK; either output may be NULL */
int lgd_copy_channel_peaks(lgd_ctx *ctx, uint32_t track, double *sample_peak, double *true_peak,
                           uint32_t cap_channels);
/* kernel-only timing of the last lgd_execute on its stream (hipEvents recorded
 * on that stream around the dominant kernel and around the whole enqueue) */
int lgd_last_kernel_ms(lgd_ctx *ctx, float *scan_ms, float *total_ms);
/* the same over the last `last_n` (<= 64, 0 = 64) executes: mean / min of the scan
 * kernel, mean of the whole enqueue; synchronises on their events */
int lgd_kernel_ms_stats(lgd_ctx *ctx, uint32_t last_n, float *scan_mean, float *scan_min,
                        float *total_mean, uint32_t *n_used);
/* lgd_kernel_ms_stats' scan figure covers every kernel that reads PCM: the scan kernels and
 * the true-peak kernel behind them; this is the scan kernels alone */
int lgd_scan_only_ms_stats(lgd_ctx *ctx, uint32_t last_n, float *mean, float *min_ms);
/* plan geometry (for DESIGN/bench reporting) */
int lgd_plan_info(lgd_ctx *ctx, uint64_t *n_segments, uint64_t *n_subblocks, uint32_t *chunk,
                  uint64_t *pcm_bytes, uint64_t *warm_bytes);

#ifdef __cplusplus
}
#endif
#endif
