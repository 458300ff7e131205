/*
 * loudscan.h -- drop-in for loudgain's scan module (the scan.h interface).
 *
 * libloudscan_hip.so exports the symbols below with the exact signatures and
 * record layout of the reference header /root/reference/src/scan.h:35-65, so a
 * loudgain build can link against it in place of scan.c + libebur128.  Behind
 * them sit the HIP kernels (include/loudscan_device.h); nothing here falls back
 * to a CPU implementation -- without a MI355X the calls fail loudly.
 *
 *   reference symbol (scan.h / scan.c)                  here
 *   scan_result                      scan.h:35-53       same field order, malloc'd, caller frees
 *   scan_init                        scan.c:66-96       picks the GPU, allocates the track table
 *   scan_deinit                      scan.c:98-108      frees names, HBM buffers, the device context
 *   scan_file                        scan.c:110-273     RIFF/WAVE reader -> S16 grid (what swr_convert
 *                                                        yields at scan.c:442) -> HBM; returns -1 only
 *                                                        for index >= nb_files (scan.c:132-135)
 *   scan_get_track_result            scan.c:275-330     gain = -18 - L + pre_gain, Opus: pre_gain - 5
 *   scan_get_album_peak              scan.c:359-378
 *   scan_set_album_result            scan.c:380-405     callable once per track like loudgain.c:340 does;
 *                                                        the album is reduced once and cached
 *   scan_album_has_different_codecs / _containers / scan_album_has_opus   scan.c:332-357
 *
 * Errors follow scan.c: fatal conditions print to stderr and _exit(EXIT_FAILURE)
 * (fail_printf, printf.c:94-102); "Index too high" prints and returns NULL.
 */
#ifndef LOUDSCAN_H
#define LOUDSCAN_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
  char *file;      /* borrowed: valid until scan_deinit (scan.c:313) */
  char *container; /* borrowed (scan.c:314) */
  int codec_id;    /* FFmpeg AVCodecID value */

  double track_gain;
  double track_peak;

  double track_loudness;
  double track_loudness_range;

  double album_gain;
  double album_peak;

  double album_loudness;
  double album_loudness_range;

  double loudness_reference;
} scan_result;

int scan_init(unsigned nb_files);
void scan_deinit(void);

int scan_album_has_different_codecs(void);
int scan_album_has_different_containers(void);
int scan_album_has_opus(void);
int scan_file(const char *file, unsigned index);

scan_result *scan_get_track_result(unsigned index, double pre_gain);
double scan_get_album_peak(void);
void scan_set_album_result(scan_result *result, double pre_amp);

/* ---- extensions of this library (not in the reference) ------------------- */
/* BASELINE.json's north_star calls the album query scan_get_album_result; the
 * reference's real name is scan_set_album_result.  Same function. */
void scan_get_album_result(scan_result *result, double pre_amp);
/* GPU to use (default 0); call before scan_init. */
int scan_set_device(int device);
/* Use n_devices GPUs of this process (0 = every visible one); call before scan_init.  The files
 * of the session are dealt round-robin over them (index mod n), every GPU scans its share and
 * the album values of scan_set_album_result / scan_get_album_peak come from the exchange of
 * partial sums that loudgain_amd/album.py runs over RCCL, here through host memory (a few MB).
 * The environment variable LOUDSCAN_DEVICES=n does the same for an unmodified caller
 * (loudgain.c:299-340 never needs to know). */
int scan_set_devices(int n_devices);
/* Raw PCM in place of a file.  Host buffers are copied to HBM (through pinned staging
 * buffers, asynchronously: the call returns once the last piece is on its way); the _device
 * form borrows an HBM pointer (16-byte aligned, must stay valid until scan_deinit).  s16 is scaled by 1/32768 like ebur128_add_frames_short. */
int scan_pcm_s16(const short *interleaved, size_t frames, unsigned channels, unsigned rate,
                 unsigned index);
int scan_pcm_f32(const float *interleaved, size_t frames, unsigned channels, unsigned rate,
                 unsigned index);
int scan_pcm_f32_device(const float *device_interleaved, size_t frames, unsigned channels,
                        unsigned rate, unsigned index);
/* Interleaved S16 already resident in HBM (16-byte aligned, valid until scan_deinit): scanned as it lies by the S16
 * kernel variants (no f32 copy exists at any time).  Host S16 (scan_pcm_s16, every file through scan_file: the reader
 * delivers the S16 that scan.c:442 converts to) stays S16 in HBM the same way -- what the reference hands to
 * ebur128_add_frames_short (scan.c:448). */
int scan_pcm_s16_device(const short *device_interleaved, size_t frames, unsigned channels,
                        unsigned rate, unsigned index);
/* The RIFF/WAVE reader of scan_file on its own, for callers that batch their own uploads
 * (loudgain_amd/batch.py): probe = header only; read = the data chunk as interleaved S16
 * exactly as scan_file stages it (8/16/24/32-bit PCM, 32/64-bit float, EXTENSIBLE).
 * Unlike scan_file these do not exit: 0 / frames read, or -1 cannot open, -2 not
 * RIFF/WAVE, -3 unknown sample format, -4 no audio data. */
typedef struct {
  int codec_id;      /* FFmpeg AVCodecID value of the PCM format */
  unsigned channels;
  unsigned rate;
  unsigned bits;
  size_t frames;     /* announced by the data chunk */
} scan_wav_info;
int scan_wav_probe(const char *file, scan_wav_info *out);
long long scan_wav_read_s16(const char *file, short *out, size_t cap_frames);
/* Per-channel peaks of a scanned file: ebur128_true_peak(state, ch) for every channel, i.e. the values
 * scan.c:300-307 takes the maximum of (true_peak[ch] = max(interpolated, sample)); sample_peak[ch] is
 * ebur128_sample_peak.  Either array may be NULL.  Returns the channel count, -1 on a bad index or cap. */
int scan_get_channel_peaks(unsigned index, double *sample_peak, double *true_peak, unsigned cap);
/* mark a track's codec (FFmpeg AVCodecID) for callers that decode themselves,
 * e.g. 0x1503C (Opus) to get scan.c's -5 dB pre-gain rule */
int scan_set_codec(unsigned index, int codec_id, const char *container);

#ifdef __cplusplus
}
#endif
#endif
