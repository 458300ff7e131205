// scan_api.cpp -- the scan.h drop-in (include/loudscan.h) on top of the device ABI.
//
// Mirrors /root/reference/src/scan.c: module-level state indexed by file order,
// scan_result records malloc'd for the caller, borrowed name strings, fatal errors
// through the fail_printf convention (print + _exit(EXIT_FAILURE), printf.c:94-102).
// The FFmpeg half of scan_file (scan.c:139-272) is replaced by a RIFF/WAVE reader
// that yields the same interleaved S16 the reference feeds to libebur128
// (scan.c:414,442,448); the libebur128 half by the HIP kernels.
//
// All files of a session are scanned in ONE batched launch at the first result
// query (tracks are independent, scan.c:126); results are cached, so loudgain's
// per-track scan_set_album_result calls (loudgain.c:339-340) cost nothing.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <unistd.h>
#include <cstdint>
#include <cstdio>
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/loudscan.h"
#include "../../include/loudscan_device.h"

extern "C" hipError_t lgd_launch_s16_to_f32(const short *in, float *out, size_t n, hipStream_t s);

// FFmpeg AVCodecID values (recalled; FFmpeg headers are absent here).  The
// reference only compares them for equality and against OPUS.
enum {
  CODEC_PCM_S16LE = 0x10000, CODEC_PCM_U8 = 0x10005, CODEC_PCM_S32LE = 0x10008,
  CODEC_PCM_S24LE = 0x1000C, CODEC_PCM_F32LE = 0x10015, CODEC_PCM_F64LE = 0x10017,
  CODEC_OPUS = 0x1503C
};

#define LUFS_TO_RG(L) (-18 - (L))

namespace {

struct Track {
  char *file = nullptr, *container = nullptr;
  int codec = 0;
  unsigned channels = 0, rate = 0;
  size_t frames = 0;
  float *dev = nullptr;  // interleaved f32 in HBM
  bool owned = false, loaded = false;
};

std::vector<Track> g_tracks;
int g_nb = 0, g_device = 0;
lgd_ctx *g_ctx = nullptr;
bool g_scanned = false;
std::vector<lgd_track_result> g_res;
lgd_album_result g_album;

[[noreturn]] void fail(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  fprintf(stderr, "[loudscan] ");
  vfprintf(stderr, fmt, ap);
  fprintf(stderr, "\n");
  va_end(ap);
  _exit(EXIT_FAILURE);
}
void errmsg(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  fprintf(stderr, "[loudscan] ");
  vfprintf(stderr, fmt, ap);
  fprintf(stderr, "\n");
  va_end(ap);
}
#define HIPFATAL(expr)                                                       \
  do {                                                                       \
    hipError_t e_ = (expr);                                                  \
    if (e_ != hipSuccess) fail("%s: %s", #expr, hipGetErrorString(e_));      \
  } while (0)

char *dupstr(const char *s) {
  size_t n = strlen(s) + 1;
  char *d = (char *)malloc(n);
  if (!d) fail("OOM");
  memcpy(d, s, n);
  return d;
}

void release_track(Track &t) {
  if (t.dev && t.owned) (void)hipFree(t.dev);
  free(t.file);
  free(t.container);
  t = Track();
}

void begin_track(unsigned index, const char *name, const char *container, int codec,
                 unsigned channels, unsigned rate, size_t frames) {
  Track &t = g_tracks[index];
  release_track(t);
  // ebur128_init's argument checks (scan.c:203-209: "Could not initialize EBU R128 scanner")
  if (channels == 0 || channels > LGD_MAX_CHANNELS || rate < 16 || rate > 2822400)
    fail("Could not initialize EBU R128 scanner");
  t.file = dupstr(name);
  t.container = dupstr(container);
  t.codec = codec;
  t.channels = channels;
  t.rate = rate;
  t.frames = frames;
  t.loaded = true;
  g_scanned = false;
}

void upload_s16(Track &t, const short *pcm) {
  const size_t n = t.frames * t.channels;
  HIPFATAL(hipSetDevice(g_device));
  HIPFATAL(hipMalloc((void **)&t.dev, (n ? n : 1) * sizeof(float)));
  t.owned = true;
  if (!n) return;
  short *tmp = nullptr;  // device S16 staging: half the PCIe bytes of f32
  HIPFATAL(hipMalloc((void **)&tmp, n * sizeof(short)));
  HIPFATAL(hipMemcpy(tmp, pcm, n * sizeof(short), hipMemcpyHostToDevice));
  HIPFATAL(lgd_launch_s16_to_f32(tmp, t.dev, n, nullptr));
  HIPFATAL(hipDeviceSynchronize());
  HIPFATAL(hipFree(tmp));
}

uint32_t rd32(const unsigned char *p) {
  return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}
uint16_t rd16(const unsigned char *p) { return (uint16_t)(p[0] | (p[1] << 8)); }
short clip16(long v) { return (short)(v < -32768 ? -32768 : (v > 32767 ? 32767 : v)); }

// the whole session in one batched device scan
void ensure_scanned() {
  if (g_scanned) return;
  for (int i = 0; i < g_nb; ++i)
    if (!g_tracks[i].loaded) fail("scan_file was not called for index %d", i);
  if (!g_ctx) {
    g_ctx = lgd_create(g_device);
    if (!g_ctx) fail("%s", lgd_last_error());
  }
  std::vector<lgd_track> lt(g_nb);
  for (int i = 0; i < g_nb; ++i) {
    lt[i].pcm = g_tracks[i].dev;
    lt[i].frames = g_tracks[i].frames;
    lt[i].channels = g_tracks[i].channels;
    lt[i].rate = g_tracks[i].rate;
  }
  g_res.assign(g_nb ? g_nb : 1, lgd_track_result());
  if (lgd_plan(g_ctx, lt.data(), (uint32_t)g_nb, LGD_FLAG_TRUE_PEAK | LGD_FLAG_ALBUM) ||
      lgd_execute(g_ctx, nullptr) || lgd_fetch(g_ctx, g_res.data(), &g_album))
    fail("%s", lgd_last_error());
  // the PCM is no longer needed: results are cached until the next scan_file
  for (int i = 0; i < g_nb; ++i) {
    Track &t = g_tracks[i];
    if (t.dev && t.owned) (void)hipFree(t.dev);
    t.dev = nullptr;
  }
  g_scanned = true;
}

}  // namespace

extern "C" int scan_set_device(int device) {
  g_device = device;
  return 0;
}

extern "C" int scan_init(unsigned nb_files) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || g_device >= n)
    fail("no MI355X / HIP device %d available (the scanner has no CPU path)", g_device);
  g_nb = (int)nb_files;
  g_tracks.assign(nb_files, Track());
  g_scanned = false;
  return 0;
}

extern "C" void scan_deinit(void) {
  for (Track &t : g_tracks) release_track(t);
  g_tracks.clear();
  g_res.clear();
  if (g_ctx) lgd_destroy(g_ctx);
  g_ctx = nullptr;
  g_nb = 0;
  g_scanned = false;
}

extern "C" int scan_pcm_s16(const short *pcm, size_t frames, unsigned channels, unsigned rate,
                            unsigned index) {
  if ((int)index >= g_nb) return -1;
  begin_track(index, "<pcm_s16>", "wav", CODEC_PCM_S16LE, channels, rate, frames);
  upload_s16(g_tracks[index], pcm);
  return 0;
}

extern "C" int scan_pcm_f32(const float *pcm, size_t frames, unsigned channels, unsigned rate,
                            unsigned index) {
  if ((int)index >= g_nb) return -1;
  begin_track(index, "<pcm_f32>", "wav", CODEC_PCM_F32LE, channels, rate, frames);
  Track &t = g_tracks[index];
  const size_t n = frames * channels;
  HIPFATAL(hipSetDevice(g_device));
  HIPFATAL(hipMalloc((void **)&t.dev, (n ? n : 1) * sizeof(float)));
  t.owned = true;
  if (n) HIPFATAL(hipMemcpy(t.dev, pcm, n * sizeof(float), hipMemcpyHostToDevice));
  return 0;
}

extern "C" int scan_pcm_f32_device(const float *dev, size_t frames, unsigned channels, unsigned rate,
                                   unsigned index) {
  if ((int)index >= g_nb) return -1;
  begin_track(index, "<pcm_f32_device>", "wav", CODEC_PCM_F32LE, channels, rate, frames);
  g_tracks[index].dev = const_cast<float *>(dev);
  g_tracks[index].owned = false;
  return 0;
}

extern "C" int scan_set_codec(unsigned index, int codec_id, const char *container) {
  if ((int)index >= g_nb || !g_tracks[index].loaded) return -1;
  g_tracks[index].codec = codec_id;
  if (container) {
    free(g_tracks[index].container);
    g_tracks[index].container = dupstr(container);
  }
  return 0;
}

namespace {

struct WavInfo {
  int codec = 0;
  unsigned channels = 0, rate = 0, bits = 0, block_align = 0;
  long data_off = 0;
  size_t frames = 0;  // as the data chunk announces; a truncated file yields fewer
};

enum { WAV_OK = 0, WAV_EOPEN = -1, WAV_ENOTWAVE = -2, WAV_ECODEC = -3, WAV_ENOAUDIO = -4 };

// RIFF/WAVE header walk (stands in for avformat_open_input / find_stream_info /
// av_find_best_stream, scan.c:139-165, for the one container readable without FFmpeg)
int wav_probe(const char *file, WavInfo *wi) {
  FILE *fp = fopen(file, "rb");
  if (!fp) return WAV_EOPEN;
  unsigned char hdr[12], ck[8], fmt[40];
  if (fread(hdr, 1, 12, fp) != 12 || memcmp(hdr, "RIFF", 4) || memcmp(hdr + 8, "WAVE", 4)) {
    fclose(fp);
    return WAV_ENOTWAVE;
  }
  unsigned tag = 0;
  bool have_fmt = false;
  int rc = WAV_ENOAUDIO;
  while (fread(ck, 1, 8, fp) == 8) {
    const uint32_t sz = rd32(ck + 4);
    if (!memcmp(ck, "fmt ", 4)) {
      const uint32_t n = sz < sizeof(fmt) ? sz : (uint32_t)sizeof(fmt);
      memset(fmt, 0, sizeof(fmt));
      if (fread(fmt, 1, n, fp) != n) break;
      if (sz > n) fseek(fp, (long)(sz - n), SEEK_CUR);
      if (sz & 1) fseek(fp, 1, SEEK_CUR);
      tag = rd16(fmt);
      wi->channels = rd16(fmt + 2);
      wi->rate = rd32(fmt + 4);
      wi->block_align = rd16(fmt + 12);
      wi->bits = rd16(fmt + 14);
      if (tag == 0xFFFE && sz >= 26) tag = rd16(fmt + 24);  // WAVE_FORMAT_EXTENSIBLE sub-format
      have_fmt = true;
    } else if (!memcmp(ck, "data", 4)) {
      if (!have_fmt || !wi->channels || !wi->block_align) break;
      const unsigned bits = wi->bits;
      if (tag == 1 && bits == 16) wi->codec = CODEC_PCM_S16LE;
      else if (tag == 1 && bits == 8) wi->codec = CODEC_PCM_U8;
      else if (tag == 1 && bits == 24) wi->codec = CODEC_PCM_S24LE;
      else if (tag == 1 && bits == 32) wi->codec = CODEC_PCM_S32LE;
      else if (tag == 3 && bits == 32) wi->codec = CODEC_PCM_F32LE;
      else if (tag == 3 && bits == 64) wi->codec = CODEC_PCM_F64LE;
      else {
        rc = WAV_ECODEC;
        break;
      }
      wi->data_off = ftell(fp);
      wi->frames = sz / wi->block_align;
      rc = WAV_OK;
      break;
    } else {
      fseek(fp, (long)(sz + (sz & 1)), SEEK_CUR);
    }
  }
  fclose(fp);
  return rc;
}

// the data chunk -> interleaved S16, as swr_convert does for every decoded frame at
// scan.c:442; returns the frames actually present (a truncated file is silently
// shortened, like the packet loop at scan.c:229-240), or -1 if the file vanished
long long wav_read_s16(const char *file, const WavInfo &wi, short *out, size_t cap_frames) {
  FILE *fp = fopen(file, "rb");
  if (!fp) return -1;
  fseek(fp, wi.data_off, SEEK_SET);
  const size_t want = wi.frames < cap_frames ? wi.frames : cap_frames;
  const unsigned ch = wi.channels, ba = wi.block_align, bps = wi.bits / 8;
  size_t done = 0;
  if (wi.codec == CODEC_PCM_S16LE && ba == ch * 2) {  // already the target grid: straight read
    done = want ? fread(out, ba, want, fp) : 0;
    fclose(fp);
    return (long long)done;
  }
  std::vector<unsigned char> raw((size_t)65536 * ba);
  while (done < want) {
    const size_t n = std::min<size_t>(65536, want - done);
    const size_t got = fread(raw.data(), ba, n, fp);
    if (!got) break;
    short *o = out + done * ch;
    for (size_t f = 0; f < got; ++f) {
      const unsigned char *p = raw.data() + f * ba;
      for (unsigned c = 0; c < ch; ++c, p += bps) {
        short v;
        switch (wi.codec) {
          case CODEC_PCM_S16LE: v = (short)rd16(p); break;
          case CODEC_PCM_U8: v = (short)(((int)p[0] - 0x80) * 256); break;
          case CODEC_PCM_S24LE:
            v = (short)((int32_t)((uint32_t)p[0] << 8 | (uint32_t)p[1] << 16 | (uint32_t)p[2] << 24) >> 16);
            break;
          case CODEC_PCM_S32LE: v = (short)((int32_t)rd32(p) >> 16); break;
          case CODEC_PCM_F32LE: {
            float x;
            const uint32_t u = rd32(p);
            memcpy(&x, &u, 4);
            v = clip16(lrintf(x * 32768.0f));
          } break;
          default: {
            double x;
            const uint64_t u = (uint64_t)rd32(p) | ((uint64_t)rd32(p + 4) << 32);
            memcpy(&x, &u, 8);
            v = clip16(lrint(x * 32768.0));
          } break;
        }
        o[f * ch + c] = v;
      }
    }
    done += got;
    if (got < n) break;
  }
  fclose(fp);
  return (long long)done;
}

}  // namespace

extern "C" int scan_wav_probe(const char *file, scan_wav_info *out) {
  if (!file || !out) return WAV_EOPEN;
  WavInfo wi;
  const int rc = wav_probe(file, &wi);
  if (rc) return rc;
  out->codec_id = wi.codec;
  out->channels = wi.channels;
  out->rate = wi.rate;
  out->bits = wi.bits;
  out->frames = wi.frames;
  return 0;
}

extern "C" long long scan_wav_read_s16(const char *file, short *out, size_t cap_frames) {
  if (!file || (!out && cap_frames)) return WAV_EOPEN;
  WavInfo wi;
  const int rc = wav_probe(file, &wi);
  if (rc) return rc;
  const long long got = wav_read_s16(file, wi, out, cap_frames);
  return got < 0 ? (long long)WAV_EOPEN : got;
}

extern "C" int scan_file(const char *file, unsigned index) {
  if ((int)index >= g_nb) {
    errmsg("Index too high");
    return -1;
  }
  WavInfo wi;
  switch (wav_probe(file, &wi)) {
    case WAV_OK: break;
    case WAV_EOPEN: fail("Could not open input: %s", file);
    case WAV_ENOTWAVE: fail("Could not find stream info: %s (only RIFF/WAVE is read without FFmpeg)", file);
    case WAV_ECODEC: fail("Could not find the codec: %s", file);
    default: fail("Could not find audio stream: %s", file);
  }
  std::vector<short> s16(wi.frames * wi.channels);
  const long long got = wav_read_s16(file, wi, s16.data(), wi.frames);
  if (got < 0) fail("Could not open input: %s", file);
  begin_track(index, file, "wav", wi.codec, wi.channels, wi.rate, (size_t)got);
  upload_s16(g_tracks[index], s16.data());
  return 0;
}

extern "C" scan_result *scan_get_track_result(unsigned index, double pre_gain) {
  if ((int)index >= g_nb) {
    errmsg("Index too high");
    return nullptr;
  }
  ensure_scanned();
  scan_result *r = (scan_result *)malloc(sizeof(scan_result));
  if (!r) fail("OOM");
  const Track &t = g_tracks[index];
  const lgd_track_result &d = g_res[index];
  // Opus is always based on -23 LUFS (scan.c:309-311)
  if (t.codec == CODEC_OPUS) pre_gain = pre_gain - 5.0f;
  r->file = t.file;
  r->container = t.container;
  r->codec_id = t.codec;
  r->track_gain = LUFS_TO_RG(d.loudness) + pre_gain;
  r->track_peak = d.peak;
  r->track_loudness = d.loudness;
  r->track_loudness_range = d.lra;
  r->album_gain = 0.f;
  r->album_peak = 0.f;
  r->album_loudness = 0.f;
  r->album_loudness_range = 0.f;
  r->loudness_reference = LUFS_TO_RG(-pre_gain);
  return r;
}

extern "C" int scan_album_has_different_containers(void) {
  for (int i = 0; i < g_nb; ++i)
    if (strcmp(g_tracks[0].container, g_tracks[i].container)) return 1;
  return 0;
}

extern "C" int scan_album_has_different_codecs(void) {
  for (int i = 0; i < g_nb; ++i)
    if (g_tracks[0].codec != g_tracks[i].codec) return 1;
  return 0;
}

extern "C" int scan_album_has_opus(void) {
  for (int i = 0; i < g_nb; ++i)
    if (g_tracks[i].codec == CODEC_OPUS) return 1;
  return 0;
}

extern "C" double scan_get_album_peak(void) {
  ensure_scanned();
  return g_nb ? g_album.peak : 0.0;
}

extern "C" void scan_set_album_result(scan_result *result, double pre_gain) {
  ensure_scanned();
  if (scan_album_has_opus()) pre_gain = pre_gain - 5.0f;
  result->album_gain = LUFS_TO_RG(g_album.loudness) + pre_gain;
  result->album_peak = g_album.peak;
  result->album_loudness = g_album.loudness;
  result->album_loudness_range = g_album.lra;
}

extern "C" void scan_get_album_result(scan_result *result, double pre_gain) {
  scan_set_album_result(result, pre_gain);
}
