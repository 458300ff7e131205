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


// FFmpeg AVCodecID values (recalled; FFmpeg headers are absent here).  The
// reference only compares them for equality and against OPUS.
enum {
  CODEC_PCM_S16LE = 0x10000, CODEC_PCM_U8 = 0x10005, CODEC_PCM_S32LE = 0x10008,
  CODEC_PCM_S24LE = 0x1000C, CODEC_PCM_F32LE = 0x10015, CODEC_PCM_F64LE = 0x10017,
  CODEC_OPUS = 0x1503C
};

#define LUFS_TO_RG(L) (-18 - (L))

namespace {

// PCM moves host -> HBM in pieces of this many bytes through two pinned buffers per GPU
const size_t STAGE_BYTES = 64u << 20;
const size_t ARENA_MIN = 64u << 20, ARENA_MAX = 1u << 30;

// One GPU of the session: its engine context, its stream, an arena the tracks' PCM (S16 or f32, as it came) lives in
// until scan_deinit (one hipMalloc per ~GB, not per file), pinned double-buffered staging.
struct Dev {
  int hip_id = 0;
  lgd_ctx *ctx = nullptr;
  hipStream_t stream = nullptr;
  std::vector<void *> blocks;  // arena blocks
  char *cur = nullptr;
  size_t cur_left = 0, next_block = ARENA_MIN;
  void *pinned[2] = {nullptr, nullptr};
  hipEvent_t ev[2] = {nullptr, nullptr};    // piece i of a buffer has left it (copy done)
  bool ev_used[2] = {false, false};
  int turn = 0;
  std::vector<int> tracks;  // file indices scanned on this GPU, ascending
};

struct Track {
  char *file = nullptr, *container = nullptr;
  int codec = 0;
  unsigned channels = 0, rate = 0;
  size_t frames = 0;
  float *dev = nullptr;  // interleaved f32 in HBM (arena of its GPU, or borrowed) -- or int16, see s16
  bool s16 = false;      // the PCM stays resident as the interleaved S16 it arrived as (LGD_PCM_S16: mono / stereo)
  int gpu = 0;           // index into g_devs
  bool loaded = false;
};

std::vector<Track> g_tracks;
std::vector<Dev> g_devs;
int g_nb = 0, g_device = 0, g_want_devices = 1;
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
  free(t.file);
  free(t.container);
  t = Track();
}

void dev_open(Dev &d) {
  HIPFATAL(hipSetDevice(d.hip_id));
  if (!d.stream) HIPFATAL(hipStreamCreateWithFlags(&d.stream, hipStreamNonBlocking));
  for (int i = 0; i < 2; ++i) {
    if (!d.pinned[i]) HIPFATAL(hipHostMalloc(&d.pinned[i], STAGE_BYTES));
    if (!d.ev[i]) HIPFATAL(hipEventCreateWithFlags(&d.ev[i], hipEventDisableTiming));
  }
}

void dev_close(Dev &d) {
  (void)hipSetDevice(d.hip_id);
  if (d.stream) (void)hipStreamSynchronize(d.stream);
  if (d.ctx) lgd_destroy(d.ctx);
  for (void *b : d.blocks) (void)hipFree(b);
  for (int i = 0; i < 2; ++i) {
    if (d.pinned[i]) (void)hipHostFree(d.pinned[i]);
    if (d.ev[i]) (void)hipEventDestroy(d.ev[i]);
  }
  if (d.stream) (void)hipStreamDestroy(d.stream);
  d = Dev();
}

// 256-B aligned piece of the GPU's arena (lives until scan_deinit: a file may be scanned again,
// like the reference allows, and the album needs every track at the first result query)
void *arena_alloc(Dev &d, size_t bytes) {
  bytes = (bytes + 255) & ~(size_t)255;
  if (bytes > d.cur_left) {
    size_t blk = std::max(bytes, d.next_block);
    void *p = nullptr;
    HIPFATAL(hipSetDevice(d.hip_id));
    HIPFATAL(hipMalloc(&p, blk));
    d.blocks.push_back(p);
    d.cur = (char *)p;
    d.cur_left = blk;
    d.next_block = std::min(ARENA_MAX, d.next_block * 2);
  }
  void *r = d.cur;
  d.cur += bytes;
  d.cur_left -= bytes;
  return r;
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
  t.gpu = (int)(index % g_devs.size());  // tracks are dealt round-robin over the GPUs
  t.loaded = true;
  g_scanned = false;
}

// Streams `total` samples into the track's buffer in HBM.  `fill(dst, first, n)` puts samples
// [first, first + n) into a pinned buffer, as S16 (the grid scan.c:414 puts every input on: half the PCIe
// bytes and half the HBM, scanned as it is) or as f32.  Two pinned buffers alternate: piece i + 1 is
// produced while piece i crosses PCIe; nothing here waits for the GPU except for a buffer's turn.
template <typename Fill>
void upload(Track &t, size_t total, bool as_s16, Fill fill) {
  Dev &d = g_devs[t.gpu];
  HIPFATAL(hipSetDevice(d.hip_id));
  // S16 input stays S16 in HBM: the kernels' S16 variants read it as it is (what ebur128_add_frames_short is
  // handed at scan.c:448) -- no widening pass, half the arena
  t.s16 = as_s16;
  t.dev = (float *)arena_alloc(d, (total ? total : 1) * (t.s16 ? sizeof(short) : sizeof(float)));
  // A piece is a whole number of FRAMES (and of 8 samples, which keeps every piece's start 16-B aligned):
  // `fill` may come from a sequential reader that delivers whole frames, so a piece
  // that ended inside a frame would shift everything behind it (round 2 cut at 2^25 samples whatever the
  // channel count: 3 / 5 / 6 / 7-channel files longer than one piece got their later pieces 2 samples late
  // -- channels rotated, the last samples lost).
  size_t piece = STAGE_BYTES / (as_s16 ? sizeof(short) : sizeof(float));
  piece -= piece % (8 * (size_t)t.channels);
  for (size_t first = 0; first < total; first += piece) {
    const size_t n = std::min(piece, total - first);
    const int b = d.turn;
    d.turn ^= 1;
    if (d.ev_used[b]) HIPFATAL(hipEventSynchronize(d.ev[b]));  // the buffer's previous piece has left it
    fill(d.pinned[b], first, n);
    if (t.s16) {
      HIPFATAL(hipMemcpyAsync((short *)t.dev + first, d.pinned[b], n * sizeof(short), hipMemcpyHostToDevice, d.stream));
    } else {
      HIPFATAL(hipMemcpyAsync(t.dev + first, d.pinned[b], n * sizeof(float), hipMemcpyHostToDevice, d.stream));
    }
    HIPFATAL(hipEventRecord(d.ev[b], d.stream));
    d.ev_used[b] = true;
  }
}

uint32_t rd32(const unsigned char *p) {
  return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}
uint16_t rd16(const unsigned char *p) { return (uint16_t)(p[0] | (p[1] << 8)); }
short clip16(long v) { return (short)(v < -32768 ? -32768 : (v > 32767 ? 32767 : v)); }

void copy_results(const std::vector<int> &idx, const std::vector<lgd_track_result> &r) {
  for (size_t k = 0; k < idx.size(); ++k) g_res[idx[k]] = r[k];
}

// The whole session in one batched scan per GPU.  One GPU: tracks and album in one plan.
// Several GPUs (scan.c's album walk over all states, scan.c:359-405, split over the devices of this
// process): every GPU scans its tracks, then the exchange of loudgain_amd/album.py through the host
// -- record 1 of every GPU (partial sums, peak, its listed 3 s energies) to all, relative gate from
// the heads in device order, second pass, record 2 to all, album loudness / range / peak on every
// GPU (identical bits) -- a few MB at most.
void ensure_scanned() {
  if (g_scanned) return;
  for (int i = 0; i < g_nb; ++i)
    if (!g_tracks[i].loaded) fail("scan_file was not called for index %d", i);
  const size_t nd = g_devs.size();
  g_res.assign(g_nb ? g_nb : 1, lgd_track_result());
  std::vector<std::vector<lgd_track>> lt(nd);
  std::vector<std::vector<uint8_t>> lf(nd);  // LGD_PCM_* per track of a GPU's plan
  for (Dev &d : g_devs) d.tracks.clear();
  for (int i = 0; i < g_nb; ++i) {
    const Track &t = g_tracks[i];
    lgd_track x;
    x.pcm = t.dev;
    x.frames = t.frames;
    x.channels = t.channels;
    x.rate = t.rate;
    lt[t.gpu].push_back(x);
    lf[t.gpu].push_back((uint8_t)(t.s16 ? LGD_PCM_S16 : LGD_PCM_F32));
    g_devs[t.gpu].tracks.push_back(i);
  }
  for (Dev &d : g_devs) {
    HIPFATAL(hipSetDevice(d.hip_id));
    if (!d.ctx) {
      d.ctx = lgd_create(d.hip_id);
      if (!d.ctx) fail("%s", lgd_last_error());
    }
  }
  if (nd == 1) {
    Dev &d = g_devs[0];
    std::vector<lgd_track_result> r(g_nb ? g_nb : 1);
    if (lgd_plan_formats(d.ctx, lf[0].data(), (uint32_t)g_nb) ||
        lgd_plan(d.ctx, lt[0].data(), (uint32_t)g_nb, LGD_FLAG_TRUE_PEAK | LGD_FLAG_ALBUM) ||
        lgd_execute(d.ctx, d.stream) || lgd_fetch(d.ctx, r.data(), &g_album))
      fail("%s", lgd_last_error());
    copy_results(d.tracks, r);
    g_scanned = true;
    return;
  }
  // ---- several GPUs
  const uint32_t flags = LGD_FLAG_TRUE_PEAK | LGD_FLAG_ALBUM_PART1;
  uint64_t slots = 0;
  for (size_t k = 0; k < nd; ++k) {  // record 1 must have one length everywhere: the longest
    Dev &d = g_devs[k];
    double *p;
    uint64_t n;
    if (lgd_set_param(d.ctx, "album_slots", 0) || lgd_set_param(d.ctx, "album_world", (long)nd) ||
        lgd_plan_formats(d.ctx, lf[k].data(), (uint32_t)lf[k].size()) ||
        lgd_plan(d.ctx, lt[k].data(), (uint32_t)lt[k].size(), flags) || lgd_album_record1(d.ctx, &p, &n))
      fail("%s", lgd_last_error());
    slots = std::max(slots, n - 4);
  }
  for (size_t k = 0; k < nd; ++k) {
    Dev &d = g_devs[k];
    if (lgd_set_param(d.ctx, "album_slots", (long)slots) || lgd_plan_formats(d.ctx, lf[k].data(), (uint32_t)lf[k].size()) ||
        lgd_plan(d.ctx, lt[k].data(), (uint32_t)lt[k].size(), flags) || lgd_execute(d.ctx, d.stream))
      fail("%s", lgd_last_error());  // (all GPUs now scan at the same time)
  }
  const uint64_t n1 = slots + 4;
  std::vector<double> all1(nd * n1), all2(nd * 2);
  std::vector<double *> d_all1(nd), d_all2(nd);
  for (size_t k = 0; k < nd; ++k) {
    Dev &d = g_devs[k];
    double *p;
    uint64_t n;
    HIPFATAL(hipSetDevice(d.hip_id));
    if (lgd_album_join(d.ctx, d.stream) || lgd_album_record1(d.ctx, &p, &n)) fail("%s", lgd_last_error());
    HIPFATAL(hipMemcpyAsync(&all1[k * n1], p, n1 * sizeof(double), hipMemcpyDeviceToHost, d.stream));
  }
  for (Dev &d : g_devs) {
    HIPFATAL(hipSetDevice(d.hip_id));
    HIPFATAL(hipStreamSynchronize(d.stream));
  }
  for (size_t k = 0; k < nd; ++k) {
    Dev &d = g_devs[k];
    HIPFATAL(hipSetDevice(d.hip_id));
    d_all1[k] = (double *)arena_alloc(d, all1.size() * sizeof(double));
    d_all2[k] = (double *)arena_alloc(d, all2.size() * sizeof(double));
    HIPFATAL(hipMemcpyAsync(d_all1[k], all1.data(), all1.size() * sizeof(double), hipMemcpyHostToDevice, d.stream));
    double *r2;
    if (lgd_album_stage2(d.ctx, d_all1[k], (uint32_t)nd, d.stream) || lgd_album_record2(d.ctx, &r2))
      fail("%s", lgd_last_error());
    HIPFATAL(hipMemcpyAsync(&all2[k * 2], r2, 2 * sizeof(double), hipMemcpyDeviceToHost, d.stream));
  }
  for (Dev &d : g_devs) {
    HIPFATAL(hipSetDevice(d.hip_id));
    HIPFATAL(hipStreamSynchronize(d.stream));
  }
  for (size_t k = 0; k < nd; ++k) {
    Dev &d = g_devs[k];
    HIPFATAL(hipSetDevice(d.hip_id));
    HIPFATAL(hipMemcpyAsync(d_all2[k], all2.data(), all2.size() * sizeof(double), hipMemcpyHostToDevice, d.stream));
    if (lgd_album_stage3(d.ctx, d_all2[k], (uint32_t)nd, d.stream)) fail("%s", lgd_last_error());
  }
  for (size_t k = 0; k < nd; ++k) {
    Dev &d = g_devs[k];
    HIPFATAL(hipSetDevice(d.hip_id));
    std::vector<lgd_track_result> r(lt[k].size() ? lt[k].size() : 1);
    lgd_album_result a;
    if (lgd_fetch(d.ctx, r.data(), &a)) fail("%s", lgd_last_error());
    copy_results(d.tracks, r);
    if (k == 0) g_album = a;  // (every GPU holds the same album numbers)
  }
  g_scanned = true;
}

}  // namespace

extern "C" int scan_set_device(int device) {
  g_device = device;
  g_want_devices = 1;
  return 0;
}

extern "C" int scan_set_devices(int n_devices) {
  g_want_devices = n_devices;  // 0 = every visible GPU
  return 0;
}

extern "C" int scan_init(unsigned nb_files) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || g_device >= n)
    fail("no MI355X / HIP device %d available (the scanner has no CPU path)", g_device);
  if (!g_devs.empty() || !g_tracks.empty()) scan_deinit();  // a session that was never closed
  int want = g_want_devices == 0 ? n : g_want_devices;
  // LOUDSCAN_DEVICES=k: use k GPUs (0 = all) without touching the caller; LOUDSCAN_VIRTUAL_DEVICES=k:
  // k engine contexts dealt over the real GPUs -- rehearses the multi-GPU album on a one-GPU box
  int virt = 0;
  if (const char *e = getenv("LOUDSCAN_DEVICES")) want = atoi(e) == 0 ? n : atoi(e);
  if (const char *e = getenv("LOUDSCAN_VIRTUAL_DEVICES")) virt = atoi(e);
  if (want > n) want = n;
  if (want < 1) want = 1;
  const int nd = virt > 0 ? virt : want;
  g_devs.assign((size_t)nd, Dev());
  for (int k = 0; k < nd; ++k) {
    g_devs[k].hip_id = want == 1 ? g_device : k % want;
    dev_open(g_devs[k]);
  }
  g_nb = (int)nb_files;
  g_tracks.assign(nb_files, Track());
  g_scanned = false;
  return 0;
}

extern "C" void scan_deinit(void) {
  for (Track &t : g_tracks) release_track(t);
  g_tracks.clear();
  g_res.clear();
  for (Dev &d : g_devs) dev_close(d);
  g_devs.clear();
  g_nb = 0;
  g_scanned = false;
}

extern "C" int scan_pcm_s16(const short *pcm, size_t frames, unsigned channels, unsigned rate,
                            unsigned index) {
  if ((int)index >= g_nb) return -1;
  begin_track(index, "<pcm_s16>", "wav", CODEC_PCM_S16LE, channels, rate, frames);
  upload(g_tracks[index], frames * channels, true,
         [&](void *dst, size_t first, size_t n) { memcpy(dst, pcm + first, n * sizeof(short)); });
  return 0;
}

extern "C" int scan_pcm_f32(const float *pcm, size_t frames, unsigned channels, unsigned rate,
                            unsigned index) {
  if ((int)index >= g_nb) return -1;
  begin_track(index, "<pcm_f32>", "wav", CODEC_PCM_F32LE, channels, rate, frames);
  upload(g_tracks[index], frames * channels, false,
         [&](void *dst, size_t first, size_t n) { memcpy(dst, pcm + first, n * sizeof(float)); });
  return 0;
}

extern "C" int scan_pcm_f32_device(const float *dev, size_t frames, unsigned channels, unsigned rate,
                                   unsigned index) {
  if ((int)index >= g_nb) return -1;
  begin_track(index, "<pcm_f32_device>", "wav", CODEC_PCM_F32LE, channels, rate, frames);
  Track &t = g_tracks[index];
  t.dev = const_cast<float *>(dev);
  // a borrowed buffer is scanned where it lives
  hipPointerAttribute_t attr;
  if (hipPointerGetAttributes(&attr, dev) == hipSuccess)
    for (size_t k = 0; k < g_devs.size(); ++k)
      if (g_devs[k].hip_id == attr.device) { t.gpu = (int)k; break; }
  return 0;
}

extern "C" int scan_pcm_s16_device(const short *dev, size_t frames, unsigned channels, unsigned rate,
                                   unsigned index) {
  if ((int)index >= g_nb) return -1;
  begin_track(index, "<pcm_s16_device>", "wav", CODEC_PCM_S16LE, channels, rate, frames);
  Track &t = g_tracks[index];
  hipPointerAttribute_t attr;
  if (hipPointerGetAttributes(&attr, dev) == hipSuccess)
    for (size_t k = 0; k < g_devs.size(); ++k)
      if (g_devs[k].hip_id == attr.device) { t.gpu = (int)k; break; }
  t.dev = reinterpret_cast<float *>(const_cast<short *>(dev));  // scanned where and as it lies
  t.s16 = true;
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
      {
        // never trust the announced size beyond the file: a streamed / piped WAV says 0 or
        // 0xFFFFFFFF (FFmpeg's demuxer then reads to the end, scan.c:225), a truncated one too much
        fseek(fp, 0, SEEK_END);
        const long end = ftell(fp);
        const size_t avail = end > wi->data_off ? (size_t)(end - wi->data_off) : 0;
        const size_t bytes = (sz == 0 || sz == 0xFFFFFFFFu || sz > avail) ? avail : sz;
        wi->frames = bytes / wi->block_align;
      }
      rc = WAV_OK;
      break;
    } else {
      fseek(fp, (long)(sz + (sz & 1)), SEEK_CUR);
    }
  }
  fclose(fp);
  return rc;
}

// any PCM flavour of the data chunk -> interleaved S16 (what swr_convert yields at scan.c:442)
void convert_frames(const WavInfo &wi, const unsigned char *raw, size_t got, short *o) {
  const unsigned ch = wi.channels, ba = wi.block_align, bps = wi.bits / 8;
  for (size_t f = 0; f < got; ++f) {
    const unsigned char *p = raw + f * ba;
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
}

// the data chunk -> interleaved S16, as swr_convert does for every decoded frame at
// scan.c:442; returns the frames actually present (a truncated file is silently
// shortened, like the packet loop at scan.c:229-240), or -1 if the file vanished
long long wav_read_s16(const char *file, const WavInfo &wi, short *out, size_t cap_frames) {
  FILE *fp = fopen(file, "rb");
  if (!fp) return -1;
  fseek(fp, wi.data_off, SEEK_SET);
  const size_t want = wi.frames < cap_frames ? wi.frames : cap_frames;
  const unsigned ch = wi.channels, ba = wi.block_align;
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
    convert_frames(wi, raw.data(), got, out + done * ch);
    done += got;
    if (got < n) break;
  }
  fclose(fp);
  return (long long)done;
}

// sequential reader over the data chunk for scan_file's piecewise upload
struct WavReader {
  FILE *fp;
  WavInfo wi;
  std::vector<unsigned char> raw;
  WavReader(const char *file, const WavInfo &w) : fp(fopen(file, "rb")), wi(w) {
    if (!fp) fail("Could not open input: %s", file);
    fseek(fp, wi.data_off, SEEK_SET);
  }
  ~WavReader() { if (fp) fclose(fp); }
  // up to `frames` frames as interleaved S16; returns the frames delivered
  size_t read(short *out, size_t frames) {
    const unsigned ch = wi.channels, ba = wi.block_align;
    if (wi.codec == CODEC_PCM_S16LE && ba == ch * 2) return frames ? fread(out, ba, frames, fp) : 0;
    size_t done = 0;
    raw.resize((size_t)65536 * ba);
    while (done < frames) {
      const size_t n = std::min<size_t>(65536, frames - done);
      const size_t got = fread(raw.data(), ba, n, fp);
      if (!got) break;
      convert_frames(wi, raw.data(), got, out + done * ch);
      done += got;
      if (got < n) break;
    }
    return done;
  }
};

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
  begin_track(index, file, "wav", wi.codec, wi.channels, wi.rate, wi.frames);
  // straight from the file into the pinned staging buffers, piece by piece (the packet loop of
  // scan.c:225-250 without the per-frame swr_init / av_malloc); a file that ends early leaves
  // silence behind, like the reference it is "silently truncated" (its frames stay as announced
  // by the -- already clamped -- data chunk)
  WavReader rd(file, wi);
  size_t at = 0;  // the reader's position in samples: pieces must arrive in order and frame-aligned
  upload(g_tracks[index], wi.frames * wi.channels, true, [&](void *dst, size_t first, size_t n) {
    if (first != at || n % wi.channels) fail("scan_file: staging piece [%zu, +%zu) is not where the reader is (%zu)", first, n, at);
    const size_t got = rd.read((short *)dst, n / wi.channels) * wi.channels;
    if (got < n) memset((short *)dst + got, 0, (n - got) * sizeof(short));
    at += n;
  });
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

// ebur128_true_peak per channel, the loop of scan.c:300-307 before its maximum is taken: what a
// caller (or a test of the channel order) needs to see every channel on its own.  Returns the
// channel count, -1 for a bad index or too small a `cap`.
extern "C" int scan_get_channel_peaks(unsigned index, double *sample_peak, double *true_peak, unsigned cap) {
  if ((int)index >= g_nb) {
    errmsg("Index too high");
    return -1;
  }
  ensure_scanned();
  const Track &t = g_tracks[index];
  if (cap < t.channels) return -1;
  Dev &d = g_devs[t.gpu];
  const auto it = std::find(d.tracks.begin(), d.tracks.end(), (int)index);
  if (it == d.tracks.end()) fail("scan_get_channel_peaks: track %u is not in its GPU's plan", index);
  HIPFATAL(hipSetDevice(d.hip_id));
  if (lgd_copy_channel_peaks(d.ctx, (uint32_t)(it - d.tracks.begin()), sample_peak, true_peak, cap))
    fail("%s", lgd_last_error());
  return (int)t.channels;
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
