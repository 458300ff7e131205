// ebur128_shim.cpp -- include/loudscan_ebur128.h on top of the device-level C ABI.
//
// Replaces, for an unmodified scan.c, the nine libebur128 entry points it imports
// (/root/reference/src/scan.c:102,203,294,297,303,371,383,388,448 and
// loudgain.c:179).  Frames are collected on the host as they arrive (S16 like
// scan.c:442 produces them, or f32).
//
// ONE batched scan per session: the first query of any state uploads every live state that
// holds frames (once: pinned double-buffered pieces on a stream of the shim's own, S16 widened
// on the device, the PCM stays in an arena until its state is destroyed) and runs ONE plan --
// all those states as tracks of one album -- whose per-state results, per-channel peaks and
// album result are cached, keyed by (state, frames).  loudgain's main (loudgain.c:299-340: all
// files scanned, then per file scan_get_track_result + scan_set_album_result, i.e. per file
// loudness_global, loudness_range, true_peak per channel and both _multiple calls over all
// states) is served by that one plan; round 2 ran one plan per state plus one per album and
// scanned every track twice.  A query that the cache does not cover (frames added since, a
// _multiple call over another set of states) plans again: exactly the states it needs.
// loudscan_ebur128_plan_count() tells how many plans ran.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <mutex>
#include <new>
#include <vector>

#include "../../include/loudscan_device.h"
#include "../../include/loudscan_ebur128.h"


struct ebur128_state_internal {
  std::vector<short> s16;   // collected frames (one of the two is used, by first call)
  std::vector<float> f32;
  bool is_float = false;
  size_t frames = 0;          // collected
  int device = 0;
  // the collected frames in HBM, interleaved f32 or int16 as they came (a piece of an arena block)
  float *dev = nullptr;
  bool dev_s16 = false;  // `dev` holds interleaved int16 (states fed by ebur128_add_frames_short)
  size_t dev_frames = 0;      // frames that are up there
  int block = -1;             // arena block the piece lives in
  // cached results for `scanned_frames` frames (true peak only if the plan had the interpolator)
  bool have = false, have_tp = false;
  size_t scanned_frames = 0;
  lgd_track_result res;
  std::vector<double> sample_peak, true_peak;
};

namespace {

std::mutex g_mu;
int g_device = 0;
lgd_ctx *g_ctx = nullptr;
int g_ctx_device = -1;
unsigned long long g_plans = 0;
std::vector<ebur128_state *> g_live;  // in creation order

// upload machinery of the session's GPU: arena blocks (freed when their last state goes), two pinned and two
// device staging buffers, a stream
const size_t STAGE_BYTES = 32u << 20, BLOCK_MIN = 256u << 20;
struct Block {
  void *base = nullptr;
  size_t size = 0, used = 0;
  int live = 0;
};
std::vector<Block> g_blocks;
hipStream_t g_stream = nullptr;
void *g_pinned[2] = {nullptr, nullptr};
hipEvent_t g_ev[2] = {nullptr, nullptr};
bool g_ev_used[2] = {false, false};
int g_turn = 0;

// album result of the last plan and the states it covered
std::vector<std::pair<const ebur128_state *, size_t>> g_album_key;  // sorted by address
lgd_album_result g_album;
bool g_album_valid = false;

void drop_machinery() {
  if (g_stream) (void)hipStreamSynchronize(g_stream);
  for (int i = 0; i < 2; ++i) {
    if (g_pinned[i]) (void)hipHostFree(g_pinned[i]);
    if (g_ev[i]) (void)hipEventDestroy(g_ev[i]);
    g_pinned[i] = nullptr;
    g_ev[i] = nullptr;
    g_ev_used[i] = false;
  }
  if (g_stream) (void)hipStreamDestroy(g_stream);
  g_stream = nullptr;
}

lgd_ctx *ctx_for(int device) {
  if (g_ctx && g_ctx_device != device) {
    drop_machinery();
    lgd_destroy(g_ctx);
    g_ctx = nullptr;
  }
  if (!g_ctx) {
    g_ctx = lgd_create(device);
    g_ctx_device = device;
  }
  return g_ctx;
}

bool machinery() {
  if (g_stream) return true;
  if (hipStreamCreateWithFlags(&g_stream, hipStreamNonBlocking) != hipSuccess) return false;
  for (int i = 0; i < 2; ++i)
    if (hipHostMalloc(&g_pinned[i], STAGE_BYTES) != hipSuccess ||
        hipEventCreateWithFlags(&g_ev[i], hipEventDisableTiming) != hipSuccess)
      return false;
  return true;
}

float *arena_alloc(size_t bytes, int *block_out) {
  bytes = (bytes + 255) & ~(size_t)255;
  for (size_t b = 0; b < g_blocks.size(); ++b)
    if (g_blocks[b].base && g_blocks[b].size - g_blocks[b].used >= bytes) {
      float *p = (float *)((char *)g_blocks[b].base + g_blocks[b].used);
      g_blocks[b].used += bytes;
      ++g_blocks[b].live;
      *block_out = (int)b;
      return p;
    }
  Block nb;
  nb.size = std::max(bytes, BLOCK_MIN);
  if (hipMalloc(&nb.base, nb.size) != hipSuccess) return nullptr;
  nb.used = bytes;
  nb.live = 1;
  size_t slot = g_blocks.size();
  for (size_t b = 0; b < g_blocks.size(); ++b)
    if (!g_blocks[b].base) { slot = b; break; }
  if (slot == g_blocks.size()) g_blocks.push_back(nb); else g_blocks[slot] = nb;
  *block_out = (int)slot;
  return (float *)nb.base;
}

void arena_release(int block) {
  if (block < 0 || block >= (int)g_blocks.size() || !g_blocks[block].base) return;
  if (--g_blocks[block].live == 0) {
    if (g_stream) (void)hipStreamSynchronize(g_stream);  // (nothing of the shim's may still read it)
    (void)hipFree(g_blocks[block].base);
    g_blocks[block] = Block();
  }
}

// bring the collected frames into HBM: asynchronous pieces on g_stream, nothing waits for the GPU here except
// for a staging buffer's turn.  A state that received frames after an upload is uploaded again as a whole.
int upload(ebur128_state *st) {
  ebur128_state_internal *d = st->d;
  if (d->dev_frames == d->frames && (d->dev || !d->frames)) return EBUR128_SUCCESS;
  if (hipSetDevice(d->device) != hipSuccess || !machinery()) return EBUR128_ERROR_NOMEM;
  const size_t n = d->frames * st->channels;
  if (d->dev) {
    arena_release(d->block);
    d->dev = nullptr;
    d->block = -1;
  }
  // frames from ebur128_add_frames_short stay S16 in HBM: the kernels' S16 variants read them as they are
  // (LGD_PCM_S16) -- no widening pass, half the arena
  const bool keep_s16 = !d->is_float;
  d->dev_s16 = keep_s16;
  d->dev = arena_alloc((n ? n : 1) * (keep_s16 ? sizeof(short) : sizeof(float)), &d->block);
  if (!d->dev) return EBUR128_ERROR_NOMEM;
  const size_t esz = d->is_float ? sizeof(float) : sizeof(short);
  size_t piece = STAGE_BYTES / esz;
  piece -= piece % 8;  // every piece starts 16-byte aligned in the device buffer
  for (size_t first = 0; first < n; first += piece) {
    const size_t m = std::min(piece, n - first);
    const int b = g_turn;
    g_turn ^= 1;
    if (g_ev_used[b] && hipEventSynchronize(g_ev[b]) != hipSuccess) return EBUR128_ERROR_NOMEM;
    bool ok;
    if (d->is_float) {
      memcpy(g_pinned[b], d->f32.data() + first, m * sizeof(float));
      ok = hipMemcpyAsync(d->dev + first, g_pinned[b], m * sizeof(float), hipMemcpyHostToDevice, g_stream) == hipSuccess;
    } else {
      memcpy(g_pinned[b], d->s16.data() + first, m * sizeof(short));
      ok = hipMemcpyAsync((short *)d->dev + first, g_pinned[b], m * sizeof(short), hipMemcpyHostToDevice, g_stream) == hipSuccess;
    }
    if (!ok || hipEventRecord(g_ev[b], g_stream) != hipSuccess) return EBUR128_ERROR_NOMEM;
    g_ev_used[b] = true;
  }
  d->dev_frames = d->frames;
  return EBUR128_SUCCESS;
}

bool fresh(const ebur128_state *st, bool want_tp) {
  const ebur128_state_internal *d = st->d;
  return d->have && d->scanned_frames == d->frames && (d->have_tp || !want_tp);
}

std::vector<std::pair<const ebur128_state *, size_t>> key_of(ebur128_state *const *sts, size_t n) {
  std::vector<std::pair<const ebur128_state *, size_t>> k;
  for (size_t i = 0; i < n; ++i) k.emplace_back(sts[i], sts[i]->d->frames);
  std::sort(k.begin(), k.end());
  return k;
}

// one plan: the given states as the tracks of one album; fills every state's cache and the album cache
int scan_states(const std::vector<ebur128_state *> &sts) {
  if (sts.empty()) return EBUR128_SUCCESS;
  const int device = sts[0]->d->device;
  bool tp = false;
  for (ebur128_state *s : sts) {
    if (s->d->device != device) return EBUR128_ERROR_INVALID_MODE;  // one GPU per session here
    tp = tp || (s->mode & EBUR128_MODE_TRUE_PEAK) == EBUR128_MODE_TRUE_PEAK;
  }
  lgd_ctx *c = ctx_for(device);
  if (!c) return EBUR128_ERROR_NOMEM;
  std::vector<lgd_track> t(sts.size());
  std::vector<uint8_t> fmt(sts.size());
  for (size_t i = 0; i < sts.size(); ++i) {
    const int rc = upload(sts[i]);
    if (rc) return rc;
    fmt[i] = (uint8_t)(sts[i]->d->dev_s16 ? LGD_PCM_S16 : LGD_PCM_F32);
    t[i].pcm = sts[i]->d->dev;
    t[i].frames = sts[i]->d->frames;
    t[i].channels = sts[i]->channels;
    t[i].rate = (uint32_t)sts[i]->samplerate;
  }
  std::vector<lgd_track_result> r(sts.size());
  g_album_valid = false;
  // (the scan follows the uploads on the shim's stream; lgd_fetch waits for it)
  if (lgd_plan_formats(c, fmt.data(), (uint32_t)sts.size()) ||
      lgd_plan(c, t.data(), (uint32_t)sts.size(), (tp ? LGD_FLAG_TRUE_PEAK : 0u) | LGD_FLAG_ALBUM) ||
      lgd_execute(c, g_stream) || lgd_fetch(c, r.data(), &g_album))
    return EBUR128_ERROR_NOMEM;
  ++g_plans;
  for (size_t i = 0; i < sts.size(); ++i) {
    ebur128_state_internal *d = sts[i]->d;
    d->res = r[i];
    d->sample_peak.assign(sts[i]->channels, 0.0);
    d->true_peak.assign(sts[i]->channels, 0.0);
    if (lgd_copy_channel_peaks(c, (uint32_t)i, d->sample_peak.data(), d->true_peak.data(), sts[i]->channels))
      return EBUR128_ERROR_NOMEM;
    d->have = true;
    d->have_tp = tp;
    d->scanned_frames = d->frames;
  }
  g_album_key = key_of(sts.data(), sts.size());
  g_album_valid = true;
  return EBUR128_SUCCESS;
}

// a single-state query: served from the cache, else by one plan over EVERY live state that holds frames not yet
// scanned (the caller is most likely about to ask for the others and for their album: loudgain.c:334-340;
// states whose results are up to date are left alone)
int scan_one(ebur128_state *st) {
  const bool want_tp = (st->mode & EBUR128_MODE_TRUE_PEAK) == EBUR128_MODE_TRUE_PEAK;
  if (fresh(st, want_tp)) return EBUR128_SUCCESS;
  std::vector<ebur128_state *> sts;
  for (ebur128_state *s : g_live) {
    const bool s_tp = (s->mode & EBUR128_MODE_TRUE_PEAK) == EBUR128_MODE_TRUE_PEAK;
    if (s == st || (s->d->frames && s->d->device == st->d->device && !fresh(s, s_tp))) sts.push_back(s);
  }
  return scan_states(sts);
}

int scan_multi(ebur128_state **sts, size_t size) {
  if (g_album_valid && key_of(sts, size) == g_album_key) return EBUR128_SUCCESS;
  std::vector<ebur128_state *> v;
  for (size_t i = 0; i < size; ++i)
    if (std::find(v.begin(), v.end(), sts[i]) == v.end()) v.push_back(sts[i]);
  if (v.size() != size) return EBUR128_ERROR_INVALID_MODE;  // the same state twice
  if (size == 0) {
    memset(&g_album, 0, sizeof(g_album));
    g_album.loudness = -HUGE_VAL;
    g_album_key.clear();
    g_album_valid = true;
    return EBUR128_SUCCESS;
  }
  return scan_states(v);
}

template <typename T>
int add_frames(ebur128_state *st, std::vector<T> &dst, const T *src, size_t frames, bool is_float) {
  ebur128_state_internal *d = st->d;
  if (d->frames && d->is_float != is_float) return EBUR128_ERROR_INVALID_MODE;  // one sample type per state
  d->is_float = is_float;
  try {
    dst.insert(dst.end(), src, src + frames * st->channels);
  } catch (const std::bad_alloc &) {
    return EBUR128_ERROR_NOMEM;
  }
  d->frames += frames;
  return EBUR128_SUCCESS;
}

}  // namespace

extern "C" int loudscan_ebur128_set_device(int device) {
  std::lock_guard<std::mutex> lk(g_mu);
  g_device = device;
  return 0;
}

extern "C" unsigned long long loudscan_ebur128_plan_count(void) {
  std::lock_guard<std::mutex> lk(g_mu);
  return g_plans;
}

extern "C" void ebur128_get_version(int *major, int *minor, int *patch) {
  *major = EBUR128_VERSION_MAJOR;
  *minor = EBUR128_VERSION_MINOR;
  *patch = EBUR128_VERSION_PATCH;
}

extern "C" ebur128_state *ebur128_init(unsigned int channels, unsigned long samplerate, int mode) {
  std::lock_guard<std::mutex> lk(g_mu);
  // libebur128 1.2.4 argument limits; the scanner itself needs rate >= 4 kHz, <= 64 channels
  if (channels == 0 || channels > LGD_MAX_CHANNELS || samplerate < 16 || samplerate > 2822400) return nullptr;
  if (samplerate < 4000) return nullptr;
  if (!ctx_for(g_device)) return nullptr;  // no HIP device: no CPU fallback
  ebur128_state *st = new (std::nothrow) ebur128_state;
  if (!st) return nullptr;
  st->d = new (std::nothrow) ebur128_state_internal;
  if (!st->d) {
    delete st;
    return nullptr;
  }
  st->mode = mode;
  st->channels = channels;
  st->samplerate = samplerate;
  st->d->device = g_device;
  g_live.push_back(st);
  return st;
}

extern "C" void ebur128_destroy(ebur128_state **st) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (!st || !*st) return;
  g_live.erase(std::remove(g_live.begin(), g_live.end(), *st), g_live.end());
  if ((*st)->d) {
    if ((*st)->d->dev) arena_release((*st)->d->block);
    delete (*st)->d;
  }
  g_album_valid = false;  // its address may be reused
  delete *st;
  *st = nullptr;
  if (g_live.empty() && g_ctx) {  // the session is over: give the GPU back
    drop_machinery();
    lgd_destroy(g_ctx);
    g_ctx = nullptr;
    g_ctx_device = -1;
  }
}

extern "C" int ebur128_add_frames_short(ebur128_state *st, const short *src, size_t frames) {
  std::lock_guard<std::mutex> lk(g_mu);
  return add_frames(st, st->d->s16, src, frames, false);
}
extern "C" int ebur128_add_frames_float(ebur128_state *st, const float *src, size_t frames) {
  std::lock_guard<std::mutex> lk(g_mu);
  return add_frames(st, st->d->f32, src, frames, true);
}

extern "C" int ebur128_loudness_global(ebur128_state *st, double *out) {
  std::lock_guard<std::mutex> lk(g_mu);
  if ((st->mode & EBUR128_MODE_I) != EBUR128_MODE_I) return EBUR128_ERROR_INVALID_MODE;
  int rc = scan_one(st);
  if (rc) return rc;
  *out = st->d->res.loudness;
  return EBUR128_SUCCESS;
}

extern "C" int ebur128_loudness_range(ebur128_state *st, double *out) {
  std::lock_guard<std::mutex> lk(g_mu);
  if ((st->mode & EBUR128_MODE_LRA) != EBUR128_MODE_LRA) return EBUR128_ERROR_INVALID_MODE;
  int rc = scan_one(st);
  if (rc) return rc;
  *out = st->d->res.lra;
  return EBUR128_SUCCESS;
}

extern "C" int ebur128_sample_peak(ebur128_state *st, unsigned int ch, double *out) {
  std::lock_guard<std::mutex> lk(g_mu);
  if ((st->mode & EBUR128_MODE_SAMPLE_PEAK) != EBUR128_MODE_SAMPLE_PEAK) return EBUR128_ERROR_INVALID_MODE;
  if (ch >= st->channels) return EBUR128_ERROR_INVALID_CHANNEL_INDEX;
  int rc = scan_one(st);
  if (rc) return rc;
  *out = st->d->sample_peak[ch];
  return EBUR128_SUCCESS;
}

extern "C" int ebur128_true_peak(ebur128_state *st, unsigned int ch, double *out) {
  std::lock_guard<std::mutex> lk(g_mu);
  if ((st->mode & EBUR128_MODE_TRUE_PEAK) != EBUR128_MODE_TRUE_PEAK) return EBUR128_ERROR_INVALID_MODE;
  if (ch >= st->channels) return EBUR128_ERROR_INVALID_CHANNEL_INDEX;
  int rc = scan_one(st);
  if (rc) return rc;
  // 1.2.4: the larger of the interpolated and the sample peak of that channel
  const double t = st->d->true_peak[ch], s = st->d->sample_peak[ch];
  *out = t > s ? t : s;
  return EBUR128_SUCCESS;
}

extern "C" int ebur128_loudness_global_multiple(ebur128_state **sts, size_t size, double *out) {
  std::lock_guard<std::mutex> lk(g_mu);
  for (size_t i = 0; i < size; ++i)
    if ((sts[i]->mode & EBUR128_MODE_I) != EBUR128_MODE_I) return EBUR128_ERROR_INVALID_MODE;
  int rc = scan_multi(sts, size);
  if (rc) return rc;
  *out = g_album.loudness;
  return EBUR128_SUCCESS;
}

extern "C" int ebur128_loudness_range_multiple(ebur128_state **sts, size_t size, double *out) {
  std::lock_guard<std::mutex> lk(g_mu);
  for (size_t i = 0; i < size; ++i)
    if ((sts[i]->mode & EBUR128_MODE_LRA) != EBUR128_MODE_LRA) return EBUR128_ERROR_INVALID_MODE;
  int rc = scan_multi(sts, size);
  if (rc) return rc;
  *out = g_album.lra;
  return EBUR128_SUCCESS;
}
