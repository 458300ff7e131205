// ebur128_shim.cpp -- include/loudscan_ebur128.h on top of the device-level C ABI.
//
// Replaces, for an unmodified scan.c, the nine libebur128 entry points it imports
// (/root/reference/src/scan.c:102,203,294,297,303,371,383,388,448 and
// loudgain.c:179).  Frames are collected on the host as they arrive (S16 like
// scan.c:442 produces them, or f32); the first query uploads them once (S16 crosses
// PCIe as 2 bytes per sample and is widened on the device) and runs one batched scan.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>
#include <mutex>
#include <new>
#include <vector>

#include "../../include/loudscan_device.h"
#include "../../include/loudscan_ebur128.h"

extern "C" hipError_t lgd_launch_s16_to_f32(const short *in, float *out, size_t n, hipStream_t s);

struct ebur128_state_internal {
  std::vector<short> s16;   // collected frames (one of the two is used, by first call)
  std::vector<float> f32;
  bool is_float = false;
  size_t frames = 0;          // collected
  float *dev = nullptr;       // all collected frames as interleaved f32 in HBM
  size_t dev_frames = 0, dev_cap = 0;
  int device = 0;
  // cached single-state results for `scanned_frames` frames
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

// result of the last _multiple call, reused by the sibling call on the same states
struct MultiKey {
  std::vector<std::pair<const ebur128_state *, size_t>> v;
  bool operator==(const MultiKey &o) const { return v == o.v; }
};
MultiKey g_multi_key;
lgd_album_result g_multi;
bool g_multi_valid = false;

lgd_ctx *ctx_for(int device) {
  if (g_ctx && g_ctx_device != device) {
    lgd_destroy(g_ctx);
    g_ctx = nullptr;
  }
  if (!g_ctx) {
    g_ctx = lgd_create(device);
    g_ctx_device = device;
  }
  return g_ctx;
}

// bring the collected frames into HBM (whole track; queries between add_frames calls
// re-upload, which a scanner that queries once at the end never pays)
int upload(ebur128_state *st) {
  ebur128_state_internal *d = st->d;
  if (d->dev_frames == d->frames && d->dev) return EBUR128_SUCCESS;
  if (hipSetDevice(d->device) != hipSuccess) return EBUR128_ERROR_NOMEM;
  const size_t n = d->frames * st->channels;
  if (n > d->dev_cap || !d->dev) {
    if (d->dev) (void)hipFree(d->dev);
    d->dev = nullptr;
    d->dev_cap = 0;
    if (hipMalloc((void **)&d->dev, (n ? n : 1) * sizeof(float)) != hipSuccess) return EBUR128_ERROR_NOMEM;
    d->dev_cap = n;
  }
  if (n) {
    if (d->is_float) {
      if (hipMemcpy(d->dev, d->f32.data(), n * sizeof(float), hipMemcpyHostToDevice) != hipSuccess)
        return EBUR128_ERROR_NOMEM;
    } else {
      short *tmp = nullptr;
      if (hipMalloc((void **)&tmp, n * sizeof(short)) != hipSuccess) return EBUR128_ERROR_NOMEM;
      bool ok = hipMemcpy(tmp, d->s16.data(), n * sizeof(short), hipMemcpyHostToDevice) == hipSuccess &&
                lgd_launch_s16_to_f32(tmp, d->dev, n, nullptr) == hipSuccess &&
                hipDeviceSynchronize() == hipSuccess;
      (void)hipFree(tmp);
      if (!ok) return EBUR128_ERROR_NOMEM;
    }
  }
  d->dev_frames = d->frames;
  return EBUR128_SUCCESS;
}

int scan_one(ebur128_state *st) {
  ebur128_state_internal *d = st->d;
  const bool want_tp = (st->mode & EBUR128_MODE_TRUE_PEAK) == EBUR128_MODE_TRUE_PEAK;
  if (d->have && d->scanned_frames == d->frames && d->have_tp == want_tp) return EBUR128_SUCCESS;
  int rc = upload(st);
  if (rc) return rc;
  lgd_ctx *c = ctx_for(d->device);
  if (!c) return EBUR128_ERROR_NOMEM;
  lgd_track t;
  t.pcm = d->dev;
  t.frames = d->frames;
  t.channels = st->channels;
  t.rate = (uint32_t)st->samplerate;
  if (lgd_plan(c, &t, 1, want_tp ? LGD_FLAG_TRUE_PEAK : 0) || lgd_execute(c, nullptr) ||
      lgd_fetch(c, &d->res, nullptr))
    return EBUR128_ERROR_NOMEM;
  d->sample_peak.assign(st->channels, 0.0);
  d->true_peak.assign(st->channels, 0.0);
  if (lgd_copy_channel_peaks(c, 0, d->sample_peak.data(), d->true_peak.data(), st->channels))
    return EBUR128_ERROR_NOMEM;
  d->have = true;
  d->have_tp = want_tp;
  d->scanned_frames = d->frames;
  return EBUR128_SUCCESS;
}

int scan_multi(ebur128_state **sts, size_t size) {
  MultiKey key;
  for (size_t i = 0; i < size; ++i) key.v.emplace_back(sts[i], sts[i]->d->frames);
  if (g_multi_valid && key == g_multi_key) return EBUR128_SUCCESS;
  g_multi_valid = false;
  std::vector<lgd_track> t(size);
  int device = size ? sts[0]->d->device : g_device;
  for (size_t i = 0; i < size; ++i) {
    if (sts[i]->d->device != device) return EBUR128_ERROR_INVALID_MODE;  // one GPU per album here
    int rc = upload(sts[i]);
    if (rc) return rc;
    t[i].pcm = sts[i]->d->dev;
    t[i].frames = sts[i]->d->frames;
    t[i].channels = sts[i]->channels;
    t[i].rate = (uint32_t)sts[i]->samplerate;
  }
  lgd_ctx *c = ctx_for(device);
  if (!c) return EBUR128_ERROR_NOMEM;
  std::vector<lgd_track_result> r(size ? size : 1);
  if (lgd_plan(c, t.data(), (uint32_t)size, LGD_FLAG_ALBUM) || lgd_execute(c, nullptr) ||
      lgd_fetch(c, r.data(), &g_multi))
    return EBUR128_ERROR_NOMEM;
  g_multi_key = key;
  g_multi_valid = true;
  return EBUR128_SUCCESS;
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
  return st;
}

extern "C" void ebur128_destroy(ebur128_state **st) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (!st || !*st) return;
  if ((*st)->d) {
    if ((*st)->d->dev) (void)hipFree((*st)->d->dev);
    delete (*st)->d;
  }
  g_multi_valid = false;  // its address may be reused
  delete *st;
  *st = nullptr;
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
  *out = g_multi.loudness;
  return EBUR128_SUCCESS;
}

extern "C" int ebur128_loudness_range_multiple(ebur128_state **sts, size_t size, double *out) {
  std::lock_guard<std::mutex> lk(g_mu);
  for (size_t i = 0; i < size; ++i)
    if ((sts[i]->mode & EBUR128_MODE_LRA) != EBUR128_MODE_LRA) return EBUR128_ERROR_INVALID_MODE;
  int rc = scan_multi(sts, size);
  if (rc) return rc;
  *out = g_multi.lra;
  return EBUR128_SUCCESS;
}
