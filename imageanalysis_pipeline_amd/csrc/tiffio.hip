// Batched plane I/O for the TIFF sequences on either side of the hot path (host code only).
//
// The reference keeps every (channel, frame, z) plane in its own baseline TIFF written by
// tifffile.imwrite(plane.astype(np.uint16)) (utility.py:155-166, 264, 355, 616) and reads the
// stack back file by file in Python (utility.py:487-524 load_stack, :558-595 load_mask).  Here a
// pool of threads parses the classic-TIFF directory of each file and preads the strips straight
// into the caller's (ideally pinned) stack buffer, so that one H2D copy follows.
//
// Supported: classic TIFF (magic 42), II or MM byte order, one sample per pixel, 16-bit unsigned,
// uncompressed, stripped (any RowsPerStrip).  Anything else is reported by name with
// IAP_E_UNSUPPORTED; the Python shim then falls back to its generic reader for that folder.
#include <errno.h>
#include <fcntl.h>
#include <string.h>
#include <sys/stat.h>
#include <unistd.h>
#include <atomic>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#include "../../include/iap.h"
#include "common.h"

namespace iap {
namespace {

struct TiffDir {
  int Y = 0, X = 0, bits = 1, sample_format = 1, compression = 1, samples = 1;
  bool big_endian = false, tiled = false;
  std::vector<uint64_t> offs, cnts;
};

struct Err {
  int code = IAP_OK;
  std::string msg;
};

static bool pread_all(int fd, void* buf, size_t n, uint64_t off) {
  char* p = (char*)buf;
  while (n) {
    ssize_t r = pread(fd, p, n, (off_t)off);
    if (r <= 0) {
      if (r < 0 && errno == EINTR) continue;
      return false;
    }
    p += r; n -= (size_t)r; off += (uint64_t)r;
  }
  return true;
}

static bool pwrite_all(int fd, const void* buf, size_t n, uint64_t off) {
  const char* p = (const char*)buf;
  while (n) {
    ssize_t r = pwrite(fd, p, n, (off_t)off);
    if (r <= 0) {
      if (r < 0 && errno == EINTR) continue;
      return false;
    }
    p += r; n -= (size_t)r; off += (uint64_t)r;
  }
  return true;
}

static inline uint16_t rd16(const unsigned char* p, bool be) {
  return be ? (uint16_t)((p[0] << 8) | p[1]) : (uint16_t)((p[1] << 8) | p[0]);
}
static inline uint32_t rd32(const unsigned char* p, bool be) {
  return be ? ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]
            : ((uint32_t)p[3] << 24) | ((uint32_t)p[2] << 16) | ((uint32_t)p[1] << 8) | p[0];
}

// first image file directory of a classic TIFF
static Err parse_dir(int fd, const char* path, TiffDir& d) {
  Err e;
  auto fail = [&](int code, const std::string& what) {
    e.code = code;
    e.msg = std::string(path) + ": " + what;
    return e;
  };
  unsigned char h[8];
  if (!pread_all(fd, h, 8, 0)) return fail(IAP_E_ARG, "shorter than a TIFF header");
  if (h[0] == 'I' && h[1] == 'I') d.big_endian = false;
  else if (h[0] == 'M' && h[1] == 'M') d.big_endian = true;
  else return fail(IAP_E_ARG, "not a TIFF file");
  const bool be = d.big_endian;
  const uint16_t magic = rd16(h + 2, be);
  if (magic == 43) return fail(IAP_E_UNSUPPORTED, "BigTIFF is not supported by the batched reader");
  if (magic != 42) return fail(IAP_E_ARG, "not a TIFF file");
  const uint64_t ifd = rd32(h + 4, be);
  unsigned char nb[2];
  if (!pread_all(fd, nb, 2, ifd)) return fail(IAP_E_ARG, "truncated image file directory");
  const int n = rd16(nb, be);
  std::vector<unsigned char> ent((size_t)n * 12);
  if (n && !pread_all(fd, ent.data(), ent.size(), ifd + 2)) return fail(IAP_E_ARG, "truncated image file directory");
  auto values = [&](const unsigned char* en, std::vector<uint64_t>& out) -> bool {
    const int typ = rd16(en + 2, be);
    const uint32_t cnt = rd32(en + 4, be);
    const int size = typ == 3 ? 2 : (typ == 4 ? 4 : (typ == 1 ? 1 : 0));
    if (!size || cnt == 0 || cnt > (1u << 24)) return false;   // (a strip per row of a 16M-row plane: nothing real comes near)
    std::vector<unsigned char> buf((size_t)size * cnt);
    if ((size_t)size * cnt <= 4) memcpy(buf.data(), en + 8, buf.size());
    else if (!pread_all(fd, buf.data(), buf.size(), rd32(en + 8, be))) return false;
    out.resize(cnt);
    for (uint32_t i = 0; i < cnt; ++i)
      out[i] = size == 2 ? rd16(&buf[2 * (size_t)i], be) : (size == 4 ? rd32(&buf[4 * (size_t)i], be) : buf[i]);
    return true;
  };
  for (int i = 0; i < n; ++i) {
    const unsigned char* en = &ent[(size_t)i * 12];
    const int tag = rd16(en, be);
    std::vector<uint64_t> v;
    switch (tag) {
      case 256: case 257: case 258: case 259: case 277: case 339: case 273: case 279:
        if (!values(en, v) || v.empty()) return fail(IAP_E_ARG, "unreadable directory entry " + std::to_string(tag));
        break;
      case 322: case 323: case 324: case 325: d.tiled = true; continue;
      default: continue;
    }
    switch (tag) {
      case 256: d.X = (int)v[0]; break;
      case 257: d.Y = (int)v[0]; break;
      case 258: d.bits = (int)v[0]; break;
      case 259: d.compression = (int)v[0]; break;
      case 277: d.samples = (int)v[0]; break;
      case 339: d.sample_format = (int)v[0]; break;
      case 273: d.offs = v; break;
      case 279: d.cnts = v; break;
    }
  }
  if (d.X <= 0 || d.Y <= 0) return fail(IAP_E_ARG, "no image size in the directory");
  return e;
}

static Err read_plane(const char* path, uint16_t* dst, int Y, int X) {
  Err e;
  int fd = open(path, O_RDONLY | O_CLOEXEC);
  if (fd < 0) {
    e.code = IAP_E_ARG;
    e.msg = std::string(path) + ": " + strerror(errno);
    return e;
  }
  TiffDir d;
  e = parse_dir(fd, path, d);
  auto fail = [&](int code, const std::string& what) {
    e.code = code;
    e.msg = std::string(path) + ": " + what;
  };
  if (e.code == IAP_OK) {
    const size_t bytes = (size_t)Y * X * 2;
    if (d.tiled) fail(IAP_E_UNSUPPORTED, "tiled TIFF is not supported by the batched reader");
    else if (d.compression != 1) fail(IAP_E_UNSUPPORTED, "compressed TIFF (scheme " + std::to_string(d.compression) + ")");
    else if (d.samples != 1 || d.bits != 16 || d.sample_format != 1)
      fail(IAP_E_UNSUPPORTED, "not a single-sample 16-bit unsigned plane (bits " + std::to_string(d.bits) + ", samples " +
                                  std::to_string(d.samples) + ", format " + std::to_string(d.sample_format) + ")");
    else if (d.Y != Y || d.X != X)
      fail(IAP_E_ARG, "plane is " + std::to_string(d.Y) + "x" + std::to_string(d.X) + ", expected " + std::to_string(Y) +
                          "x" + std::to_string(X));
    else if (d.offs.empty() || d.offs.size() != d.cnts.size()) fail(IAP_E_ARG, "strip offsets and byte counts disagree");
    else {
      size_t done = 0;
      for (size_t s = 0; s < d.offs.size() && e.code == IAP_OK; ++s) {
        size_t c = (size_t)d.cnts[s];
        if (done + c > bytes) c = bytes - done;   // a padded last strip
        if (c && !pread_all(fd, (char*)dst + done, c, d.offs[s])) fail(IAP_E_ARG, "truncated strip data");
        done += c;
      }
      if (e.code == IAP_OK && done != bytes) fail(IAP_E_ARG, "strips hold fewer bytes than the plane");
      if (e.code == IAP_OK && d.big_endian) {
        const size_t npx = (size_t)Y * X;
        for (size_t i = 0; i < npx; ++i) dst[i] = (uint16_t)((dst[i] << 8) | (dst[i] >> 8));
      }
    }
  }
  close(fd);
  return e;
}

static void put16(std::vector<unsigned char>& o, uint16_t v) { o.push_back(v & 0xff); o.push_back(v >> 8); }
static void put32(std::vector<unsigned char>& o, uint32_t v) {
  for (int i = 0; i < 4; ++i) o.push_back((v >> (8 * i)) & 0xff);
}

// little-endian classic TIFF, one strip, the ten tags tifffile writes for a grayscale plane
static Err write_plane(const char* path, const uint16_t* src, int Y, int X) {
  Err e;
  const size_t bytes = (size_t)Y * X * 2;
  if (bytes + 256 >= (1ull << 32)) {
    e.code = IAP_E_UNSUPPORTED;
    e.msg = std::string(path) + ": plane too large for a classic TIFF";
    return e;
  }
  int fd = open(path, O_WRONLY | O_CREAT | O_TRUNC | O_CLOEXEC, 0644);
  if (fd < 0) {
    e.code = IAP_E_ARG;
    e.msg = std::string(path) + ": " + strerror(errno);
    return e;
  }
  const uint32_t ifd_off = (uint32_t)(8 + bytes);
  std::vector<unsigned char> head, dir;
  head.push_back('I'); head.push_back('I');
  put16(head, 42);
  put32(head, ifd_off);
  const uint32_t tags[10][3] = {{256, 4, (uint32_t)X}, {257, 4, (uint32_t)Y}, {258, 3, 16}, {259, 3, 1}, {262, 3, 1},
                                {273, 4, 8}, {277, 3, 1}, {278, 4, (uint32_t)Y}, {279, 4, (uint32_t)bytes}, {339, 3, 1}};
  put16(dir, 10);
  for (auto& t : tags) {
    put16(dir, (uint16_t)t[0]);
    put16(dir, (uint16_t)t[1]);
    put32(dir, 1);
    if (t[1] == 3) { put16(dir, (uint16_t)t[2]); put16(dir, 0); }
    else put32(dir, t[2]);
  }
  put32(dir, 0);
  if (!pwrite_all(fd, head.data(), head.size(), 0) || !pwrite_all(fd, src, bytes, 8) ||
      !pwrite_all(fd, dir.data(), dir.size(), ifd_off)) {
    e.code = IAP_E_ARG;
    e.msg = std::string(path) + ": write failed: " + strerror(errno);
  }
  if (close(fd) != 0 && e.code == IAP_OK) {
    e.code = IAP_E_ARG;
    e.msg = std::string(path) + ": close failed: " + strerror(errno);
  }
  return e;
}

template <typename Fn>
static int run_pool(int64_t n, int threads, Fn fn) {
  if (threads < 1) threads = 1;
  if (threads > 64) threads = 64;
  if ((int64_t)threads > n) threads = (int)n;
  std::atomic<int64_t> next(0);
  std::atomic<bool> stop(false);
  std::mutex mu;
  Err first;
  auto work = [&]() {
    for (;;) {
      const int64_t i = next.fetch_add(1);
      if (i >= n || stop.load()) return;
      Err e;
      try {
        e = fn(i);
      } catch (const std::exception& ex) {   // e.g. bad_alloc on a directory that claims a huge table: an error, not a terminate()
        e.code = IAP_E_ARG;
        e.msg = "plane " + std::to_string(i) + ": " + ex.what();
      }
      if (e.code != IAP_OK) {
        std::lock_guard<std::mutex> lk(mu);
        if (first.code == IAP_OK) first = e;
        stop.store(true);
        return;
      }
    }
  };
  std::vector<std::thread> pool;
  for (int t = 1; t < threads; ++t) pool.emplace_back(work);
  work();
  for (auto& t : pool) t.join();
  if (first.code != IAP_OK) set_error("%s", first.msg.c_str());
  return first.code;
}

}  // namespace
}  // namespace iap

using namespace iap;

extern "C" int iap_tiff_probe(const char* path, int* Y, int* X, int* bits, int* sample_format, int* compression,
                              int* samples) {
  IAP_REQUIRE(path, IAP_E_ARG, "iap_tiff_probe: null path");
  int fd = open(path, O_RDONLY | O_CLOEXEC);
  IAP_REQUIRE(fd >= 0, IAP_E_ARG, "%s: %s", path, strerror(errno));
  TiffDir d;
  Err e;
  try {
    e = parse_dir(fd, path, d);
  } catch (const std::exception& ex) {
    e.code = IAP_E_ARG;
    e.msg = std::string(path) + ": " + ex.what();
  }
  close(fd);
  if (e.code != IAP_OK) {
    set_error("%s", e.msg.c_str());
    return e.code;
  }
  if (Y) *Y = d.Y;
  if (X) *X = d.X;
  if (bits) *bits = d.bits;
  if (sample_format) *sample_format = d.sample_format;
  if (compression) *compression = d.tiled ? -1 : d.compression;
  if (samples) *samples = d.samples;
  return IAP_OK;
}

extern "C" int iap_tiff_read_u16(const char* const* paths, int64_t n, uint16_t* dst, int Y, int X, int threads) {
  IAP_REQUIRE(paths && dst, IAP_E_ARG, "iap_tiff_read_u16: null pointer");
  IAP_REQUIRE(n > 0 && Y > 0 && X > 0, IAP_E_ARG, "iap_tiff_read_u16: non-positive size");
  for (int64_t i = 0; i < n; ++i) IAP_REQUIRE(paths[i], IAP_E_ARG, "iap_tiff_read_u16: null path at index %lld", (long long)i);
  const size_t px = (size_t)Y * X;
  return run_pool(n, threads, [&](int64_t i) { return read_plane(paths[i], dst + (size_t)i * px, Y, X); });
}

extern "C" int iap_tiff_write_u16(const char* const* paths, int64_t n, const uint16_t* src, int Y, int X, int threads) {
  IAP_REQUIRE(paths && src, IAP_E_ARG, "iap_tiff_write_u16: null pointer");
  IAP_REQUIRE(n > 0 && Y > 0 && X > 0, IAP_E_ARG, "iap_tiff_write_u16: non-positive size");
  for (int64_t i = 0; i < n; ++i) IAP_REQUIRE(paths[i], IAP_E_ARG, "iap_tiff_write_u16: null path at index %lld", (long long)i);
  const size_t px = (size_t)Y * X;
  return run_pool(n, threads, [&](int64_t i) { return write_plane(paths[i], src + (size_t)i * px, Y, X); });
}
