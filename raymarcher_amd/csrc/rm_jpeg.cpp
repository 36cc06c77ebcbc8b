// rm_jpeg.cpp — baseline (SOF0/SOF1, Huffman, 8-bit) JPEG → RGBA8, for the reference's JPEG assets (the "Beach" sky
// box, scenefiles/texture_store/cube_map/beach/*.jpg, loaded with QImage in Realtime::initCubeMap,
// src/realtimerender.cpp:557-589).  QImage decodes through libjpeg with its defaults — the accurate integer IDCT
// ("islow"), triangle-filter ("fancy") chroma upsampling, 16-bit fixed-point YCbCr → RGB — and every stage here
// follows those definitions so that the pixels equal that decoder's, value for value (tests/test_abi.py compares
// with the libjpeg-turbo build inside Pillow).  Progressive, arithmetic-coded, 12-bit and CMYK files are refused.
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/raymarcher_amd.h"
#include "rm_internal.h"

namespace rm {
namespace {

struct Huff {
  // canonical decode: codes of length l are in [mincode[l], maxcode[l]], values start at valptr[l]
  int mincode[17] = {}, maxcode[18] = {}, valptr[17] = {};
  uint8_t vals[256] = {};
  bool present = false;
};
struct Comp { int id, h, v, tq, td, ta, wBlocks, hBlocks, dw, dh, pred; std::vector<uint8_t> plane; int stride; };

const uint8_t kZigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                             41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                             30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct BitReader {
  const uint8_t *p, *end;
  uint32_t acc = 0;
  int bits = 0;
  bool hitMarker = false;
  void fill() {
    while (bits <= 24) {
      int b = 0;
      if (!hitMarker && p < end) {
        b = *p;
        if (b == 0xFF) {
          if (p + 1 < end && p[1] == 0x00) p += 2;           // stuffed byte
          else { hitMarker = true; b = 0; }                  // a marker: feed zeros, leave p on it
        } else p++;
      }
      acc |= (uint32_t)b << (24 - bits);
      bits += 8;
    }
  }
  int get(int n) {
    if (n == 0) return 0;
    fill();
    int v = (int)(acc >> (32 - n));
    acc <<= n;
    bits -= n;
    return v;
  }
  int peek16() { fill(); return (int)(acc >> 16); }
  void reset() { acc = 0; bits = 0; hitMarker = false; }
};

bool buildHuff(Huff &h, const uint8_t counts[16], const uint8_t *vals, int nvals) {
  int code = 0, k = 0;
  for (int l = 1; l <= 16; l++) {
    h.valptr[l] = k;
    h.mincode[l] = code;
    code += counts[l - 1];
    k += counts[l - 1];
    h.maxcode[l] = counts[l - 1] ? code - 1 : -1;
    code <<= 1;
  }
  h.maxcode[17] = 0x7fffffff;
  if (k != nvals || k > 256) return false;
  std::memcpy(h.vals, vals, (size_t)k);
  h.present = true;
  return true;
}
int decodeSym(BitReader &br, const Huff &h) {
  int look = br.peek16(), code = 0;
  for (int l = 1; l <= 16; l++) {
    code = look >> (16 - l);
    if (h.maxcode[l] >= 0 && code <= h.maxcode[l] && code >= h.mincode[l]) {
      br.get(l);
      return h.vals[h.valptr[l] + code - h.mincode[l]];
    }
  }
  return -1;
}
inline int extend(int v, int n) { return (v < (1 << (n - 1))) ? v - (1 << n) + 1 : v; }

// post-IDCT range limiting of libjpeg (prepare_range_limit_table, indexed through & RANGE_MASK): a clamp of x + 128 to
// [0, 255] for |x| < 384, with libjpeg's wrap-around behaviour beyond (never reached by well-formed data).
inline uint8_t rangeLimit(int x) {
  x &= 1023;
  if (x < 128) return (uint8_t)(x + 128);
  if (x < 512) return 255;
  if (x < 896) return 0;
  return (uint8_t)(x - 896);
}

// jidctint.c (jpeg_idct_islow): CONST_BITS = 13, PASS1_BITS = 2.
void idctIslow(const int16_t coef[64], const uint16_t q[64], uint8_t *out, int stride) {
  constexpr int CB = 13, P1 = 2;
  constexpr int64_t F0298 = 2446, F0390 = 3196, F0541 = 4433, F0765 = 6270, F0899 = 7373, F1175 = 9633, F1501 = 12299,
                    F1847 = 15137, F1961 = 16069, F2053 = 16819, F2562 = 20995, F3072 = 25172;
  auto descale = [](int64_t x, int n) { return (int)((x + ((int64_t)1 << (n - 1))) >> n); };
  int ws[64];
  for (int c = 0; c < 8; c++) {
    auto in = [&](int r) { return (int64_t)coef[r * 8 + c] * q[r * 8 + c]; };
    if (!coef[8 + c] && !coef[16 + c] && !coef[24 + c] && !coef[32 + c] && !coef[40 + c] && !coef[48 + c] && !coef[56 + c]) {
      int dc = (int)(in(0) * (1 << P1));
      for (int r = 0; r < 8; r++) ws[r * 8 + c] = dc;
      continue;
    }
    int64_t z2 = in(2), z3 = in(6);
    int64_t z1 = (z2 + z3) * F0541;
    int64_t tmp2 = z1 + z3 * (-F1847), tmp3 = z1 + z2 * F0765;
    z2 = in(0); z3 = in(4);
    int64_t tmp0 = (z2 + z3) * (1 << CB), tmp1 = (z2 - z3) * (1 << CB);
    int64_t tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
    tmp0 = in(7); tmp1 = in(5); tmp2 = in(3); tmp3 = in(1);
    z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
    int64_t z4 = tmp1 + tmp3, z5 = (z3 + z4) * F1175;
    tmp0 *= F0298; tmp1 *= F2053; tmp2 *= F3072; tmp3 *= F1501;
    z1 *= -F0899; z2 *= -F2562; z3 *= -F1961; z4 *= -F0390;
    z3 += z5; z4 += z5;
    tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
    ws[0 * 8 + c] = descale(tmp10 + tmp3, CB - P1); ws[7 * 8 + c] = descale(tmp10 - tmp3, CB - P1);
    ws[1 * 8 + c] = descale(tmp11 + tmp2, CB - P1); ws[6 * 8 + c] = descale(tmp11 - tmp2, CB - P1);
    ws[2 * 8 + c] = descale(tmp12 + tmp1, CB - P1); ws[5 * 8 + c] = descale(tmp12 - tmp1, CB - P1);
    ws[3 * 8 + c] = descale(tmp13 + tmp0, CB - P1); ws[4 * 8 + c] = descale(tmp13 - tmp0, CB - P1);
  }
  for (int r = 0; r < 8; r++) {
    const int *w = ws + r * 8;
    uint8_t *o = out + (size_t)r * stride;
    int64_t z2 = w[2], z3 = w[6];
    int64_t z1 = (z2 + z3) * F0541;
    int64_t tmp2 = z1 + z3 * (-F1847), tmp3 = z1 + z2 * F0765;
    int64_t tmp0 = ((int64_t)w[0] + w[4]) * (1 << CB), tmp1 = ((int64_t)w[0] - w[4]) * (1 << CB);
    int64_t tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
    tmp0 = w[7]; tmp1 = w[5]; tmp2 = w[3]; tmp3 = w[1];
    z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
    int64_t z4 = tmp1 + tmp3, z5 = (z3 + z4) * F1175;
    tmp0 *= F0298; tmp1 *= F2053; tmp2 *= F3072; tmp3 *= F1501;
    z1 *= -F0899; z2 *= -F2562; z3 *= -F1961; z4 *= -F0390;
    z3 += z5; z4 += z5;
    tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
    constexpr int S = CB + P1 + 3;
    o[0] = rangeLimit(descale(tmp10 + tmp3, S)); o[7] = rangeLimit(descale(tmp10 - tmp3, S));
    o[1] = rangeLimit(descale(tmp11 + tmp2, S)); o[6] = rangeLimit(descale(tmp11 - tmp2, S));
    o[2] = rangeLimit(descale(tmp12 + tmp1, S)); o[5] = rangeLimit(descale(tmp12 - tmp1, S));
    o[3] = rangeLimit(descale(tmp13 + tmp0, S)); o[4] = rangeLimit(descale(tmp13 - tmp0, S));
  }
}

// jdsample.c: chroma planes to full resolution.  `src` has dw × dh real samples (stride `stride`).
// h2v1 / h2v2 use the "fancy" triangle filters (h2v1_fancy_upsample, h2v2_fancy_upsample); rows beyond the
// component's real height replicate its last row, as jdmainct.c's context-row handling does.
std::vector<uint8_t> upsample(const Comp &c, int hs, int vs, int W, int H) {
  std::vector<uint8_t> out((size_t)W * H);
  const int dw = c.dw, dh = c.dh;
  auto row = [&](int y) { return c.plane.data() + (size_t)(y < 0 ? 0 : (y >= dh ? dh - 1 : y)) * c.stride; };
  if (hs == 1 && vs == 1) {
    for (int y = 0; y < H; y++) std::memcpy(&out[(size_t)y * W], row(y), (size_t)W);
    return out;
  }
  std::vector<uint8_t> line((size_t)dw * 2 + 2);
  // jinit_upsampler picks the fancy filters only when downsampled_width > 2; narrower planes are box-replicated
  if (hs == 2 && vs == 1 && dw > 2) {
    for (int y = 0; y < H; y++) {
      const uint8_t *in = row(y);
      {
        line[0] = in[0];
        line[1] = (uint8_t)((in[0] * 3 + in[1] + 2) >> 2);
        for (int x = 1; x < dw - 1; x++) {
          int v = in[x] * 3;
          line[2 * x] = (uint8_t)((v + in[x - 1] + 1) >> 2);
          line[2 * x + 1] = (uint8_t)((v + in[x + 1] + 2) >> 2);
        }
        line[2 * (dw - 1)] = (uint8_t)((in[dw - 1] * 3 + in[dw - 2] + 1) >> 2);
        line[2 * (dw - 1) + 1] = in[dw - 1];
      }
      std::memcpy(&out[(size_t)y * W], line.data(), (size_t)W);
    }
    return out;
  }
  if (hs == 2 && vs == 2 && dw > 2) {
    for (int y = 0; y < H; y++) {
      const int sy = y >> 1;
      const uint8_t *in0 = row(sy), *in1 = row((y & 1) ? sy + 1 : sy - 1);  // nearer neighbour row
      {
        int thisc = in0[0] * 3 + in1[0], nextc = in0[1] * 3 + in1[1], lastc;
        line[0] = (uint8_t)((thisc * 4 + 8) >> 4);
        line[1] = (uint8_t)((thisc * 3 + nextc + 7) >> 4);
        lastc = thisc; thisc = nextc;
        for (int x = 1; x < dw - 1; x++) {
          nextc = in0[x + 1] * 3 + in1[x + 1];
          line[2 * x] = (uint8_t)((thisc * 3 + lastc + 8) >> 4);
          line[2 * x + 1] = (uint8_t)((thisc * 3 + nextc + 7) >> 4);
          lastc = thisc; thisc = nextc;
        }
        line[2 * (dw - 1)] = (uint8_t)((thisc * 3 + lastc + 8) >> 4);
        line[2 * (dw - 1) + 1] = (uint8_t)((thisc * 4 + 7) >> 4);
      }
      std::memcpy(&out[(size_t)y * W], line.data(), (size_t)W);
    }
    return out;
  }
  // everything else — h1v2, other integral ratios, planes at most 2 samples wide: box replication (h2v1_upsample,
  // h2v2_upsample, int_upsample)
  for (int y = 0; y < H; y++) {
    const uint8_t *in = row(y / vs);
    for (int x = 0; x < W; x++) out[(size_t)y * W + x] = in[x / hs < dw ? x / hs : dw - 1];
  }
  return out;
}

inline uint8_t clamp8(int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

}  // namespace

// Decodes a whole file image.  Returns RM_OK and fills rgba (top row first), or an error status with rm_last_error set.
int jpeg_decode(const std::vector<uint8_t> &file, std::vector<uint8_t> &rgba, int &W, int &H) {
  auto fail = [](int st, const char *msg) { set_error(std::string("JPEG: ") + msg); return st; };
  if (file.size() < 4 || file[0] != 0xFF || file[1] != 0xD8) return fail(RM_ERR_PARSE, "missing SOI");
  uint16_t qt[4][64] = {};
  bool qtPresent[4] = {false, false, false, false};
  Huff dc[4], ac[4];
  std::vector<Comp> comps;
  int restart = 0, adobeTransform = -1;
  bool haveSOF = false;
  W = H = 0;
  size_t pos = 2;
  while (pos + 4 <= file.size()) {
    if (file[pos] != 0xFF) return fail(RM_ERR_PARSE, "marker expected");
    while (pos < file.size() && file[pos] == 0xFF) pos++;  // fill bytes
    if (pos >= file.size()) break;
    const int m = file[pos++];
    if (m == 0xD9) break;
    if (m == 0x01 || (m >= 0xD0 && m <= 0xD7)) continue;
    if (pos + 2 > file.size()) return fail(RM_ERR_PARSE, "truncated segment");
    const size_t len = ((size_t)file[pos] << 8) | file[pos + 1];
    if (len < 2 || pos + len > file.size()) return fail(RM_ERR_PARSE, "truncated segment");
    const uint8_t *d = &file[pos + 2];
    const size_t n = len - 2;
    if (m == 0xDB) {  // DQT
      size_t i = 0;
      while (i < n) {
        const int pq = d[i] >> 4, tq = d[i] & 15;
        i++;
        if (tq > 3 || i + (pq ? 128 : 64) > n) return fail(RM_ERR_PARSE, "bad DQT");
        for (int k = 0; k < 64; k++) {
          qt[tq][kZigzag[k]] = pq ? (uint16_t)((d[i] << 8) | d[i + 1]) : d[i];
          i += pq ? 2 : 1;
        }
        qtPresent[tq] = true;
      }
    } else if (m == 0xC4) {  // DHT
      size_t i = 0;
      while (i + 17 <= n) {
        const int tc = d[i] >> 4, th = d[i] & 15;
        int total = 0;
        for (int k = 0; k < 16; k++) total += d[i + 1 + k];
        if (tc > 1 || th > 3 || i + 17 + (size_t)total > n) return fail(RM_ERR_PARSE, "bad DHT");
        if (!buildHuff(tc ? ac[th] : dc[th], &d[i + 1], &d[i + 17], total)) return fail(RM_ERR_PARSE, "bad DHT");
        i += 17 + (size_t)total;
      }
    } else if (m == 0xC0 || m == 0xC1) {  // SOF0 / SOF1
      if (n < 6 || d[0] != 8) return fail(RM_ERR_UNSUPPORTED, "only 8-bit samples");
      H = (d[1] << 8) | d[2];
      W = (d[3] << 8) | d[4];
      const int nc = d[5];
      if (W <= 0 || H <= 0 || W > 32768 || H > 32768) return fail(RM_ERR_PARSE, "bad size");
      if ((nc != 1 && nc != 3) || n < 6 + 3 * (size_t)nc) return fail(RM_ERR_UNSUPPORTED, "only grayscale and 3-component files");
      comps.resize(nc);
      for (int k = 0; k < nc; k++) {
        comps[k].id = d[6 + 3 * k];
        comps[k].h = d[7 + 3 * k] >> 4;
        comps[k].v = d[7 + 3 * k] & 15;
        comps[k].tq = d[8 + 3 * k];
        if (comps[k].h < 1 || comps[k].h > 4 || comps[k].v < 1 || comps[k].v > 4 || comps[k].tq > 3) return fail(RM_ERR_PARSE, "bad SOF");
        comps[k].td = comps[k].ta = -1;  // assigned by the scan header
        for (int j = 0; j < k; j++)
          if (comps[j].id == comps[k].id) return fail(RM_ERR_PARSE, "SOF lists a component id twice");
      }
      haveSOF = true;
    } else if (m == 0xC2 || (m >= 0xC3 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC)) {
      return fail(RM_ERR_UNSUPPORTED, "progressive / lossless / arithmetic-coded files are not supported (baseline only)");
    } else if (m == 0xDD) {
      if (n >= 2) restart = (d[0] << 8) | d[1];
    } else if (m == 0xEE) {
      if (n >= 12 && !std::memcmp(d, "Adobe", 5)) adobeTransform = d[11];
    } else if (m == 0xDA) {  // SOS: the single scan of a baseline file
      if (!haveSOF) return fail(RM_ERR_PARSE, "SOS before SOF");
      const int ns = d[0];
      if (ns != (int)comps.size() || n < 1 + 2 * (size_t)ns + 3) return fail(RM_ERR_UNSUPPORTED, "non-interleaved scans are not supported");
      for (int k = 0; k < ns; k++) {
        Comp *c = nullptr;
        for (auto &cc : comps) if (cc.id == d[1 + 2 * k]) c = &cc;
        if (!c) return fail(RM_ERR_PARSE, "bad SOS");
        c->td = d[2 + 2 * k] >> 4;
        c->ta = d[2 + 2 * k] & 15;
        if (c->td > 3 || c->ta > 3 || !dc[c->td].present || !ac[c->ta].present || !qtPresent[c->tq]) return fail(RM_ERR_PARSE, "missing table");
      }
      for (auto &cc : comps)  // every component of the frame got its tables from this scan (a scan naming one id twice leaves another without)
        if (cc.td < 0 || cc.ta < 0) return fail(RM_ERR_PARSE, "a component is missing from the scan");
      int hmax = 1, vmax = 1;
      for (auto &c : comps) { hmax = c.h > hmax ? c.h : hmax; vmax = c.v > vmax ? c.v : vmax; }
      if (comps.size() == 1) { comps[0].h = comps[0].v = 1; hmax = vmax = 1; }  // a single-component scan is never interleaved
      const int mcuW = 8 * hmax, mcuH = 8 * vmax, mcusX = (W + mcuW - 1) / mcuW, mcusY = (H + mcuH - 1) / mcuH;
      for (auto &c : comps) {
        if (hmax % c.h || vmax % c.v) return fail(RM_ERR_UNSUPPORTED, "fractional sampling ratios are not supported");
        c.wBlocks = mcusX * c.h; c.hBlocks = mcusY * c.v;
        c.stride = c.wBlocks * 8;
        c.plane.assign((size_t)c.stride * c.hBlocks * 8, 0);
        c.dw = (W * c.h + hmax - 1) / hmax; c.dh = (H * c.v + vmax - 1) / vmax;
        c.pred = 0;
      }
      BitReader br;
      br.p = d + n; br.end = file.data() + file.size();
      int16_t block[64];
      int untilRestart = restart, nextRst = 0;
      for (int my = 0; my < mcusY; my++)
        for (int mx = 0; mx < mcusX; mx++) {
          if (restart && untilRestart == 0) {
            // byte-align, expect RSTn
            br.reset();
            const uint8_t *q = br.p;
            while (q + 1 < br.end && !(q[0] == 0xFF && q[1] >= 0xD0 && q[1] <= 0xD7)) q++;
            if (q + 1 >= br.end || q[1] != 0xD0 + nextRst) return fail(RM_ERR_PARSE, "restart marker missing");
            br.p = q + 2;
            nextRst = (nextRst + 1) & 7;
            untilRestart = restart;
            for (auto &c : comps) c.pred = 0;
          }
          for (auto &c : comps)
            for (int by = 0; by < c.v; by++)
              for (int bx = 0; bx < c.h; bx++) {
                std::memset(block, 0, sizeof block);
                int s = decodeSym(br, dc[c.td]);
                if (s < 0 || s > 11) return fail(RM_ERR_PARSE, "bad DC code");
                if (s) c.pred += extend(br.get(s), s);
                block[0] = (int16_t)c.pred;
                for (int k = 1; k < 64;) {
                  int rs = decodeSym(br, ac[c.ta]);
                  if (rs < 0) return fail(RM_ERR_PARSE, "bad AC code");
                  const int r = rs >> 4, sz = rs & 15;
                  if (sz == 0) {
                    if (r != 15) break;
                    k += 16;
                    continue;
                  }
                  k += r;
                  if (k > 63) return fail(RM_ERR_PARSE, "AC run past the block");
                  block[kZigzag[k]] = (int16_t)extend(br.get(sz), sz);
                  k++;
                }
                const int X = (mx * c.h + bx) * 8, Y = (my * c.v + by) * 8;
                idctIslow(block, qt[c.tq], &c.plane[(size_t)Y * c.stride + X], c.stride);
              }
          if (restart) untilRestart--;
        }
      // colour
      rgba.assign((size_t)W * H * 4, 255);
      if (comps.size() == 1) {
        for (int y = 0; y < H; y++)
          for (int x = 0; x < W; x++) {
            uint8_t g = comps[0].plane[(size_t)y * comps[0].stride + x];
            uint8_t *o = &rgba[((size_t)y * W + x) * 4];
            o[0] = o[1] = o[2] = g;
          }
        return RM_OK;
      }
      std::vector<uint8_t> full[3];
      for (int k = 0; k < 3; k++) full[k] = upsample(comps[k], hmax / comps[k].h, vmax / comps[k].v, W, H);
      const bool ycc = adobeTransform != 0;  // JFIF / Adobe transform 1: YCbCr; Adobe transform 0: RGB
      // jdcolor.c build_ycc_rgb_table: SCALEBITS = 16, ONE_HALF = 1 << 15, FIX(x) = (int)(x·65536 + 0.5)
      constexpr int64_t FIX_1_40200 = 91881, FIX_1_77200 = 116130, FIX_0_71414 = 46802, FIX_0_34414 = 22554, HALF = 32768;
      for (size_t i = 0; i < (size_t)W * H; i++) {
        uint8_t *o = &rgba[i * 4];
        if (!ycc) { o[0] = full[0][i]; o[1] = full[1][i]; o[2] = full[2][i]; continue; }
        const int y = full[0][i], cb = full[1][i] - 128, cr = full[2][i] - 128;
        const int crR = (int)((FIX_1_40200 * cr + HALF) >> 16), cbB = (int)((FIX_1_77200 * cb + HALF) >> 16);
        const int g = (int)(((-FIX_0_34414) * cb + HALF + (-FIX_0_71414) * cr) >> 16);
        o[0] = clamp8(y + crR); o[1] = clamp8(y + g); o[2] = clamp8(y + cbB);
      }
      return RM_OK;
    }
    pos += len;
  }
  return fail(RM_ERR_PARSE, "no scan found");
}

}  // namespace rm
