// rm_gif.cpp — first frame of a GIF87a / GIF89a file → RGBA8 (one of the reference's textures,
// scenefiles/texture_store/breakfast.gif, is a single-frame GIF read by QImage).  LZW with variable code size, global or
// local colour table, interlaced rows, the transparency index of a preceding graphic-control extension (alpha 0).
// The frame is placed on a canvas of the logical screen size; pixels it does not cover stay (0,0,0,0).
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/raymarcher_amd.h"
#include "rm_internal.h"

namespace rm {

int gif_decode(const std::vector<uint8_t> &f, std::vector<uint8_t> &rgba, int &W, int &H) {
  auto fail = [](int st, const char *msg) { set_error(std::string("GIF: ") + msg); return st; };
  if (f.size() < 13 || (std::memcmp(f.data(), "GIF87a", 6) && std::memcmp(f.data(), "GIF89a", 6))) return fail(RM_ERR_PARSE, "bad signature");
  W = f[6] | (f[7] << 8);
  H = f[8] | (f[9] << 8);
  if (W <= 0 || H <= 0) return fail(RM_ERR_PARSE, "bad size");
  size_t pos = 13;
  const uint8_t *gct = nullptr;
  int gctSize = 0;
  if (f[10] & 0x80) {
    gctSize = 1 << ((f[10] & 7) + 1);
    if (pos + 3 * (size_t)gctSize > f.size()) return fail(RM_ERR_PARSE, "truncated colour table");
    gct = &f[pos];
    pos += 3 * (size_t)gctSize;
  }
  int transparent = -1;
  while (pos < f.size()) {
    const uint8_t b = f[pos++];
    if (b == 0x3B) break;
    if (b == 0x21) {  // extension
      if (pos >= f.size()) break;
      const uint8_t label = f[pos++];
      if (label == 0xF9 && pos + 5 < f.size() && f[pos] == 4) transparent = (f[pos + 1] & 1) ? f[pos + 4] : -1;
      while (pos < f.size() && f[pos]) pos += (size_t)f[pos] + 1;  // sub-blocks
      pos++;
      continue;
    }
    if (b != 0x2C) return fail(RM_ERR_PARSE, "unexpected block");
    if (pos + 9 > f.size()) return fail(RM_ERR_PARSE, "truncated image descriptor");
    const int ix = f[pos] | (f[pos + 1] << 8), iy = f[pos + 2] | (f[pos + 3] << 8);
    const int iw = f[pos + 4] | (f[pos + 5] << 8), ih = f[pos + 6] | (f[pos + 7] << 8);
    const uint8_t flags = f[pos + 8];
    pos += 9;
    const uint8_t *ct = gct;
    int ctSize = gctSize;
    if (flags & 0x80) {
      ctSize = 1 << ((flags & 7) + 1);
      if (pos + 3 * (size_t)ctSize > f.size()) return fail(RM_ERR_PARSE, "truncated colour table");
      ct = &f[pos];
      pos += 3 * (size_t)ctSize;
    }
    if (!ct) return fail(RM_ERR_PARSE, "no colour table");
    if (pos >= f.size()) return fail(RM_ERR_PARSE, "truncated image data");
    const int minCode = f[pos++];
    if (minCode < 2 || minCode > 8) return fail(RM_ERR_PARSE, "bad LZW code size");
    std::vector<uint8_t> data;
    while (pos < f.size() && f[pos]) {
      const size_t n = f[pos];
      if (pos + 1 + n > f.size()) return fail(RM_ERR_PARSE, "truncated image data");
      data.insert(data.end(), &f[pos + 1], &f[pos + 1 + n]);
      pos += n + 1;
    }
    // LZW
    std::vector<uint8_t> idx;
    idx.reserve((size_t)iw * ih);
    const int clear = 1 << minCode, eoi = clear + 1;
    int codeSize = minCode + 1, next = eoi + 1, prev = -1;
    std::vector<uint16_t> prefix(4096);
    std::vector<uint8_t> suffix(4096), stack(4097);
    for (int i = 0; i < clear; i++) { prefix[i] = 0xFFFF; suffix[i] = (uint8_t)i; }
    uint32_t acc = 0;
    int bits = 0;
    size_t dp = 0;
    uint8_t first = 0;
    while (idx.size() < (size_t)iw * ih) {
      while (bits < codeSize && dp < data.size()) { acc |= (uint32_t)data[dp++] << bits; bits += 8; }
      if (bits < codeSize) break;
      int code = (int)(acc & ((1u << codeSize) - 1));
      acc >>= codeSize;
      bits -= codeSize;
      if (code == clear) { codeSize = minCode + 1; next = eoi + 1; prev = -1; continue; }
      if (code == eoi) break;
      int sp = 0, cur = code;
      if (prev == -1) {
        if (code >= clear) return fail(RM_ERR_PARSE, "bad LZW stream");
        idx.push_back((uint8_t)code);
        first = (uint8_t)code;
        prev = code;
        continue;
      }
      if (code >= next) {  // the KwKwK case
        if (code > next) return fail(RM_ERR_PARSE, "bad LZW stream");
        stack[sp++] = first;
        cur = prev;
      }
      while (cur >= clear) {
        if (cur >= 4096 || sp >= 4096) return fail(RM_ERR_PARSE, "bad LZW stream");
        stack[sp++] = suffix[cur];
        cur = prefix[cur];
      }
      first = (uint8_t)cur;
      stack[sp++] = first;
      while (sp) idx.push_back(stack[--sp]);
      if (next < 4096) {
        prefix[next] = (uint16_t)prev;
        suffix[next] = first;
        next++;
        if (next == (1 << codeSize) && codeSize < 12) codeSize++;
      }
      prev = code;
    }
    idx.resize((size_t)iw * ih, 0);
    rgba.assign((size_t)W * H * 4, 0);
    // interlaced rows: passes start 0,4,2,1 with steps 8,8,4,2
    std::vector<int> rowOf(ih);
    if (flags & 0x40) {
      int r = 0;
      const int start[4] = {0, 4, 2, 1}, step[4] = {8, 8, 4, 2};
      for (int p = 0; p < 4; p++)
        for (int y = start[p]; y < ih; y += step[p]) rowOf[r++] = y;
    } else {
      for (int y = 0; y < ih; y++) rowOf[y] = y;
    }
    for (int r = 0; r < ih; r++) {
      const int y = iy + rowOf[r];
      if (y < 0 || y >= H) continue;
      for (int x = 0; x < iw; x++) {
        const int X = ix + x;
        if (X < 0 || X >= W) continue;
        const int c = idx[(size_t)r * iw + x];
        uint8_t *o = &rgba[((size_t)y * W + X) * 4];
        if (c == transparent) { o[0] = o[1] = o[2] = o[3] = 0; continue; }
        const uint8_t *e = ct + 3 * (size_t)(c < ctSize ? c : 0);
        o[0] = e[0]; o[1] = e[1]; o[2] = e[2]; o[3] = 255;
      }
    }
    return RM_OK;
  }
  return fail(RM_ERR_PARSE, "no image");
}

}  // namespace rm
