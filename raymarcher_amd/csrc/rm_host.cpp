// rm_host.cpp — host side kept from the reference: status/error plumbing, Settings defaults, camera
// matrices, row-tile partition helpers and the PNG writer.  No GPU work happens in this file.
//
// Reference counterparts: src/settings.h:19-55, src/camera/camera.cpp:8-133,
// src/realtime.cpp:284-350 (saveViewportImage).
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/raymarcher_amd.h"
#include "rm_internal.h"
#include "rm_mat4.h"

namespace rm {
namespace {
thread_local std::string t_error;
std::atomic<int> g_rootRelief{0};  // rm_set_root_relief: the partition's root relief, process-wide
}
void set_error(const std::string &msg) { t_error = msg; }
int root_relief() { return g_rootRelief.load(); }

}  // namespace rm

using namespace rm;

extern "C" {

int rm_abi_version(void) { return RM_ABI_VERSION; }

const char *rm_skybox_face_path(int which, int face) {
  // RayMarchScene::getCubeMapWithType, raymarchscene.cpp:50-86 (lists as written there)
  static const char *const kFaces[3][6] = {
      {"texture_store/cube_map/beach/+x.jpg", "texture_store/cube_map/beach/-x.jpg", "texture_store/cube_map/beach/+y.jpg",
       "texture_store/cube_map/beach/-y.jpg", "texture_store/cube_map/beach/+z.jpg", "texture_store/cube_map/beach/-z.jpg"},
      {"texture_store/cube_map/night/-x.png", "texture_store/cube_map/night/+x.png", "texture_store/cube_map/night/-y.png",
       "texture_store/cube_map/night/+y.png", "texture_store/cube_map/night/+z.png", "texture_store/cube_map/night/-z.png"},
      {"texture_store/cube_map/island/+x.png", "texture_store/cube_map/island/-x.png", "texture_store/cube_map/island/+y.png",
       "texture_store/cube_map/island/-y.png", "texture_store/cube_map/island/+z.png", "texture_store/cube_map/island/-z.png"}};
  if (which < 1 || which > 3 || face < 0 || face > 5) return nullptr;
  return kFaces[which - 1][face];
}

void rm_ltc_quantise(const float *table, uint8_t *out, int texels) {
  for (int i = 0; i < texels * 4; i++) {
    float v = table[i];
    v = (v < 0.0f) ? 0.0f : ((v > 1.0f) ? 1.0f : v);
    if (v != v) v = 0.0f;
    out[i] = (uint8_t)std::nearbyintf(v * 255.0f);
  }
}

const char *rm_status_string(int status) {
  switch (status) {
    case RM_OK: return "RM_OK";
    case RM_ERR_INVALID_ARGUMENT: return "RM_ERR_INVALID_ARGUMENT";
    case RM_ERR_CAPACITY: return "RM_ERR_CAPACITY";
    case RM_ERR_UNSUPPORTED: return "RM_ERR_UNSUPPORTED";
    case RM_ERR_DEVICE: return "RM_ERR_DEVICE";
    case RM_ERR_IO: return "RM_ERR_IO";
    case RM_ERR_PARSE: return "RM_ERR_PARSE";
    default: return "RM_ERR_UNKNOWN";
  }
}
const char *rm_last_error(void) { return t_error.c_str(); }

void rm_settings_default(RmSettings *s) {
  if (!s) return;
  std::memset(s, 0, sizeof(*s));
  s->maxSteps = 256;     // frag:28
  s->fractalIters = 20;  // frag:29
  s->mengerLevels = 4;   // frag:1056
  s->numReflection = 1;  // frag:45
  s->features = RM_FEAT_REFERENCE_DEFAULT;
}

void rm_host_settings_default(RmHostSettings *s) {
  if (!s) return;
  std::memset(s, 0, sizeof(*s));
  s->screenWidth = 1024;  // settings.h:21-22
  s->screenHeight = 768;
  s->nearPlane = 0.1f;    // GUI defaults, mainwindow.cpp:129-130
  s->farPlane = 100.0f;
  s->power = 8.0f;        // settings.h:47
}

// camera.cpp:8-34 (initializeCamera) → :74-97 (view = R·T) and :105-133 (proj = remap·unhinge·scale),
// then invProjView = inverse(proj·view) as configureCameraUniforms does (realtimerender.cpp:596-615).
int rm_camera_build(const RmCameraData *cd, int W, int H, float nearPlane, float farPlane, float view[16],
                    float proj[16], RmCamera *out) {
  if (!cd || W <= 0 || H <= 0 || !(farPlane > 0.0f)) { set_error("bad camera arguments"); return RM_ERR_INVALID_ARGUMENT; }
  const float lx = cd->look[0], ly = cd->look[1], lz = cd->look[2];
  const float ux = cd->up[0], uy = cd->up[1], uz = cd->up[2];
  // w = −normalize(look); v = normalize(up − dot(up,w)·w); u = v × w
  const float ll = std::sqrt(lx * lx + ly * ly + lz * lz);
  if (!(ll > 0.0f)) { set_error("camera look vector has zero length"); return RM_ERR_INVALID_ARGUMENT; }
  const float il = 1.0f / ll;
  const float wx = -(lx * il), wy = -(ly * il), wz = -(lz * il);
  const float duw = ux * wx + uy * wy + uz * wz;
  float vx = ux - duw * wx, vy = uy - duw * wy, vz = uz - duw * wz;
  const float vl = std::sqrt(vx * vx + vy * vy + vz * vz);
  if (!(vl > 0.0f)) { set_error("camera up vector is parallel to look"); return RM_ERR_INVALID_ARGUMENT; }
  const float iv = 1.0f / vl;
  vx *= iv; vy *= iv; vz *= iv;
  const float uxx = vy * wz - wy * vz, uyy = vz * wx - wz * vx, uzz = vx * wy - wx * vy;
  Mat4 T = mat_identity();
  T.at(0, 3) = -cd->pos[0]; T.at(1, 3) = -cd->pos[1]; T.at(2, 3) = -cd->pos[2];
  Mat4 R = mat_identity();
  R.at(0, 0) = uxx; R.at(0, 1) = uyy; R.at(0, 2) = uzz;
  R.at(1, 0) = vx; R.at(1, 1) = vy; R.at(1, 2) = vz;
  R.at(2, 0) = wx; R.at(2, 1) = wy; R.at(2, 2) = wz;
  const Mat4 V = mat_mul(R, T);
  // projection
  const float aspect = (float)W / (float)H;
  const float vh = 2.0f * farPlane * std::tan(cd->heightAngle / 2.0f);
  const float vw = aspect * vh;
  Mat4 S = mat_identity();
  S.at(0, 0) = 2.0f / vw; S.at(1, 1) = 2.0f / vh; S.at(2, 2) = 1.0f / farPlane;
  const float c = -nearPlane / farPlane;
  Mat4 U = mat_identity();
  U.at(2, 2) = 1.0f / (1.0f + c); U.at(3, 2) = -1.0f; U.at(2, 3) = -c / (1.0f + c); U.at(3, 3) = 0.0f;
  Mat4 G = mat_identity();
  G.at(2, 2) = -2.0f; G.at(2, 3) = -1.0f;
  const Mat4 P = mat_mul(mat_mul(G, U), S);
  const Mat4 inv = mat_inverse(mat_mul(P, V));
  if (view) std::memcpy(view, V.m, sizeof(V.m));
  if (proj) std::memcpy(proj, P.m, sizeof(P.m));
  if (out) {
    std::memcpy(out->invProjView, inv.m, sizeof(inv.m));
    out->initialFar = farPlane;
    out->eyePosition[0] = cd->pos[0]; out->eyePosition[1] = cd->pos[1]; out->eyePosition[2] = cd->pos[2];
    out->eyePosition[3] = cd->pos[3];
  }
  return RM_OK;
}

int rm_shard_rows(int H, int tileRows, int shard, int numShards) {
  if (H <= 0 || tileRows <= 0 || numShards <= 0 || shard < 0 || shard >= numShards) return -1;
  return shard_rows(H, tileRows, shard, numShards, root_relief());
}
int rm_shard_row_to_frame(int H, int tileRows, int shard, int numShards, int localRow) {
  if (H <= 0 || tileRows <= 0 || numShards <= 0 || shard < 0 || shard >= numShards) return -1;
  if (localRow < 0 || localRow >= shard_rows(H, tileRows, shard, numShards, root_relief())) return -1;
  return tile_of(shard, localRow / tileRows, numShards, root_relief()) * tileRows + (localRow % tileRows);
}
int rm_set_root_relief(int K) {
  if (K != 0 && (K < 2 || K > 64)) { set_error("root relief must be 0 (off) or 2..64"); return RM_ERR_INVALID_ARGUMENT; }
  g_rootRelief.store(K);
  return RM_OK;
}
int rm_get_root_relief(void) { return g_rootRelief.load(); }

// ---- PNG reader (stands in for QImage::load → RGBA8888 → mirrored(), raymarchscene.cpp:198-209) ------------
namespace {
inline uint32_t be32r(const uint8_t *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
inline int paeth(int a, int b, int c) {
  int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
  return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}
}  // namespace

int rm_image_load(const char *path, int flipVertical, uint8_t **outPixels, int *w, int *h) {
  if (!path || !outPixels || !w || !h) { set_error("null argument"); return RM_ERR_INVALID_ARGUMENT; }
  *outPixels = nullptr;
  FILE *f = std::fopen(path, "rb");
  if (!f) { set_error(std::string("could not open ") + path); return RM_ERR_IO; }
  std::vector<uint8_t> file;
  uint8_t buf[65536];
  size_t n;
  while ((n = std::fread(buf, 1, sizeof buf, f)) > 0) file.insert(file.end(), buf, buf + n);
  std::fclose(f);
  const bool isJpeg = file.size() >= 4 && file[0] == 0xFF && file[1] == 0xD8;
  const bool isGif = file.size() >= 6 && !std::memcmp(file.data(), "GIF8", 4);
  if (isJpeg || isGif) {  // rm_jpeg.cpp / rm_gif.cpp
    std::vector<uint8_t> px;
    int jw = 0, jh = 0;
    int st = isJpeg ? jpeg_decode(file, px, jw, jh) : gif_decode(file, px, jw, jh);
    if (st != RM_OK) return st;
    uint8_t *o = static_cast<uint8_t *>(std::malloc(px.size()));
    if (!o) { set_error("out of memory"); return RM_ERR_IO; }
    const size_t rowBytes = (size_t)jw * 4;
    for (int y = 0; y < jh; y++) std::memcpy(o + (size_t)(flipVertical ? jh - 1 - y : y) * rowBytes, &px[(size_t)y * rowBytes], rowBytes);
    *outPixels = o; *w = jw; *h = jh;
    return RM_OK;
  }
  static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n'};
  if (file.size() < 33 || std::memcmp(file.data(), sig, 8) != 0) {
    set_error(std::string(path) + ": not a PNG, JPEG or GIF file");
    return RM_ERR_UNSUPPORTED;
  }
  uint32_t W = 0, H = 0;
  int depth = 0, ctype = 0, interlace = 0;
  std::vector<uint8_t> idat, plte, trns;
  size_t pos = 8;
  while (pos + 12 <= file.size()) {
    uint32_t len = be32r(&file[pos]);
    const char *tag = reinterpret_cast<const char *>(&file[pos + 4]);
    if (pos + 12 + (size_t)len > file.size()) { set_error("truncated PNG chunk"); return RM_ERR_PARSE; }
    const uint8_t *d = &file[pos + 8];
    if (!std::memcmp(tag, "IHDR", 4) && len >= 13) {
      W = be32r(d); H = be32r(d + 4); depth = d[8]; ctype = d[9]; interlace = d[12];
    } else if (!std::memcmp(tag, "PLTE", 4)) plte.assign(d, d + len);
    else if (!std::memcmp(tag, "tRNS", 4)) trns.assign(d, d + len);
    else if (!std::memcmp(tag, "IDAT", 4)) idat.insert(idat.end(), d, d + len);
    else if (!std::memcmp(tag, "IEND", 4)) break;
    pos += 12 + (size_t)len;
  }
  if (W == 0 || H == 0 || W > 32768 || H > 32768) { set_error("bad PNG size"); return RM_ERR_PARSE; }
  if (interlace != 0) { set_error("interlaced PNG not supported"); return RM_ERR_UNSUPPORTED; }
  int channels = (ctype == 0) ? 1 : (ctype == 2) ? 3 : (ctype == 3) ? 1 : (ctype == 4) ? 2 : (ctype == 6) ? 4 : 0;
  if (!channels || !(depth == 8 || depth == 16 || (ctype == 3 && (depth == 1 || depth == 2 || depth == 4)) ||
                     (ctype == 0 && (depth == 1 || depth == 2 || depth == 4)))) {
    set_error("unsupported PNG colour type / bit depth");
    return RM_ERR_UNSUPPORTED;
  }
  const size_t bpp = (size_t)std::max(1, channels * depth / 8), stride = ((size_t)W * channels * depth + 7) / 8;
  std::vector<uint8_t> raw((stride + 1) * H);
  uLongf rawLen = raw.size();
  if (uncompress(raw.data(), &rawLen, idat.data(), idat.size()) != Z_OK || rawLen != raw.size()) {
    set_error("PNG inflate failed");
    return RM_ERR_PARSE;
  }
  std::vector<uint8_t> img(stride * H);
  for (uint32_t y = 0; y < H; y++) {
    const uint8_t *in = &raw[(stride + 1) * y];
    uint8_t *cur = &img[stride * y];
    const uint8_t *up = y ? &img[stride * (y - 1)] : nullptr;
    const int ft = in[0];
    for (size_t x = 0; x < stride; x++) {
      int a = x >= bpp ? cur[x - bpp] : 0, b = up ? up[x] : 0, c = (up && x >= bpp) ? up[x - bpp] : 0, v = in[1 + x];
      switch (ft) {
        case 0: break;
        case 1: v += a; break;
        case 2: v += b; break;
        case 3: v += (a + b) / 2; break;
        case 4: v += paeth(a, b, c); break;
        default: set_error("bad PNG filter"); return RM_ERR_PARSE;
      }
      cur[x] = (uint8_t)v;
    }
  }
  uint8_t *out = static_cast<uint8_t *>(std::malloc((size_t)W * H * 4));
  if (!out) { set_error("out of memory"); return RM_ERR_IO; }
  auto sample = [&](const uint8_t *row, size_t idx) -> int {  // idx-th sample of the row, scaled to 8 bits
    if (depth == 8) return row[idx];
    if (depth == 16) return row[idx * 2];  // high byte (what an 8-bit conversion keeps)
    const int per = 8 / depth, shift = (per - 1 - (int)(idx % per)) * depth;
    const int v = (row[idx / per] >> shift) & ((1 << depth) - 1);
    return (ctype == 3) ? v : v * 255 / ((1 << depth) - 1);
  };
  for (uint32_t y = 0; y < H; y++) {
    const uint8_t *row = &img[stride * y];
    uint8_t *o = out + (size_t)(flipVertical ? (H - 1 - y) : y) * W * 4;
    for (uint32_t x = 0; x < W; x++, o += 4) {
      if (ctype == 3) {
        const int i = sample(row, x);
        if ((size_t)i * 3 + 2 >= plte.size() + 0 && (size_t)i * 3 + 2 >= plte.size()) { o[0] = o[1] = o[2] = 0; o[3] = 255; continue; }
        o[0] = plte[i * 3]; o[1] = plte[i * 3 + 1]; o[2] = plte[i * 3 + 2];
        o[3] = ((size_t)i < trns.size()) ? trns[i] : 255;
      } else if (ctype == 0) { o[0] = o[1] = o[2] = (uint8_t)sample(row, x); o[3] = 255; }
      else if (ctype == 4) { o[0] = o[1] = o[2] = (uint8_t)sample(row, x * 2); o[3] = (uint8_t)sample(row, x * 2 + 1); }
      else if (ctype == 2) { o[0] = (uint8_t)sample(row, x * 3); o[1] = (uint8_t)sample(row, x * 3 + 1); o[2] = (uint8_t)sample(row, x * 3 + 2); o[3] = 255; }
      else { o[0] = (uint8_t)sample(row, x * 4); o[1] = (uint8_t)sample(row, x * 4 + 1); o[2] = (uint8_t)sample(row, x * 4 + 2); o[3] = (uint8_t)sample(row, x * 4 + 3); }
    }
  }
  *outPixels = out; *w = (int)W; *h = (int)H;
  return RM_OK;
}
void rm_image_free(uint8_t *pixels) { std::free(pixels); }

// RGBA8, filter 0 on every row, one zlib stream.
int rm_write_png(const char *path, const uint8_t *rgba, int W, int H) {
  if (!path || !rgba || W <= 0 || H <= 0) { set_error("bad png arguments"); return RM_ERR_INVALID_ARGUMENT; }
  std::vector<uint8_t> raw((size_t)H * ((size_t)W * 4 + 1));
  for (int y = 0; y < H; y++) {
    raw[(size_t)y * (W * 4 + 1)] = 0;
    std::memcpy(&raw[(size_t)y * (W * 4 + 1) + 1], rgba + (size_t)y * W * 4, (size_t)W * 4);
  }
  uLongf zlen = compressBound(raw.size());
  std::vector<uint8_t> z(zlen);
  if (compress2(z.data(), &zlen, raw.data(), raw.size(), 1) != Z_OK) { set_error("zlib compress failed"); return RM_ERR_IO; }
  FILE *f = std::fopen(path, "wb");
  if (!f) { set_error(std::string("cannot open ") + path); return RM_ERR_IO; }
  auto be32 = [](uint8_t *p, uint32_t v) { p[0] = v >> 24; p[1] = v >> 16; p[2] = v >> 8; p[3] = v; };
  auto chunk = [&](const char *tag, const uint8_t *data, uint32_t len) {
    uint8_t hdr[8];
    be32(hdr, len);
    std::memcpy(hdr + 4, tag, 4);
    std::fwrite(hdr, 1, 8, f);
    if (len) std::fwrite(data, 1, len, f);
    uLong crc = crc32(0L, reinterpret_cast<const Bytef *>(tag), 4);
    if (len) crc = crc32(crc, data, len);
    uint8_t c[4];
    be32(c, (uint32_t)crc);
    std::fwrite(c, 1, 4, f);
  };
  const uint8_t sig[8] = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n'};
  std::fwrite(sig, 1, 8, f);
  uint8_t ihdr[13];
  be32(ihdr, (uint32_t)W); be32(ihdr + 4, (uint32_t)H);
  ihdr[8] = 8; ihdr[9] = 6; ihdr[10] = 0; ihdr[11] = 0; ihdr[12] = 0;
  chunk("IHDR", ihdr, 13);
  chunk("IDAT", z.data(), (uint32_t)zlen);
  chunk("IEND", nullptr, 0);
  const bool ok = std::fclose(f) == 0;
  if (!ok) { set_error("short write"); return RM_ERR_IO; }
  return RM_OK;
}

}  // extern "C"
