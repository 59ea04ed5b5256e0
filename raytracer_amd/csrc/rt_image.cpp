// rt_image.cpp -- texture images for the host scene builder: PNG (8-bit, non-interlaced; zlib inflate) and baseline JPEG
// (Huffman, 8-bit, one or three components, any sampling up to 2x2) decoded to RGB8, written here because the image has
// neither libpng nor libjpeg headers and the reference's own texture loader is an empty stub (material.cu:281-283: "fill
// tex with what is in the image").  rt_image_load in the C ABI lets the tests compare the decoders with other decoders.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <new>
#include <stdexcept>
#include <string>
#include <vector>

#include <zlib.h>

#include "rt_internal.h"

namespace {

bool read_file(const std::string& path, std::vector<uint8_t>& out) {
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) return false;
  fseek(f, 0, SEEK_END);
  const long n = ftell(f);
  fseek(f, 0, SEEK_SET);
  out.resize(n > 0 ? (size_t)n : 0);
  const size_t got = out.empty() ? 0 : fread(out.data(), 1, out.size(), f);
  fclose(f);
  return got == out.size();
}

uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
uint32_t be16(const uint8_t* p) { return ((uint32_t)p[0] << 8) | p[1]; }

// ---------------------------------------------------------------------------------------------------------------- PNG
bool decode_png(const std::vector<uint8_t>& d, int& W, int& H, std::vector<uint8_t>& rgb, std::string& err) {
  size_t at = 8;
  uint32_t w = 0, h = 0; int depth = 0, ctype = -1, interlace = 0;
  std::vector<uint8_t> idat, plte;
  while (at + 12 <= d.size()) {
    const uint32_t len = be32(&d[at]);
    const uint8_t* tag = &d[at + 4];
    if (at + 12 + (size_t)len > d.size()) { err = "truncated PNG chunk"; return false; }
    const uint8_t* body = &d[at + 8];
    if (!memcmp(tag, "IHDR", 4) && len >= 13) { w = be32(body); h = be32(body + 4); depth = body[8]; ctype = body[9]; interlace = body[12]; }
    else if (!memcmp(tag, "PLTE", 4)) plte.assign(body, body + len);
    else if (!memcmp(tag, "IDAT", 4)) idat.insert(idat.end(), body, body + len);
    else if (!memcmp(tag, "IEND", 4)) break;
    at += 12 + (size_t)len;
  }
  const int channels = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
  if (!w || !h || depth != 8 || !channels || interlace) { err = "PNG flavour not supported (8-bit, non-interlaced only)"; return false; }
  if (w > 16384 || h > 16384) { err = "PNG larger than 16384 pixels a side"; return false; }
  const size_t stride = (size_t)w * channels;
  std::vector<uint8_t> raw((stride + 1) * h);
  uLongf raw_len = (uLongf)raw.size();
  if (uncompress(raw.data(), &raw_len, idat.data(), (uLong)idat.size()) != Z_OK || raw_len != raw.size()) { err = "PNG data does not inflate to the image size"; return false; }
  std::vector<uint8_t> img(stride * h);
  for (uint32_t y = 0; y < h; y++) {   // undo the scanline filters
    const uint8_t* in = &raw[(stride + 1) * y];
    uint8_t* out = &img[stride * y];
    const uint8_t* up = y ? out - stride : nullptr;
    const int filter = in[0];
    for (size_t x = 0; x < stride; x++) {
      const int a = x >= (size_t)channels ? out[x - channels] : 0, b = up ? up[x] : 0, c = (up && x >= (size_t)channels) ? up[x - channels] : 0;
      int pred = 0;
      if (filter == 1) pred = a;
      else if (filter == 2) pred = b;
      else if (filter == 3) pred = (a + b) >> 1;
      else if (filter == 4) { const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c); pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c); }
      else if (filter != 0) { err = "PNG scanline filter out of range"; return false; }
      out[x] = (uint8_t)(in[1 + x] + pred);
    }
  }
  W = (int)w; H = (int)h;
  rgb.resize((size_t)w * h * 3);
  for (size_t p = 0; p < (size_t)w * h; p++) {
    const uint8_t* s = &img[p * channels];
    uint8_t* o = &rgb[3 * p];
    if (ctype == 0 || ctype == 4) o[0] = o[1] = o[2] = s[0];
    else if (ctype == 3) { const size_t k = 3 * (size_t)s[0]; if (k + 2 < plte.size()) { o[0] = plte[k]; o[1] = plte[k + 1]; o[2] = plte[k + 2]; } else o[0] = o[1] = o[2] = 0; }
    else { o[0] = s[0]; o[1] = s[1]; o[2] = s[2]; }
  }
  return true;
}

// --------------------------------------------------------------------------------------------------------------- JPEG
struct Huff {
  uint8_t bits[17] = {0};
  uint8_t vals[256] = {0};
  int mincode[17], maxcode[18], valptr[17];
  bool present = false;
  void build() {
    int code = 0, k = 0;
    for (int l = 1; l <= 16; l++) {
      valptr[l] = k; mincode[l] = code;
      code += bits[l]; k += bits[l];
      maxcode[l] = bits[l] ? code - 1 : -1;
      code <<= 1;
    }
    maxcode[17] = 0x7fffffff;
    present = true;
  }
};

struct BitReader {
  const uint8_t* p; const uint8_t* end;
  uint32_t acc = 0; int n = 0; bool hit_marker = false;
  void fill() {
    while (n <= 24) {
      int b = 0;
      if (!hit_marker && p < end) {
        b = *p;
        if (b == 0xff) {
          if (p + 1 < end && p[1] == 0x00) p += 2;
          else { hit_marker = true; b = 0; }   // a marker: feed zeros (the scan ends or a restart follows)
        } else p++;
      }
      acc |= (uint32_t)b << (24 - n);
      n += 8;
    }
  }
  int get(int k) { if (!k) return 0; if (n < k) fill(); const int v = (int)(acc >> (32 - k)); acc <<= k; n -= k; return v; }
  int decode(const Huff& h) {
    if (n < 16) fill();
    int code = 0;
    for (int l = 1; l <= 16; l++) {
      code = (code << 1) | (int)((acc >> 31) & 1u); acc <<= 1; n--;
      if (h.maxcode[l] >= 0 && code <= h.maxcode[l] && code >= h.mincode[l]) return h.vals[h.valptr[l] + code - h.mincode[l]];
    }
    return -1;
  }
  void restart() { acc = 0; n = 0; hit_marker = false; }
};

int extend(int v, int t) { return (t && v < (1 << (t - 1))) ? v - (1 << t) + 1 : v; }

const uint8_t kZigzag[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                             35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

// The definition of the inverse transform, in double, rounded once: f(x,y) = 1/4 sum C(u) C(v) F(u,v) cos(..) cos(..) + 128.
void idct8x8(const int* coef, uint8_t* out, size_t stride) {
  static double basis[8][8];
  static bool ready = false;
  if (!ready) {
    for (int x = 0; x < 8; x++) for (int u = 0; u < 8; u++) basis[x][u] = (u ? 1.0 : std::sqrt(0.5)) * std::cos((2 * x + 1) * u * M_PI / 16.0) / 2.0;
    ready = true;
  }
  double tmp[64];
  for (int v = 0; v < 8; v++)        // rows of coefficients -> along x
    for (int x = 0; x < 8; x++) { double s = 0; for (int u = 0; u < 8; u++) s += basis[x][u] * coef[8 * v + u]; tmp[8 * v + x] = s; }
  for (int x = 0; x < 8; x++)
    for (int y = 0; y < 8; y++) {
      double s = 0; for (int v = 0; v < 8; v++) s += basis[y][v] * tmp[8 * v + x];
      const long r = std::lround(s + 128.0);
      out[y * stride + x] = (uint8_t)(r < 0 ? 0 : (r > 255 ? 255 : r));
    }
}

struct Component { int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0; int bw = 0, bh = 0; std::vector<uint8_t> plane; size_t stride = 0; int pred = 0; };

// Chroma planes stored at half resolution are brought to full size with the triangle filter JPEG decoders commonly use
// (3/4 of the nearer sample, 1/4 of the farther one, in each direction that is halved).
void upsample(const Component& c, int hmax, int vmax, int W, int H, std::vector<uint8_t>& full) {
  const int cw = (W * c.h + hmax - 1) / hmax, ch = (H * c.v + vmax - 1) / vmax;   // samples that carry picture
  full.resize((size_t)W * H);
  const int fx = hmax / c.h, fy = vmax / c.v;
  if (fx == 1 && fy == 1) { for (int y = 0; y < H; y++) memcpy(&full[(size_t)y * W], &c.plane[(size_t)y * c.stride], (size_t)W); return; }
  auto S = [&](int x, int y) { x = x < 0 ? 0 : (x >= cw ? cw - 1 : x); y = y < 0 ? 0 : (y >= ch ? ch - 1 : y); return (int)c.plane[(size_t)y * c.stride + x]; };
  for (int y = 0; y < H; y++) {
    const int sy = fy == 2 ? y >> 1 : y, oy = fy == 2 ? ((y & 1) ? sy + 1 : sy - 1) : sy;
    for (int x = 0; x < W; x++) {
      const int sx = fx == 2 ? x >> 1 : x, ox = fx == 2 ? ((x & 1) ? sx + 1 : sx - 1) : sx;
      int v;
      if (fx == 2 && fy == 2) {
        const int near_col = 3 * S(sx, sy) + S(sx, oy), far_col = 3 * S(ox, sy) + S(ox, oy);
        v = (3 * near_col + far_col + ((x & 1) ? 7 : 8)) >> 4;
      } else if (fx == 2) v = (3 * S(sx, sy) + S(ox, sy) + ((x & 1) ? 2 : 1)) >> 2;
      else v = (3 * S(sx, sy) + S(sx, oy) + ((y & 1) ? 2 : 1)) >> 2;
      full[(size_t)y * W + x] = (uint8_t)v;
    }
  }
}

bool decode_jpeg(const std::vector<uint8_t>& d, int& W, int& H, std::vector<uint8_t>& rgb, std::string& err) {
  int qt[4][64]; bool have_q[4] = {false, false, false, false};
  Huff dc[4], ac[4];
  std::vector<Component> comps;
  int restart_interval = 0, adobe_transform = -1;
  bool have_frame = false;
  size_t at = 2;
  while (at + 4 <= d.size()) {
    if (d[at] != 0xff) { err = "JPEG marker expected"; return false; }
    const int m = d[at + 1];
    if (m == 0xff) { at++; continue; }
    if (m == 0xd8 || (m >= 0xd0 && m <= 0xd7) || m == 0x01) { at += 2; continue; }
    if (m == 0xd9) break;
    const size_t len = be16(&d[at + 2]);
    if (len < 2 || at + 2 + len > d.size()) { err = "truncated JPEG segment"; return false; }
    const uint8_t* s = &d[at + 4];
    const size_t n = len - 2;
    if (m == 0xdb) {   // quantisation tables
      size_t k = 0;
      while (k < n) {
        const int pq = s[k] >> 4, tq = s[k] & 15; k++;
        if (tq > 3 || k + (pq ? 128 : 64) > n) { err = "bad JPEG quantisation table"; return false; }
        for (int i = 0; i < 64; i++) { qt[tq][kZigzag[i]] = pq ? (int)be16(&s[k + 2 * i]) : s[k + i]; }
        k += pq ? 128 : 64; have_q[tq] = true;
      }
    } else if (m == 0xc4) {   // Huffman tables
      size_t k = 0;
      while (k + 17 <= n) {
        const int tc = s[k] >> 4, th = s[k] & 15;
        if (tc > 1 || th > 3) { err = "bad JPEG Huffman table"; return false; }
        Huff& h = tc ? ac[th] : dc[th];
        int total = 0;
        for (int l = 1; l <= 16; l++) { h.bits[l] = s[k + l]; total += h.bits[l]; }
        if (total > 256 || k + 17 + (size_t)total > n) { err = "bad JPEG Huffman table"; return false; }
        memcpy(h.vals, &s[k + 17], (size_t)total);
        h.build();
        k += 17 + (size_t)total;
      }
    } else if (m == 0xc0 || m == 0xc1) {   // baseline / extended sequential frame, Huffman
      if (n < 6 || s[0] != 8) { err = "JPEG sample precision not supported (8 bits only)"; return false; }
      H = (int)be16(&s[1]); W = (int)be16(&s[3]);
      const int nc = s[5];
      if ((nc != 1 && nc != 3) || n < 6 + 3 * (size_t)nc || W <= 0 || H <= 0) { err = "JPEG with other than one or three components"; return false; }
      if (W > 16384 || H > 16384) { err = "JPEG larger than 16384 pixels a side"; return false; }
      comps.assign(nc, Component());
      for (int c = 0; c < nc; c++) {
        comps[c].id = s[6 + 3 * c]; comps[c].h = s[7 + 3 * c] >> 4; comps[c].v = s[7 + 3 * c] & 15; comps[c].tq = s[8 + 3 * c];
        if (comps[c].h < 1 || comps[c].h > 2 || comps[c].v < 1 || comps[c].v > 2 || comps[c].tq > 3) { err = "JPEG sampling factors above 2 not supported"; return false; }
      }
      have_frame = true;
    } else if (m == 0xc2 || (m >= 0xc3 && m <= 0xcf && m != 0xc4 && m != 0xc8 && m != 0xcc)) {
      err = "JPEG flavour not supported (baseline Huffman only; this file is progressive, lossless or arithmetic)"; return false;
    } else if (m == 0xdd && n >= 2) restart_interval = (int)be16(s);
    else if (m == 0xee && n >= 12 && !memcmp(s, "Adobe", 5)) adobe_transform = s[11];
    else if (m == 0xda) {   // scan: the entropy-coded data follows the header
      if (!have_frame) { err = "JPEG scan before its frame header"; return false; }
      if (n < 1) { err = "JPEG scan header cut short"; return false; }
      const int ns = s[0];
      if (ns != (int)comps.size() || n < 1 + 2 * (size_t)ns + 3) { err = "JPEG with several scans not supported"; return false; }
      for (int k = 0; k < ns; k++) {
        const int id = s[1 + 2 * k];
        Component* c = nullptr;
        for (auto& cc : comps) if (cc.id == id) c = &cc;
        if (!c) { err = "JPEG scan names an unknown component"; return false; }
        c->td = s[2 + 2 * k] >> 4; c->ta = s[2 + 2 * k] & 15;
        if (c->td > 3 || c->ta > 3 || !dc[c->td].present || !ac[c->ta].present || !have_q[c->tq]) { err = "JPEG scan uses a table that was not defined"; return false; }
      }
      int hmax = 1, vmax = 1;
      for (auto& c : comps) { hmax = std::max(hmax, c.h); vmax = std::max(vmax, c.v); }
      const int mcux = (W + 8 * hmax - 1) / (8 * hmax), mcuy = (H + 8 * vmax - 1) / (8 * vmax);
      for (auto& c : comps) {
        c.bw = mcux * c.h; c.bh = mcuy * c.v; c.stride = (size_t)c.bw * 8;
        c.plane.assign(c.stride * (size_t)c.bh * 8, 0); c.pred = 0;
      }
      BitReader br{d.data() + std::min(at + 2 + len, d.size()), d.data() + d.size()};   // (a scan header that ends the file: no data, zeros are fed)
      int until_restart = restart_interval;
      for (int my = 0; my < mcuy; my++)
        for (int mx = 0; mx < mcux; mx++) {
          if (restart_interval && until_restart == 0) {   // RSTn: byte-align, skip the marker, reset the predictions
            const uint8_t* q = br.p;
            while (q + 1 < br.end && !(q[0] == 0xff && q[1] >= 0xd0 && q[1] <= 0xd7)) q++;
            br.p = q + 2 <= br.end ? q + 2 : br.end;
            br.restart();
            for (auto& c : comps) c.pred = 0;
            until_restart = restart_interval;
          }
          for (auto& c : comps)
            for (int by = 0; by < c.v; by++)
              for (int bx = 0; bx < c.h; bx++) {
                int coef[64] = {0};
                const int t = br.decode(dc[c.td]);
                if (t < 0 || t > 11) { err = "bad JPEG Huffman code"; return false; }
                c.pred += extend(br.get(t), t);
                coef[0] = c.pred * qt[c.tq][0];
                for (int k = 1; k < 64;) {
                  const int rs = br.decode(ac[c.ta]);
                  if (rs < 0) { err = "bad JPEG Huffman code"; return false; }
                  const int r = rs >> 4, sz = rs & 15;
                  if (!sz) { if (r == 15) { k += 16; continue; } break; }
                  k += r;
                  if (k > 63) { err = "JPEG coefficient run beyond the block"; return false; }
                  coef[kZigzag[k]] = extend(br.get(sz), sz) * qt[c.tq][kZigzag[k]];
                  k++;
                }
                const size_t ox = ((size_t)mx * c.h + bx) * 8, oy = ((size_t)my * c.v + by) * 8;
                idct8x8(coef, &c.plane[oy * c.stride + ox], c.stride);
              }
          if (restart_interval) until_restart--;
        }
      // to RGB
      rgb.resize((size_t)W * H * 3);
      if (comps.size() == 1) {
        for (int y = 0; y < H; y++) for (int x = 0; x < W; x++) { const uint8_t g = comps[0].plane[(size_t)y * comps[0].stride + x]; uint8_t* o = &rgb[3 * ((size_t)y * W + x)]; o[0] = o[1] = o[2] = g; }
        return true;
      }
      std::vector<uint8_t> full[3];
      for (int c = 0; c < 3; c++) upsample(comps[c], hmax, vmax, W, H, full[c]);
      const bool ycc = adobe_transform != 0;   // Adobe transform 0 with three components: the planes are R, G, B already
      for (size_t p = 0; p < (size_t)W * H; p++) {
        uint8_t* o = &rgb[3 * p];
        if (!ycc) { o[0] = full[0][p]; o[1] = full[1][p]; o[2] = full[2][p]; continue; }
        const int y = full[0][p], cb = full[1][p] - 128, cr = full[2][p] - 128;
        // ITU-R BT.601 as JFIF prescribes it, in 16-bit fixed point
        const int r = y + ((91881 * cr + 32768) >> 16);
        const int g = y + ((-22554 * cb - 46802 * cr + 32768) >> 16);
        const int b = y + ((116130 * cb + 32768) >> 16);
        o[0] = (uint8_t)(r < 0 ? 0 : r > 255 ? 255 : r); o[1] = (uint8_t)(g < 0 ? 0 : g > 255 ? 255 : g); o[2] = (uint8_t)(b < 0 ? 0 : b > 255 ? 255 : b);
      }
      return true;
    }
    at += 2 + len;
  }
  err = "JPEG without a scan";
  return false;
}

}  // namespace

// RGB8, row 0 on top.  Returns false with a message in `err`.
bool rt_decode_image_file(const std::string& path, int& W, int& H, std::vector<uint8_t>& rgb, std::string& err) {
  std::vector<uint8_t> d;
  if (!read_file(path, d)) { err = "No " + path + " has been found"; return false; }
  static const uint8_t png_sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
  if (d.size() >= 8 && !memcmp(d.data(), png_sig, 8)) return decode_png(d, W, H, rgb, err);
  if (d.size() >= 4 && d[0] == 0xff && d[1] == 0xd8) return decode_jpeg(d, W, H, rgb, err);
  err = "texture " + path + " is neither a PNG nor a JPEG file";
  return false;
}

extern "C" int rt_image_load(const char* path, int* width, int* height, uint8_t* rgb, size_t capacity) {
  if (!path || !width || !height) { rt_set_error("rt_image_load: NULL argument"); return RT_ERR_ARG; }
  try {   // (no exception leaves an extern "C" function: a decode that runs out of memory is an error code)
    int W = 0, H = 0; std::vector<uint8_t> px; std::string err;
    if (!rt_decode_image_file(path, W, H, px, err)) { rt_set_error(err); return err.rfind("No ", 0) == 0 ? RT_ERR_IO : RT_ERR_PARSE; }
    *width = W; *height = H;
    if (rgb) {
      if (capacity < px.size()) { rt_set_error("rt_image_load: the buffer is smaller than width * height * 3"); return RT_ERR_ARG; }
      memcpy(rgb, px.data(), px.size());
    }
    return RT_OK;
  } catch (const std::bad_alloc&) {
    rt_set_error("rt_image_load: out of memory");
    return RT_ERR_NOMEM;
  } catch (const std::exception& e) {
    rt_set_error(std::string("rt_image_load: ") + e.what());
    return RT_ERR_PARSE;
  }
}
