// Host-side texture ingest: replaces the reference's Image (PathTracing/src/image.cpp:1-91, which leans
// on stb_image / stb_image_resize) with a small own decoder.  Off the per-sample path: images are
// decoded once to RGBA8 and uploaded into the device texel atlas by PathTracer::BuildBVH.
//
// Formats: binary PNM (P5 / P6, maxval <= 255 or 16-bit) and non-interlaced PNG (colour types 0, 2, 3,
// 4, 6; bit depths 1-16) through zlib.  Everything is expanded to 4 channels the way stbi_load(..., 4)
// does (grey -> g,g,g,255; 16-bit -> high byte).  Images with a side > 1024 are reduced so the
// longest side is 1024 (image.cpp:47-60) — with an area-average filter, not stb_image_resize's
// Mitchell kernel: the resulting SIZE is pinned by the golden vectors, the filtered texel values are
// "parity unpinned".
#include <zlib.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "pathtracer.h"

namespace {

bool read_file(const std::string& path, std::vector<unsigned char>& out)
{
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) return false;
    std::fseek(f, 0, SEEK_END);
    long n = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    if (n < 0) { std::fclose(f); return false; }
    out.resize((size_t)n);
    size_t got = n ? std::fread(out.data(), 1, (size_t)n, f) : 0;
    std::fclose(f);
    return got == (size_t)n;
}

// ---- PNM ---------------------------------------------------------------------------------------
bool pnm_token(const std::vector<unsigned char>& d, size_t& pos, int& value)
{
    for (;;)
    {
        while (pos < d.size() && (d[pos] == ' ' || d[pos] == '\t' || d[pos] == '\n' || d[pos] == '\r')) pos++;
        if (pos < d.size() && d[pos] == '#') { while (pos < d.size() && d[pos] != '\n') pos++; continue; }
        break;
    }
    if (pos >= d.size() || d[pos] < '0' || d[pos] > '9') return false;
    long v = 0;
    while (pos < d.size() && d[pos] >= '0' && d[pos] <= '9') { v = v * 10 + (d[pos] - '0'); if (v > 1 << 24) return false; pos++; }
    value = (int)v;
    return true;
}

bool decode_pnm(const std::vector<unsigned char>& d, int& w, int& h, std::vector<unsigned char>& rgba)
{
    if (d.size() < 3 || d[0] != 'P' || (d[1] != '5' && d[1] != '6')) return false;
    int comp = d[1] == '6' ? 3 : 1;
    size_t pos = 2;
    int maxv = 0;
    if (!pnm_token(d, pos, w) || !pnm_token(d, pos, h) || !pnm_token(d, pos, maxv)) return false;
    if (w <= 0 || h <= 0 || maxv <= 0 || maxv > 65535) return false;
    pos++;   // single whitespace after maxval
    size_t bps = maxv > 255 ? 2 : 1;
    size_t need = (size_t)w * h * comp * bps;
    if (pos + need > d.size()) return false;
    rgba.resize((size_t)w * h * 4);
    const unsigned char* p = d.data() + pos;
    for (size_t i = 0; i < (size_t)w * h; i++)
    {
        unsigned char c[3];
        for (int k = 0; k < comp; k++) c[k] = p[(i * comp + k) * bps];     // 16-bit: high byte first
        if (comp == 1) c[1] = c[2] = c[0];
        rgba[i * 4] = c[0]; rgba[i * 4 + 1] = c[1]; rgba[i * 4 + 2] = c[2]; rgba[i * 4 + 3] = 255;
    }
    return true;
}

// ---- PNG ---------------------------------------------------------------------------------------
uint32_t be32(const unsigned char* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

int paeth(int a, int b, int c)
{
    int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
    if (pa <= pb && pa <= pc) return a;
    if (pb <= pc) return b;
    return c;
}

bool decode_png(const std::vector<unsigned char>& d, int& w, int& h, std::vector<unsigned char>& rgba)
{
    static const unsigned char sig[8] = { 137, 80, 78, 71, 13, 10, 26, 10 };
    if (d.size() < 33 || std::memcmp(d.data(), sig, 8) != 0) return false;
    size_t pos = 8;
    int depth = 0, ctype = 0, interlace = 0;
    std::vector<unsigned char> idat, plte, trns;
    bool have_ihdr = false;
    while (pos + 12 <= d.size())
    {
        uint32_t len = be32(&d[pos]);
        const unsigned char* type = &d[pos + 4];
        if (pos + 12 + (size_t)len > d.size()) return false;
        const unsigned char* body = &d[pos + 8];
        if (!std::memcmp(type, "IHDR", 4))
        {
            if (len < 13) return false;
            w = (int)be32(body); h = (int)be32(body + 4);
            depth = body[8]; ctype = body[9]; interlace = body[12];
            have_ihdr = true;
        }
        else if (!std::memcmp(type, "PLTE", 4)) plte.assign(body, body + len);
        else if (!std::memcmp(type, "tRNS", 4)) trns.assign(body, body + len);
        else if (!std::memcmp(type, "IDAT", 4)) idat.insert(idat.end(), body, body + len);
        else if (!std::memcmp(type, "IEND", 4)) break;
        pos += 12 + (size_t)len;
    }
    if (!have_ihdr || w <= 0 || h <= 0 || w > (1 << 15) || h > (1 << 15) || interlace != 0) return false;
    int channels;
    switch (ctype) { case 0: channels = 1; break; case 2: channels = 3; break; case 3: channels = 1; break;
                     case 4: channels = 2; break; case 6: channels = 4; break; default: return false; }
    if (!(depth == 1 || depth == 2 || depth == 4 || depth == 8 || depth == 16)) return false;
    if (depth < 8 && !(ctype == 0 || ctype == 3)) return false;
    size_t bpp_bits = (size_t)channels * depth;
    size_t stride = ((size_t)w * bpp_bits + 7) / 8;
    size_t bpp = (bpp_bits + 7) / 8;
    std::vector<unsigned char> raw((stride + 1) * (size_t)h);
    uLongf dlen = (uLongf)raw.size();
    if (uncompress(raw.data(), &dlen, idat.data(), (uLong)idat.size()) != Z_OK || dlen != raw.size()) return false;
    // unfilter in place
    std::vector<unsigned char> prev(stride, 0);
    std::vector<unsigned char> img(stride * (size_t)h);
    for (int y = 0; y < h; y++)
    {
        const unsigned char* src = &raw[(stride + 1) * (size_t)y];
        unsigned char* cur = &img[stride * (size_t)y];
        int ft = src[0];
        for (size_t x = 0; x < stride; x++)
        {
            int a = x >= bpp ? cur[x - bpp] : 0, b = prev[x], c = x >= bpp ? prev[x - bpp] : 0;
            int v = src[1 + x];
            switch (ft) { case 0: break; case 1: v += a; break; case 2: v += b; break;
                          case 3: v += (a + b) / 2; break; case 4: v += paeth(a, b, c); break; default: return false; }
            cur[x] = (unsigned char)v;
        }
        std::memcpy(prev.data(), cur, stride);
    }
    rgba.resize((size_t)w * h * 4);
    for (int y = 0; y < h; y++)
    {
        const unsigned char* row = &img[stride * (size_t)y];
        for (int x = 0; x < w; x++)
        {
            unsigned char s[4] = { 0, 0, 0, 255 };
            for (int k = 0; k < channels; k++)
            {
                if (depth == 8) s[k] = row[(size_t)x * channels + k];
                else if (depth == 16) s[k] = row[((size_t)x * channels + k) * 2];
                else
                {
                    size_t bit = (size_t)x * depth;
                    int v = (row[bit / 8] >> (8 - depth - (bit % 8))) & ((1 << depth) - 1);
                    s[k] = ctype == 3 ? (unsigned char)v : (unsigned char)(v * 255 / ((1 << depth) - 1));
                }
            }
            unsigned char* o = &rgba[((size_t)y * w + x) * 4];
            if (ctype == 0) { o[0] = o[1] = o[2] = s[0]; o[3] = 255; }
            else if (ctype == 2) { o[0] = s[0]; o[1] = s[1]; o[2] = s[2]; o[3] = 255; }
            else if (ctype == 4) { o[0] = o[1] = o[2] = s[0]; o[3] = s[1]; }
            else if (ctype == 6) { o[0] = s[0]; o[1] = s[1]; o[2] = s[2]; o[3] = s[3]; }
            else
            {
                size_t idx = s[0];
                if (idx * 3 + 2 < plte.size()) { o[0] = plte[idx * 3]; o[1] = plte[idx * 3 + 1]; o[2] = plte[idx * 3 + 2]; }
                else { o[0] = o[1] = o[2] = 0; }
                o[3] = idx < trns.size() ? trns[idx] : 255;
            }
        }
    }
    return true;
}

void downscale_area(const std::vector<unsigned char>& src, int w, int h, int nw, int nh, std::vector<unsigned char>& dst)
{
    dst.resize((size_t)nw * nh * 4);
    for (int y = 0; y < nh; y++)
    {
        double y0 = (double)y * h / nh, y1 = (double)(y + 1) * h / nh;
        for (int x = 0; x < nw; x++)
        {
            double x0 = (double)x * w / nw, x1 = (double)(x + 1) * w / nw;
            double acc[4] = { 0, 0, 0, 0 }, wsum = 0;
            for (int sy = (int)y0; sy < h && sy < y1; sy++)
            {
                double wy = std::min(y1, (double)sy + 1) - std::max(y0, (double)sy);
                for (int sx = (int)x0; sx < w && sx < x1; sx++)
                {
                    double wx = std::min(x1, (double)sx + 1) - std::max(x0, (double)sx);
                    const unsigned char* p = &src[((size_t)sy * w + sx) * 4];
                    double ww = wx * wy;
                    for (int k = 0; k < 4; k++) acc[k] += ww * p[k];
                    wsum += ww;
                }
            }
            for (int k = 0; k < 4; k++) dst[((size_t)y * nw + x) * 4 + k] = (unsigned char)std::lround(acc[k] / (wsum > 0 ? wsum : 1));
        }
    }
}

}  // namespace

Image::Image() : mWidth(0), mHeight(0)
{
    mFilename = "";
    mData = 0;
}

Image::Image(const std::string& filename)
{
    mFilename = filename;
    mWidth = mHeight = 0;
    mData = 0;
    Load(mFilename);
}

Image::~Image()
{
    if (mData) std::free(mData);
}

const int Image::width() const { return mWidth; }
const int Image::height() const { return mHeight; }
unsigned char* Image::data() { return mData; }

// image.cpp:38-61
void Image::Load(const std::string& filename)
{
    if (mData) { std::free(mData); mData = 0; }
    mFilename = filename;
    mWidth = mHeight = 0;
    std::vector<unsigned char> file, rgba;
    int w = 0, h = 0;
    if (!read_file(filename, file)) return;                       // missing file -> mData == 0 -> sampler returns 0
    if (!decode_pnm(file, w, h, rgba) && !decode_png(file, w, h, rgba)) return;
    if (w > 1024 || h > 1024)
    {
        float scale = 1024.f / fmax(w, h);                        // image.cpp:49
        int nw = w * scale;
        int nh = h * scale;
        if (nw < 1) nw = 1;
        if (nh < 1) nh = 1;
        std::vector<unsigned char> small;
        downscale_area(rgba, w, h, nw, nh, small);
        rgba.swap(small);
        w = nw; h = nh;
    }
    mData = (unsigned char*)std::malloc(rgba.size());
    if (!mData) return;
    std::memcpy(mData, rgba.data(), rgba.size());
    mWidth = w; mHeight = h;
}

// image.cpp:63-86 — host restatement of the sampler the kernel implements (used by tests / tools)
glm::vec4 Image::tex2D(const glm::vec2& uv)
{
    if (!mData) return glm::vec4(0.0f);
    float u = fmodf(uv.x, 1.0f);
    float v = fmodf(uv.y, 1.0f);
    if (u < 0.0f) u += 1.0f;
    if (v < 0.0f) v += 1.0f;
    int cx = (int)(mWidth * u), cy = (int)(mHeight * v);
    if (cx > mWidth - 1) cx = mWidth - 1;                          // the reference over-reads here when u rounds to 1
    if (cy > mHeight - 1) cy = mHeight - 1;
    if (cx < 0) cx = 0;
    if (cy < 0) cy = 0;
    const unsigned char* p = mData + (4 * (cy * mWidth + cx));
    return glm::vec4((float)p[0] / 255.0f, (float)p[1] / 255.0f, (float)p[2] / 255.0f, (float)p[3] / 255.0f);
}

// PNG export of the RGB8 hand-off buffer: the reference's ExportAt (main.cpp:760-771) writes texData with
// stbi_flip_vertically_on_write(true), i.e. the bottom-up buffer becomes a top-down image.  Stored
// (filter 0) scanlines, zlib-compressed; 8-bit RGB, no alpha.
static void put_be32(std::vector<unsigned char>& v, uint32_t x)
{
    v.push_back((unsigned char)(x >> 24)); v.push_back((unsigned char)(x >> 16));
    v.push_back((unsigned char)(x >> 8)); v.push_back((unsigned char)x);
}
static void put_chunk(std::vector<unsigned char>& out, const char* type, const std::vector<unsigned char>& body)
{
    put_be32(out, (uint32_t)body.size());
    size_t start = out.size();
    out.insert(out.end(), type, type + 4);
    out.insert(out.end(), body.begin(), body.end());
    uint32_t crc = (uint32_t)crc32(0L, out.data() + start, (uInt)(out.size() - start));
    put_be32(out, crc);
}
bool ptk_write_png_rgb8_bottom_up(const char* path, const unsigned char* rgb, int w, int h)
{
    if (!path || !rgb || w <= 0 || h <= 0) return false;
    std::vector<unsigned char> raw((size_t)h * ((size_t)w * 3 + 1));
    for (int y = 0; y < h; y++)
    {
        unsigned char* row = &raw[(size_t)y * ((size_t)w * 3 + 1)];
        row[0] = 0;
        std::memcpy(row + 1, rgb + (size_t)(h - 1 - y) * w * 3, (size_t)w * 3);     // vertical flip
    }
    uLongf clen = compressBound((uLong)raw.size());
    std::vector<unsigned char> comp(clen);
    if (compress2(comp.data(), &clen, raw.data(), (uLong)raw.size(), 6) != Z_OK) return false;
    comp.resize(clen);
    std::vector<unsigned char> out = { 137, 80, 78, 71, 13, 10, 26, 10 };
    std::vector<unsigned char> ihdr;
    put_be32(ihdr, (uint32_t)w); put_be32(ihdr, (uint32_t)h);
    ihdr.push_back(8); ihdr.push_back(2); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);
    put_chunk(out, "IHDR", ihdr);
    put_chunk(out, "IDAT", comp);
    put_chunk(out, "IEND", std::vector<unsigned char>());
    FILE* f = std::fopen(path, "wb");
    if (!f) return false;
    size_t n = std::fwrite(out.data(), 1, out.size(), f);
    std::fclose(f);
    return n == out.size();
}
