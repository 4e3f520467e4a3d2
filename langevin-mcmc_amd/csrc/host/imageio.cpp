// Minimal image I/O for the drop-in front end (replaces OpenImageIO, which the
// reference uses in /root/reference/src/image.cpp:6-60 and bitmaptexture.h):
//   * OpenEXR scanline files, compression NONE/ZIPS/ZIP, HALF or FLOAT channels (read);
//     3-channel HALF, ZIP (write) -- the format of the reference's outputs (image.cpp:47-61).
//   * PNG (8/16-bit gray, RGB, palette, +alpha; non-interlaced) (read).
// zlib comes from the system library (the reference links -lz too, src/Tupfile).
#include "imageio.h"

#include <zlib.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <stdexcept>

namespace lmc {

static std::vector<uint8_t> ReadFile(const std::string &fn) {
    FILE *f = fopen(fn.c_str(), "rb");
    if (!f) throw std::runtime_error("File not found: " + fn);
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    std::vector<uint8_t> b((size_t)n);
    if (n > 0 && fread(b.data(), 1, (size_t)n, f) != (size_t)n) {
        fclose(f);
        throw std::runtime_error("short read: " + fn);
    }
    fclose(f);
    return b;
}

float HalfToFloat(uint16_t h) {
    uint32_t s = (h >> 15) & 1, e = (h >> 10) & 0x1f, m = h & 0x3ff, o;
    if (e == 0) {
        if (m == 0)
            o = s << 31;
        else {
            int sh = 0;
            while (!(m & 0x400)) {
                m <<= 1;
                sh++;
            }
            m &= 0x3ff;
            o = (s << 31) | ((uint32_t)(127 - 15 - sh + 1) << 23) | (m << 13);
        }
    } else if (e == 31)
        o = (s << 31) | 0x7f800000u | (m << 13);
    else
        o = (s << 31) | ((e - 15 + 127) << 23) | (m << 13);
    float f;
    memcpy(&f, &o, 4);
    return f;
}

uint16_t FloatToHalf(float f) {  // round to nearest even, overflow -> inf
    uint32_t x;
    memcpy(&x, &f, 4);
    uint32_t s = (x >> 16) & 0x8000u;
    int32_t e = (int32_t)((x >> 23) & 0xff) - 127 + 15;
    uint32_t m = x & 0x7fffffu;
    if (((x >> 23) & 0xff) == 0xff) return (uint16_t)(s | 0x7c00u | (m ? 0x200u : 0));
    if (e >= 31) return (uint16_t)(s | 0x7c00u);
    if (e <= 0) {
        if (e < -10) return (uint16_t)s;
        m |= 0x800000u;
        int shift = 14 - e;
        uint32_t r = m >> shift;
        uint32_t rem = m & ((1u << shift) - 1), half = 1u << (shift - 1);
        if (rem > half || (rem == half && (r & 1))) r++;
        return (uint16_t)(s | r);
    }
    uint32_t r = ((uint32_t)e << 10) | (m >> 13);
    uint32_t rem = m & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (r & 1))) r++;
    return (uint16_t)(s | r);
}

// ------------------------------------------------------------------------------------------ EXR
namespace {
struct Reader {
    const uint8_t *p, *end;
    template <class T>
    T get() {
        if (p + sizeof(T) > end) throw std::runtime_error("EXR: truncated");
        T v;
        memcpy(&v, p, sizeof(T));
        p += sizeof(T);
        return v;
    }
    std::string str() {
        std::string s;
        while (p < end && *p) s.push_back((char)*p++);
        if (p >= end) throw std::runtime_error("EXR: truncated string");
        p++;
        return s;
    }
};
struct Chan {
    std::string name;
    int type;  // 0 uint, 1 half, 2 float
};
}  // namespace

Image3f ReadEXR(const std::string &fn) {
    std::vector<uint8_t> buf = ReadFile(fn);
    Reader r{buf.data(), buf.data() + buf.size()};
    if (r.get<uint32_t>() != 20000630u) throw std::runtime_error("EXR: bad magic " + fn);
    uint32_t ver = r.get<uint32_t>();
    if (ver & 0x200u) throw std::runtime_error("EXR: tiled files unsupported");
    if (ver & 0x1800u) throw std::runtime_error("EXR: deep/multipart unsupported");
    std::vector<Chan> chans;
    int compression = 0, xmin = 0, ymin = 0, xmax = -1, ymax = -1, lineOrder = 0;
    for (;;) {
        std::string name = r.str();
        if (name.empty()) break;
        std::string type = r.str();
        int32_t size = r.get<int32_t>();
        const uint8_t *a = r.p;
        if (r.p + size > r.end) throw std::runtime_error("EXR: truncated attribute");
        if (name == "channels") {
            Reader c{a, a + size};
            for (;;) {
                std::string cn = c.str();
                if (cn.empty()) break;
                Chan ch;
                ch.name = cn;
                ch.type = c.get<int32_t>();
                c.get<uint32_t>();  // pLinear + reserved
                int xs = c.get<int32_t>(), ys = c.get<int32_t>();
                if (xs != 1 || ys != 1) throw std::runtime_error("EXR: subsampled channels unsupported");
                chans.push_back(ch);
            }
        } else if (name == "compression")
            compression = a[0];
        else if (name == "dataWindow") {
            memcpy(&xmin, a, 4), memcpy(&ymin, a + 4, 4), memcpy(&xmax, a + 8, 4), memcpy(&ymax, a + 12, 4);
        } else if (name == "lineOrder")
            lineOrder = a[0];
        r.p += size;
    }
    (void)lineOrder;
    const int W = xmax - xmin + 1, H = ymax - ymin + 1;
    if (W <= 0 || H <= 0 || chans.empty()) throw std::runtime_error("EXR: bad header");
    int linesPerBlock;
    switch (compression) {
        case 0: case 2: linesPerBlock = 1; break;
        case 3: linesPerBlock = 16; break;
        default: throw std::runtime_error("EXR: unsupported compression " + std::to_string(compression));
    }
    size_t bytesPerLine = 0;
    for (auto &c : chans) bytesPerLine += (size_t)W * (c.type == 1 ? 2 : 4);
    const int nBlocks = (H + linesPerBlock - 1) / linesPerBlock;
    std::vector<uint64_t> offs(nBlocks);
    for (int i = 0; i < nBlocks; i++) offs[i] = r.get<uint64_t>();
    // channel -> rgb slot
    std::vector<int> slot(chans.size(), -1);
    bool hasRGB = false;
    for (size_t i = 0; i < chans.size(); i++) {
        if (chans[i].name == "R") slot[i] = 0, hasRGB = true;
        if (chans[i].name == "G") slot[i] = 1;
        if (chans[i].name == "B") slot[i] = 2;
    }
    if (!hasRGB) {  // single channel -> grey
        if (chans.size() == 1 || chans[0].name == "Y") slot[0] = 3;
        else throw std::runtime_error("EXR: no R/G/B or Y channel");
    }
    Image3f img;
    img.width = W, img.height = H;
    img.data.assign((size_t)W * H * 3, 0.f);
    std::vector<uint8_t> raw, tmp;
    for (int b = 0; b < nBlocks; b++) {
        Reader c{buf.data() + offs[b], buf.data() + buf.size()};
        int y0 = c.get<int32_t>() - ymin;
        int32_t dsize = c.get<int32_t>();
        if (c.p + dsize > c.end) throw std::runtime_error("EXR: truncated chunk");
        int nl = std::min(linesPerBlock, H - y0);
        size_t usize = bytesPerLine * (size_t)nl;
        raw.resize(usize);
        if (compression == 0 || (size_t)dsize >= usize) {
            memcpy(raw.data(), c.p, std::min<size_t>(usize, (size_t)dsize));
        } else {
            tmp.resize(usize);
            uLongf dl = (uLongf)usize;
            if (uncompress(tmp.data(), &dl, c.p, (uLong)dsize) != Z_OK || dl != usize) throw std::runtime_error("EXR: zlib error");
            for (size_t i = 1; i < usize; i++) tmp[i] = (uint8_t)(tmp[i - 1] + tmp[i] - 128);  // predictor
            const uint8_t *t1 = tmp.data(), *t2 = tmp.data() + (usize + 1) / 2;           // de-interleave
            for (size_t i = 0; i < usize;) {
                raw[i++] = *t1++;
                if (i < usize) raw[i++] = *t2++;
            }
        }
        const uint8_t *q = raw.data();
        for (int l = 0; l < nl; l++) {
            int y = y0 + l;
            for (size_t ci = 0; ci < chans.size(); ci++) {
                for (int x = 0; x < W; x++) {
                    float v;
                    if (chans[ci].type == 1) {
                        uint16_t h;
                        memcpy(&h, q, 2);
                        q += 2;
                        v = HalfToFloat(h);
                    } else if (chans[ci].type == 2) {
                        memcpy(&v, q, 4);
                        q += 4;
                    } else {
                        uint32_t u;
                        memcpy(&u, q, 4);
                        q += 4;
                        v = (float)u;
                    }
                    float *px = &img.data[((size_t)y * W + x) * 3];
                    if (slot[ci] == 3) px[0] = px[1] = px[2] = v;
                    else if (slot[ci] >= 0) px[slot[ci]] = v;
                }
            }
        }
    }
    return img;
}

void WriteEXRHalf(const std::string &fn, const float *rgb, int W, int H) {
    std::vector<uint8_t> out;
    auto put = [&](const void *p, size_t n) { out.insert(out.end(), (const uint8_t *)p, (const uint8_t *)p + n); };
    auto puts = [&](const char *s) { put(s, strlen(s) + 1); };
    auto puti = [&](int32_t v) { put(&v, 4); };
    uint32_t magic = 20000630u, ver = 2;
    put(&magic, 4), put(&ver, 4);
    puts("channels"), puts("chlist"), puti(3 * 18 + 1);
    for (const char *cn : {"B", "G", "R"}) {
        puts(cn);
        puti(1);  // HALF
        puti(0), puti(1), puti(1);
    }
    out.push_back(0);
    puts("compression"), puts("compression"), puti(1), out.push_back(3);  // ZIP
    int32_t box[4] = {0, 0, W - 1, H - 1};
    puts("dataWindow"), puts("box2i"), puti(16), put(box, 16);
    puts("displayWindow"), puts("box2i"), puti(16), put(box, 16);
    puts("lineOrder"), puts("lineOrder"), puti(1), out.push_back(0);
    float one = 1.f, zero2[2] = {0.f, 0.f};
    puts("pixelAspectRatio"), puts("float"), puti(4), put(&one, 4);
    puts("screenWindowCenter"), puts("v2f"), puti(8), put(zero2, 8);
    puts("screenWindowWidth"), puts("float"), puti(4), put(&one, 4);
    out.push_back(0);
    const int nBlocks = (H + 15) / 16;
    size_t tableAt = out.size();
    out.resize(out.size() + (size_t)nBlocks * 8);
    std::vector<uint8_t> raw, tmp, comp;
    for (int b = 0; b < nBlocks; b++) {
        uint64_t off = out.size();
        memcpy(&out[tableAt + (size_t)b * 8], &off, 8);
        int y0 = b * 16, nl = std::min(16, H - y0);
        size_t usize = (size_t)nl * W * 6;
        raw.resize(usize);
        uint8_t *q = raw.data();
        for (int l = 0; l < nl; l++)
            for (int ch = 2; ch >= 0; ch--)  // B, G, R
                for (int x = 0; x < W; x++) {
                    uint16_t h = FloatToHalf(rgb[((size_t)(y0 + l) * W + x) * 3 + ch]);
                    memcpy(q, &h, 2);
                    q += 2;
                }
        tmp.resize(usize);
        uint8_t *t1 = tmp.data(), *t2 = tmp.data() + (usize + 1) / 2;
        for (size_t i = 0; i < usize;) {
            *t1++ = raw[i++];
            if (i < usize) *t2++ = raw[i++];
        }
        {
            uint8_t prev = tmp[0];
            for (size_t i = 1; i < usize; i++) {
                uint8_t cur = tmp[i];
                tmp[i] = (uint8_t)(cur - prev + 128);
                prev = cur;
            }
        }
        uLongf cl = compressBound((uLong)usize);
        comp.resize(cl);
        if (compress2(comp.data(), &cl, tmp.data(), (uLong)usize, 6) != Z_OK) throw std::runtime_error("EXR: compress failed");
        puti(y0);
        if (cl < usize) {
            puti((int32_t)cl);
            put(comp.data(), cl);
        } else {
            puti((int32_t)usize);
            put(raw.data(), usize);
        }
    }
    FILE *f = fopen(fn.c_str(), "wb");
    if (!f) throw std::runtime_error("Fail to create file " + fn);
    fwrite(out.data(), 1, out.size(), f);
    fclose(f);
}

// ------------------------------------------------------------------------------------------ PNG
static int Paeth(int a, int b, int c) {
    int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

Image3f ReadPNG(const std::string &fn, bool *is8bit) {
    std::vector<uint8_t> buf = ReadFile(fn);
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    if (buf.size() < 8 || memcmp(buf.data(), sig, 8)) throw std::runtime_error("PNG: bad signature " + fn);
    size_t p = 8;
    uint32_t W = 0, H = 0;
    int depth = 0, ctype = 0, interlace = 0;
    std::vector<uint8_t> idat, plte;
    auto be32 = [&](size_t at) { return ((uint32_t)buf[at] << 24) | ((uint32_t)buf[at + 1] << 16) | ((uint32_t)buf[at + 2] << 8) | buf[at + 3]; };
    while (p + 12 <= buf.size()) {
        uint32_t len = be32(p);
        std::string type((const char *)&buf[p + 4], 4);
        const uint8_t *d = &buf[p + 8];
        if (p + 12 + len > buf.size()) throw std::runtime_error("PNG: truncated");
        if (type == "IHDR") {
            W = be32(p + 8), H = be32(p + 12);
            depth = d[8], ctype = d[9], interlace = d[12];
        } else if (type == "PLTE")
            plte.assign(d, d + len);
        else if (type == "IDAT")
            idat.insert(idat.end(), d, d + len);
        else if (type == "IEND")
            break;
        p += 12 + len;
    }
    if (interlace) throw std::runtime_error("PNG: interlaced files unsupported");
    int nch = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : 4;
    int bpp = std::max(1, nch * depth / 8);
    size_t stride = ((size_t)W * nch * depth + 7) / 8;
    std::vector<uint8_t> raw((stride + 1) * H);
    uLongf dl = (uLongf)raw.size();
    if (uncompress(raw.data(), &dl, idat.data(), (uLong)idat.size()) != Z_OK) throw std::runtime_error("PNG: zlib error");
    std::vector<uint8_t> pix(stride * H);
    for (uint32_t y = 0; y < H; y++) {
        const uint8_t *in = &raw[(stride + 1) * y];
        uint8_t *cur = &pix[stride * y];
        const uint8_t *up = y ? &pix[stride * (y - 1)] : nullptr;
        int ft = in[0];
        for (size_t i = 0; i < stride; i++) {
            int a = i >= (size_t)bpp ? cur[i - bpp] : 0, b = up ? up[i] : 0, c = (up && i >= (size_t)bpp) ? up[i - bpp] : 0;
            int x = in[1 + i];
            switch (ft) {
                case 0: break;
                case 1: x += a; break;
                case 2: x += b; break;
                case 3: x += (a + b) / 2; break;
                case 4: x += Paeth(a, b, c); break;
                default: throw std::runtime_error("PNG: bad filter");
            }
            cur[i] = (uint8_t)x;
        }
    }
    Image3f img;
    img.width = (int)W, img.height = (int)H;
    img.data.resize((size_t)W * H * 3);
    if (is8bit) *is8bit = depth <= 8;
    auto sample = [&](uint32_t y, size_t idx) -> float {  // idx-th sample of row y, normalised
        const uint8_t *row = &pix[stride * y];
        if (depth == 8) return row[idx] / 255.f;
        if (depth == 16) return ((row[idx * 2] << 8) | row[idx * 2 + 1]) / 65535.f;
        size_t bit = idx * depth;
        int v = (row[bit / 8] >> (8 - depth - (bit % 8))) & ((1 << depth) - 1);
        return ctype == 3 ? (float)v : v / (float)((1 << depth) - 1);
    };
    for (uint32_t y = 0; y < H; y++)
        for (uint32_t x = 0; x < W; x++) {
            float *o = &img.data[((size_t)y * W + x) * 3];
            if (ctype == 3) {
                int v = depth == 8 ? pix[stride * y + x] : (int)sample(y, x);
                for (int k = 0; k < 3; k++) o[k] = (size_t)(v * 3 + k) < plte.size() ? plte[v * 3 + k] / 255.f : 0.f;
            } else if (nch <= 2) {
                o[0] = o[1] = o[2] = sample(y, (size_t)x * nch);
            } else
                for (int k = 0; k < 3; k++) o[k] = sample(y, (size_t)x * nch + k);
        }
    return img;
}

Image3f ReadImage(const std::string &fn, bool *is8bit) {
    std::string ext = fn.size() >= 4 ? fn.substr(fn.size() - 4) : "";
    std::transform(ext.begin(), ext.end(), ext.begin(), ::tolower);
    if (is8bit) *is8bit = false;
    if (ext == ".exr") return ReadEXR(fn);
    if (ext == ".png") return ReadPNG(fn, is8bit);
    std::string ext5 = fn.size() >= 5 ? fn.substr(fn.size() - 5) : "";
    std::transform(ext5.begin(), ext5.end(), ext5.begin(), ::tolower);
    if (ext == ".jpg" || ext5 == ".jpeg") {
        if (is8bit) *is8bit = true;  // 8-bit file: gamma 2.2 in the texture lookup (bitmaptexture.h:135-144)
        return ReadJPEG(fn);
    }
    throw std::runtime_error("Unsupported image format (EXR, PNG and JPEG are implemented): " + fn);
}

}  // namespace lmc
