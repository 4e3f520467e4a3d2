// JPEG decoder for bitmap textures (/root/reference/src/parsescene.cpp:414-432 reads them through OpenImageIO, which hands
// 8-bit JFIF files to libjpeg).  Written against ITU T.81: baseline and progressive Huffman-coded DCT, 8-bit samples, 1 or 3
// components, any sampling factors up to 2x2 chroma subsampling, restart intervals.  The reconstruction follows libjpeg's
// default decompression path so that texels match what the reference sees: the "slow" integer inverse DCT (Loeffler,
// Ligtenberg, Moschytz; 13-bit constants, 2 extra bits after the first pass), triangle-filter ("fancy") chroma upsampling
// and the 16-bit fixed-point YCbCr -> RGB conversion.  tests/test_host.py compares the decoder with Pillow (libjpeg-turbo)
// on every JPEG of the shipped veach-door scene.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <stdexcept>

#include "imageio.h"

namespace lmc {
namespace {

const int kZigzag[64 + 16] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13,
                              6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31,
                              39, 46, 53, 60, 61, 54, 47, 55, 62, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63};

struct Huff {
    bool present = false;
    uint8_t bits[17] = {0};
    uint8_t vals[256] = {0};
    int mincode[17], maxcode[18], valptr[17];
    uint8_t lookNbits[256], lookSym[256];  // 8-bit look-ahead
    void Build() {
        int code = 0, k = 0;
        for (int l = 1; l <= 16; l++) {
            valptr[l] = k;
            mincode[l] = code;
            code += bits[l];
            k += bits[l];
            maxcode[l] = bits[l] ? code - 1 : -1;
            code <<= 1;
        }
        maxcode[17] = 0x7fffffff;
        memset(lookNbits, 0, sizeof(lookNbits));
        int p = 0;
        code = 0;
        for (int l = 1; l <= 8; l++) {
            for (int i = 0; i < bits[l]; i++, p++, code++) {
                int look = code << (8 - l);
                for (int c = 0; c < (1 << (8 - l)); c++) lookNbits[look + c] = (uint8_t)l, lookSym[look + c] = vals[p];
            }
            code <<= 1;
        }
    }
};

struct Component {
    int id = 0, h = 1, v = 1, tq = 0;
    int td = 0, ta = 0;            // tables of the current scan
    int blocksW = 0, blocksH = 0;  // padded to whole MCUs
    int dcPred = 0;
    std::vector<int16_t> coef;  // blocksW * blocksH * 64, natural order
    std::vector<uint8_t> plane; // blocksW*8 x blocksH*8 samples
};

struct Decoder {
    const uint8_t *d;
    size_t n, pos = 0;
    int W = 0, H = 0, nComp = 0;
    bool progressive = false;
    Component comp[4];
    uint16_t qt[4][64];
    Huff dcTab[4], acTab[4];
    int restartInterval = 0;
    int hmax = 1, vmax = 1, mcuW = 0, mcuH = 0;
    bool adobe = false;
    int adobeTransform = -1;
    // entropy decoder state
    uint32_t bitBuf = 0;
    int bitCnt = 0;
    int eobrun = 0;
    bool hitMarker = false;

    [[noreturn]] void Fail(const char *m) { throw std::runtime_error(std::string("JPEG: ") + m); }
    int U8() {
        if (pos >= n) Fail("truncated file");
        return d[pos++];
    }
    int U16() {
        int a = U8();
        return (a << 8) | U8();
    }

    // ---- bit reader over the entropy-coded segment (0xFF00 stuffing; a marker ends the data: further bits read as 0)
    void FillBits() {
        while (bitCnt <= 24) {
            int b = 0;
            if (!hitMarker && pos < n) {
                b = d[pos];
                if (b == 0xFF) {
                    int b2 = pos + 1 < n ? d[pos + 1] : 0xD9;
                    if (b2 == 0) pos += 2;
                    else {
                        hitMarker = true;
                        b = 0;
                    }
                } else
                    pos++;
            }
            bitBuf |= (uint32_t)b << (24 - bitCnt);
            bitCnt += 8;
        }
    }
    int GetBits(int nb) {
        if (nb == 0) return 0;
        if (bitCnt < nb) FillBits();
        int v = (int)(bitBuf >> (32 - nb));
        bitBuf <<= nb;
        bitCnt -= nb;
        return v;
    }
    int GetBit() { return GetBits(1); }
    int DecodeSym(const Huff &h) {
        if (bitCnt < 16) FillBits();
        int look = (int)(bitBuf >> 24);
        int nb = h.lookNbits[look];
        if (nb) {
            bitBuf <<= nb;
            bitCnt -= nb;
            return h.lookSym[look];
        }
        int code = (int)(bitBuf >> 23);  // 9 bits
        int l = 9;
        while (l <= 16 && code > h.maxcode[l]) {
            l++;
            code = (int)(bitBuf >> (32 - l));
        }
        if (l > 16) Fail("bad Huffman code");
        bitBuf <<= l;
        bitCnt -= l;
        return h.vals[h.valptr[l] + code - h.mincode[l]];
    }
    static int Extend(int v, int t) { return v < (1 << (t - 1)) ? v - (1 << t) + 1 : v; }
    void ResetEntropy() {
        bitBuf = 0, bitCnt = 0, eobrun = 0, hitMarker = false;
        for (int i = 0; i < nComp; i++) comp[i].dcPred = 0;
    }
    void ProcessRestart() {
        // skip to the RSTn marker
        bitBuf = 0, bitCnt = 0;
        while (pos + 1 < n && !(d[pos] == 0xFF && d[pos + 1] >= 0xD0 && d[pos + 1] <= 0xD7)) pos++;
        if (pos + 1 < n) pos += 2;
        eobrun = 0, hitMarker = false;
        for (int i = 0; i < nComp; i++) comp[i].dcPred = 0;
    }

    // ---- block decoders; coefficients are stored un-dequantised in natural order
    void BaselineBlock(Component &c, int16_t *blk) {
        const Huff &dc = dcTab[c.td], &ac = acTab[c.ta];
        int t = DecodeSym(dc);
        if (t > 11) Fail("bad DC category");  // 8-bit samples: at most 11 magnitude bits (shifts below stay defined)
        int diff = t ? Extend(GetBits(t), t) : 0;
        c.dcPred += diff;
        blk[0] = (int16_t)c.dcPred;
        for (int k = 1; k < 64;) {
            int rs = DecodeSym(ac), r = rs >> 4, s = rs & 15;
            if (s == 0) {
                if (r == 15) {
                    k += 16;
                    continue;
                }
                break;
            }
            k += r;
            if (k > 63) Fail("bad AC run");
            if (s > 10) Fail("bad AC size");
            blk[kZigzag[k]] = (int16_t)Extend(GetBits(s), s);
            k++;
        }
    }
    void DcFirst(Component &c, int16_t *blk, int Al) {
        int t = DecodeSym(dcTab[c.td]);
        if (t > 11) Fail("bad DC category");
        int diff = t ? Extend(GetBits(t), t) : 0;
        c.dcPred += diff;
        blk[0] = (int16_t)(c.dcPred * (1 << Al));
    }
    void DcRefine(int16_t *blk, int Al) {
        if (GetBit()) blk[0] |= (int16_t)(1 << Al);
    }
    void AcFirst(Component &c, int16_t *blk, int Ss, int Se, int Al) {
        if (eobrun > 0) {
            eobrun--;
            return;
        }
        const Huff &ac = acTab[c.ta];
        for (int k = Ss; k <= Se;) {
            int rs = DecodeSym(ac), r = rs >> 4, s = rs & 15;
            if (s == 0) {
                if (r < 15) {
                    eobrun = (1 << r) - 1;
                    if (r) eobrun += GetBits(r);
                    break;
                }
                k += 16;
                continue;
            }
            k += r;
            if (k > Se) Fail("bad AC run");
            if (s > 10) Fail("bad AC size");
            blk[kZigzag[k]] = (int16_t)(Extend(GetBits(s), s) * (1 << Al));
            k++;
        }
    }
    void AcRefine(Component &c, int16_t *blk, int Ss, int Se, int Al) {
        const int p1 = 1 << Al, m1 = -1 * (1 << Al);
        const Huff &ac = acTab[c.ta];
        int k = Ss;
        if (eobrun <= 0) {
            for (; k <= Se;) {
                int rs = DecodeSym(ac), r = rs >> 4, s = rs & 15;
                int val = 0;
                if (s == 0) {
                    if (r < 15) {
                        eobrun = (1 << r);
                        if (r) eobrun += GetBits(r);
                        break;  // the rest of the block is handled by the EOB logic below
                    }
                } else {
                    if (s != 1) Fail("bad refinement code");
                    val = GetBit() ? p1 : m1;
                }
                // advance over already-nonzero coefficients (refining them) and r zero-history coefficients
                for (; k <= Se; k++) {
                    int16_t &co = blk[kZigzag[k]];
                    if (co != 0) {
                        if (GetBit())
                            if ((co & p1) == 0) co = (int16_t)(co >= 0 ? co + p1 : co + m1);
                    } else {
                        if (--r < 0) break;
                    }
                }
                if (val && k <= Se) blk[kZigzag[k]] = (int16_t)val;
                k++;
            }
        }
        if (eobrun > 0) {
            for (; k <= Se; k++) {
                int16_t &co = blk[kZigzag[k]];
                if (co != 0)
                    if (GetBit())
                        if ((co & p1) == 0) co = (int16_t)(co >= 0 ? co + p1 : co + m1);
            }
            eobrun--;
        }
    }

    // ---- one scan (SOS ... entropy data)
    void Scan() {
        int len = U16();
        int ns = U8();
        if (len != 6 + 2 * ns || ns < 1 || ns > 4) Fail("bad SOS");
        int idx[4];
        for (int i = 0; i < ns; i++) {
            int cid = U8(), tt = U8();
            int ci = -1;
            for (int k = 0; k < nComp; k++)
                if (comp[k].id == cid) ci = k;
            if (ci < 0) Fail("SOS: unknown component");
            idx[i] = ci;
            comp[ci].td = tt >> 4, comp[ci].ta = tt & 15;
            if (comp[ci].td > 3 || comp[ci].ta > 3) Fail("bad table index");
        }
        int Ss = U8(), Se = U8(), AhAl = U8(), Ah = AhAl >> 4, Al = AhAl & 15;
        if (!progressive) Ss = 0, Se = 63, Ah = Al = 0;
        // the spectral band indexes the 64-entry zig-zag table and the coefficient block; Al is a shift count
        if (Ss > Se || Se > 63 || Al > 13 || (Ss == 0 && progressive && Se != 0)) Fail("bad spectral selection / successive approximation");
        for (int i = 0; i < ns; i++) {  // a scan may only use Huffman tables that a DHT segment has defined (mincode / maxcode / valptr)
            const Component &c = comp[idx[i]];
            const bool needDc = !progressive || Ss == 0, needAc = !progressive || Ss > 0;
            if (needDc && !(progressive && Ah != 0) && !dcTab[c.td].present) Fail("scan uses an undefined DC Huffman table");
            if (needAc && !acTab[c.ta].present) Fail("scan uses an undefined AC Huffman table");
        }
        ResetEntropy();
        int restartsLeft = restartInterval;
        auto block = [&](Component &c, int bx, int by) {
            int16_t *blk = &c.coef[((size_t)by * c.blocksW + bx) * 64];
            if (!progressive) BaselineBlock(c, blk);
            else if (Ss == 0) {
                if (Ah == 0) DcFirst(c, blk, Al);
                else
                    DcRefine(blk, Al);
            } else {
                if (Ah == 0) AcFirst(c, blk, Ss, Se, Al);
                else
                    AcRefine(c, blk, Ss, Se, Al);
            }
        };
        if (ns == 1) {  // non-interleaved: the component's own blocks, ceil(size/8) per dimension
            Component &c = comp[idx[0]];
            const int cw = (W * c.h + hmax - 1) / hmax, ch = (H * c.v + vmax - 1) / vmax;
            const int bw = (cw + 7) / 8, bh = (ch + 7) / 8;
            for (int by = 0; by < bh; by++)
                for (int bx = 0; bx < bw; bx++) {
                    if (restartInterval && restartsLeft == 0) {
                        ProcessRestart();
                        restartsLeft = restartInterval;
                    }
                    block(c, bx, by);
                    restartsLeft--;
                }
        } else {
            for (int my = 0; my < mcuH; my++)
                for (int mx = 0; mx < mcuW; mx++) {
                    if (restartInterval && restartsLeft == 0) {
                        ProcessRestart();
                        restartsLeft = restartInterval;
                    }
                    for (int i = 0; i < ns; i++) {
                        Component &c = comp[idx[i]];
                        for (int v = 0; v < c.v; v++)
                            for (int h = 0; h < c.h; h++) block(c, mx * c.h + h, my * c.v + v);
                    }
                    restartsLeft--;
                }
        }
        // position `pos` at the marker that ended the scan
        if (!hitMarker)
            while (pos + 1 < n && !(d[pos] == 0xFF && d[pos + 1] != 0 && !(d[pos + 1] >= 0xD0 && d[pos + 1] <= 0xD7))) pos++;
    }

    // ---- libjpeg's jidctint.c (ISLOW): constants scaled by 2^13, first pass keeps 2 extra bits
    static inline int Desc(int64_t x, int n) { return (int)((x + ((int64_t)1 << (n - 1))) >> n); }
    void Idct(const int16_t *in, const uint16_t *q, uint8_t *out, int stride) {
        const int CB = 13, P1 = 2;
        const int F0298 = 2446, F0390 = 3196, F0541 = 4433, F0765 = 6270, F0899 = 7373, F1175 = 9633, F1501 = 12299, F1847 = 15137, F1961 = 16069,
                  F2053 = 16819, F2562 = 20995, F3072 = 25172;
        int ws[64];
        for (int c = 0; c < 8; c++) {
            const int16_t *ip = in + c;
            const uint16_t *qp = q + c;
            int *wp = ws + c;
            if (ip[8] == 0 && ip[16] == 0 && ip[24] == 0 && ip[32] == 0 && ip[40] == 0 && ip[48] == 0 && ip[56] == 0) {
                int dc = (ip[0] * qp[0]) * (1 << P1);
                for (int r = 0; r < 8; r++) wp[8 * r] = dc;
                continue;
            }
            int64_t z2 = ip[16] * qp[16], z3 = ip[48] * qp[48];
            int64_t z1 = (z2 + z3) * F0541;
            int64_t tmp2 = z1 + z3 * (-F1847), tmp3 = z1 + z2 * F0765;
            z2 = ip[0] * qp[0], z3 = ip[32] * qp[32];
            int64_t tmp0 = (z2 + z3) * (1 << CB), tmp1 = (z2 - z3) * (1 << CB);
            int64_t tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
            tmp0 = ip[56] * qp[56], tmp1 = ip[40] * qp[40], tmp2 = ip[24] * qp[24], tmp3 = ip[8] * qp[8];
            z1 = tmp0 + tmp3, z2 = tmp1 + tmp2, z3 = tmp0 + tmp2;
            int64_t z4 = tmp1 + tmp3, z5 = (z3 + z4) * F1175;
            tmp0 *= F0298, tmp1 *= F2053, tmp2 *= F3072, tmp3 *= F1501;
            z1 *= -F0899, z2 *= -F2562, z3 *= -F1961, z4 *= -F0390;
            z3 += z5, z4 += z5;
            tmp0 += z1 + z3, tmp1 += z2 + z4, tmp2 += z2 + z3, tmp3 += z1 + z4;
            wp[0] = Desc(tmp10 + tmp3, CB - P1), wp[56] = Desc(tmp10 - tmp3, CB - P1);
            wp[8] = Desc(tmp11 + tmp2, CB - P1), wp[48] = Desc(tmp11 - tmp2, CB - P1);
            wp[16] = Desc(tmp12 + tmp1, CB - P1), wp[40] = Desc(tmp12 - tmp1, CB - P1);
            wp[24] = Desc(tmp13 + tmp0, CB - P1), wp[32] = Desc(tmp13 - tmp0, CB - P1);
        }
        auto clamp = [](int v) { return (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v); };
        for (int r = 0; r < 8; r++) {
            const int *wp = ws + 8 * r;
            uint8_t *op = out + (size_t)r * stride;
            int64_t z2 = wp[2], z3 = wp[6];
            int64_t z1 = (z2 + z3) * F0541;
            int64_t tmp2 = z1 + z3 * (-F1847), tmp3 = z1 + z2 * F0765;
            int64_t tmp0 = ((int64_t)wp[0] + wp[4]) * (1 << CB), tmp1 = ((int64_t)wp[0] - wp[4]) * (1 << CB);
            int64_t tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
            tmp0 = wp[7], tmp1 = wp[5], tmp2 = wp[3], tmp3 = wp[1];
            z1 = tmp0 + tmp3, z2 = tmp1 + tmp2, z3 = tmp0 + tmp2;
            int64_t z4 = tmp1 + tmp3, z5 = (z3 + z4) * F1175;
            tmp0 *= F0298, tmp1 *= F2053, tmp2 *= F3072, tmp3 *= F1501;
            z1 *= -F0899, z2 *= -F2562, z3 *= -F1961, z4 *= -F0390;
            z3 += z5, z4 += z5;
            tmp0 += z1 + z3, tmp1 += z2 + z4, tmp2 += z2 + z3, tmp3 += z1 + z4;
            const int S = CB + P1 + 3;
            op[0] = clamp(Desc(tmp10 + tmp3, S) + 128), op[7] = clamp(Desc(tmp10 - tmp3, S) + 128);
            op[1] = clamp(Desc(tmp11 + tmp2, S) + 128), op[6] = clamp(Desc(tmp11 - tmp2, S) + 128);
            op[2] = clamp(Desc(tmp12 + tmp1, S) + 128), op[5] = clamp(Desc(tmp12 - tmp1, S) + 128);
            op[3] = clamp(Desc(tmp13 + tmp0, S) + 128), op[4] = clamp(Desc(tmp13 - tmp0, S) + 128);
        }
    }

    void Frame(int marker) {
        int len = U16();
        if (U8() != 8) Fail("only 8-bit samples are supported");
        H = U16(), W = U16(), nComp = U8();
        if (len != 8 + 3 * nComp || W <= 0 || H <= 0 || (nComp != 1 && nComp != 3)) Fail("unsupported frame header");
        progressive = marker == 0xC2;
        for (int i = 0; i < nComp; i++) {
            comp[i].id = U8();
            int hv = U8();
            comp[i].h = hv >> 4, comp[i].v = hv & 15, comp[i].tq = U8();
            if (comp[i].h < 1 || comp[i].h > 2 || comp[i].v < 1 || comp[i].v > 2 || comp[i].tq > 3) Fail("unsupported sampling factors");
            hmax = std::max(hmax, comp[i].h), vmax = std::max(vmax, comp[i].v);
        }
        mcuW = (W + 8 * hmax - 1) / (8 * hmax), mcuH = (H + 8 * vmax - 1) / (8 * vmax);
        for (int i = 0; i < nComp; i++) {
            comp[i].blocksW = mcuW * comp[i].h, comp[i].blocksH = mcuH * comp[i].v;
            comp[i].coef.assign((size_t)comp[i].blocksW * comp[i].blocksH * 64, 0);
        }
    }

    // libjpeg's "fancy" upsampling of one chroma plane to full resolution (jdsample.c: h2v1 / h2v2 triangle filters);
    // the input plane is the component's downsampled size (edge samples replicated beyond it, like libjpeg's context rows)
    std::vector<uint8_t> Upsample(const Component &c) {
        const int cw = (W * c.h + hmax - 1) / hmax, ch = (H * c.v + vmax - 1) / vmax;
        const int stride = c.blocksW * 8;
        std::vector<uint8_t> out((size_t)W * H);
        const int hs = hmax / c.h, vs = vmax / c.v;
        auto in = [&](int x, int y) -> int {
            x = x < 0 ? 0 : (x >= cw ? cw - 1 : x);
            y = y < 0 ? 0 : (y >= ch ? ch - 1 : y);
            return c.plane[(size_t)y * stride + x];
        };
        if (hs == 1 && vs == 1) {
            for (int y = 0; y < H; y++)
                for (int x = 0; x < W; x++) out[(size_t)y * W + x] = (uint8_t)in(x, y);
        } else if (hs == 2 && vs == 1) {
            for (int y = 0; y < H; y++)
                for (int x = 0; x < W; x++) {
                    const int xi = x >> 1;
                    int v;
                    if (cw == 1) v = in(0, y);
                    else if (x == 0) v = in(0, y);
                    else if (x == 2 * cw - 1) v = in(cw - 1, y);
                    else if (x & 1) v = (3 * in(xi, y) + in(xi + 1, y) + 2) >> 2;
                    else
                        v = (3 * in(xi, y) + in(xi - 1, y) + 1) >> 2;
                    out[(size_t)y * W + x] = (uint8_t)v;
                }
        } else if (hs == 2 && vs == 2) {
            for (int y = 0; y < H; y++) {
                const int yi = y >> 1, yn = (y & 1) ? yi + 1 : yi - 1;  // nearer row yi, further row yn
                for (int x = 0; x < W; x++) {
                    const int xi = x >> 1;
                    auto colsum = [&](int xx) { return 3 * in(xx, yi) + in(xx, yn); };
                    int v;
                    if (cw == 1) v = (colsum(0) * 4 + 8) >> 4;
                    else if (x == 0) v = (colsum(0) * 4 + 8) >> 4;
                    else if (x == 2 * cw - 1) v = (colsum(cw - 1) * 4 + 7) >> 4;
                    else if (x & 1) v = (3 * colsum(xi) + colsum(xi + 1) + 7) >> 4;
                    else
                        v = (3 * colsum(xi) + colsum(xi - 1) + 8) >> 4;
                    out[(size_t)y * W + x] = (uint8_t)v;
                }
            }
        } else if (hs == 1 && vs == 2) {  // libjpeg has no fancy h1v2 in its classic code path: replicate rows
            for (int y = 0; y < H; y++)
                for (int x = 0; x < W; x++) out[(size_t)y * W + x] = (uint8_t)in(x, y >> 1);
        } else
            Fail("unsupported subsampling");
        return out;
    }

    Image3f Decode() {
        if (n < 4 || d[0] != 0xFF || d[1] != 0xD8) Fail("not a JPEG file");
        pos = 2;
        bool haveFrame = false, done = false;
        memset(qt, 0, sizeof(qt));
        while (!done) {
            // next marker
            while (pos < n && d[pos] != 0xFF) pos++;
            while (pos < n && d[pos] == 0xFF) pos++;
            if (pos >= n) break;
            int m = d[pos++];
            switch (m) {
                case 0xD9: done = true; break;
                case 0xDB: {  // DQT
                    int len = U16() - 2;
                    while (len > 0) {
                        int pt = U8(), prec = pt >> 4, id = pt & 15;
                        if (id > 3) Fail("bad DQT");
                        for (int i = 0; i < 64; i++) qt[id][kZigzag[i]] = (uint16_t)(prec ? U16() : U8());
                        len -= 1 + 64 * (prec ? 2 : 1);
                    }
                    break;
                }
                case 0xC4: {  // DHT
                    int len = U16() - 2;
                    while (len > 0) {
                        int tc = U8(), cls = tc >> 4, id = tc & 15;
                        if (id > 3 || cls > 1) Fail("bad DHT");
                        Huff &h = cls ? acTab[id] : dcTab[id];
                        int total = 0;
                        h.bits[0] = 0;
                        for (int i = 1; i <= 16; i++) h.bits[i] = (uint8_t)U8(), total += h.bits[i];
                        if (total > 256) Fail("bad DHT");
                        for (int i = 0; i < total; i++) h.vals[i] = (uint8_t)U8();
                        h.present = true;
                        h.Build();
                        len -= 17 + total;
                    }
                    break;
                }
                case 0xC0: case 0xC1: case 0xC2:
                    Frame(m);
                    haveFrame = true;
                    break;
                case 0xC3: case 0xC5: case 0xC6: case 0xC7: case 0xC9: case 0xCA: case 0xCB: case 0xCD: case 0xCE: case 0xCF:
                    Fail("unsupported JPEG process (lossless / hierarchical / arithmetic)");
                case 0xDD: {
                    U16();
                    restartInterval = U16();
                    break;
                }
                case 0xDA:
                    if (!haveFrame) Fail("SOS before SOF");
                    Scan();
                    break;
                case 0xEE: {  // Adobe
                    int len = U16();
                    if (len < 2 || pos + (size_t)(len - 2) > n) Fail("bad segment length");
                    size_t end = pos + len - 2;
                    if (len >= 14 && pos + 12 <= n && memcmp(d + pos, "Adobe", 5) == 0) adobe = true, adobeTransform = d[pos + 11];
                    pos = end;
                    break;
                }
                default:
                    if (m >= 0xD0 && m <= 0xD7) break;  // stray RSTn
                    if (m == 0x01 || m == 0x00) break;
                    {
                        int len = U16();
                        if (len < 2 || pos + (size_t)(len - 2) > n) Fail("bad segment length");
                        pos += len - 2;
                    }
            }
        }
        if (!haveFrame) Fail("no frame");
        for (int i = 0; i < nComp; i++) {
            Component &c = comp[i];
            c.plane.assign((size_t)c.blocksW * 8 * c.blocksH * 8, 0);
            for (int by = 0; by < c.blocksH; by++)
                for (int bx = 0; bx < c.blocksW; bx++)
                    Idct(&c.coef[((size_t)by * c.blocksW + bx) * 64], qt[c.tq], &c.plane[((size_t)by * 8) * (c.blocksW * 8) + bx * 8], c.blocksW * 8);
        }
        Image3f img;
        img.width = W, img.height = H;
        img.data.resize((size_t)W * H * 3);
        if (nComp == 1) {
            std::vector<uint8_t> y = Upsample(comp[0]);
            for (size_t i = 0; i < (size_t)W * H; i++) img.data[3 * i] = img.data[3 * i + 1] = img.data[3 * i + 2] = y[i] / 255.f;
            return img;
        }
        std::vector<uint8_t> p0 = Upsample(comp[0]), p1 = Upsample(comp[1]), p2 = Upsample(comp[2]);
        const bool ycc = adobe ? adobeTransform != 0 : !(comp[0].id == 'R' && comp[1].id == 'G' && comp[2].id == 'B');
        auto clamp = [](int v) { return v < 0 ? 0 : v > 255 ? 255 : v; };
        for (size_t i = 0; i < (size_t)W * H; i++) {
            int r, g, b;
            if (ycc) {  // jdcolor.c: SCALEBITS 16, ONE_HALF
                const int y = p0[i], cb = p1[i] - 128, cr = p2[i] - 128;
                const int crR = (int)((91881LL * cr + 32768) >> 16);   // FIX(1.40200)
                const int cbB = (int)((116130LL * cb + 32768) >> 16);  // FIX(1.77200)
                const int crG = -46802 * cr, cbG = -22554 * cb + 32768; // FIX(0.71414), FIX(0.34414)
                r = clamp(y + crR), g = clamp(y + (int)((cbG + crG) >> 16)), b = clamp(y + cbB);
            } else
                r = p0[i], g = p1[i], b = p2[i];
            img.data[3 * i] = r / 255.f, img.data[3 * i + 1] = g / 255.f, img.data[3 * i + 2] = b / 255.f;
        }
        return img;
    }
};

}  // namespace

Image3f ReadJPEG(const std::string &fn) {
    FILE *f = fopen(fn.c_str(), "rb");
    if (!f) throw std::runtime_error("Cannot open image file: " + fn);
    std::vector<uint8_t> buf;
    uint8_t tmp[65536];
    size_t r;
    while ((r = fread(tmp, 1, sizeof(tmp), f)) > 0) buf.insert(buf.end(), tmp, tmp + r);
    fclose(f);
    Decoder dec;
    dec.d = buf.data(), dec.n = buf.size();
    return dec.Decode();
}

}  // namespace lmc
