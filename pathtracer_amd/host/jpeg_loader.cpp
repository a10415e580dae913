// jpeg_loader.cpp — baseline and progressive JPEG (ITU-T T.81, 8-bit, Huffman) to RGBA8, for the textures that
// Model::createTextures hands to stbi_load(..., STBI_rgb_alpha) in the reference (/root/reference/src/Model.cpp:350).
// Written from the standard; everything the standard leaves to the decoder is done the way the reference's
// decoder (external/stb/stb_image.h v2.16, JPEG part :1650-3705) does it, because the texels must come out
// identical (checked against its output on the fixtures in tests/golden/obj/textures):
//   * inverse DCT: the 13-bit fixed-point "islow" factorisation with 2 extra bits kept between the passes,
//     +128 level shift folded into the rounding constant, column pass short-cut for all-zero AC columns;
//   * chroma upsampling: triangle filter (3/4, 1/4) for 2x horizontal / 2x vertical / 2x2, replication otherwise,
//     with the row pairing of its line state machine (first and last rows pair with themselves);
//   * YCbCr -> RGB: 20-bit fixed point with the coefficients rounded to 12 bits first and the Cb->G product
//     truncated to its upper 16 bits; RGB when the component ids are 'R','G','B' or Adobe transform 0 without
//     JFIF; 4 components: CMYK (Adobe transform 0), YCCK (2), else the first three as YCbCr;
//   * entropy-coded data that ends early (marker or end of file) reads as zero bits; a restart interval that
//     ends without an RSTn marker ends the scan; bytes between a scan and the next marker are skipped;
//   * not supported, like there: 12-bit samples, arithmetic coding, lossless / hierarchical frames.
#include <cstring>

#include "image_loader.h"

namespace prt_host {

namespace {

const uint8_t kZigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                             41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                             30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
inline int natural_of(int k) { return kZigzag[k < 64 ? k : 63]; }  // a run past the end lands on the last one

struct ByteSource
{
    const uint8_t* p;
    size_t         n, pos = 0;
    int  u8() { return pos < n ? p[pos++] : 0; }  // reads past the end yield 0
    int  u16()
    {
        const int hi = u8();
        return (hi << 8) | u8();
    }
    void skip(int k)
    {
        if(k > 0) pos = (pos + (size_t)k < n) ? pos + (size_t)k : n;
    }
    bool eof() const { return pos >= n; }
};

// canonical Huffman table (T.81 Annex C / F.2.2.3)
struct HuffTable
{
    bool     defined = false;
    uint8_t  value[256];
    int32_t  mincode[17], maxcode[17], valptr[17];  // maxcode = -1: no code of that length
    bool     build(const int count[16])
    {
        int code = 0, k = 0;
        for(int len = 1; len <= 16; ++len)
        {
            valptr[len]  = k;
            mincode[len] = code;
            if(count[len - 1])
            {
                code += count[len - 1];
                k += count[len - 1];
                if(code - 1 >= (1 << len)) return false;  // more codes than the length can hold
                maxcode[len] = code - 1;
            }
            else
                maxcode[len] = -1;
            code <<= 1;
        }
        defined = true;
        return true;
    }
};

struct Component
{
    int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0;
    int dcPred = 0;
    int w = 0, hgt = 0;      // samples that belong to the image
    int stride = 0, rows = 0;  // plane size: whole MCUs
    int blocksW = 0;
    std::vector<uint8_t> plane;
    std::vector<int16_t> coeff;  // progressive only: 64 per block, natural order
};

struct Decoder
{
    ByteSource s;
    HuffTable  dc[4], ac[4];
    uint16_t   quant[4][64];  // natural order
    Component  comp[4];
    int        ncomp = 0, width = 0, height = 0;
    int        hmax = 1, vmax = 1, mcuX = 0, mcuY = 0;
    bool       progressive = false, jfif = false;
    int        adobeTransform = -1, rgbIds = 0;
    int        restartInterval = 0;
    // scan state
    int      scanN = 0, order[4] = {0, 0, 0, 0};
    int      ss = 0, se = 63, ah = 0, al = 0;
    uint32_t bitBuf = 0;
    int      bitCnt = 0;
    int      pendingMarker = 0xff;  // 0xff: none
    bool     noMore = false;
    int      eobRun = 0, todo = 0;

    // ---- bit reader ----
    void fill()
    {
        do
        {
            int b = noMore ? 0 : s.u8();
            if(b == 0xff)
            {
                int c = s.u8();
                while(c == 0xff) c = s.u8();
                if(c != 0)
                {
                    pendingMarker = c;
                    noMore        = true;
                    return;
                }
            }
            bitBuf |= (uint32_t)b << (24 - bitCnt);
            bitCnt += 8;
        } while(bitCnt <= 24);
    }
    int getBits(int n)  // n in 0..16
    {
        if(n == 0) return 0;
        if(bitCnt < n) fill();
        const uint32_t v = bitBuf >> (32 - n);
        bitBuf <<= n;
        bitCnt -= n;
        return (int)v;
    }
    int getBit() { return getBits(1); }
    int receiveExtend(int n)
    {
        if(n == 0) return 0;
        if(bitCnt < n) fill();
        const bool     positive = (bitBuf >> 31) != 0;
        const int      v        = getBits(n);
        return positive ? v : v - (1 << n) + 1;
    }
    int decodeSymbol(const HuffTable& t)
    {
        if(bitCnt < 16) fill();
        const uint32_t top = bitBuf >> 16;  // the next 16 bits
        for(int len = 1; len <= 16; ++len)
        {
            const int code = (int)(top >> (16 - len));
            if(t.maxcode[len] >= 0 && code <= t.maxcode[len] && code >= t.mincode[len])
            {
                if(len > bitCnt) return -1;
                bitBuf <<= len;
                bitCnt -= len;
                return t.value[t.valptr[len] + code - t.mincode[len]];
            }
        }
        bitCnt -= 16;
        return -1;
    }
    void resetEntropy()
    {
        bitBuf        = 0;
        bitCnt        = 0;
        noMore        = false;
        pendingMarker = 0xff;
        for(Component& c : comp) c.dcPred = 0;
        todo   = restartInterval ? restartInterval : 0x7fffffff;
        eobRun = 0;
    }

    // ---- markers ----
    int nextMarker()
    {
        if(pendingMarker != 0xff)
        {
            const int m   = pendingMarker;
            pendingMarker = 0xff;
            return m;
        }
        int x = s.u8();
        if(x != 0xff) return 0xff;
        while(x == 0xff) x = s.u8();
        return x;
    }
    bool tablesOrMisc(int m)
    {
        switch(m)
        {
        case 0xff: return false;  // no marker where one must be
        case 0xdd:
            if(s.u16() != 4) return false;
            restartInterval = s.u16();
            return true;
        case 0xdb: {
            int len = s.u16() - 2;
            while(len > 0)
            {
                const int q = s.u8(), prec = q >> 4, t = q & 15;
                if(prec > 1 || t > 3) return false;
                for(int i = 0; i < 64; ++i) quant[t][kZigzag[i]] = (uint16_t)(prec ? s.u16() : s.u8());
                len -= prec ? 129 : 65;
            }
            return len == 0;
        }
        case 0xc4: {
            int len = s.u16() - 2;
            while(len > 0)
            {
                const int q = s.u8(), tc = q >> 4, th = q & 15;
                if(tc > 1 || th > 3) return false;
                int count[16], n = 0;
                for(int i = 0; i < 16; ++i)
                {
                    count[i] = s.u8();
                    n += count[i];
                }
                len -= 17;
                HuffTable& t = tc ? ac[th] : dc[th];
                if(!t.build(count)) return false;
                for(int i = 0; i < n; ++i)
                {
                    const int v = s.u8();
                    if(i < 256) t.value[i] = (uint8_t)v;
                }
                len -= n;
            }
            return len == 0;
        }
        default: break;
        }
        if((m >= 0xe0 && m <= 0xef) || m == 0xfe)
        {
            int len = s.u16();
            if(len < 2) return false;
            len -= 2;
            if(m == 0xe0 && len >= 5)
            {
                static const char tag[5] = {'J', 'F', 'I', 'F', 0};
                bool              ok     = true;
                for(int i = 0; i < 5; ++i)
                    if(s.u8() != (uint8_t)tag[i]) ok = false;
                len -= 5;
                if(ok) jfif = true;
            }
            else if(m == 0xee && len >= 12)
            {
                static const char tag[6] = {'A', 'd', 'o', 'b', 'e', 0};
                bool              ok     = true;
                for(int i = 0; i < 6; ++i)
                    if(s.u8() != (uint8_t)tag[i]) ok = false;
                len -= 6;
                if(ok)
                {
                    s.u8();
                    s.u16();
                    s.u16();
                    adobeTransform = s.u8();
                    len -= 6;
                }
            }
            s.skip(len);
            return true;
        }
        return false;  // unknown marker
    }
    bool frameHeader()
    {
        const int lf = s.u16();
        if(lf < 11) return false;
        if(s.u8() != 8) return false;  // 8-bit samples only
        height = s.u16();
        width  = s.u16();
        if(height == 0 || width == 0) return false;
        ncomp = s.u8();
        if(ncomp != 1 && ncomp != 3 && ncomp != 4) return false;
        if(lf != 8 + 3 * ncomp) return false;
        rgbIds = 0;
        for(int i = 0; i < ncomp; ++i)
        {
            static const char rgb[3] = {'R', 'G', 'B'};
            Component&        c      = comp[i];
            c.id                     = s.u8();
            if(ncomp == 3 && c.id == rgb[i]) ++rgbIds;
            const int q = s.u8();
            c.h         = q >> 4;
            c.v         = q & 15;
            if(c.h < 1 || c.h > 4 || c.v < 1 || c.v > 4) return false;
            c.tq = s.u8();
            if(c.tq > 3) return false;
        }
        if((uint64_t)width * (uint64_t)height * (uint64_t)ncomp > 0x7fffffffull) return false;
        hmax = vmax = 1;
        for(int i = 0; i < ncomp; ++i)
        {
            if(comp[i].h > hmax) hmax = comp[i].h;
            if(comp[i].v > vmax) vmax = comp[i].v;
        }
        mcuX = (width + hmax * 8 - 1) / (hmax * 8);
        mcuY = (height + vmax * 8 - 1) / (vmax * 8);
        for(int i = 0; i < ncomp; ++i)
        {
            Component& c = comp[i];
            c.w          = (width * c.h + hmax - 1) / hmax;
            c.hgt        = (height * c.v + vmax - 1) / vmax;
            c.stride     = mcuX * c.h * 8;
            c.rows       = mcuY * c.v * 8;
            c.blocksW    = c.stride / 8;
            c.plane.assign((size_t)c.stride * c.rows, 0);
            if(progressive) c.coeff.assign((size_t)c.stride * c.rows, 0);
        }
        return true;
    }
    bool scanHeader()
    {
        const int ls = s.u16();
        scanN        = s.u8();
        if(scanN < 1 || scanN > 4 || scanN > ncomp) return false;
        if(ls != 6 + 2 * scanN) return false;
        for(int i = 0; i < scanN; ++i)
        {
            const int id = s.u8(), q = s.u8();
            int       which = 0;
            while(which < ncomp && comp[which].id != id) ++which;
            if(which == ncomp) return false;
            comp[which].td = q >> 4;
            comp[which].ta = q & 15;
            if(comp[which].td > 3 || comp[which].ta > 3) return false;
            order[i] = which;
        }
        ss            = s.u8();
        se            = s.u8();
        const int aa  = s.u8();
        ah            = aa >> 4;
        al            = aa & 15;
        if(progressive)
        {
            if(ss > 63 || se > 63 || ss > se || ah > 13 || al > 13) return false;
        }
        else
        {
            if(ss != 0 || ah != 0 || al != 0) return false;
            se = 63;
        }
        return true;
    }

    // ---- blocks ----
    bool baselineBlock(int16_t blk[64], Component& c)
    {
        const uint16_t* q = quant[c.tq];
        if(bitCnt < 16) fill();
        const int t = decodeSymbol(dc[c.td]);
        if(t < 0) return false;
        memset(blk, 0, 64 * sizeof(int16_t));
        const int diff = t ? receiveExtend(t > 16 ? 16 : t) : 0;
        c.dcPred += diff;
        blk[0] = (int16_t)(c.dcPred * q[0]);
        int k  = 1;
        do
        {
            const int rs = decodeSymbol(ac[c.ta]);
            if(rs < 0) return false;
            const int size = rs & 15, run = rs >> 4;
            if(size == 0)
            {
                if(rs != 0xf0) break;  // end of block
                k += 16;
            }
            else
            {
                k += run;
                const int z = natural_of(k++);
                blk[z]      = (int16_t)(receiveExtend(size) * q[z]);
            }
        } while(k < 64);
        return true;
    }
    bool progressiveDC(int16_t* blk, Component& c)
    {
        if(se != 0) return false;  // a scan holds either DC or AC
        if(bitCnt < 16) fill();
        if(ah == 0)
        {
            memset(blk, 0, 64 * sizeof(int16_t));
            const int t    = decodeSymbol(dc[c.td]);
            const int diff = t > 0 ? receiveExtend(t > 16 ? 16 : t) : 0;
            c.dcPred += diff;
            blk[0] = (int16_t)(c.dcPred * (1 << al));  // (a shift of a negative value is undefined before C++20)
        }
        else if(getBit())
            blk[0] = (int16_t)(blk[0] + (int16_t)(1 << al));
        return true;
    }
    static void refine(int16_t& v, int16_t bit)
    {
        if((v & bit) == 0) v = (int16_t)(v > 0 ? v + bit : v - bit);
    }
    bool progressiveAC(int16_t* blk, Component& c)
    {
        if(ss == 0) return false;
        const HuffTable& t = ac[c.ta];
        if(ah == 0)
        {
            if(eobRun)
            {
                --eobRun;
                return true;
            }
            int k = ss;
            do
            {
                const int rs = decodeSymbol(t);
                if(rs < 0) return false;
                const int size = rs & 15, run = rs >> 4;
                if(size == 0)
                {
                    if(run < 15)
                    {
                        eobRun = 1 << run;
                        if(run) eobRun += getBits(run);
                        --eobRun;
                        break;
                    }
                    k += 16;
                }
                else
                {
                    k += run;
                    blk[natural_of(k++)] = (int16_t)(receiveExtend(size) * (1 << al));
                }
            } while(k <= se);
            return true;
        }
        // refinement pass
        const int16_t bit = (int16_t)(1 << al);
        if(eobRun)
        {
            --eobRun;
            for(int k = ss; k <= se; ++k)
            {
                int16_t& v = blk[natural_of(k)];
                if(v != 0 && getBit()) refine(v, bit);
            }
            return true;
        }
        int k = ss;
        do
        {
            const int rs = decodeSymbol(t);
            if(rs < 0) return false;
            int size = rs & 15, run = rs >> 4;
            int newValue = 0;
            if(size == 0)
            {
                if(run < 15)
                {
                    eobRun = (1 << run) - 1;
                    if(run) eobRun += getBits(run);
                    run = 64;  // the rest of the band only takes correction bits
                }
            }
            else
            {
                if(size != 1) return false;
                newValue = getBit() ? bit : -bit;
            }
            while(k <= se)
            {
                int16_t& v = blk[natural_of(k++)];
                if(v != 0)
                {
                    if(getBit()) refine(v, bit);
                }
                else
                {
                    if(run == 0)
                    {
                        v = (int16_t)newValue;
                        break;
                    }
                    --run;
                }
            }
        } while(k <= se);
        return true;
    }

    // ---- inverse DCT: 8x8, input dequantised, output clamped bytes ----
    struct Pass
    {
        int x0, x1, x2, x3, t0, t1, t2, t3;
    };
    static inline int fx(double c) { return (int)(c * 4096 + 0.5); }
    static Pass       butterfly(int s0, int s1, int s2, int s3, int s4, int s5, int s6, int s7)
    {
        Pass r;
        // even part
        int p1 = (s2 + s6) * fx(0.5411961f);
        int e2 = p1 + s6 * fx(-1.847759065f);
        int e3 = p1 + s2 * fx(0.765366865f);
        int e0 = (s0 + s4) * 4096;
        int e1 = (s0 - s4) * 4096;
        r.x0   = e0 + e3;
        r.x3   = e0 - e3;
        r.x1   = e1 + e2;
        r.x2   = e1 - e2;
        // odd part
        int o0 = s7, o1 = s5, o2 = s3, o3 = s1;
        int q3 = o0 + o2, q4 = o1 + o3, q1 = o0 + o3, q2 = o1 + o2;
        int q5 = (q3 + q4) * fx(1.175875602f);
        o0 *= fx(0.298631336f);
        o1 *= fx(2.053119869f);
        o2 *= fx(3.072711026f);
        o3 *= fx(1.501321110f);
        q1 = q5 + q1 * fx(-0.899976223f);
        q2 = q5 + q2 * fx(-2.562915447f);
        q3 *= fx(-1.961570560f);
        q4 *= fx(-0.390180644f);
        r.t3 = o3 + q1 + q4;
        r.t2 = o2 + q2 + q3;
        r.t1 = o1 + q2 + q4;
        r.t0 = o0 + q1 + q3;
        return r;
    }
    static inline uint8_t clampByte(int x) { return (uint8_t)(x < 0 ? 0 : (x > 255 ? 255 : x)); }
    static void           idct(uint8_t* out, int stride, const int16_t d[64])
    {
        int tmp[64];
        for(int c = 0; c < 8; ++c)
        {
            const int16_t* p = d + c;
            if(!p[8] && !p[16] && !p[24] && !p[32] && !p[40] && !p[48] && !p[56])
            {
                const int v = p[0] * 4;
                for(int r = 0; r < 8; ++r) tmp[r * 8 + c] = v;
                continue;
            }
            Pass b = butterfly(p[0], p[8], p[16], p[24], p[32], p[40], p[48], p[56]);
            b.x0 += 512; b.x1 += 512; b.x2 += 512; b.x3 += 512;
            tmp[0 * 8 + c] = (b.x0 + b.t3) >> 10;
            tmp[7 * 8 + c] = (b.x0 - b.t3) >> 10;
            tmp[1 * 8 + c] = (b.x1 + b.t2) >> 10;
            tmp[6 * 8 + c] = (b.x1 - b.t2) >> 10;
            tmp[2 * 8 + c] = (b.x2 + b.t1) >> 10;
            tmp[5 * 8 + c] = (b.x2 - b.t1) >> 10;
            tmp[3 * 8 + c] = (b.x3 + b.t0) >> 10;
            tmp[4 * 8 + c] = (b.x3 - b.t0) >> 10;
        }
        const int bias = 65536 + (128 << 17);
        for(int r = 0; r < 8; ++r)
        {
            const int* v = tmp + r * 8;
            uint8_t*   o = out + (size_t)r * stride;
            Pass       b = butterfly(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]);
            b.x0 += bias; b.x1 += bias; b.x2 += bias; b.x3 += bias;
            o[0] = clampByte((b.x0 + b.t3) >> 17);
            o[7] = clampByte((b.x0 - b.t3) >> 17);
            o[1] = clampByte((b.x1 + b.t2) >> 17);
            o[6] = clampByte((b.x1 - b.t2) >> 17);
            o[2] = clampByte((b.x2 + b.t1) >> 17);
            o[5] = clampByte((b.x2 - b.t1) >> 17);
            o[3] = clampByte((b.x3 + b.t0) >> 17);
            o[4] = clampByte((b.x3 - b.t0) >> 17);
        }
    }

    // ---- one scan ----
    // returns false on a decoding error; true also when the scan ends early at a missing restart marker
    bool restartDue()  // after every MCU; true: stop the scan
    {
        if(--todo <= 0)
        {
            if(bitCnt < 24) fill();
            if(!(pendingMarker >= 0xd0 && pendingMarker <= 0xd7)) return true;
            resetEntropy();
        }
        return false;
    }
    bool scanData()
    {
        resetEntropy();
        int16_t blk[64];
        if(scanN == 1)
        {
            Component& c  = comp[order[0]];
            const int  bw = (c.w + 7) >> 3, bh = (c.hgt + 7) >> 3;
            for(int by = 0; by < bh; ++by)
                for(int bx = 0; bx < bw; ++bx)
                {
                    if(!progressive)
                    {
                        if(!baselineBlock(blk, c)) return false;
                        idct(&c.plane[(size_t)c.stride * by * 8 + bx * 8], c.stride, blk);
                    }
                    else
                    {
                        int16_t* b = &c.coeff[64 * ((size_t)bx + (size_t)by * c.blocksW)];
                        if(!(ss == 0 ? progressiveDC(b, c) : progressiveAC(b, c))) return false;
                    }
                    if(restartDue()) return true;
                }
            return true;
        }
        for(int my = 0; my < mcuY; ++my)
            for(int mx = 0; mx < mcuX; ++mx)
            {
                for(int k = 0; k < scanN; ++k)
                {
                    Component& c = comp[order[k]];
                    for(int y = 0; y < c.v; ++y)
                        for(int x = 0; x < c.h; ++x)
                        {
                            const int bx = mx * c.h + x, by = my * c.v + y;
                            if(!progressive)
                            {
                                if(!baselineBlock(blk, c)) return false;
                                idct(&c.plane[(size_t)c.stride * by * 8 + bx * 8], c.stride, blk);
                            }
                            else if(!progressiveDC(&c.coeff[64 * ((size_t)bx + (size_t)by * c.blocksW)], c))
                                return false;
                        }
                }
                if(restartDue()) return true;
            }
        return true;
    }
    void finishProgressive()
    {
        for(int i = 0; i < ncomp; ++i)
        {
            Component& c  = comp[i];
            const int  bw = (c.w + 7) >> 3, bh = (c.hgt + 7) >> 3;
            for(int by = 0; by < bh; ++by)
                for(int bx = 0; bx < bw; ++bx)
                {
                    int16_t* b = &c.coeff[64 * ((size_t)bx + (size_t)by * c.blocksW)];
                    for(int k = 0; k < 64; ++k) b[k] = (int16_t)(b[k] * quant[c.tq][k]);
                    idct(&c.plane[(size_t)c.stride * by * 8 + bx * 8], c.stride, b);
                }
        }
    }

    bool decode()
    {
        memset(quant, 0, sizeof quant);
        if(nextMarker() != 0xd8) return false;
        int m = nextMarker();
        while(m != 0xc0 && m != 0xc1 && m != 0xc2)
        {
            if(!tablesOrMisc(m)) return false;
            m = nextMarker();
            while(m == 0xff)
            {
                if(s.eof()) return false;
                m = nextMarker();
            }
        }
        progressive = m == 0xc2;
        if(!frameHeader()) return false;
        m = nextMarker();
        while(m != 0xd9)
        {
            if(m == 0xda)
            {
                if(!scanHeader()) return false;
                if(!scanData()) return false;
                if(pendingMarker == 0xff)
                    while(!s.eof())  // stray bytes after the entropy-coded segment
                        if(s.u8() == 0xff)
                        {
                            pendingMarker = s.u8();
                            break;
                        }
            }
            else if(m == 0xdc)
            {
                s.u16();
                s.u16();
            }
            else if(!tablesOrMisc(m))
                return false;
            m = nextMarker();
        }
        if(progressive) finishProgressive();
        return true;
    }
};

// ---- upsampling of one row ----
inline uint8_t q4(int x) { return (uint8_t)(x >> 2); }
inline uint8_t q16(int x) { return (uint8_t)(x >> 4); }
const uint8_t* upsampleRow(uint8_t* out, const uint8_t* nearRow, const uint8_t* farRow, int w, int hs, int vs)
{
    if(hs == 1 && vs == 1) return nearRow;
    if(hs == 1 && vs == 2)
    {
        for(int i = 0; i < w; ++i) out[i] = q4(3 * nearRow[i] + farRow[i] + 2);
        return out;
    }
    if(hs == 2 && vs == 1)
    {
        const uint8_t* in = nearRow;
        if(w == 1)
        {
            out[0] = out[1] = in[0];
            return out;
        }
        out[0] = in[0];
        out[1] = q4(in[0] * 3 + in[1] + 2);
        int i  = 1;
        for(; i < w - 1; ++i)
        {
            const int n    = 3 * in[i] + 2;
            out[i * 2]     = q4(n + in[i - 1]);
            out[i * 2 + 1] = q4(n + in[i + 1]);
        }
        out[i * 2]     = q4(in[w - 2] * 3 + in[w - 1] + 2);
        out[i * 2 + 1] = in[w - 1];
        return out;
    }
    if(hs == 2 && vs == 2)
    {
        if(w == 1)
        {
            out[0] = out[1] = q4(3 * nearRow[0] + farRow[0] + 2);
            return out;
        }
        int t1 = 3 * nearRow[0] + farRow[0];
        out[0] = q4(t1 + 2);
        for(int i = 1; i < w; ++i)
        {
            const int t0   = t1;
            t1             = 3 * nearRow[i] + farRow[i];
            out[i * 2 - 1] = q16(3 * t0 + t1 + 8);
            out[i * 2]     = q16(3 * t1 + t0 + 8);
        }
        out[w * 2 - 1] = q4(t1 + 2);
        return out;
    }
    for(int i = 0; i < w; ++i)
        for(int j = 0; j < hs; ++j) out[i * hs + j] = nearRow[i];
    return out;
}

inline int     c12(float x) { return ((int)(x * 4096.0f + 0.5f)) << 8; }
inline uint8_t sat(int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }
void           ycc_to_rgba(uint8_t* out, const uint8_t* y, const uint8_t* cb, const uint8_t* cr, int count)
{
    for(int i = 0; i < count; ++i, out += 4)
    {
        const int yf = (y[i] << 20) + (1 << 19);
        const int r_ = cr[i] - 128, b_ = cb[i] - 128;
        const int r  = yf + r_ * c12(1.40200f);
        const int g  = yf + (r_ * -c12(0.71414f)) + (int)((uint32_t)(b_ * -c12(0.34414f)) & 0xffff0000u);
        const int b  = yf + b_ * c12(1.77200f);
        out[0]       = sat(r >> 20);
        out[1]       = sat(g >> 20);
        out[2]       = sat(b >> 20);
        out[3]       = 255;
    }
}
inline uint8_t mul255(uint8_t x, uint8_t y)
{
    const unsigned t = (unsigned)x * y + 128;
    return (uint8_t)((t + (t >> 8)) >> 8);
}

}  // namespace

bool is_jpeg(const std::vector<uint8_t>& f)
{
    // 0xff, any number of 0xff fill bytes, then SOI
    size_t i = 0;
    if(f.empty() || f[0] != 0xff) return false;
    while(i < f.size() && f[i] == 0xff) ++i;
    return i < f.size() && f[i] == 0xd8;
}

bool load_jpeg(const std::vector<uint8_t>& f, Image& img)
{
    Decoder d;
    d.s.p = f.data();
    d.s.n = f.size();
    if(!d.decode()) return false;
    const int  W = d.width, H = d.height;
    const bool isRgb = d.ncomp == 3 && (d.rgbIds == 3 || (d.adobeTransform == 0 && !d.jfif));
    struct RowState
    {
        int hs, vs, ystep, ypos, line0, line1, wLores;
    } st[4];
    std::vector<uint8_t> line[4];
    for(int k = 0; k < d.ncomp; ++k)
    {
        const Component& c = d.comp[k];
        st[k].hs           = d.hmax / c.h;
        st[k].vs           = d.vmax / c.v;
        st[k].ystep        = st[k].vs >> 1;
        st[k].wLores       = (W + st[k].hs - 1) / st[k].hs;
        st[k].ypos         = 0;
        st[k].line0 = st[k].line1 = 0;
        line[k].assign((size_t)W + 8, 0);
    }
    img.width  = (uint32_t)W;
    img.height = (uint32_t)H;
    img.rgba.assign((size_t)W * H * 4, 0);
    const uint8_t* row[4] = {nullptr, nullptr, nullptr, nullptr};
    for(int j = 0; j < H; ++j)
    {
        uint8_t* out = &img.rgba[(size_t)j * W * 4];
        for(int k = 0; k < d.ncomp; ++k)
        {
            const Component& c    = d.comp[k];
            RowState&        r    = st[k];
            const bool       yBot = r.ystep >= (r.vs >> 1);
            const uint8_t*   l0   = &c.plane[(size_t)r.line0 * c.stride];
            const uint8_t*   l1   = &c.plane[(size_t)r.line1 * c.stride];
            row[k]                = upsampleRow(line[k].data(), yBot ? l1 : l0, yBot ? l0 : l1, r.wLores, r.hs, r.vs);
            if(++r.ystep >= r.vs)
            {
                r.ystep = 0;
                r.line0 = r.line1;
                if(++r.ypos < c.hgt) r.line1 += 1;
            }
        }
        if(d.ncomp == 3)
        {
            if(isRgb)
                for(int i = 0; i < W; ++i, out += 4)
                {
                    out[0] = row[0][i];
                    out[1] = row[1][i];
                    out[2] = row[2][i];
                    out[3] = 255;
                }
            else
                ycc_to_rgba(out, row[0], row[1], row[2], W);
        }
        else if(d.ncomp == 4)
        {
            if(d.adobeTransform == 0)  // CMYK
                for(int i = 0; i < W; ++i, out += 4)
                {
                    const uint8_t m = row[3][i];
                    out[0]          = mul255(row[0][i], m);
                    out[1]          = mul255(row[1][i], m);
                    out[2]          = mul255(row[2][i], m);
                    out[3]          = 255;
                }
            else if(d.adobeTransform == 2)  // YCCK
            {
                ycc_to_rgba(out, row[0], row[1], row[2], W);
                for(int i = 0; i < W; ++i, out += 4)
                {
                    const uint8_t m = row[3][i];
                    out[0]          = mul255((uint8_t)(255 - out[0]), m);
                    out[1]          = mul255((uint8_t)(255 - out[1]), m);
                    out[2]          = mul255((uint8_t)(255 - out[2]), m);
                }
            }
            else
                ycc_to_rgba(out, row[0], row[1], row[2], W);
        }
        else
            for(int i = 0; i < W; ++i, out += 4)
            {
                out[0] = out[1] = out[2] = row[0][i];
                out[3]                   = 255;
            }
    }
    return true;
}

}  // namespace prt_host
