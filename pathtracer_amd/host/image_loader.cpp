// image_loader.cpp — RGBA8 decoding of the texture files Model::createTextures hands to stbi_load(...,
// STBI_rgb_alpha) in the reference (/root/reference/src/Model.cpp:345-386). Written from the file-format
// specifications (PNG 1.2 / RFC 1950-1951, Truevision TGA 2.0, Windows BMP, Netpbm); the conversion rules to
// four 8-bit channels follow what the reference's decoder yields (checked against its output on the
// fixtures in tests/golden/obj/textures):
//   grey -> (g,g,g,255), grey+alpha -> (g,g,g,a), RGB -> (r,g,b,255); 16-bit samples keep their high byte;
//   1/2/4-bit grey scales to 0..255; PNG tRNS becomes alpha; TGA 15/16-bit is 5-5-5; BMP 32-bit with an
//   all-zero alpha channel is opaque.
// JPEG lives in jpeg_loader.cpp. GIF, PSD, HDR, PIC are not decoded: load_image() fails and the caller falls back
// to the 1x1 white texture exactly as the reference does for a file it cannot read (Model.cpp:352-359).
#include "image_loader.h"

#include <cstdio>
#include <cstring>

namespace prt_host {

namespace {

// ---- RFC 1951 inflate ------------------------------------------------------------------------------------
struct BitReader
{
    const uint8_t* p;
    size_t         n, pos = 0;
    uint32_t       bitbuf = 0;
    int            bitcnt = 0;
    bool           bad    = false;
    uint32_t bits(int k)
    {
        while(bitcnt < k)
        {
            uint32_t b = 0;
            if(pos < n)
                b = p[pos++];
            else
                bad = true;
            bitbuf |= b << bitcnt;
            bitcnt += 8;
        }
        const uint32_t v = bitbuf & ((k == 32) ? 0xffffffffu : ((1u << k) - 1u));
        bitbuf >>= k;
        bitcnt -= k;
        return v;
    }
    void align() { bitbuf = 0; bitcnt = 0; }
};

struct Huffman
{
    uint16_t count[16];
    uint16_t symbol[288];
    bool     build(const uint8_t* lengths, int n)
    {
        memset(count, 0, sizeof count);
        for(int i = 0; i < n; ++i) count[lengths[i]]++;
        count[0] = 0;
        int left = 1;
        for(int len = 1; len < 16; ++len)
        {
            left <<= 1;
            left -= count[len];
            if(left < 0) return false;
        }
        uint16_t offs[16];
        offs[1] = 0;
        for(int len = 1; len < 15; ++len) offs[len + 1] = (uint16_t)(offs[len] + count[len]);
        for(int i = 0; i < n; ++i)
            if(lengths[i]) symbol[offs[lengths[i]]++] = (uint16_t)i;
        return true;
    }
    int decode(BitReader& br) const
    {
        int code = 0, first = 0, index = 0;
        for(int len = 1; len < 16; ++len)
        {
            code |= (int)br.bits(1);
            const int c = count[len];
            if(code - c < first) return symbol[index + (code - first)];
            index += c;
            first += c;
            first <<= 1;
            code <<= 1;
            if(br.bad) return -1;
        }
        return -1;
    }
};

// `limit`: the decoder stops (successfully) once it has produced this many bytes. A PNG knows how many bytes its
// scanlines take; what a stream holds beyond them is never looked at (the reference's decoder ignores it too), and a
// decompression bomb costs no more memory than the image it claims to be.
bool inflate_raw(const uint8_t* src, size_t n, std::vector<uint8_t>& out, size_t limit)
{
    static const uint16_t lbase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
    static const uint16_t lext[29]  = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
    static const uint16_t dbase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
    static const uint16_t dext[30]  = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
    static const uint8_t  order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    BitReader br{src, n};
    for(;;)
    {
        const uint32_t last = br.bits(1), type = br.bits(2);
        if(br.bad) return false;
        if(type == 0)
        {
            br.align();
            if(br.pos + 4 > n) return false;
            const uint32_t len = src[br.pos] | (src[br.pos + 1] << 8), nlen = src[br.pos + 2] | (src[br.pos + 3] << 8);
            br.pos += 4;
            if((len ^ 0xffffu) != nlen || br.pos + len > n) return false;
            const size_t take = out.size() + len > limit ? limit - out.size() : (size_t)len;
            out.insert(out.end(), src + br.pos, src + br.pos + take);
            br.pos += len;
            if(out.size() >= limit) return true;
        }
        else if(type == 1 || type == 2)
        {
            Huffman lit, dist;
            uint8_t lengths[320];
            if(type == 1)
            {
                int i = 0;
                for(; i < 144; ++i) lengths[i] = 8;
                for(; i < 256; ++i) lengths[i] = 9;
                for(; i < 280; ++i) lengths[i] = 7;
                for(; i < 288; ++i) lengths[i] = 8;
                lit.build(lengths, 288);
                for(i = 0; i < 30; ++i) lengths[i] = 5;
                dist.build(lengths, 30);
            }
            else
            {
                const int nlen = (int)br.bits(5) + 257, ndist = (int)br.bits(5) + 1, ncode = (int)br.bits(4) + 4;
                if(nlen > 286 || ndist > 30) return false;
                uint8_t cl[19] = {0};
                for(int i = 0; i < ncode; ++i) cl[order[i]] = (uint8_t)br.bits(3);
                Huffman clh;
                if(!clh.build(cl, 19)) return false;
                int idx = 0;
                while(idx < nlen + ndist)
                {
                    const int sym = clh.decode(br);
                    if(sym < 0) return false;
                    if(sym < 16)
                        lengths[idx++] = (uint8_t)sym;
                    else
                    {
                        int     rep;
                        uint8_t val = 0;
                        if(sym == 16)
                        {
                            if(idx == 0) return false;
                            val = lengths[idx - 1];
                            rep = 3 + (int)br.bits(2);
                        }
                        else if(sym == 17)
                            rep = 3 + (int)br.bits(3);
                        else
                            rep = 11 + (int)br.bits(7);
                        if(idx + rep > nlen + ndist) return false;
                        while(rep--) lengths[idx++] = val;
                    }
                }
                if(!lit.build(lengths, nlen)) return false;
                dist.build(lengths + nlen, ndist);  // incomplete distance codes are legal
            }
            for(;;)
            {
                const int sym = lit.decode(br);
                if(sym < 0 || br.bad) return false;
                if(sym < 256)
                {
                    out.push_back((uint8_t)sym);
                    if(out.size() >= limit) return true;
                }
                else if(sym == 256)
                    break;
                else
                {
                    const int ls = sym - 257;
                    if(ls >= 29) return false;
                    const int len = lbase[ls] + (int)br.bits(lext[ls]);
                    const int ds  = dist.decode(br);
                    if(ds < 0 || ds >= 30) return false;
                    const size_t d = dbase[ds] + br.bits(dext[ds]);
                    if(d > out.size()) return false;
                    const size_t from = out.size() - d;
                    for(int k = 0; k < len && out.size() < limit; ++k) out.push_back(out[from + k]);
                    if(out.size() >= limit) return true;
                }
            }
        }
        else
            return false;
        if(last) break;
    }
    return true;
}

bool zlib_inflate(const uint8_t* src, size_t n, std::vector<uint8_t>& out, size_t limit)
{
    if(n < 6) return false;
    if((src[0] & 0x0f) != 8 || ((src[0] << 8) | src[1]) % 31 != 0 || (src[1] & 0x20)) return false;
    return inflate_raw(src + 2, n - 2, out, limit);  // the Adler-32 trailer is not verified (neither does the reference)
}

// ---- PNG -----------------------------------------------------------------------------------------------------
uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

int paeth(int a, int b, int c)
{
    const int p = a + b - c, pa = p > a ? p - a : a - p, pb = p > b ? p - b : b - p, pc = p > c ? p - c : c - p;
    if(pa <= pb && pa <= pc) return a;
    if(pb <= pc) return b;
    return c;
}

// un-filters one (sub-)image of w x h pixels; `raw` holds h rows of 1 + stride bytes
bool png_unfilter(const uint8_t* raw, size_t rawLen, uint32_t w, uint32_t h, int bpp /*bytes per complete pixel, >= 1*/,
                  size_t stride, std::vector<uint8_t>& pix)
{
    if(rawLen < (stride + 1) * (size_t)h) return false;
    pix.assign(stride * (size_t)h, 0);
    (void)w;
    for(uint32_t y = 0; y < h; ++y)
    {
        const uint8_t* in   = raw + (stride + 1) * (size_t)y;
        const int      ft   = in[0];
        uint8_t*       cur  = pix.data() + stride * (size_t)y;
        const uint8_t* prev = y ? cur - stride : nullptr;
        ++in;
        for(size_t x = 0; x < stride; ++x)
        {
            const int a = x >= (size_t)bpp ? cur[x - bpp] : 0, b = prev ? prev[x] : 0,
                      c = (prev && x >= (size_t)bpp) ? prev[x - bpp] : 0;
            int v = in[x];
            switch(ft)
            {
                case 0: break;
                case 1: v += a; break;
                case 2: v += b; break;
                case 3: v += (a + b) >> 1; break;
                case 4: v += paeth(a, b, c); break;
                default: return false;
            }
            cur[x] = (uint8_t)v;
        }
    }
    return true;
}

bool load_png(const std::vector<uint8_t>& f, Image& img)
{
    static const uint8_t sig[8] = {137, 80, 78, 71, 13, 10, 26, 10};
    if(f.size() < 8 || memcmp(f.data(), sig, 8) != 0) return false;
    size_t               pos = 8;
    uint32_t             w = 0, h = 0;
    int                  depth = 0, ctype = 0, interlace = 0;
    std::vector<uint8_t> idat, plte, trns;
    bool                 haveHdr = false;
    while(pos + 12 <= f.size())
    {
        const uint32_t len = be32(&f[pos]);
        const uint8_t* tp  = &f[pos + 4];
        if(pos + 12 + (size_t)len > f.size()) return false;
        const uint8_t* d = &f[pos + 8];
        if(!memcmp(tp, "IHDR", 4))
        {
            if(len != 13) return false;
            w = be32(d); h = be32(d + 4);
            depth = d[8]; ctype = d[9]; interlace = d[12];
            if(d[10] != 0 || d[11] != 0) return false;
            haveHdr = true;
        }
        else if(!memcmp(tp, "PLTE", 4))
            plte.assign(d, d + len);
        else if(!memcmp(tp, "tRNS", 4))
            trns.assign(d, d + len);
        else if(!memcmp(tp, "IDAT", 4))
            idat.insert(idat.end(), d, d + len);
        else if(!memcmp(tp, "IEND", 4))
            break;
        pos += 12 + (size_t)len;
    }
    if(!haveHdr || w == 0 || h == 0 || w > (1u << 24) || h > (1u << 24)) return false;
    int chans;
    switch(ctype)
    {
        case 0: chans = 1; break;
        case 2: chans = 3; break;
        case 3: chans = 1; break;
        case 4: chans = 2; break;
        case 6: chans = 4; break;
        default: return false;
    }
    if(!(depth == 8 || depth == 16 || ((ctype == 0 || ctype == 3) && (depth == 1 || depth == 2 || depth == 4)))) return false;
    if(ctype == 3 && (depth == 16 || plte.empty())) return false;
    if(interlace > 1) return false;
    if((uint64_t)w * h > (1ull << 28)) return false;  // 2^28 pixels: beyond any texture, and it bounds every allocation below
    const int bitsPerPixel = depth * chans;
    // bytes the scanlines (filter byte + packed samples) of the whole image take: known before anything is allocated
    size_t expected = 0;
    {
        auto lines = [&](uint32_t pw, uint32_t ph) { return pw && ph ? ((size_t)((uint64_t)pw * bitsPerPixel + 7) / 8 + 1) * ph : (size_t)0; };
        if(!interlace)
            expected = lines(w, h);
        else
        {
            static const uint32_t ax0[7] = {0, 4, 0, 2, 0, 1, 0}, ay0[7] = {0, 0, 4, 0, 2, 0, 1}, adx[7] = {8, 8, 4, 4, 2, 2, 1},
                                  ady[7] = {8, 8, 8, 4, 4, 2, 2};
            for(int p = 0; p < 7; ++p)
                expected += lines(w > ax0[p] ? (w - ax0[p] + adx[p] - 1) / adx[p] : 0, h > ay0[p] ? (h - ay0[p] + ady[p] - 1) / ady[p] : 0);
        }
    }
    std::vector<uint8_t> raw;
    if(!zlib_inflate(idat.data(), idat.size(), raw, expected)) return false;
    if(raw.size() < expected) return false;  // cut short: nothing is allocated for an image that is not there

    // samples (one uint16 per sample, native value) of the whole image
    std::vector<uint16_t> smp((size_t)w * h * chans, 0);
    const int             bpp          = bitsPerPixel >= 8 ? bitsPerPixel / 8 : 1;
    auto scatter = [&](const std::vector<uint8_t>& pix, size_t stride, uint32_t pw, uint32_t ph, uint32_t x0, uint32_t y0,
                       uint32_t dx, uint32_t dy) {
        for(uint32_t y = 0; y < ph; ++y)
        {
            const uint8_t* row = pix.data() + stride * (size_t)y;
            for(uint32_t x = 0; x < pw; ++x)
                for(int c = 0; c < chans; ++c)
                {
                    const size_t si = (size_t)x * chans + c;
                    uint16_t     v;
                    if(depth == 16)
                        v = (uint16_t)((row[2 * si] << 8) | row[2 * si + 1]);
                    else if(depth == 8)
                        v = row[si];
                    else
                    {
                        const size_t bit = si * depth;
                        v = (uint16_t)((row[bit >> 3] >> (8 - depth - (bit & 7))) & ((1 << depth) - 1));
                    }
                    smp[((size_t)(y0 + y * dy) * w + (x0 + x * dx)) * chans + c] = v;
                }
        }
    };
    std::vector<uint8_t> pix;
    if(!interlace)
    {
        const size_t stride = ((size_t)w * bitsPerPixel + 7) / 8;
        if(!png_unfilter(raw.data(), raw.size(), w, h, bpp, stride, pix)) return false;
        scatter(pix, stride, w, h, 0, 0, 1, 1);
    }
    else
    {
        static const int xs[7] = {0, 4, 0, 2, 0, 1, 0}, ys[7] = {0, 0, 4, 0, 2, 0, 1}, dxs[7] = {8, 8, 4, 4, 2, 2, 1},
                         dys[7] = {8, 8, 8, 4, 4, 2, 2};
        size_t off = 0;
        for(int p = 0; p < 7; ++p)
        {
            const uint32_t pw = (w - xs[p] + dxs[p] - 1) / dxs[p], ph = (h - ys[p] + dys[p] - 1) / dys[p];
            if((int)w <= xs[p] || (int)h <= ys[p] || pw == 0 || ph == 0) continue;
            const size_t stride = ((size_t)pw * bitsPerPixel + 7) / 8;
            if(off + (stride + 1) * ph > raw.size()) return false;
            if(!png_unfilter(raw.data() + off, raw.size() - off, pw, ph, bpp, stride, pix)) return false;
            scatter(pix, stride, pw, ph, xs[p], ys[p], dxs[p], dys[p]);
            off += (stride + 1) * ph;
        }
    }
    img.width  = w;
    img.height = h;
    img.rgba.assign((size_t)w * h * 4, 255);
    const int scale = depth == 1 ? 255 : depth == 2 ? 85 : depth == 4 ? 17 : 1;
    auto      to8   = [&](uint16_t v) -> uint8_t { return depth == 16 ? (uint8_t)(v >> 8) : (uint8_t)v; };
    for(size_t i = 0; i < (size_t)w * h; ++i)
    {
        const uint16_t* s = &smp[i * chans];
        uint8_t*        o = &img.rgba[i * 4];
        switch(ctype)
        {
            case 0:
            {
                o[0] = o[1] = o[2] = depth < 8 ? (uint8_t)(s[0] * scale) : to8(s[0]);
                if(trns.size() >= 2 && s[0] == (uint16_t)((trns[0] << 8) | trns[1])) o[3] = 0;
                break;
            }
            case 2:
                o[0] = to8(s[0]); o[1] = to8(s[1]); o[2] = to8(s[2]);
                if(trns.size() >= 6 && s[0] == (uint16_t)((trns[0] << 8) | trns[1]) && s[1] == (uint16_t)((trns[2] << 8) | trns[3])
                   && s[2] == (uint16_t)((trns[4] << 8) | trns[5]))
                    o[3] = 0;
                break;
            case 3:
            {
                const size_t k = s[0];
                if(3 * k + 2 < plte.size())
                {
                    o[0] = plte[3 * k]; o[1] = plte[3 * k + 1]; o[2] = plte[3 * k + 2];
                }
                else
                    o[0] = o[1] = o[2] = 0;
                if(k < trns.size()) o[3] = trns[k];
                break;
            }
            case 4: o[0] = o[1] = o[2] = to8(s[0]); o[3] = to8(s[1]); break;
            case 6: o[0] = to8(s[0]); o[1] = to8(s[1]); o[2] = to8(s[2]); o[3] = to8(s[3]); break;
        }
    }
    return true;
}

// ---- TGA -----------------------------------------------------------------------------------------------------
void tga_pixel(const uint8_t* p, int bits, uint8_t* o)
{
    switch(bits)
    {
        case 8: o[0] = o[1] = o[2] = p[0]; o[3] = 255; break;
        case 15:
        case 16:
        {
            const unsigned v = p[0] | (p[1] << 8);
            const unsigned r = (v >> 10) & 31, g = (v >> 5) & 31, b = v & 31;
            o[0] = (uint8_t)((r * 255) / 31); o[1] = (uint8_t)((g * 255) / 31); o[2] = (uint8_t)((b * 255) / 31);
            o[3] = 255;
            break;
        }
        case 24: o[0] = p[2]; o[1] = p[1]; o[2] = p[0]; o[3] = 255; break;
        case 32: o[0] = p[2]; o[1] = p[1]; o[2] = p[0]; o[3] = p[3]; break;
    }
}

bool load_tga(const std::vector<uint8_t>& f, Image& img)
{
    if(f.size() < 18) return false;
    const int idLen = f[0], cmapType = f[1], type = f[2];
    const int cmapStart = f[3] | (f[4] << 8), cmapLen = f[5] | (f[6] << 8), cmapBits = f[7];
    const int w = f[12] | (f[13] << 8), h = f[14] | (f[15] << 8), bits = f[16], desc = f[17];
    const bool rle = type >= 8;
    const int  base = type & 7;
    if(w <= 0 || h <= 0 || cmapType > 1) return false;
    if(!(base == 1 || base == 2 || base == 3)) return false;
    if(base == 1 && (!cmapType || !(bits == 8 || bits == 16))) return false;
    if(base == 2 && !(bits == 15 || bits == 16 || bits == 24 || bits == 32)) return false;
    if(base == 3 && !(bits == 8 || bits == 16)) return false;
    if(cmapType && !(cmapBits == 8 || cmapBits == 15 || cmapBits == 16 || cmapBits == 24 || cmapBits == 32)) return false;
    size_t pos = 18 + (size_t)idLen;
    const int cmapBytes = (cmapBits + 7) / 8;
    const uint8_t* cmap = nullptr;
    if(base == 1 && cmapLen == 0) return false;  // indices into an empty palette
    if(cmapType)
    {
        if(pos + (size_t)cmapLen * cmapBytes > f.size()) return false;
        cmap = f.data() + pos;
        pos += (size_t)cmapLen * cmapBytes;
    }
    const int bytes = (bits + 7) / 8;
    img.width = (uint32_t)w; img.height = (uint32_t)h;
    img.rgba.assign((size_t)w * h * 4, 255);
    auto emit = [&](const uint8_t* p, uint8_t* o) -> bool {
        if(base == 1)
        {
            int idx = bits == 8 ? p[0] : (p[0] | (p[1] << 8));
            idx -= cmapStart;
            if(idx < 0 || idx >= cmapLen) idx = 0;
            tga_pixel(cmap + (size_t)idx * cmapBytes, cmapBits, o);
        }
        else if(base == 3 && bits == 16)
        {
            o[0] = o[1] = o[2] = p[0];
            o[3] = p[1];
        }
        else
            tga_pixel(p, bits, o);
        return true;
    };
    size_t  i = 0, total = (size_t)w * h;
    uint8_t px[4];
    std::vector<uint8_t> lin(total * 4);
    if(!rle)
    {
        if(pos + total * bytes > f.size()) return false;
        for(; i < total; ++i, pos += bytes) emit(&f[pos], &lin[i * 4]);
    }
    else
        while(i < total)
        {
            if(pos >= f.size()) return false;
            const int hd = f[pos++], cnt = (hd & 127) + 1;
            if(hd & 128)
            {
                if(pos + bytes > f.size()) return false;
                emit(&f[pos], px);
                pos += bytes;
                for(int k = 0; k < cnt && i < total; ++k, ++i) memcpy(&lin[i * 4], px, 4);
            }
            else
                for(int k = 0; k < cnt && i < total; ++k, ++i)
                {
                    if(pos + bytes > f.size()) return false;
                    emit(&f[pos], &lin[i * 4]);
                    pos += bytes;
                }
        }
    const bool topDown = (desc & 0x20) != 0, rightLeft = (desc & 0x10) != 0;
    for(int y = 0; y < h; ++y)
        for(int x = 0; x < w; ++x)
        {
            const int sy = topDown ? y : h - 1 - y, sx = rightLeft ? w - 1 - x : x;
            memcpy(&img.rgba[((size_t)y * w + x) * 4], &lin[((size_t)sy * w + sx) * 4], 4);
        }
    return true;
}

// ---- BMP -----------------------------------------------------------------------------------------------------
uint32_t le32(const uint8_t* p) { return p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24); }
uint32_t le16(const uint8_t* p) { return p[0] | (p[1] << 8); }

int shift_of(uint32_t m) { int s = 0; if(!m) return 0; while(!(m & 1)) { m >>= 1; ++s; } return s; }
int width_of(uint32_t m) { int n = 0; if(!m) return 0; m >>= shift_of(m); while(m & 1) { m >>= 1; ++n; } return n; }
uint8_t expand_bits(uint32_t v, int bits)
{
    if(bits <= 0) return 0;
    if(bits >= 8) return (uint8_t)(v >> (bits - 8));
    // replicate the bit pattern to fill 8 bits
    uint32_t r = 0;
    int      have = 0;
    while(have < 8)
    {
        r = (r << bits) | v;
        have += bits;
    }
    return (uint8_t)(r >> (have - 8));
}

bool load_bmp(const std::vector<uint8_t>& f, Image& img)
{
    if(f.size() < 26 || f[0] != 'B' || f[1] != 'M') return false;
    const uint32_t dataOff = le32(&f[10]), hsz = le32(&f[14]);
    int32_t        w, h;
    int            bpp;
    uint32_t       comp = 0, mr = 0, mg = 0, mb = 0, ma = 0;
    if(hsz == 12)
    {
        w = (int32_t)le16(&f[18]); h = (int32_t)le16(&f[20]); bpp = (int)le16(&f[24]);
    }
    else if(hsz >= 40 && f.size() >= 14 + 40)
    {
        w = (int32_t)le32(&f[18]); h = (int32_t)le32(&f[22]); bpp = (int)le16(&f[28]); comp = le32(&f[30]);
        if(comp == 3)
        {
            const size_t mo = hsz == 40 ? 54 : 14 + 40;
            if(f.size() < mo + 12) return false;
            mr = le32(&f[mo]); mg = le32(&f[mo + 4]); mb = le32(&f[mo + 8]);
            if(hsz >= 56 && f.size() >= mo + 16) ma = le32(&f[mo + 12]);
        }
        else if(comp != 0)
            return false;  // RLE-compressed bitmaps are not decoded
    }
    else
        return false;
    const bool flip = h > 0;
    if(h < 0) h = -h;
    if(w <= 0 || h <= 0) return false;
    if(!(bpp == 1 || bpp == 4 || bpp == 8 || bpp == 16 || bpp == 24 || bpp == 32)) return false;
    if(comp == 0 && bpp == 16) { mr = 31u << 10; mg = 31u << 5; mb = 31u; }
    if(comp == 0 && bpp == 32) { mr = 0xffu << 16; mg = 0xffu << 8; mb = 0xffu; ma = 0xffu << 24; }
    const uint8_t* pal  = nullptr;
    int            npal = 0, palStride = hsz == 12 ? 3 : 4;
    if(bpp <= 8)
    {
        const size_t po = 14 + (size_t)hsz + (comp == 3 && hsz == 40 ? 12 : 0);
        npal            = (int)((dataOff - po) / palStride);
        if(npal <= 0 || po + (size_t)npal * palStride > f.size()) return false;
        pal = &f[po];
    }
    const size_t stride = (((size_t)w * bpp + 31) / 32) * 4;
    if((size_t)dataOff + stride * h > f.size()) return false;
    img.width = (uint32_t)w; img.height = (uint32_t)h;
    img.rgba.assign((size_t)w * h * 4, 255);
    bool anyAlpha = false;
    for(int y = 0; y < h; ++y)
    {
        const uint8_t* row = &f[dataOff + stride * (size_t)y];
        uint8_t*       o   = &img.rgba[(size_t)(flip ? h - 1 - y : y) * w * 4];
        for(int x = 0; x < w; ++x, o += 4)
        {
            if(bpp <= 8)
            {
                int idx;
                if(bpp == 8) idx = row[x];
                else if(bpp == 4) idx = (row[x >> 1] >> ((x & 1) ? 0 : 4)) & 15;
                else idx = (row[x >> 3] >> (7 - (x & 7))) & 1;
                if(idx >= npal) idx = 0;
                o[0] = pal[idx * palStride + 2]; o[1] = pal[idx * palStride + 1]; o[2] = pal[idx * palStride];
            }
            else if(bpp == 24)
            {
                o[0] = row[3 * x + 2]; o[1] = row[3 * x + 1]; o[2] = row[3 * x];
            }
            else
            {
                const uint32_t v = bpp == 16 ? le16(&row[2 * x]) : le32(&row[4 * x]);
                o[0] = expand_bits((v & mr) >> shift_of(mr), width_of(mr));
                o[1] = expand_bits((v & mg) >> shift_of(mg), width_of(mg));
                o[2] = expand_bits((v & mb) >> shift_of(mb), width_of(mb));
                if(ma)
                {
                    o[3] = expand_bits((v & ma) >> shift_of(ma), width_of(ma));
                    anyAlpha |= o[3] != 0;
                }
            }
        }
    }
    if(ma && !anyAlpha)  // a 32-bit bitmap whose alpha channel is all zero is opaque
        for(size_t i = 3; i < img.rgba.size(); i += 4) img.rgba[i] = 255;
    return true;
}

// ---- PNM (binary P5 / P6, 8 bits) -------------------------------------------------------------------------------
bool load_pnm(const std::vector<uint8_t>& f, Image& img)
{
    if(f.size() < 7 || f[0] != 'P' || (f[1] != '5' && f[1] != '6')) return false;
    const int chans = f[1] == '6' ? 3 : 1;
    size_t    pos   = 2;
    auto      next  = [&]() -> long {
        for(;;)
        {
            while(pos < f.size() && (f[pos] == ' ' || f[pos] == '\t' || f[pos] == '\n' || f[pos] == '\r' || f[pos] == '\f' || f[pos] == '\v')) ++pos;
            if(pos < f.size() && f[pos] == '#')
                while(pos < f.size() && f[pos] != '\n' && f[pos] != '\r') ++pos;
            else
                break;
        }
        long v = 0;
        bool any = false;
        while(pos < f.size() && f[pos] >= '0' && f[pos] <= '9')
        {
            v   = v * 10 + (f[pos++] - '0');
            any = true;
            if(v > (1l << 30)) return -1;
        }
        return any ? v : -1;
    };
    const long w = next(), h = next(), mx = next();
    if(w <= 0 || h <= 0 || mx <= 0 || mx > 255) return false;
    ++pos;  // the single whitespace after maxval
    if(pos + (size_t)w * h * chans > f.size()) return false;
    img.width = (uint32_t)w; img.height = (uint32_t)h;
    img.rgba.assign((size_t)w * h * 4, 255);
    for(size_t i = 0; i < (size_t)w * h; ++i)
    {
        const uint8_t* s = &f[pos + i * chans];
        uint8_t*       o = &img.rgba[i * 4];
        if(chans == 3) { o[0] = s[0]; o[1] = s[1]; o[2] = s[2]; }
        else o[0] = o[1] = o[2] = s[0];
    }
    return true;
}

}  // namespace

bool load_image_memory(const std::vector<uint8_t>& f, Image& img)
{
    if(is_jpeg(f)) return load_jpeg(f, img);  // the reference's decoder tries JPEG first
    if(f.size() >= 8 && f[0] == 137 && f[1] == 'P') return load_png(f, img);
    if(f.size() >= 2 && f[0] == 'B' && f[1] == 'M') return load_bmp(f, img);
    if(f.size() >= 2 && f[0] == 'P' && (f[1] == '5' || f[1] == '6')) return load_pnm(f, img);
    if(f.size() >= 4 && !memcmp(f.data(), "GIF8", 4)) return false;
    return load_tga(f, img);  // TGA has no signature: tried last
}

bool load_image(const std::string& path, Image& img)
{
    FILE* fp = fopen(path.c_str(), "rb");
    if(!fp) return false;
    std::vector<uint8_t> f;
    uint8_t              buf[65536];
    size_t               n;
    while((n = fread(buf, 1, sizeof buf, fp)) > 0) f.insert(f.end(), buf, buf + n);
    fclose(fp);
    return load_image_memory(f, img);
}

}  // namespace prt_host
