// host_mask_image.hip - host-only code (no device kernel): decodes the feature-mask image the reference loads with
// cv::imread(str_mask) + cv::cvtColor(BGR2GRAY) in ORB_GPU::ORB_GPU (src/cuda/orb_gpu.cpp:64-75; yaml keys mask.left / mask.right,
// src/Tracking.cpp:137-141, e.g. Examples/Stereo/stereo_rig_realsense.yaml:7-8 names two PNG files).  OpenCV is not a dependency of
// libjsorb, so the two formats a mask realistically comes in are decoded here: PNG (non-interlaced, 8/16 bit, gray / gray+alpha /
// RGB / RGBA / palette) and binary PGM / PPM.  What imread's default flag (IMREAD_COLOR) delivers is a 3-channel 8-bit BGR image
// (alpha dropped, 16-bit samples reduced to their high byte, gray replicated); BGR2GRAY's 8-bit form is
// (B*3735 + G*19235 + R*9798 + 2^14) >> 15 - the identity on gray input (the weights sum to 2^15).
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/jsorb.h"

namespace {

// ---- zlib / DEFLATE (RFC 1950 / 1951) ----
struct BitReader {
    const uint8_t *p, *end;
    uint32_t acc = 0;
    int n = 0;
    bool ok = true;
    uint32_t bits(int k)
    {
        while (n < k) {
            if (p >= end) { ok = false; return 0; }
            acc |= (uint32_t)*p++ << n;
            n += 8;
        }
        const uint32_t v = acc & ((k == 32) ? 0xFFFFFFFFu : ((1u << k) - 1u));
        acc = k == 32 ? 0 : acc >> k;
        n -= k;
        return v;
    }
    void align() { acc = 0; n = 0; }
};

struct Huffman {
    uint16_t count[16], symbol[288];
    void build(const uint8_t *len, int nsym)
    {
        memset(count, 0, sizeof count);
        for (int i = 0; i < nsym; i++) count[len[i]]++;
        count[0] = 0;
        uint16_t offs[16];
        offs[1] = 0;
        for (int i = 1; i < 15; i++) offs[i + 1] = offs[i] + count[i];
        for (int i = 0; i < nsym; i++)
            if (len[i]) symbol[offs[len[i]]++] = (uint16_t)i;
    }
    int decode(BitReader &br) const
    {
        int code = 0, first = 0, index = 0;
        for (int l = 1; l <= 15; l++) {
            code |= (int)br.bits(1);
            if (!br.ok) return -1;
            const int c = count[l];
            if (code - c < first) return symbol[index + (code - first)];
            index += c;
            first += c;
            first <<= 1;
            code <<= 1;
        }
        return -1;
    }
};

bool inflate_zlib(const std::vector<uint8_t> &in, std::vector<uint8_t> &out, size_t expected)
{
    if (in.size() < 6 || (in[0] & 0x0F) != 8 || ((in[0] << 8 | in[1]) % 31) != 0 || (in[1] & 0x20)) return false;
    BitReader br{in.data() + 2, in.data() + in.size()};
    out.clear();
    out.reserve(expected);
    static const uint16_t lbase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
    static const uint8_t lext[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
    static const uint16_t dbase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
    static const uint8_t dext[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
    for (;;) {
        const int last = (int)br.bits(1), type = (int)br.bits(2);
        if (!br.ok) return false;
        if (type == 0) {
            br.align();
            if (br.end - br.p < 4) return false;
            const unsigned len = br.p[0] | br.p[1] << 8, nlen = br.p[2] | br.p[3] << 8;
            br.p += 4;
            if ((len ^ 0xFFFFu) != nlen || (size_t)(br.end - br.p) < len) return false;
            out.insert(out.end(), br.p, br.p + len);
            br.p += len;
        } else if (type == 1 || type == 2) {
            Huffman lit, dist;
            uint8_t lens[320];
            if (type == 1) {
                for (int i = 0; i < 144; i++) lens[i] = 8;
                for (int i = 144; i < 256; i++) lens[i] = 9;
                for (int i = 256; i < 280; i++) lens[i] = 7;
                for (int i = 280; i < 288; i++) lens[i] = 8;
                lit.build(lens, 288);
                for (int i = 0; i < 30; i++) lens[i] = 5;
                dist.build(lens, 30);
            } else {
                const int nlen = (int)br.bits(5) + 257, ndist = (int)br.bits(5) + 1, ncode = (int)br.bits(4) + 4;
                if (!br.ok || nlen > 286 || ndist > 30) return false;
                static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
                uint8_t cl[19] = {0};
                for (int i = 0; i < ncode; i++) cl[order[i]] = (uint8_t)br.bits(3);
                Huffman clh;
                clh.build(cl, 19);
                int i = 0;
                while (i < nlen + ndist) {
                    const int sym = clh.decode(br);
                    if (sym < 0) return false;
                    if (sym < 16) lens[i++] = (uint8_t)sym;
                    else {
                        int rep, val = 0;
                        if (sym == 16) { if (i == 0) return false; val = lens[i - 1]; rep = 3 + (int)br.bits(2); }
                        else if (sym == 17) rep = 3 + (int)br.bits(3);
                        else rep = 11 + (int)br.bits(7);
                        if (i + rep > nlen + ndist) return false;
                        while (rep--) lens[i++] = (uint8_t)val;
                    }
                }
                lit.build(lens, nlen);
                dist.build(lens + nlen, ndist);
            }
            for (;;) {
                const int sym = lit.decode(br);
                if (sym < 0 || !br.ok) return false;
                if (sym < 256) out.push_back((uint8_t)sym);
                else if (sym == 256) break;
                else {
                    if (sym > 285) return false;
                    const int len = lbase[sym - 257] + (int)br.bits(lext[sym - 257]);
                    const int ds = dist.decode(br);
                    if (ds < 0 || ds > 29) return false;
                    const size_t d = dbase[ds] + br.bits(dext[ds]);
                    if (!br.ok || d > out.size()) return false;
                    for (int k = 0; k < len; k++) out.push_back(out[out.size() - d]);
                }
            }
        } else
            return false;
        if (last) break;
    }
    return true;
}

inline uint8_t bgr2gray(unsigned r, unsigned g, unsigned b) { return (uint8_t)((b * 3735u + g * 19235u + r * 9798u + (1u << 14)) >> 15); }
inline uint32_t be32(const uint8_t *p) { return (uint32_t)p[0] << 24 | (uint32_t)p[1] << 16 | (uint32_t)p[2] << 8 | p[3]; }

bool decode_png(const std::vector<uint8_t> &f, int &w, int &h, std::vector<uint8_t> &gray, std::string &err)
{
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    if (f.size() < 33 || memcmp(f.data(), sig, 8) != 0) { err = "not a PNG file"; return false; }
    size_t pos = 8;
    int depth = 0, ctype = 0, interlace = 0;
    std::vector<uint8_t> idat, plte;
    bool have_ihdr = false;
    while (pos + 12 <= f.size()) {
        const uint32_t len = be32(&f[pos]);
        const uint8_t *type = &f[pos + 4], *data = &f[pos + 8];
        if (pos + 12 + (size_t)len > f.size()) { err = "truncated PNG chunk"; return false; }
        if (!memcmp(type, "IHDR", 4) && len == 13) {
            w = (int)be32(data); h = (int)be32(data + 4);
            depth = data[8]; ctype = data[9]; interlace = data[12];
            have_ihdr = true;
        } else if (!memcmp(type, "PLTE", 4)) plte.assign(data, data + len);
        else if (!memcmp(type, "IDAT", 4)) idat.insert(idat.end(), data, data + len);
        else if (!memcmp(type, "IEND", 4)) break;
        pos += 12 + len;
    }
    if (!have_ihdr || w <= 0 || h <= 0 || w > 32767 || h > 32767) { err = "bad PNG header"; return false; }
    if (interlace) { err = "interlaced PNG masks are not supported"; return false; }
    const int channels = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
    if (!channels || !(depth == 8 || depth == 16 || (depth < 8 && (ctype == 0 || ctype == 3) && (depth == 1 || depth == 2 || depth == 4))) ||
        (ctype == 3 && depth == 16)) { err = "unsupported PNG colour type / bit depth"; return false; }
    const size_t bpp_bits = (size_t)channels * depth, stride = ((size_t)w * bpp_bits + 7) / 8, bpp = bpp_bits >= 8 ? bpp_bits / 8 : 1;
    std::vector<uint8_t> raw;
    if (!inflate_zlib(idat, raw, (stride + 1) * h) || raw.size() < (stride + 1) * (size_t)h) { err = "PNG data does not inflate"; return false; }
    // undo the scanline filters in place (PNG specification, section 9)
    std::vector<uint8_t> zero(stride, 0);
    for (int y = 0; y < h; y++) {
        uint8_t *cur = &raw[(size_t)y * (stride + 1) + 1];
        const uint8_t *up = y ? &raw[(size_t)(y - 1) * (stride + 1) + 1] : zero.data();
        const int ft = cur[-1];
        for (size_t i = 0; i < stride; i++) {
            const int a = i >= bpp ? cur[i - bpp] : 0, b = up[i], c = i >= bpp ? up[i - bpp] : 0;
            int pred = 0;
            if (ft == 1) pred = a;
            else if (ft == 2) pred = b;
            else if (ft == 3) pred = (a + b) >> 1;
            else if (ft == 4) {
                const int p = a + b - c, pa = p > a ? p - a : a - p, pb = p > b ? p - b : b - p, pc = p > c ? p - c : c - p;
                pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
            } else if (ft != 0) { err = "bad PNG filter type"; return false; }
            cur[i] = (uint8_t)(cur[i] + pred);
        }
    }
    gray.resize((size_t)w * h);
    const int sb = depth == 16 ? 2 : 1;      // bytes per sample; 16-bit samples are reduced to their high byte
    for (int y = 0; y < h; y++) {
        const uint8_t *row = &raw[(size_t)y * (stride + 1) + 1];
        for (int x = 0; x < w; x++) {
            uint8_t v;
            if (depth < 8) {
                const int per = 8 / depth, sh = (per - 1 - (x % per)) * depth;
                const unsigned s = (row[x / per] >> sh) & ((1u << depth) - 1u);
                if (ctype == 3) {
                    if (3 * s + 2 >= plte.size()) { err = "PNG palette index out of range"; return false; }
                    v = bgr2gray(plte[3 * s], plte[3 * s + 1], plte[3 * s + 2]);
                } else v = (uint8_t)(s * 255u / ((1u << depth) - 1u));      // libpng expands low-bit-depth gray to the full 8-bit range
            } else if (ctype == 3) {
                const unsigned s = row[x];
                if (3 * s + 2 >= plte.size()) { err = "PNG palette index out of range"; return false; }
                v = bgr2gray(plte[3 * s], plte[3 * s + 1], plte[3 * s + 2]);
            } else if (ctype == 0 || ctype == 4) v = row[(size_t)x * channels * sb];
            else { const uint8_t *p = row + (size_t)x * channels * sb; v = bgr2gray(p[0], p[sb], p[2 * sb]); }
            gray[(size_t)y * w + x] = v;
        }
    }
    return true;
}

bool decode_pnm(const std::vector<uint8_t> &f, int &w, int &h, std::vector<uint8_t> &gray, std::string &err)
{
    size_t pos = 2;
    auto token = [&](int &v) {
        while (pos < f.size() && (f[pos] == '#' || f[pos] == ' ' || f[pos] == '\n' || f[pos] == '\r' || f[pos] == '\t')) {
            if (f[pos] == '#') while (pos < f.size() && f[pos] != '\n') pos++;
            else pos++;
        }
        v = 0;
        bool any = false;
        while (pos < f.size() && f[pos] >= '0' && f[pos] <= '9') { v = v * 10 + (f[pos++] - '0'); any = true; if (v > 1000000) return false; }
        return any;
    };
    int maxv = 0;
    if (f.size() < 8 || f[0] != 'P' || (f[1] != '5' && f[1] != '6') || !token(w) || !token(h) || !token(maxv) || w <= 0 || h <= 0 || maxv <= 0 || maxv > 255) {
        err = "not a binary PGM / PPM file with maxval <= 255";
        return false;
    }
    pos++;                                            // the single whitespace byte after maxval
    const int ch = f[1] == '6' ? 3 : 1;
    if (f.size() < pos + (size_t)w * h * ch) { err = "truncated PNM file"; return false; }
    gray.resize((size_t)w * h);
    for (size_t i = 0; i < gray.size(); i++) gray[i] = ch == 1 ? f[pos + i] : bgr2gray(f[pos + 3 * i], f[pos + 3 * i + 1], f[pos + 3 * i + 2]);
    return true;
}

thread_local std::string g_mask_err;

} // namespace

extern "C" {

const char *jsorb_mask_image_last_error(void) { return g_mask_err.c_str(); }

int jsorb_read_mask_image(const char *path, int *width, int *height, uint8_t *gray_out, size_t capacity)
{
    if (!path || !width || !height) return JSORB_ERR_INVALID;
    FILE *fp = fopen(path, "rb");
    if (!fp) { g_mask_err = std::string("cannot open ") + path; return JSORB_ERR_STATE; }      // the reference treats an unreadable file as "no mask"
    std::vector<uint8_t> f;
    uint8_t buf[65536];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, fp)) > 0) f.insert(f.end(), buf, buf + n);
    fclose(fp);
    std::vector<uint8_t> gray;
    int w = 0, h = 0;
    std::string err;
    const bool ok = (f.size() >= 2 && f[0] == 'P') ? decode_pnm(f, w, h, gray, err) : decode_png(f, w, h, gray, err);
    if (!ok) { g_mask_err = std::string(path) + ": " + err; return JSORB_ERR_UNSUPPORTED; }
    *width = w; *height = h;
    if (gray_out) {
        if (capacity < gray.size()) { g_mask_err = "destination too small"; return JSORB_ERR_INVALID; }
        memcpy(gray_out, gray.data(), gray.size());
    }
    return JSORB_OK;
}

} // extern "C"
