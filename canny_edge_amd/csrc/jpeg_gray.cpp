// jpeg_gray.cpp -- baseline JPEG -> 8-bit gray on the host (include/canny_frames.h).
//
// Stands in for cv::imread(path, IMREAD_GRAYSCALE), which is how the reference's tests obtain their frame
// (StevenChang5/Canny_Edge tests/utils/test_utils.cpp:49); OpenCV asks libjpeg for JCS_GRAYSCALE, i.e. the luminance
// plane straight out of the inverse DCT, no colour conversion.  So only the luminance blocks are reconstructed here;
// the chroma blocks are entropy-decoded (their bits have to be consumed) and dropped.
//
// Format: ITU-T T.81 (sequential DCT, Huffman, 8-bit).  The inverse DCT is the Loeffler-Ligtenberg-Moschytz
// factorisation in 13-bit fixed point with the two-pass scaling libjpeg's default ("islow") method uses; identical
// arithmetic means identical bytes, which tests/test_jpeg_gray.py checks against PIL's libjpeg.
#include "canny_frames.h"

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <exception>
#include <string>
#include <vector>

namespace {

thread_local std::string g_err;

int fail(int code, const std::string &msg)
{
    g_err = msg;
    return code;
}

// position k of the zig-zag scan -> index in the 8x8 block (row*8+col)
const uint8_t kZigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                             41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                             30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct HuffTable {
    bool present = false;
    // canonical code: for length L (1..16) codes run from first_code[L]; values listed in order
    int32_t max_code[18];   // largest code of length L, -1 if none; max_code[17] is a sentinel
    int32_t val_offset[17]; // index of the first value of length L minus its first code
    uint8_t values[256];
    // 9-bit prefix lookup: (length << 8) | value, 0 when the code is longer than 9 bits
    uint16_t fast[512];

    bool build(const uint8_t counts[16], const uint8_t *vals, int n_vals)
    {
        std::memcpy(values, vals, (size_t)n_vals);
        std::memset(fast, 0, sizeof(fast));
        int32_t code = 0;
        int k = 0;
        for (int len = 1; len <= 16; len++) {
            const int n = counts[len - 1];
            val_offset[len] = k - code;
            if (n) {
                if (code + n > (1 << len)) return false; // over-subscribed
                if (len <= 9)
                    for (int i = 0; i < n; i++) {
                        const int first = (code + i) << (9 - len);
                        for (int j = 0; j < (1 << (9 - len)); j++)
                            fast[first + j] = (uint16_t)((len << 8) | values[k + i]);
                    }
                code += n;
                k += n;
                max_code[len] = code - 1;
            } else {
                max_code[len] = -1;
            }
            code <<= 1;
        }
        max_code[17] = 0x7fffffff;
        present = true;
        return true;
    }
};

struct Component {
    int id = 0, h = 1, v = 1, tq = 0;
    int td = 0, ta = 0; // Huffman table selectors of the current scan
    int pred = 0;       // DC predictor
    int blocks_w = 0, blocks_h = 0; // size in 8x8 blocks, padded to whole MCUs
};

// Entropy-coded segment reader: removes the 0xFF00 stuffing, stops at markers.  Past a marker (or the end of the
// buffer) it feeds zero bits and counts them, so a short stream neither runs off the buffer nor goes unnoticed.
struct BitReader {
    const uint8_t *p, *end;
    uint64_t acc = 0;
    int bits = 0;
    int marker = 0; // marker met in the entropy-coded data (0xD0..0xD7, 0xD9, ...), 0 if none yet
    int pad = 0;    // zero bits fed since the last restart

    void fill()
    {
        while (bits <= 56) {
            uint32_t byte = 0;
            bool real = false;
            if (!marker && p < end) {
                byte = *p++;
                real = true;
                if (byte == 0xFF) {
                    while (p < end && *p == 0xFF) p++; // fill bytes
                    if (p < end && *p == 0x00) {
                        p++;
                    } else {
                        marker = p < end ? *p++ : 0xD9;
                        byte = 0;
                        real = false;
                    }
                }
            } else if (!marker) {
                marker = 0xD9; // ran out of data: behave as at EOI
            }
            if (!real) pad += 8;
            acc |= (uint64_t)byte << (56 - bits);
            bits += 8;
        }
    }
    inline uint32_t peek(int n) // n <= 16
    {
        if (bits < n) fill();
        return (uint32_t)(acc >> (64 - n));
    }
    inline void drop(int n)
    {
        acc <<= n;
        bits -= n;
    }
    inline int receive_extend(int n) // T.81 F.2.2.1: n-bit magnitude category -> signed value
    {
        if (n == 0) return 0;
        const int v = (int)peek(n);
        drop(n);
        return v < (1 << (n - 1)) ? v - (1 << n) + 1 : v;
    }
    inline int decode(const HuffTable &t)
    {
        if (bits < 16) fill();
        const uint32_t look = (uint32_t)(acc >> 48); // 16 bits
        const uint16_t f = t.fast[look >> 7];
        if (f) {
            drop(f >> 8);
            return f & 255;
        }
        for (int len = 10; len <= 16; len++) {
            const int32_t code = (int32_t)(look >> (16 - len));
            if (code <= t.max_code[len]) {
                drop(len);
                return t.values[(code + t.val_offset[len]) & 255];
            }
        }
        return -1; // not a code of this table
    }
    // bits are consumed in order, so the zero bits go last: fewer bits left than zero bits fed = some were used
    bool overrun() const { return pad > bits; }
    // restart marker: discard the partial byte, then expect RSTn
    bool restart(int expected)
    {
        if (overrun()) return false;
        acc = 0;
        bits = 0;
        pad = 0;
        if (!marker) { // the marker has not been consumed by the look-ahead: it is next in the stream
            while (p + 1 < end && !(p[0] == 0xFF && p[1] != 0x00 && p[1] != 0xFF)) p++;
            if (p + 1 >= end) return false;
            marker = p[1];
            p += 2;
        }
        if (marker != 0xD0 + expected) return false;
        marker = 0;
        return true;
    }
};

// One 8x8 block: dequantised coefficients (row-major) -> samples, libjpeg's default integer method.
void inverse_dct(const int32_t *coef, uint8_t *out, int stride)
{
    constexpr int kConstBits = 13, kPass1Bits = 2;
    constexpr int64_t c0_298 = 2446, c0_390 = 3196, c0_541 = 4433, c0_765 = 6270, c0_899 = 7373, c1_175 = 9633,
                      c1_501 = 12299, c1_847 = 15137, c1_961 = 16069, c2_053 = 16819, c2_562 = 20995, c3_072 = 25172;
    int64_t ws[64];
    // one 1-D transform; `sh` is the descale shift of the pass
    auto lane = [&](const int64_t in[8], int64_t res[8], int sh) {
        // even part: inputs 0, 2, 4, 6
        int64_t z1 = (in[2] + in[6]) * c0_541;
        const int64_t e2 = z1 - in[6] * c1_847;
        const int64_t e3 = z1 + in[2] * c0_765;
        const int64_t e0 = (in[0] + in[4]) * ((int64_t)1 << kConstBits);
        const int64_t e1 = (in[0] - in[4]) * ((int64_t)1 << kConstBits);
        const int64_t a0 = e0 + e3, a3 = e0 - e3, a1 = e1 + e2, a2 = e1 - e2;
        // odd part: inputs 7, 5, 3, 1
        int64_t o0 = in[7], o1 = in[5], o2 = in[3], o3 = in[1];
        z1 = o0 + o3;
        int64_t z2 = o1 + o2, z3 = o0 + o2, z4 = o1 + o3;
        const int64_t z5 = (z3 + z4) * c1_175;
        o0 *= c0_298;
        o1 *= c2_053;
        o2 *= c3_072;
        o3 *= c1_501;
        z1 *= -c0_899;
        z2 *= -c2_562;
        z3 = z3 * -c1_961 + z5;
        z4 = z4 * -c0_390 + z5;
        o0 += z1 + z3;
        o1 += z2 + z4;
        o2 += z2 + z3;
        o3 += z1 + z4;
        const int64_t half = (int64_t)1 << (sh - 1);
        res[0] = (a0 + o3 + half) >> sh;
        res[7] = (a0 - o3 + half) >> sh;
        res[1] = (a1 + o2 + half) >> sh;
        res[6] = (a1 - o2 + half) >> sh;
        res[2] = (a2 + o1 + half) >> sh;
        res[5] = (a2 - o1 + half) >> sh;
        res[3] = (a3 + o0 + half) >> sh;
        res[4] = (a3 - o0 + half) >> sh;
    };
    int64_t in[8], res[8];
    for (int c = 0; c < 8; c++) { // columns, results kept scaled up by 2^kPass1Bits
        for (int r = 0; r < 8; r++) in[r] = coef[r * 8 + c];
        lane(in, res, kConstBits - kPass1Bits);
        for (int r = 0; r < 8; r++) ws[r * 8 + c] = res[r];
    }
    for (int r = 0; r < 8; r++) { // rows; the extra 3 bits remove the 8x of the 2-D transform
        lane(ws + r * 8, res, kConstBits + kPass1Bits + 3);
        for (int c = 0; c < 8; c++) {
            // level shift and clamp.  (libjpeg's C code clamps through a lookup table that wraps beyond +-512, its SIMD
            // code saturates; the two only part on coefficients no encoder produces -- this follows the SIMD code.)
            const int64_t v = res[c] + 128;
            out[r * stride + c] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
        }
    }
}

struct Decoder {
    const uint8_t *data;
    size_t size;
    int width = 0, height = 0;
    std::vector<Component> comps;
    uint16_t quant[4][64]; // row-major (de-zigzagged)
    bool quant_present[4] = {false, false, false, false};
    HuffTable dc[4], ac[4];
    int restart_interval = 0;
    int hmax = 1, vmax = 1;
    bool adobe = false;
    int adobe_transform = -1;
    bool have_frame = false;

    static int be16(const uint8_t *p) { return (p[0] << 8) | p[1]; }

    int parse_dqt(const uint8_t *p, int len)
    {
        while (len > 0) {
            const int pq = p[0] >> 4, tq = p[0] & 15;
            if (tq > 3 || pq > 1) return fail(CANNY_FRAMES_ERR_FORMAT, "bad quantisation table header");
            const int need = 1 + 64 * (pq + 1);
            if (len < need) return fail(CANNY_FRAMES_ERR_FORMAT, "short quantisation table");
            for (int k = 0; k < 64; k++)
                quant[tq][kZigzag[k]] = (uint16_t)(pq ? be16(p + 1 + 2 * k) : p[1 + k]);
            quant_present[tq] = true;
            p += need;
            len -= need;
        }
        return CANNY_FRAMES_OK;
    }

    int parse_dht(const uint8_t *p, int len)
    {
        while (len > 0) {
            if (len < 17) return fail(CANNY_FRAMES_ERR_FORMAT, "short Huffman table");
            const int tc = p[0] >> 4, th = p[0] & 15;
            if (tc > 1 || th > 3) return fail(CANNY_FRAMES_ERR_FORMAT, "bad Huffman table header");
            int n = 0;
            for (int i = 0; i < 16; i++) n += p[1 + i];
            if (n > 256 || len < 17 + n) return fail(CANNY_FRAMES_ERR_FORMAT, "bad Huffman table size");
            if (!(tc ? ac[th] : dc[th]).build(p + 1, p + 17, n))
                return fail(CANNY_FRAMES_ERR_FORMAT, "over-subscribed Huffman table");
            p += 17 + n;
            len -= 17 + n;
        }
        return CANNY_FRAMES_OK;
    }

    int parse_sof(const uint8_t *p, int len)
    {
        if (len < 6) return fail(CANNY_FRAMES_ERR_FORMAT, "short frame header");
        if (have_frame) return fail(CANNY_FRAMES_ERR_FORMAT, "second frame header");
        if (p[0] != 8) return fail(CANNY_FRAMES_ERR_UNSUPPORTED, "only 8-bit samples are supported");
        height = be16(p + 1);
        width = be16(p + 3);
        const int n = p[5];
        if (height == 0) return fail(CANNY_FRAMES_ERR_UNSUPPORTED, "height given by a DNL marker is not supported");
        if (width == 0) return fail(CANNY_FRAMES_ERR_FORMAT, "zero width");
        if (n != 1 && n != 3)
            return fail(CANNY_FRAMES_ERR_UNSUPPORTED, "only grayscale and YCbCr files are supported (" +
                                                          std::to_string(n) + " components)");
        if (len < 6 + 3 * n) return fail(CANNY_FRAMES_ERR_FORMAT, "short frame header");
        comps.resize((size_t)n);
        for (int i = 0; i < n; i++) {
            Component &c = comps[(size_t)i];
            c.id = p[6 + 3 * i];
            c.h = p[7 + 3 * i] >> 4;
            c.v = p[7 + 3 * i] & 15;
            c.tq = p[8 + 3 * i];
            if (c.h < 1 || c.h > 4 || c.v < 1 || c.v > 4 || c.tq > 3)
                return fail(CANNY_FRAMES_ERR_FORMAT, "bad component description");
            hmax = c.h > hmax ? c.h : hmax;
            vmax = c.v > vmax ? c.v : vmax;
        }
        if (n == 1) { // a single component is never interleaved: its sampling factors carry no meaning
            comps[0].h = comps[0].v = 1;
            hmax = vmax = 1;
        }
        if (comps[0].h != hmax || comps[0].v != vmax)
            return fail(CANNY_FRAMES_ERR_UNSUPPORTED, "subsampled luminance plane");
        const int mcux = (width + 8 * hmax - 1) / (8 * hmax), mcuy = (height + 8 * vmax - 1) / (8 * vmax);
        for (Component &c : comps) {
            c.blocks_w = mcux * c.h;
            c.blocks_h = mcuy * c.v;
        }
        have_frame = true;
        return CANNY_FRAMES_OK;
    }

    // Walks the marker segments up to (and including) the first SOS when `stop_at_frame` is false.
    // On CANNY_FRAMES_OK with *scan set, *scan points at the SOS segment's payload.
    int headers(size_t &pos, bool stop_at_frame, const uint8_t **scan, int *scan_len)
    {
        *scan = nullptr;
        while (true) {
            while (pos < size && data[pos] != 0xFF) pos++; // tolerate garbage between segments as libjpeg does
            while (pos < size && data[pos] == 0xFF) pos++;
            if (pos >= size) return fail(CANNY_FRAMES_ERR_FORMAT, "truncated file (no image data)");
            const int m = data[pos++];
            if (m == 0x00 || m == 0xD8 || m == 0x01 || (m >= 0xD0 && m <= 0xD7)) continue; // stuffed byte / no payload
            if (m == 0xD9) return fail(CANNY_FRAMES_ERR_FORMAT, "end of image before any scan");
            if (pos + 2 > size) return fail(CANNY_FRAMES_ERR_FORMAT, "truncated marker segment");
            const int len = be16(data + pos) - 2;
            if (len < 0 || pos + 2 + (size_t)len > size) return fail(CANNY_FRAMES_ERR_FORMAT, "truncated marker segment");
            const uint8_t *p = data + pos + 2;
            pos += 2 + (size_t)len;
            int st = CANNY_FRAMES_OK;
            switch (m) {
            case 0xC0: // baseline
            case 0xC1: // extended sequential, Huffman
                st = parse_sof(p, len);
                if (!st && stop_at_frame) return st;
                break;
            case 0xC2: return fail(CANNY_FRAMES_ERR_UNSUPPORTED, "progressive JPEG is not supported");
            case 0xC3: case 0xC5: case 0xC6: case 0xC7: case 0xC9: case 0xCA: case 0xCB: case 0xCD: case 0xCE: case 0xCF:
                return fail(CANNY_FRAMES_ERR_UNSUPPORTED, "lossless, hierarchical and arithmetic-coded JPEG are not supported");
            case 0xC4: st = parse_dht(p, len); break;
            case 0xDB: st = parse_dqt(p, len); break;
            case 0xDD:
                if (len < 2) return fail(CANNY_FRAMES_ERR_FORMAT, "short restart-interval segment");
                restart_interval = be16(p);
                break;
            case 0xEE: // Adobe: says whether three components are YCbCr (1) or RGB (0)
                if (len >= 12 && !std::memcmp(p, "Adobe", 5)) {
                    adobe = true;
                    adobe_transform = p[11];
                }
                break;
            case 0xDA:
                if (!have_frame) return fail(CANNY_FRAMES_ERR_FORMAT, "scan before the frame header");
                *scan = p;
                *scan_len = len;
                return CANNY_FRAMES_OK;
            default: break; // APPn, COM, ...
            }
            if (st) return st;
        }
    }

    int check_colour_space()
    {
        if (comps.size() != 3) return CANNY_FRAMES_OK;
        // libjpeg's rules (jdapimin.c default_decompress_parms): Adobe transform 0 -> RGB; without JFIF/Adobe markers,
        // component ids 'R','G','B' -> RGB; everything else YCbCr
        const bool rgb_ids = comps[0].id == 'R' && comps[1].id == 'G' && comps[2].id == 'B';
        if ((adobe && adobe_transform == 0) || (!adobe && rgb_ids))
            return fail(CANNY_FRAMES_ERR_UNSUPPORTED, "RGB-coded JPEG is not supported");
        return CANNY_FRAMES_OK;
    }

    // Entropy-decodes one block.  `coef` (64, zeroed by the caller) is filled for the luminance component only.
    int decode_block(BitReader &br, Component &c, int32_t *coef)
    {
        const HuffTable &tdc = dc[c.td], &tac = ac[c.ta];
        int s = br.decode(tdc);
        if (s < 0 || s > 11) return fail(CANNY_FRAMES_ERR_FORMAT, "bad DC code");
        c.pred = (int)((unsigned)c.pred + (unsigned)br.receive_extend(s)); // wraps on a hostile stream, never overflows
        const uint16_t *q = quant[c.tq];
        if (coef) coef[0] = (int32_t)(uint32_t)((int64_t)c.pred * q[0]);
        for (int k = 1; k < 64;) {
            const int rs = br.decode(tac);
            if (rs < 0) return fail(CANNY_FRAMES_ERR_FORMAT, "bad AC code");
            const int run = rs >> 4, cat = rs & 15;
            if (cat == 0) {
                if (run != 15) break; // end of block
                k += 16;
                continue;
            }
            k += run;
            if (k > 63) return fail(CANNY_FRAMES_ERR_FORMAT, "AC run past the end of the block");
            const int v = br.receive_extend(cat);
            if (coef) coef[kZigzag[k]] = v * (int32_t)q[kZigzag[k]];
            k++;
        }
        return CANNY_FRAMES_OK;
    }

    int decode(uint8_t *out)
    {
        size_t pos = 0;
        if (size < 4 || data[0] != 0xFF || data[1] != 0xD8) return fail(CANNY_FRAMES_ERR_FORMAT, "not a JPEG file");
        std::vector<uint8_t> plane; // luminance, padded to whole MCUs
        int stride = 0;
        bool luma_done = false;
        while (!luma_done) {
            const uint8_t *scan;
            int scan_len = 0;
            int st = headers(pos, false, &scan, &scan_len);
            if (st) return st;
            if (plane.empty()) {
                st = check_colour_space();
                if (st) return st;
                stride = comps[0].blocks_w * 8;
                // Untrusted header: every 8x8 block costs at least one bit of entropy-coded data per component, so a
                // file cannot describe more than 8 * size luminance blocks -- refuse before allocating gigabytes for
                // a 65535 x 65535 header on a tiny file (and before decoding its phantom blocks).
                if ((unsigned long long)comps[0].blocks_w * comps[0].blocks_h > 8ull * size)
                    return fail(CANNY_FRAMES_ERR_FORMAT, "frame size implausible for the file size (damaged header)");
                plane.assign((size_t)stride * comps[0].blocks_h * 8, 0);
            }
            // scan header: which components, which tables
            if (scan_len < 1) return fail(CANNY_FRAMES_ERR_FORMAT, "short scan header");
            const int ns = scan[0];
            if (ns < 1 || ns > (int)comps.size() || scan_len < 1 + 2 * ns + 3)
                return fail(CANNY_FRAMES_ERR_FORMAT, "bad scan header");
            std::vector<Component *> in_scan;
            for (int i = 0; i < ns; i++) {
                Component *c = nullptr;
                for (Component &k : comps)
                    if (k.id == scan[1 + 2 * i]) c = &k;
                if (!c) return fail(CANNY_FRAMES_ERR_FORMAT, "scan names an unknown component");
                c->td = scan[2 + 2 * i] >> 4;
                c->ta = scan[2 + 2 * i] & 15;
                if (c->td > 3 || c->ta > 3 || !dc[c->td].present || !ac[c->ta].present)
                    return fail(CANNY_FRAMES_ERR_FORMAT, "scan uses a Huffman table the file does not define");
                if (!quant_present[c->tq]) return fail(CANNY_FRAMES_ERR_FORMAT, "missing quantisation table");
                c->pred = 0;
                in_scan.push_back(c);
            }
            const uint8_t *tail = scan + 1 + 2 * ns;
            if (tail[0] != 0 || tail[1] != 63 || tail[2] != 0)
                return fail(CANNY_FRAMES_ERR_UNSUPPORTED, "spectral selection / successive approximation in a sequential file");

            BitReader br;
            br.p = data + pos;
            br.end = data + size;
            int32_t coef[64];
            int restarts_left = restart_interval, next_rst = 0;
            auto maybe_restart = [&]() -> bool {
                if (!restart_interval) return true;
                if (restarts_left == 0) {
                    if (!br.restart(next_rst)) return false;
                    next_rst = (next_rst + 1) & 7;
                    restarts_left = restart_interval;
                    for (Component *c : in_scan) c->pred = 0;
                }
                restarts_left--;
                return true;
            };
            if (ns == 1) {
                // non-interleaved: the component's own blocks in raster order, only those that cover the image
                Component &c = *in_scan[0];
                const bool luma = &c == &comps[0];
                const int cw = (width * c.h + hmax - 1) / hmax, ch = (height * c.v + vmax - 1) / vmax;
                const int bw = (cw + 7) / 8, bh = (ch + 7) / 8;
                for (int by = 0; by < bh; by++) {
                    if (br.overrun())
                        return fail(CANNY_FRAMES_ERR_FORMAT, "entropy-coded data ends early (truncated or damaged file)");
                    for (int bx = 0; bx < bw; bx++) {
                        if (!maybe_restart()) return fail(CANNY_FRAMES_ERR_FORMAT, "missing restart marker (truncated or damaged file)");
                        if (luma) std::memset(coef, 0, sizeof(coef));
                        st = decode_block(br, c, luma ? coef : nullptr);
                        if (st) return st;
                        if (luma) inverse_dct(coef, plane.data() + (size_t)by * 8 * stride + bx * 8, stride);
                    }
                }
                luma_done = luma_done || luma;
            } else {
                const int mcux = comps[0].blocks_w / comps[0].h, mcuy = comps[0].blocks_h / comps[0].v;
                bool has_luma = false;
                for (Component *c : in_scan) has_luma = has_luma || c == &comps[0];
                for (int my = 0; my < mcuy; my++) {
                    if (br.overrun()) // once per MCU row: a cut file stops here, not after a scan of zero-bit blocks
                        return fail(CANNY_FRAMES_ERR_FORMAT, "entropy-coded data ends early (truncated or damaged file)");
                    for (int mx = 0; mx < mcux; mx++) {
                        if (!maybe_restart()) return fail(CANNY_FRAMES_ERR_FORMAT, "missing restart marker (truncated or damaged file)");
                        for (Component *c : in_scan) {
                            const bool luma = c == &comps[0];
                            for (int v = 0; v < c->v; v++)
                                for (int h = 0; h < c->h; h++) {
                                    if (luma) std::memset(coef, 0, sizeof(coef));
                                    st = decode_block(br, *c, luma ? coef : nullptr);
                                    if (st) return st;
                                    if (luma)
                                        inverse_dct(coef,
                                                    plane.data() + (size_t)(my * c->v + v) * 8 * stride +
                                                        (size_t)(mx * c->h + h) * 8,
                                                    stride);
                                }
                        }
                    }
                }
                luma_done = luma_done || has_luma;
            }
            if (br.overrun())
                return fail(CANNY_FRAMES_ERR_FORMAT, "entropy-coded data ends early (truncated or damaged file)");
            // continue after the entropy-coded data: at the marker the reader met, or search for the next one
            pos = (size_t)(br.p - data);
            if (br.marker) pos -= 2; // re-read it as a segment marker
        }
        for (int y = 0; y < height; y++) std::memcpy(out + (size_t)y * width, plane.data() + (size_t)y * stride, (size_t)width);
        return CANNY_FRAMES_OK;
    }
};

} // namespace

extern "C" int canny_frames_jpeg_info(const void *data, size_t bytes, int *height, int *width)
{
    g_err.clear();
    if (!data || !height || !width) return fail(CANNY_FRAMES_ERR_ARG, "null argument");
    Decoder d;
    d.data = (const uint8_t *)data;
    d.size = bytes;
    if (bytes < 4 || d.data[0] != 0xFF || d.data[1] != 0xD8) return fail(CANNY_FRAMES_ERR_FORMAT, "not a JPEG file");
    size_t pos = 0;
    const uint8_t *scan;
    int scan_len;
    const int st = d.headers(pos, true, &scan, &scan_len);
    if (st) return st;
    *height = d.height;
    *width = d.width;
    return CANNY_FRAMES_OK;
}

extern "C" int canny_frames_jpeg_decode_gray(const void *data, size_t bytes, unsigned char *out, size_t out_bytes,
                                             int *height, int *width)
{
    int h = 0, w = 0;
    int st = canny_frames_jpeg_info(data, bytes, &h, &w);
    if (st) return st;
    if (!out || out_bytes < (size_t)h * (size_t)w)
        return fail(CANNY_FRAMES_ERR_ARG, "output buffer too small for " + std::to_string(w) + "x" + std::to_string(h));
    try {
        Decoder d;
        d.data = (const uint8_t *)data;
        d.size = bytes;
        st = d.decode(out);
    } catch (const std::exception &e) { // bad_alloc on a 65535 x 65535 header
        return fail(CANNY_FRAMES_ERR_FORMAT, std::string("decoder: ") + e.what());
    }
    if (st) return st;
    *height = h;
    *width = w;
    return CANNY_FRAMES_OK;
}

extern "C" const char *canny_frames_last_error(void) { return g_err.c_str(); }
