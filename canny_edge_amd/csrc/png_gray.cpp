// png_gray.cpp -- 8-bit gray frames to disk: binary PGM or PNG (include/canny_frames.h).
//
// The reference shows its planes in windows (cv::imshow, src/utils.cpp:440-486); headless, they go to files.  PGM is the
// default; PNG is for viewers that do not read PGM.  The PNG is written without compression (zlib "stored" blocks, RFC
// 1950/1951): an edge map is not worth a deflate implementation, and every PNG reader accepts it.
#include "canny_frames.h"

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace {

struct Crc32 {
    uint32_t table[256];
    Crc32()
    {
        for (uint32_t n = 0; n < 256; n++) {
            uint32_t c = n;
            for (int k = 0; k < 8; k++) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            table[n] = c;
        }
    }
    uint32_t run(uint32_t crc, const uint8_t *p, size_t n) const
    {
        for (size_t i = 0; i < n; i++) crc = table[(crc ^ p[i]) & 255u] ^ (crc >> 8);
        return crc;
    }
};

void put32(std::vector<uint8_t> &v, uint32_t x)
{
    for (int s = 24; s >= 0; s -= 8) v.push_back((uint8_t)(x >> s));
}

bool write_chunk(FILE *f, const Crc32 &crc, const char type[4], const std::vector<uint8_t> &body)
{
    std::vector<uint8_t> head;
    put32(head, (uint32_t)body.size());
    head.insert(head.end(), type, type + 4);
    uint32_t c = crc.run(0xFFFFFFFFu, head.data() + 4, 4);
    c = crc.run(c, body.data(), body.size()) ^ 0xFFFFFFFFu;
    std::vector<uint8_t> tail;
    put32(tail, c);
    return fwrite(head.data(), 1, head.size(), f) == head.size() &&
           (body.empty() || fwrite(body.data(), 1, body.size(), f) == body.size()) &&
           fwrite(tail.data(), 1, 4, f) == 4;
}

bool write_png(FILE *f, const unsigned char *px, int height, int width)
{
    static const Crc32 crc;
    static const uint8_t magic[8] = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1A, '\n'};
    if (fwrite(magic, 1, 8, f) != 8) return false;
    std::vector<uint8_t> ihdr;
    put32(ihdr, (uint32_t)width);
    put32(ihdr, (uint32_t)height);
    const uint8_t kind[5] = {8, 0, 0, 0, 0}; // 8 bits, gray, deflate, adaptive filtering (all rows: filter 0), no interlace
    ihdr.insert(ihdr.end(), kind, kind + 5);
    if (!write_chunk(f, crc, "IHDR", ihdr)) return false;

    // the filtered image: every row preceded by its filter type 0; Adler-32 runs over exactly these bytes
    const size_t row = (size_t)width + 1, total = row * (size_t)height;
    uint32_t a = 1, b = 0;
    std::vector<uint8_t> raw;
    raw.reserve(total);
    for (int y = 0; y < height; y++) {
        raw.push_back(0);
        raw.insert(raw.end(), px + (size_t)y * width, px + (size_t)(y + 1) * width);
    }
    for (size_t i = 0; i < total;) { // modulo taken every 5552 bytes, the largest run that cannot overflow 32 bits
        const size_t n = total - i < 5552 ? total - i : 5552;
        for (size_t k = 0; k < n; k++) {
            a += raw[i + k];
            b += a;
        }
        a %= 65521u;
        b %= 65521u;
        i += n;
    }
    // zlib stream of stored blocks, cut into IDAT chunks of at most 1 MiB of payload
    std::vector<uint8_t> z;
    z.reserve(total + total / 65535 * 5 + 16);
    z.push_back(0x78);
    z.push_back(0x01);
    for (size_t i = 0; i < total || i == 0;) {
        const size_t n = total - i < 65535 ? total - i : 65535;
        z.push_back(i + n >= total ? 1 : 0); // BFINAL, BTYPE = 00
        z.push_back((uint8_t)(n & 255));
        z.push_back((uint8_t)(n >> 8));
        z.push_back((uint8_t)(~n & 255));
        z.push_back((uint8_t)((~n >> 8) & 255));
        z.insert(z.end(), raw.begin() + (ptrdiff_t)i, raw.begin() + (ptrdiff_t)(i + n));
        i += n;
        if (n == 0) break;
    }
    put32(z, (b << 16) | a);
    for (size_t i = 0; i < z.size(); i += (size_t)1 << 20) {
        const size_t n = z.size() - i < ((size_t)1 << 20) ? z.size() - i : (size_t)1 << 20;
        if (!write_chunk(f, crc, "IDAT", std::vector<uint8_t>(z.begin() + (ptrdiff_t)i, z.begin() + (ptrdiff_t)(i + n))))
            return false;
    }
    return write_chunk(f, crc, "IEND", {});
}

bool ends_with_png(const char *path)
{
    const size_t n = std::strlen(path);
    if (n < 4) return false;
    const char *e = path + n - 4;
    return e[0] == '.' && (e[1] | 32) == 'p' && (e[2] | 32) == 'n' && (e[3] | 32) == 'g';
}

} // namespace

extern "C" int canny_frames_write_gray(const char *path, const unsigned char *px, int height, int width)
{
    if (!path || !px || height < 1 || width < 1) return CANNY_FRAMES_ERR_ARG;
    FILE *f = std::fopen(path, "wb");
    if (!f) return CANNY_FRAMES_ERR_ARG;
    bool ok;
    if (ends_with_png(path)) {
        ok = write_png(f, px, height, width);
    } else {
        ok = std::fprintf(f, "P5\n%d %d\n255\n", width, height) > 0 &&
             fwrite(px, 1, (size_t)height * width, f) == (size_t)height * width;
    }
    ok = (std::fclose(f) == 0) && ok;
    return ok ? CANNY_FRAMES_OK : CANNY_FRAMES_ERR_FORMAT;
}
