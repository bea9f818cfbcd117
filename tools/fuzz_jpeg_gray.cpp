// fuzz_jpeg_gray.cpp -- mutation fuzz of the JPEG frame source under AddressSanitizer / UBSan (CPU only):
//   g++ -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=all -Iinclude tools/fuzz_jpeg_gray.cpp \
//       canny_edge_amd/csrc/jpeg_gray.cpp -o tools/bin/fuzz_jpeg_gray && tools/bin/fuzz_jpeg_gray file.jpg... [rounds]
// Every input file is decoded as is, truncated at every length (short files) or at 2000 random lengths, and with 1..8
// random bytes overwritten, `rounds` times (default 20000).  Any status is fine; a sanitizer report is the failure.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iterator>
#include <vector>

#include "canny_frames.h"

static uint64_t s = 0x9E3779B97F4A7C15ull;
static uint32_t rnd()
{
    s ^= s << 7;
    s ^= s >> 9;
    return (uint32_t)(s >> 16);
}

static int one(const std::vector<unsigned char> &d, size_t n, long stats[4])
{
    int h = 0, w = 0;
    int st = canny_frames_jpeg_info(d.data(), n, &h, &w);
    if (!st && (size_t)h * w <= (size_t)1 << 26) { // mutated headers may claim 65535 x 65535: skip those decodes
        std::vector<unsigned char> out((size_t)h * w);
        st = canny_frames_jpeg_decode_gray(d.data(), n, out.data(), out.size(), &h, &w);
    }
    stats[st & 3]++;
    return st;
}

int main(int argc, char **argv)
{
    long rounds = 20000;
    long stats[4] = {0, 0, 0, 0};
    for (int a = 1; a < argc; a++) {
        std::ifstream f(argv[a], std::ios::binary);
        if (!f) {
            char *e = nullptr;
            rounds = strtol(argv[a], &e, 10);
            continue;
        }
        std::vector<unsigned char> file((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
        one(file, file.size(), stats);
        if (file.size() <= 4096)
            for (size_t n = 0; n < file.size(); n++) one(file, n, stats);
        else
            for (int i = 0; i < 2000; i++) one(file, rnd() % file.size(), stats);
        for (long r = 0; r < rounds; r++) {
            std::vector<unsigned char> m = file;
            const int k = 1 + (int)(rnd() % 8);
            for (int i = 0; i < k; i++) m[rnd() % m.size()] = (unsigned char)rnd();
            one(m, m.size(), stats);
        }
    }
    printf("decodes: ok %ld, arg %ld, format %ld, unsupported %ld -- no sanitizer report\n", stats[0], stats[1], stats[2],
           stats[3]);
    return 0;
}
