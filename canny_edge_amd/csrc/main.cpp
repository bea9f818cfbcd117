// main.cpp -- the reference's command line (StevenChang5/Canny_Edge src/main.cpp:18-142) without
// the webcam and the GUI:  ./Main sigma minVal maxVal [-c] [-s] [-i in.pgm|in.jpg] [-o dir] [-p] [-n WxH] [-b dir]
//
// Kept from the reference: the three positionals may appear anywhere relative to the flags
// (src/main.cpp:29-46); exactly three are required, otherwise the usage text is printed and the
// program exits with status 0 (:48-56); maxVal must exceed minVal and both must lie in [0,255]
// (:63-76), again exiting 0 with the reference's messages; -s shows the steps, -c selects the GPU
// entry point (cuda_canny) instead of canny().  In this build both run on the MI355X.
// Replaced: VideoCapture(0) 640x480 (:78-115) -> a binary PGM or a baseline JPEG given with -i (the JPEG is read as
// cv::imread(..., IMREAD_GRAYSCALE) reads it, include/canny_frames.h), or a synthetic frame of the webcam's size
// (-n overrides the size); imshow -> PGM (or, with -p, PNG) files in the -o directory.
#include <algorithm>
#include <cctype>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <filesystem>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

#include "canny_frames.h"
#include "canny_hip.h"
#include "utils.h"
#include "cuda.h"

#define WIDTH 640
#define HEIGHT 480

using namespace std;

static bool read_pgm(const string &path, vector<unsigned char> &px, int &height, int &width)
{
    ifstream f(path, ios::binary);
    if (!f) return false;
    string magic;
    f >> magic;
    if (magic != "P5") return false;
    auto next_int = [&](int &v) {
        f >> ws;
        while (f.peek() == '#') {
            string line;
            getline(f, line);
            f >> ws;
        }
        return (bool)(f >> v);
    };
    int maxv = 0;
    if (!next_int(width) || !next_int(height) || !next_int(maxv)) return false;
    if (width < 1 || height < 1 || maxv < 1 || maxv > 255) return false;
    f.get(); // single whitespace after maxval
    px.resize((size_t)width * height);
    f.read((char *)px.data(), (streamsize)px.size());
    return (size_t)f.gcount() == px.size();
}

// A frame file: binary PGM, or a JPEG (told by its first two bytes, not by its name).
static bool read_frame(const string &path, vector<unsigned char> &px, int &height, int &width)
{
    ifstream f(path, ios::binary);
    if (!f) return false;
    unsigned char magic[2] = {0, 0};
    f.read((char *)magic, 2);
    if (f.gcount() == 2 && magic[0] == 0xFF && magic[1] == 0xD8) {
        f.seekg(0, ios::end);
        const streamoff len = f.tellg();
        if (len < 2) return false; // tellg() failed (-1: a pipe, a directory) or nothing behind the magic
        vector<unsigned char> file((size_t)len);
        f.seekg(0);
        f.read((char *)file.data(), (streamsize)file.size());
        int st = canny_frames_jpeg_info(file.data(), file.size(), &height, &width);
        if (!st) {
            px.resize((size_t)width * height);
            st = canny_frames_jpeg_decode_gray(file.data(), file.size(), px.data(), px.size(), &height, &width);
        }
        if (st) cout << "ERROR: " << path << ": " << canny_frames_last_error() << endl;
        return st == 0;
    }
    f.close();
    return read_pgm(path, px, height, width);
}

static bool is_frame_file(const std::filesystem::path &p)
{
    string ext = p.extension().string();
    transform(ext.begin(), ext.end(), ext.begin(), [](unsigned char c) { return (char)tolower(c); });
    return ext == ".pgm" || ext == ".jpg" || ext == ".jpeg";
}

// Deterministic test card: gray background, filled rectangles, a little noise (xorshift).
static void synthetic_frame(vector<unsigned char> &px, int height, int width)
{
    px.assign((size_t)width * height, 30);
    unsigned s = 42u;
    auto rnd = [&]() {
        s ^= s << 13;
        s ^= s >> 17;
        s ^= s << 5;
        return s;
    };
    int rects = (int)(((long long)width * height) / 8000 + 4);
    for (int i = 0; i < rects; i++) {
        int w = 8 + (int)(rnd() % (unsigned)(width / 6 > 8 ? width / 6 - 7 : 1));
        int h = 8 + (int)(rnd() % (unsigned)(height / 6 > 8 ? height / 6 - 7 : 1));
        int x0 = (int)(rnd() % (unsigned)width), y0 = (int)(rnd() % (unsigned)height);
        unsigned char lv = (unsigned char)(rnd() & 255u);
        for (int y = y0; y < y0 + h && y < height; y++)
            for (int x = x0; x < x0 + w && x < width; x++) px[(size_t)y * width + x] = lv;
    }
    for (auto &p : px) {
        int v = (int)p + (int)(rnd() % 17u) - 8;
        p = (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v));
    }
}

static bool png_output = false; // -p: results as PNG instead of PGM

static bool write_frame(const string &path, const unsigned char *px, int height, int width)
{
    return canny_frames_write_gray(path.c_str(), px, height, width) == CANNY_FRAMES_OK;
}

// -b dir: every *.pgm / *.jpg / *.jpeg of the directory (sorted by name, all of one size) goes through the stream-overlapped
// batch entry point in one call; the 0/255 edge maps come back as bytes and are written as <name>_edges.pgm.
// The reference has no such mode (it loops over webcam frames, src/main.cpp:120-137); SURVEY.md 8(f) item 1.
static int run_batch(const string &dir, const string &outdir, float sigma, int minVal, int maxVal)
{
    namespace fs = std::filesystem;
    vector<fs::path> files;
    error_code ec;
    for (const auto &e : fs::directory_iterator(dir, ec))
        if (e.is_regular_file() && is_frame_file(e.path())) files.push_back(e.path());
    if (ec || files.empty()) {
        cout << "ERROR: no .pgm / .jpg frames in " << dir << endl;
        return -1;
    }
    sort(files.begin(), files.end());
    int height = 0, width = 0;
    vector<unsigned char> frames, one;
    for (size_t i = 0; i < files.size(); i++) {
        int h = 0, w = 0;
        if (!read_frame(files[i].string(), one, h, w) || (i > 0 && (h != height || w != width))) {
            cout << "ERROR: Failed to open " << files[i].string() << " (or its size differs from the first frame)" << endl;
            return -1;
        }
        height = h;
        width = w;
        frames.insert(frames.end(), one.begin(), one.end());
    }
    const size_t frame_px = (size_t)height * width;
    vector<unsigned char> edges(frames.size());
    canny_hip_ctx *ctx = nullptr;
    int st = canny_hip_ctx_create(&ctx, 0);
    const auto t0 = chrono::steady_clock::now();
    if (!st)
        st = canny_hip_canny_batch_u8(ctx, frames.data(), (int)files.size(), sigma, minVal, maxVal, height, width,
                                      edges.data());
    const chrono::duration<double> dt = chrono::steady_clock::now() - t0;
    if (st) {
        fprintf(stderr, "ERROR: %s\n", ctx ? canny_hip_last_error(ctx) : canny_hip_status_string(st));
        if (ctx) canny_hip_ctx_destroy(ctx);
        return 1;
    }
    canny_hip_ctx_destroy(ctx);
    const fs::path out = outdir.empty() ? fs::path(dir) : fs::path(outdir);
    for (size_t i = 0; i < files.size(); i++) {
        const fs::path dst = out / (files[i].stem().string() + (png_output ? "_edges.png" : "_edges.pgm"));
        if (!write_frame(dst.string(), edges.data() + i * frame_px, height, width)) {
            cout << "ERROR: Failed to write " << dst.string() << endl;
            return -1;
        }
    }
    cout << "Execution time: " << dt.count() << " seconds (" << files.size() << " frames of " << width << "x" << height
         << ")\n";
    return 0;
}

int main(int argc, char *argv[])
{
    // The batch pipeline wants its upload, compute and download streams on separate hardware queues; HIP reads this
    // when its runtime initialises (first HIP call, below), and the library leaves the environment alone.
    setenv("GPU_MAX_HW_QUEUES", "8", /*overwrite=*/0);
    float sigma;
    int minVal;
    int maxVal;
    bool use_gpu_entry = false;
    bool show_steps = false;
    string input, outdir, batch_dir;
    int width = WIDTH, height = HEIGHT;
    vector<string> values;

    for (int i = 1; i < argc; i++) {
        string arg = argv[i];
        if (arg == "-c") {
            use_gpu_entry = true;
        } else if (arg == "-s") {
            show_steps = true;
        } else if (arg == "-p") {
            png_output = true;
        } else if (arg == "-i" && i + 1 < argc) {
            input = argv[++i];
        } else if (arg == "-o" && i + 1 < argc) {
            outdir = argv[++i];
        } else if (arg == "-b" && i + 1 < argc) {
            batch_dir = argv[++i];
        } else if (arg == "-n" && i + 1 < argc) {
            if (sscanf(argv[++i], "%dx%d", &width, &height) != 2 || width < 2 || height < 2) {
                fprintf(stderr, "ERROR: -n expects WIDTHxHEIGHT\n");
                exit(0);
            }
        } else {
            values.push_back(arg);
        }
    }

    if (values.size() != 3) {
        fprintf(stderr, "USAGE: %s sigma minVal maxVal\n", argv[0]);
        fprintf(stderr, "   sigma: Standard deviation used for the gaussian blurring kernel\n");
        fprintf(stderr, "   minVal: The minimum threshold value used for hysteresis\n");
        fprintf(stderr, "           Must be in the range of [0,255]\n");
        fprintf(stderr, "   maxVal: The maximum threshold value used for hysteresis\n");
        fprintf(stderr, "           Must be in the range of [0,255]\n");
        fprintf(stderr, "   -c: use the GPU entry point (cuda_canny)   -s: write every step\n");
        fprintf(stderr, "   -i frame: input frame (binary PGM or baseline JPEG)   -n WxH: synthetic frame size   -o dir: output dir\n");
        fprintf(stderr, "   -p: write PNG files instead of PGM\n");
        fprintf(stderr, "   -b dir: run every .pgm / .jpg of dir as one batch, write <name>_edges.pgm\n");
        exit(0);
    }

    try {
        sigma = stof(values[0]);
        minVal = stoi(values[1]);
        maxVal = stoi(values[2]);
    } catch (const exception &) {
        fprintf(stderr, "ERROR: sigma, minVal and maxVal must be numbers\n");
        exit(0);
    }

    if (maxVal <= minVal) {
        fprintf(stderr, "ERROR: minVal must be less than maxVal\n");
        exit(0);
    }
    if (minVal < 0 or minVal > 255) {
        fprintf(stderr, "ERROR: minVal must be in the range of [0,255]");
        exit(0);
    }
    if (maxVal < 0 or maxVal > 255) {
        fprintf(stderr, "ERROR: maxVal must be in the range of [0,255]");
        exit(0);
    }

    if (!batch_dir.empty()) return run_batch(batch_dir, outdir, sigma, minVal, maxVal);

    vector<unsigned char> frame;
    if (!input.empty()) {
        if (!read_frame(input, frame, height, width)) {
            cout << "ERROR: Failed to open " << input << endl;
            return -1;
        }
    } else {
        synthetic_frame(frame, height, width);
    }
    if (!outdir.empty()) setenv("CANNY_OUTPUT_DIR", outdir.c_str(), 1);
    if (png_output) setenv("CANNY_OUTPUT_FORMAT", "png", 1);

    try {
        if (use_gpu_entry)
            cuda_canny(frame.data(), sigma, minVal, maxVal, height, width, show_steps);
        else
            canny(frame.data(), sigma, minVal, maxVal, height, width, show_steps);
    } catch (const exception &e) {
        fprintf(stderr, "ERROR: %s\n", e.what());
        return 1;
    }
    return 0;
}
