// hostio.hpp -- the small amount of file I/O the wass_stereo boundary needs, without OpenCV / Boost:
//   logger            src/include/log.hpp:105-168        "<scope> [info ] message" to stdout and <wd>/wass_stereo_log.txt
//   OpenCV XML matrix src/include/utils.hpp:32-66        first top-level node of a FileStorage XML
//   matrix .txt       src/include/utils.hpp:69-92        precision(16), scientific, no trailing newline (App. B.4)
//   PNG gray8         cv::imread(IMREAD_GRAYSCALE)       wass_stereo.cpp:393,396  (zlib inflate + PNG unfilter)
//   PLY / xyzbin      src/wass_stereo/PovMesh.cpp:346-375,463-517 (App. B.2, B.6)
#pragma once

#include <dlfcn.h>
#include <fcntl.h>
#include <unistd.h>
#include <zlib.h>

#include <cerrno>
#include <charconv>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace wasshost {

// ------------------------------------------------------------------ logger
struct LogState {
    std::string scope;
    std::unique_ptr<std::ofstream> file;
    // set: lines are collected here instead of going to stdout / the file.  The pipelined sequence driver works on one frame
    // from three threads at different times (decode, GPU submission, output); each binds the frame's buffer while it does,
    // and the frame's wass_stereo_log.txt is written once, in order, when the frame is finished.
    std::string* sink = nullptr;
    static LogState& get() { static thread_local LogState s; return s; }     // per thread: a sequence driver runs frames on several threads
};
struct LogSinkScope {                      // binds a frame's log buffer to the calling thread for the lifetime of the object
    std::string* prev_sink; std::string prev_scope;
    explicit LogSinkScope(std::string* s) : prev_sink(LogState::get().sink), prev_scope(LogState::get().scope) { LogState::get().sink = s; }
    ~LogSinkScope() { LogState::get().sink = prev_sink; LogState::get().scope = prev_scope; }
};
inline void setup_logger(const std::string& filename = std::string())
{
    if (!filename.empty()) LogState::get().file.reset(new std::ofstream(filename.c_str()));
}
class LogLine {
public:
    explicit LogLine(const char* sev) { emit(LogState::get().scope + " [" + sev + "] "); }
    ~LogLine()
    {
        auto& st = LogState::get();
        if (st.sink) { st.sink->push_back('\n'); return; }
        std::cout << std::endl; auto& f = st.file; if (f) (*f) << std::endl;
    }
    template <typename T> LogLine& operator<<(const T& v) { std::ostringstream os; os << v; emit(os.str()); return *this; }
private:
    static void emit(const std::string& s)
    {
        auto& st = LogState::get();
        if (st.sink) { st.sink->append(s); return; }
        std::cout << s; auto& f = st.file; if (f) (*f) << s;
    }
};
#define WLOG_SCOPE(name) (wasshost::LogState::get().scope = std::string(name))
#define WLOGI wasshost::LogLine("info ")
#define WLOGE wasshost::LogLine("error")

// ------------------------------------------------------------------ matrices
struct Mat {
    int rows = 0, cols = 0;
    std::vector<double> d;
    Mat() {}
    Mat(int r, int c) : rows(r), cols(c), d((size_t)r * c, 0.0) {}
    double& operator()(int i, int j) { return d[(size_t)i * cols + j]; }
    double operator()(int i, int j) const { return d[(size_t)i * cols + j]; }
    bool empty() const { return d.empty(); }
    static Mat eye(int n) { Mat m(n, n); for (int i = 0; i < n; ++i) m(i, i) = 1; return m; }
};
inline Mat matmul(const Mat& a, const Mat& b)
{
    Mat r(a.rows, b.cols);
    for (int i = 0; i < a.rows; ++i)
        for (int j = 0; j < b.cols; ++j) { double s = 0; for (int k = 0; k < a.cols; ++k) s += a(i, k) * b(k, j); r(i, j) = s; }
    return r;
}
inline Mat transpose(const Mat& a) { Mat r(a.cols, a.rows); for (int i = 0; i < a.rows; ++i) for (int j = 0; j < a.cols; ++j) r(j, i) = a(i, j); return r; }
inline Mat scaled(const Mat& a, double s) { Mat r = a; for (auto& v : r.d) v *= s; return r; }
inline double det3(const Mat& m)
{
    return m(0, 0) * (m(1, 1) * m(2, 2) - m(1, 2) * m(2, 1)) - m(0, 1) * (m(1, 0) * m(2, 2) - m(1, 2) * m(2, 0)) +
           m(0, 2) * (m(1, 0) * m(2, 1) - m(1, 1) * m(2, 0));
}
inline Mat inv3(const Mat& m)
{
    const double d = 1.0 / det3(m);
    Mat r(3, 3);
    r(0, 0) = (m(1, 1) * m(2, 2) - m(1, 2) * m(2, 1)) * d; r(0, 1) = (m(0, 2) * m(2, 1) - m(0, 1) * m(2, 2)) * d; r(0, 2) = (m(0, 1) * m(1, 2) - m(0, 2) * m(1, 1)) * d;
    r(1, 0) = (m(1, 2) * m(2, 0) - m(1, 0) * m(2, 2)) * d; r(1, 1) = (m(0, 0) * m(2, 2) - m(0, 2) * m(2, 0)) * d; r(1, 2) = (m(0, 2) * m(1, 0) - m(0, 0) * m(1, 2)) * d;
    r(2, 0) = (m(1, 0) * m(2, 1) - m(1, 1) * m(2, 0)) * d; r(2, 1) = (m(0, 1) * m(2, 0) - m(0, 0) * m(2, 1)) * d; r(2, 2) = (m(0, 0) * m(1, 1) - m(0, 1) * m(1, 0)) * d;
    return r;
}

// OpenCV FileStorage XML: <opencv_storage><NAME type_id="opencv-matrix"><rows>..<cols>..<dt>d</dt><data>..</data></NAME>
inline Mat load_matrix_xml(const std::string& filename)
{
    std::ifstream ifs(filename.c_str());
    if (!ifs.is_open()) { WLOGE << "Unable to load " << filename; return Mat(); }
    std::stringstream ss; ss << ifs.rdbuf();
    const std::string s = ss.str();
    auto tag = [&](const std::string& name, size_t from, std::string& out) -> size_t {
        const size_t a = s.find("<" + name + ">", from);
        if (a == std::string::npos) return std::string::npos;
        const size_t b = s.find("</" + name + ">", a);
        if (b == std::string::npos) return std::string::npos;
        out = s.substr(a + name.size() + 2, b - a - name.size() - 2);
        return b;
    };
    const size_t root = s.find("<opencv_storage>");
    if (root == std::string::npos) { WLOGE << filename << " is not an OpenCV storage file"; return Mat(); }
    std::string rows, cols, dt, data;
    size_t p = tag("rows", root, rows);                       // first top-level node = first <rows> after the root
    if (p == std::string::npos || tag("cols", root, cols) == std::string::npos || tag("dt", root, dt) == std::string::npos ||
        tag("data", root, data) == std::string::npos) { WLOGE << filename << ": no opencv-matrix node"; return Mat(); }
    Mat m(std::stoi(rows), std::stoi(cols));
    std::istringstream ds(data);
    for (auto& v : m.d) if (!(ds >> v)) { WLOGE << filename << ": matrix data truncated"; return Mat(); }
    return m;
}
inline bool save_matrix_xml(const std::string& filename, const std::string& node, const Mat& m)   // test fixtures
{
    std::ofstream ofs(filename.c_str());
    if (ofs.fail()) return false;
    ofs << "<?xml version=\"1.0\"?>\n<opencv_storage>\n<" << node << " type_id=\"opencv-matrix\">\n  <rows>" << m.rows << "</rows>\n  <cols>"
        << m.cols << "</cols>\n  <dt>d</dt>\n  <data>\n   ";
    ofs << std::setprecision(17);
    for (double v : m.d) ofs << " " << v;
    ofs << "</data></" << node << ">\n</opencv_storage>\n";
    return true;
}
// The small text files a frame's PREPARATION leaves in its workdir (stereo_config.txt, P0cam.txt ... scale.txt, H0_rect.txt) go through
// commit_text_file: written at once -- or, while a DeferredFiles sink is installed for the thread, kept in memory until the caller writes
// them (the resident worker prepares and computes a frame BEFORE anybody asked for it and must not touch its workdir until then:
// stereo_server.hpp, "speculation").
struct DeferredFiles {
    std::vector<std::pair<std::string, std::string>> files;
    bool write_all() const
    {
        bool ok = true;
        for (const auto& f : files) { std::ofstream ofs(f.first.c_str(), std::ios::binary); ofs.write(f.second.data(), (std::streamsize)f.second.size()); ok = ok && !ofs.fail(); }
        return ok;
    }
};
inline DeferredFiles*& deferred_files_sink() { static thread_local DeferredFiles* sink = nullptr; return sink; }
struct DeferredFilesScope {
    DeferredFiles* prev;
    explicit DeferredFilesScope(DeferredFiles* s) : prev(deferred_files_sink()) { deferred_files_sink() = s; }
    ~DeferredFilesScope() { deferred_files_sink() = prev; }
};
inline bool commit_text_file(const std::string& filename, const std::string& content)
{
    if (DeferredFiles* s = deferred_files_sink()) { s->files.emplace_back(filename, content); return true; }
    std::ofstream ofs(filename.c_str(), std::ios::binary);
    if (ofs.fail()) return false;
    ofs.write(content.data(), (std::streamsize)content.size());
    return !ofs.fail();
}
inline bool save_matrix_txt(const std::string& filename, const Mat& m)
{
    std::ostringstream ofs;
    ofs.precision(16);
    ofs << std::scientific;
    for (int i = 0; i < m.rows; ++i) {
        for (int j = 0; j < m.cols; ++j) { ofs << m(i, j); if (j != m.cols - 1) ofs << " "; }
        if (i != m.rows - 1) ofs << std::endl;
    }
    return commit_text_file(filename, ofs.str());
}

// ------------------------------------------------------------------ zlib streams, fast
// A frame's host time is mostly zlib: two 5-megapixel PNGs inflated (2 x 25 ms with libz), two previews deflated.  libdeflate
// does the same streams 2-3 times faster; the image ships its runtime (libdeflate.so.0) but no header, so the handful of entry
// points used here are declared locally and resolved with dlopen -- and libz remains the fallback when the library is not
// there (WASS_NO_LIBDEFLATE=1 forces it).  Same bytes out of inflate either way; deflate output differs between the two
// (any valid stream is a valid PNG), which is why only files whose PIXELS matter are written through it.
struct FastZ {
    void* (*alloc_d)() = nullptr;
    int (*zlib_d)(void*, const void*, size_t, void*, size_t, size_t*) = nullptr;
    void (*free_d)(void*) = nullptr;
    void* (*alloc_c)(int) = nullptr;
    size_t (*zlib_c)(void*, const void*, size_t, void*, size_t) = nullptr;
    size_t (*bound_c)(void*, size_t) = nullptr;
    void (*free_c)(void*) = nullptr;
    bool ok = false;
    FastZ()
    {
        const char* off = getenv("WASS_NO_LIBDEFLATE");
        if (off && atoi(off) != 0) return;
        void* h = nullptr;
        for (const char* n : { "libdeflate.so.0", "libdeflate.so" }) if ((h = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
        if (!h) return;
        alloc_d = (void* (*)())dlsym(h, "libdeflate_alloc_decompressor");
        zlib_d = (int (*)(void*, const void*, size_t, void*, size_t, size_t*))dlsym(h, "libdeflate_zlib_decompress");
        free_d = (void (*)(void*))dlsym(h, "libdeflate_free_decompressor");
        alloc_c = (void* (*)(int))dlsym(h, "libdeflate_alloc_compressor");
        zlib_c = (size_t (*)(void*, const void*, size_t, void*, size_t))dlsym(h, "libdeflate_zlib_compress");
        bound_c = (size_t (*)(void*, size_t))dlsym(h, "libdeflate_zlib_compress_bound");
        free_c = (void (*)(void*))dlsym(h, "libdeflate_free_compressor");
        ok = alloc_d && zlib_d && free_d && alloc_c && zlib_c && bound_c && free_c;
    }
    static const FastZ& get() { static const FastZ z; return z; }
};
// a whole zlib stream -> exactly outn bytes
inline bool zlib_inflate_exact(const uint8_t* in, size_t n, uint8_t* out, size_t outn)
{
    const FastZ& z = FastZ::get();
    if (z.ok) {
        struct D { const FastZ& z; void* d; ~D() { if (d) z.free_d(d); } };
        static thread_local D dec{ z, nullptr };                   // a decompressor is not thread-safe; one per thread, reused
        if (!dec.d) dec.d = z.alloc_d();
        if (dec.d) {
            size_t got = 0;
            if (z.zlib_d(dec.d, in, n, out, outn, &got) == 0 && got == outn) return true;
            // (a stream libdeflate refuses -- e.g. one with a preset dictionary -- gets its second chance below)
        }
    }
    uLongf len = (uLongf)outn;
    return uncompress(out, &len, in, (uLong)n) == Z_OK && len == outn;
}
// bytes -> a zlib stream; level 1 .. 9.  NOT byte-stable across the two libraries: for files whose pixels matter, not their bytes.
inline bool zlib_deflate_fast(const uint8_t* in, size_t n, int level, std::vector<uint8_t>& out)
{
    const FastZ& z = FastZ::get();
    if (z.ok) {
        if (void* c = z.alloc_c(level)) {
            out.resize(z.bound_c(c, n));
            const size_t got = z.zlib_c(c, in, n, out.data(), out.size());
            z.free_c(c);
            if (got) { out.resize(got); return true; }
        }
    }
    uLongf clen = compressBound((uLong)n);
    out.resize(clen);
    if (compress2(out.data(), &clen, in, (uLong)n, level) != Z_OK) return false;
    out.resize(clen);
    return true;
}
inline bool read_whole_file(const std::string& filename, std::vector<uint8_t>& f)
{
    FILE* fp = fopen(filename.c_str(), "rb");
    if (!fp) return false;
    bool ok = fseek(fp, 0, SEEK_END) == 0;
    const long n = ok ? ftell(fp) : -1;
    ok = ok && n >= 0 && fseek(fp, 0, SEEK_SET) == 0;
    if (ok) { f.resize((size_t)n); ok = n == 0 || fread(f.data(), 1, (size_t)n, fp) == (size_t)n; }
    fclose(fp);
    return ok;
}
// One buffer -> one file (created or truncated), without a stream buffer in between.  (Cutting the buffer into ranges written side by
// side with pwrite was measured for the resident worker's callers, who wait for this very file: slower on tmpfs, 16 against 12 ms for a
// frame's 41 MB with three threads -- the page allocations serialise; profiles/r05g_cli_ab.log.)
inline bool write_whole_file(const std::string& filename, const void* data, size_t n)
{
    const int fd = open(filename.c_str(), O_WRONLY | O_CREAT | O_TRUNC | O_CLOEXEC, 0666);
    if (fd < 0) return false;
    const char* p = (const char*)data;
    bool ok = true;
    for (size_t a = 0; a < n && ok;) {
        const ssize_t k = write(fd, p + a, std::min<size_t>(n - a, (size_t)8 << 20));
        if (k < 0 && errno == EINTR) continue;
        if (k <= 0) ok = false; else a += (size_t)k;
    }
    return (close(fd) == 0) && ok;
}

// ------------------------------------------------------------------ images
struct Image {
    int w = 0, h = 0;
    std::vector<uint8_t> px;
    Image() {}
    Image(int w_, int h_, uint8_t v = 0) : w(w_), h(h_), px((size_t)w_ * h_, v) {}
    bool empty() const { return px.empty(); }
    uint8_t& at(int y, int x) { return px[(size_t)y * w + x]; }
    uint8_t at(int y, int x) const { return px[(size_t)y * w + x]; }
};

inline uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

// 8-bit grey PNG (what wass_prepare writes, wass_prepare.cpp:92,275); 8-bit grey+alpha / RGB / RGBA are converted
// like cv::imread(IMREAD_GRAYSCALE) (BT.601 fixed point); anything else is rejected with a clear message.
inline Image read_png_gray(const std::string& filename)
{
    std::vector<uint8_t> f;
    if (!read_whole_file(filename, f)) throw std::runtime_error("unable to open " + filename);
    static const uint8_t sig[8] = { 0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a };
    if (f.size() < 33 || memcmp(f.data(), sig, 8) != 0) throw std::runtime_error(filename + " is not a PNG file");
    int w = 0, h = 0, depth = 0, ctype = 0, interlace = 0;
    // the IDAT payloads, moved together in place (the zlib stream may be cut into chunks anywhere): no second buffer
    size_t zpos = 0, zlen = 0;
    for (size_t p = 8; p + 12 <= f.size();) {
        const uint32_t len = be32(&f[p]);
        const char* type = (const char*)&f[p + 4];
        if (p + 12 + len > f.size()) throw std::runtime_error(filename + ": truncated PNG chunk");
        if (!memcmp(type, "IHDR", 4)) { if (len < 13) throw std::runtime_error(filename + ": bad IHDR"); w = (int)be32(&f[p + 8]); h = (int)be32(&f[p + 12]); depth = f[p + 16]; ctype = f[p + 17]; interlace = f[p + 20]; }
        else if (!memcmp(type, "IDAT", 4)) {
            if (zlen == 0) zpos = p + 8; else memmove(&f[zpos + zlen], &f[p + 8], len);     // always towards lower addresses
            zlen += len;
        }
        else if (!memcmp(type, "IEND", 4)) break;
        p += 12 + len;
    }
    if (w <= 0 || h <= 0) throw std::runtime_error(filename + ": missing IHDR");
    if (depth != 8 || interlace != 0 || (ctype != 0 && ctype != 2 && ctype != 4 && ctype != 6))
        throw std::runtime_error(filename + ": only non-interlaced 8-bit grey/RGB(A) PNG files are supported");
    const int ch = ctype == 0 ? 1 : (ctype == 4 ? 2 : (ctype == 2 ? 3 : 4));
    const size_t stride = (size_t)w * ch;
    const size_t rawn = (stride + 1) * h;
    // A grey picture whose zlib stream is stored blocks throughout and whose rows carry the filter None -- what this product's own
    // wass_prepare writes (write_png_gray level 0) -- goes from the file's bytes straight into the picture: no inflate, no second buffer.
    if (ch == 1 && zlen >= 7 && (f[zpos] & 0x0f) == 8 && (f[zpos + 1] & 0x20) == 0 && (f[zpos + 2] & 6) == 0) {
        Image img(w, h);
        const uint8_t* z = f.data() + zpos;
        size_t at = 2, pos = 0;                                  // offset in the stream / in the unfiltered-row sequence
        bool good = true, last = false;
        while (good && !last && at + 5 <= zlen) {
            last = z[at] & 1;
            if (z[at] & 6) { good = false; break; }              // a deflated block after all: the general path
            const size_t len = z[at + 1] | ((size_t)z[at + 2] << 8), nlen = z[at + 3] | ((size_t)z[at + 4] << 8);
            at += 5;
            if ((len ^ nlen) != 0xFFFF || at + len > zlen || pos + len > rawn) { good = false; break; }
            for (size_t done = 0; done < len;) {
                const size_t row = (pos + done) / (stride + 1), col = (pos + done) % (stride + 1);
                if (col == 0) { if (z[at + done] != 0) { good = false; break; } ++done; continue; }
                const size_t n = std::min(len - done, stride + 1 - col);
                memcpy(&img.px[row * stride + (col - 1)], z + at + done, n);
                done += n;
            }
            at += len; pos += len;
        }
        if (good && last && pos == rawn) return img;            // (the Adler-32 is not checked: the chunk's CRC is not either, as in every fast path here)
    }
    std::unique_ptr<uint8_t[]> rawbuf(new uint8_t[rawn]);      // (not a vector: zero-filling 5 MB that inflate overwrites is a millisecond per picture)
    uint8_t* const raw = rawbuf.get();
    if (!zlib_inflate_exact(f.data() + zpos, zlen, raw, rawn))
        throw std::runtime_error(filename + ": zlib inflate failed");
    std::vector<uint8_t> cur(stride), prev(stride, 0);
    Image img(w, h);
    bool prev_in_picture = false;                            // the previous row went straight into the picture: `prev` is stale
    for (int y = 0; y < h; ++y) {
        const uint8_t ft = raw[(stride + 1) * y];
        const uint8_t* in = &raw[(stride + 1) * y + 1];
        if (ch == 1 && ft <= 2) {
            // grey rows with the filters None / Sub / Up (what fast encoders write): straight into the picture
            uint8_t* o = &img.px[(size_t)y * w];
            if (ft == 0) memcpy(o, in, stride);
            else if (ft == 1) { uint8_t a = 0; for (size_t i = 0; i < stride; ++i) { a = (uint8_t)(in[i] + a); o[i] = a; } }
            else { const uint8_t* up = prev_in_picture ? o - w : prev.data(); for (size_t i = 0; i < stride; ++i) o[i] = (uint8_t)(in[i] + up[i]); }
            prev_in_picture = true;
            continue;
        }
        if (prev_in_picture) { memcpy(prev.data(), &img.px[(size_t)(y - 1) * w], stride); prev_in_picture = false; }   // (ch == 1 here)
        for (size_t i = 0; i < stride; ++i) {
            const int a = i >= (size_t)ch ? cur[i - ch] : 0, b = prev[i], c = i >= (size_t)ch ? prev[i - ch] : 0;
            int v = in[i];
            switch (ft) {
                case 0: break;
                case 1: v += a; break;
                case 2: v += b; break;
                case 3: v += (a + b) / 2; break;
                case 4: { const int pp = a + b - c, pa = std::abs(pp - a), pb = std::abs(pp - b), pc = std::abs(pp - c);
                          v += (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c); break; }
                default: throw std::runtime_error(filename + ": bad PNG filter");
            }
            cur[i] = (uint8_t)v;
        }
        for (int x = 0; x < w; ++x) {
            const uint8_t* p = &cur[(size_t)x * ch];
            img.at(y, x) = ch <= 2 ? p[0] : (uint8_t)((p[0] * 4899 + p[1] * 9617 + p[2] * 1868 + 8192) >> 14);
        }
        prev.swap(cur);
    }
    return img;
}

// cv::resize(src, dst, Size(nw, nh), 0, 0, INTER_CUBIC) for CV_8UC1 (wass_stereo.cpp:413,416), restated from OpenCV
// 4.5.5 imgproc/resize.cpp (unpinned): source position (d + 0.5) * scale - 0.5 in float, Keys a = -0.75 weights as
// int16 * 2^11, replicated borders, horizontal pass to int32 rows, vertical pass (sum + 2^21) >> 22.  (OpenCV's SIMD
// vertical pass rounds in float and can differ from this scalar form by one grey level.)
inline Image resize_cubic(const Image& src, int nw, int nh)
{
    auto coeffs = [](float x, short* c) {
        const float A = -0.75f;
        float f[4];
        f[0] = ((A * (x + 1) - 5 * A) * (x + 1) + 8 * A) * (x + 1) - 4 * A;
        f[1] = ((A + 2) * x - (A + 3)) * x * x + 1;
        f[2] = ((A + 2) * (1 - x) - (A + 3)) * (1 - x) * (1 - x) + 1;
        f[3] = 1.f - f[0] - f[1] - f[2];
        for (int k = 0; k < 4; ++k) { const long r = lrintf(f[k] * 2048.f); c[k] = (short)(r < -32768 ? -32768 : (r > 32767 ? 32767 : r)); }
    };
    auto clip = [](int v, int n) { return v < 0 ? 0 : (v >= n ? n - 1 : v); };
    Image dst(nw, nh);
    if (src.empty() || nw <= 0 || nh <= 0) return dst;
    const double sx = (double)src.w / nw, sy = (double)src.h / nh;
    std::vector<int> xo((size_t)nw * 4);
    std::vector<short> xa((size_t)nw * 4);
    for (int dx = 0; dx < nw; ++dx) {
        float fx = (float)((dx + 0.5) * sx - 0.5);
        const int ix = (int)std::floor(fx);
        fx -= ix;
        coeffs(fx, &xa[(size_t)dx * 4]);
        for (int k = 0; k < 4; ++k) xo[(size_t)dx * 4 + k] = clip(ix - 1 + k, src.w);
    }
    std::vector<int> rows[4];
    int have[4] = { -1, -1, -1, -1 };
    for (auto& r : rows) r.resize(nw);
    for (int dy = 0; dy < nh; ++dy) {
        float fy = (float)((dy + 0.5) * sy - 0.5);
        const int iy = (int)std::floor(fy);
        fy -= iy;
        short b[4];
        coeffs(fy, b);
        const int* R[4];
        for (int k = 0; k < 4; ++k) {
            const int yy = clip(iy - 1 + k, src.h);
            int slot = -1;
            for (int q = 0; q < 4; ++q) if (have[q] == yy) slot = q;
            if (slot < 0) {                                   // a row not in the small cache: horizontal pass
                slot = 0;
                for (int q = 0; q < 4; ++q) {                 // evict a row this output line does not need
                    bool needed = false;
                    for (int kk = 0; kk < 4; ++kk) needed |= have[q] == clip(iy - 1 + kk, src.h);
                    if (!needed) { slot = q; break; }
                }
                have[slot] = yy;
                const uint8_t* S = &src.px[(size_t)yy * src.w];
                for (int dx = 0; dx < nw; ++dx) {
                    const int* o = &xo[(size_t)dx * 4];
                    const short* a = &xa[(size_t)dx * 4];
                    rows[slot][dx] = S[o[0]] * a[0] + S[o[1]] * a[1] + S[o[2]] * a[2] + S[o[3]] * a[3];
                }
            }
            R[k] = rows[slot].data();
        }
        for (int dx = 0; dx < nw; ++dx) {
            const int v = (R[0][dx] * b[0] + R[1][dx] * b[1] + R[2][dx] * b[2] + R[3][dx] * b[3] + (1 << 21)) >> 22;
            dst.at(dy, dx) = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
        }
    }
    return dst;
}

// level 1 .. 9: deflated (1 = cv::imwrite's default, Z_BEST_SPEED).  level 0: a zlib stream of STORED blocks -- still a PNG every reader
// takes (libpng, cv::imread, this file's reader), 0.03 % larger than the pixels, and its "inflate" is a memcpy: what this product's own
// wass_prepare writes undistorted/*.png with (two 5-megapixel pictures cost a frame 2 x 17 ms of inflate at level 1, a third of its host
// time; nothing downstream reads the files' bytes, only their pixels).
inline int prepared_png_level()                 // WASS_PREPARE_PNG_LEVEL=1: deflated like the reference's cv::imwrite; default 0: stored
{
    const char* e = getenv("WASS_PREPARE_PNG_LEVEL");
    return e && *e ? std::max(0, std::min(9, atoi(e))) : 0;
}
inline bool write_png_gray(const std::string& filename, const Image& img, int level = 1)
{
    std::vector<uint8_t> raw((size_t)(img.w + 1) * img.h);
    for (int y = 0; y < img.h; ++y) { raw[(size_t)(img.w + 1) * y] = 0; memcpy(&raw[(size_t)(img.w + 1) * y + 1], &img.px[(size_t)y * img.w], img.w); }
    // a valid PNG with these pixels is the contract, not its bytes
    std::vector<uint8_t> comp;
    if (level <= 0) {
        const size_t n = raw.size(), nblk = (n + 65534) / 65535;
        comp.reserve(n + 5 * std::max<size_t>(nblk, 1) + 6);
        comp.push_back(0x78); comp.push_back(0x01);             // CM = 8, 32 K window, no dictionary, fastest
        for (size_t at = 0, b = 0; b < std::max<size_t>(nblk, 1); ++b) {
            const size_t len = std::min<size_t>(65535, n - at);
            comp.push_back(at + len == n ? 1 : 0);             // BFINAL, BTYPE = 00
            comp.push_back((uint8_t)len); comp.push_back((uint8_t)(len >> 8));
            comp.push_back((uint8_t)~len); comp.push_back((uint8_t)(~len >> 8));
            comp.insert(comp.end(), raw.begin() + (long)at, raw.begin() + (long)(at + len));
            at += len;
        }
        const uLong ad = adler32(adler32(0L, Z_NULL, 0), raw.data(), (uInt)n);
        for (int s = 24; s >= 0; s -= 8) comp.push_back((uint8_t)(ad >> s));
    } else if (!zlib_deflate_fast(raw.data(), raw.size(), level, comp)) return false;
    const size_t clen = comp.size();
    std::ofstream ofs(filename.c_str(), std::ios::binary);
    if (ofs.fail()) return false;
    auto chunk = [&](const char* type, const uint8_t* data, uint32_t len) {
        uint8_t hdr[8] = { (uint8_t)(len >> 24), (uint8_t)(len >> 16), (uint8_t)(len >> 8), (uint8_t)len, (uint8_t)type[0], (uint8_t)type[1], (uint8_t)type[2], (uint8_t)type[3] };
        ofs.write((const char*)hdr, 8);
        if (len) ofs.write((const char*)data, len);
        uLong crc = crc32(0L, (const Bytef*)type, 4);
        if (len) crc = crc32(crc, data, len);
        const uint8_t c[4] = { (uint8_t)(crc >> 24), (uint8_t)(crc >> 16), (uint8_t)(crc >> 8), (uint8_t)crc };
        ofs.write((const char*)c, 4);
    };
    static const uint8_t sig[8] = { 0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a };
    ofs.write((const char*)sig, 8);
    uint8_t ihdr[13] = { (uint8_t)(img.w >> 24), (uint8_t)(img.w >> 16), (uint8_t)(img.w >> 8), (uint8_t)img.w,
                         (uint8_t)(img.h >> 24), (uint8_t)(img.h >> 16), (uint8_t)(img.h >> 8), (uint8_t)img.h, 8, 0, 0, 0, 0 };
    chunk("IHDR", ihdr, 13);
    chunk("IDAT", comp.data(), (uint32_t)clen);
    chunk("IEND", nullptr, 0);
    return !ofs.fail();
}

// ------------------------------------------------------------------ "%g" of a double, fast
// plane_refinement_inliers.xyz holds ~1.4 million numbers per 5-megapixel frame in the stream's default format (what
// printf("%g") prints: six significant digits, trailing zeros removed, scientific outside [1e-4, 1e6)).  std::to_chars
// is exact but costs ~0.3 us per number -- 0.4 s of CPU per frame.  fmt_g6 takes the common case in ~25 ns: scale into
// [1e5, 1e6) by an exactly representable power of ten (one rounding, relative error <= 2^-53), round to an integer, and
// accept the result only when the scaled value is far enough from a rounding boundary that neither that error nor a tie
// can change the digits; everything else (boundaries, tiny / huge / non-finite values) goes to std::to_chars, which is
// specified to produce printf's digits.  tests/test_hostio.py compares the two on tens of millions of values.
inline char* fmt_g6(char* out, char* end, double v)
{
    static const double P10[11] = { 1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10 };
    auto slow = [&]() { return std::to_chars(out, end, v, std::chars_format::general, 6).ptr; };
    const double a = std::fabs(v);
    if (!(a >= 1e-4 && a < 999999.0)) {                     // zero, subnormal-ish, scientific range, inf, nan
        if (v == 0.0) { char* q = out; if (std::signbit(v)) *q++ = '-'; *q++ = '0'; return q; }
        return slow();
    }
    if (end - out < 16) return slow();
    int e;                                                   // decimal exponent: 10^e <= a < 10^(e+1)
    if (a >= 1.0) e = a < 1e1 ? 0 : a < 1e2 ? 1 : a < 1e3 ? 2 : a < 1e4 ? 3 : a < 1e5 ? 4 : 5;
    else e = a >= 1e-1 ? -1 : a >= 1e-2 ? -2 : a >= 1e-3 ? -3 : -4;
    const double sc = a * P10[5 - e];                       // in [1e5, 1e6) up to one rounding
    const double fl = std::floor(sc), fr = sc - fl;
    if (std::fabs(fr - 0.5) < 1e-7 || sc < 100000.0 || sc >= 999999.5) return slow();     // boundary cases: exact path
    uint32_t m = (uint32_t)fl + (fr > 0.5 ? 1u : 0u);       // six digits
    char dg[6];
    for (int i = 5; i >= 0; --i) { dg[i] = (char)('0' + m % 10); m /= 10; }
    int nd = 6;
    while (nd > 1 && dg[nd - 1] == '0') --nd;               // %g strips trailing zeros
    char* q = out;
    if (v < 0) *q++ = '-';
    if (e >= 0) {
        const int ip = e + 1;                                // digits in front of the point
        for (int i = 0; i < ip; ++i) *q++ = dg[i];           // (zeros that belong to the integer part are never stripped: nd < ip is padded)
        if (nd > ip) { *q++ = '.'; for (int i = ip; i < nd; ++i) *q++ = dg[i]; }
    } else {
        *q++ = '0'; *q++ = '.';
        for (int i = 0; i < -e - 1; ++i) *q++ = '0';
        for (int i = 0; i < nd; ++i) *q++ = dg[i];
    }
    return q;
}

// ------------------------------------------------------------------ point-cloud writers (host copies of the mesh)
inline bool save_xyz_binary(const std::string& filename, const std::vector<uint8_t>& valid, const std::vector<double>& p3d)
{
    std::ofstream ofs(filename.c_str(), std::ios::binary);
    if (ofs.fail()) return false;
    std::vector<float> pts;
    for (size_t i = 0; i < valid.size(); ++i)
        if (valid[i]) { pts.push_back((float)p3d[3 * i]); pts.push_back((float)p3d[3 * i + 1]); pts.push_back((float)p3d[3 * i + 2]); }
    const uint32_t n = (uint32_t)(pts.size() / 3);
    ofs.write((const char*)&n, 4);
    ofs.write((const char*)pts.data(), (std::streamsize)(pts.size() * 4));
    return !ofs.fail();
}
inline bool save_ply_points(const std::string& filename, const std::vector<uint8_t>& valid, const std::vector<double>& p3d,
                            const std::vector<uint8_t>& gray)
{
    std::ofstream ofs(filename.c_str(), std::ios::binary);
    if (ofs.fail()) return false;
    size_t n = 0;
    for (uint8_t v : valid) n += v ? 1 : 0;
    ofs << "ply\nformat binary_little_endian 1.0\nelement vertex " << n
        << "\nproperty float x\nproperty float y\nproperty float z\nproperty uchar red\nproperty uchar green\nproperty uchar blue\nend_header\n";
    for (size_t i = 0; i < valid.size(); ++i)
        if (valid[i]) {
            const float p[3] = { (float)p3d[3 * i], (float)p3d[3 * i + 1], (float)p3d[3 * i + 2] };
            const uint8_t c[3] = { gray[i], gray[i], gray[i] };
            ofs.write((const char*)p, 12);
            ofs.write((const char*)c, 3);
        }
    return !ofs.fail();
}

}  // namespace wasshost
