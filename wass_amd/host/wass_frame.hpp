// wass_frame.hpp -- one frame of the reference's wass_stereo stage (/root/reference/src/wass_stereo/wass_stereo.cpp:1833-2147,
// everything main() does once argv has been checked), calling libwassgpu through its C ABI.  Shared by the drop-in
// executable (wass_stereo.cpp: one frame per process, as wasscli launches it) and by the sequence driver
// (wass_stereo_batch.cpp: one process per GPU, one persistent context, many frames).
//
// Same argv, exit codes, config format, workdir inputs/outputs, stdout progress markers and log format as the
// reference.  There is NO CPU implementation of the hot path here: without a GPU (or
// without libwassgpu.so) the program fails with exit code -1.
//
// Divergences from the reference, all listed in DESIGN.md: the debug pictures are PNG, not JPEG (render.hpp);
// --measure (interactive GUI) is rejected.
#pragma once

#include <charconv>
#include <sys/stat.h>
#include <sys/time.h>

#include <cstdlib>
#include <ctime>
#include <memory>
#include <thread>

#include "../../include/wass_gpu.h"
#include "config.hpp"
#include "hostio.hpp"
#include "tiff.hpp"
#include "rectify.hpp"
#include "render.hpp"

using namespace wasshost;

#ifndef WASS_AMD_VERSION
#define WASS_AMD_VERSION "1.26-mi355x"
#endif

namespace wassframe {

struct Timer {           // cvlab::HiresTimer (src/wass_lib/hires_timer.cpp:76-131)
    double t0 = 0, tend = 0;
    std::vector<std::pair<double, std::string>> events;
    static double now() { timeval tv; gettimeofday(&tv, nullptr); return (double)tv.tv_sec + (double)tv.tv_usec / 1e6; }
    void start() { t0 = now(); }
    double elapsed() const { return (tend > 0 ? tend : now()) - t0; }
    void stop() { tend = now(); }
    void operator<<(const std::string& name) { events.emplace_back(elapsed(), name); }
};

struct Env {             // StereoMatchEnv (wass_stereo.cpp:202-335)
    Timer timer;
    std::string workdir;
    Mat K_left, K_right, K0, K1, R, T, Rinv, Tinv, P0, P1, Rpose0, Tpose0, Rpose1, Tpose1, HL, HR, HLi, HRi;
    bool use_custom = true;
    double rec_R1[9] = {}, rec_R2[9] = {}, rec_P1[12] = {}, rec_P2[12] = {};                 // cv::stereoRectify outputs (:248-251)
    Image left, right, left_crop, right_crop;
    int left_index = 0, right_index = 1;
    double cam_distance = 1.0, disparity_compensation = 0.0;
    Rect roi_l, roi_r;
};

std::string path_join(const std::string& a, const std::string& b) { return a.empty() || a.back() == '/' ? a + b : a + "/" + b; }
bool exists(const std::string& p) { struct stat st; return stat(p.c_str(), &st) == 0; }

Mat stack_RT(const Mat& R, const Mat& T) { Mat m(3, 4); for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) m(i, j) = R(i, j); m(i, 3) = T(i, 0); } return m; }
void invert_RT(Mat& R, Mat& T) { R = transpose(R); T = scaled(matmul(R, T), -1.0); }     // :196-200
void computeP(Env& e) { e.P0 = matmul(e.K0, stack_RT(e.Rpose0, e.Tpose0)); e.P1 = matmul(e.K1, stack_RT(e.Rpose1, e.Tpose1)); }

void swapLeftRight(Env& e)                                                                // :264-297
{
    std::swap(e.left_index, e.right_index);
    std::swap(e.left, e.right);
    std::swap(e.K_left, e.K_right);
    std::swap(e.R, e.Rinv);
    std::swap(e.T, e.Tinv);
    std::swap(e.Rpose0, e.Rpose1);
    std::swap(e.Tpose0, e.Tpose1);
    invert_RT(e.Rpose0, e.Tpose0);
    invert_RT(e.Rpose1, e.Tpose1);
    computeP(e);
}

// A host thread that is always joined (on every return path and by exceptions).
struct BgTask {
    std::thread t;
    template <typename F> void run(F&& f) { wait(); t = std::thread(std::forward<F>(f)); }
    void wait() { if (t.joinable()) t.join(); }
    ~BgTask() { wait(); }
};
// File writes that may finish while the NEXT frame is being processed (sequence driver); one job in flight.
struct AsyncWriter {
    BgTask task;
    bool failed = false;
    template <typename F> void submit(F&& f) { task.run([this, f]() { if (!f()) failed = true; }); }
    void wait() { task.wait(); }
};

// The two input pictures of a workdir, decoded ahead of time: a sequence driver inflates the next frame's PNGs on a host
// thread while the current frame is on the GPU (zlib is the whole "Data load" time, ~50 ms at 2456 x 2058).
struct Preload {
    std::string workdir;
    Image left, right;
    std::string error;             // non-empty: what read_png_gray threw (reported by load_data as if it had read the files itself)
};
inline void preload_images(const std::string& workdir, Preload& p)
{
    p.workdir = workdir; p.error.clear();
    std::string err1;
    std::thread t1([&]() { try { p.right = read_png_gray(path_join(workdir, "undistorted/00000001.png")); } catch (const std::exception& e) { err1 = e.what(); } });
    try { p.left = read_png_gray(path_join(workdir, "undistorted/00000000.png")); } catch (const std::exception& e) { p.error = e.what(); }
    t1.join();
    if (p.error.empty()) p.error = err1;
}

// load_data() (:337-445) in three parts, in the reference's order: calibration, pictures, the SAVE_INPUT_SCALE outputs.  The
// pipelined driver's prepare-less mode (frame_pipeline.hpp) runs the first, decodes RAW camera frames instead of the second
// and writes the previews of the third once the GPU has undistorted them.
bool load_calibration(Env& env)                                                           // :340-391
{
    WLOG_SCOPE("load_data");
    env.R = load_matrix_xml(path_join(env.workdir, "ext_R.xml"));
    if (env.R.rows != 3 || env.R.cols != 3) { WLOGE << "invalid extrinsic rotation matrix (ext_R.xml)"; return false; }
    env.T = load_matrix_xml(path_join(env.workdir, "ext_T.xml"));
    if (env.T.cols != 1 || env.T.rows != 3) { WLOGE << "invalid extrinsic translation vector (ext_T.xml)"; return false; }
    env.Rinv = env.R; env.Tinv = env.T;
    invert_RT(env.Rinv, env.Tinv);
    const double cur = std::sqrt(env.T(0, 0) * env.T(0, 0) + env.T(1, 0) * env.T(1, 0) + env.T(2, 0) * env.T(2, 0));
    for (int i = 0; i < 3; ++i) { env.T(i, 0) = env.T(i, 0) / cur * env.cam_distance; env.Tinv(i, 0) = env.Tinv(i, 0) / cur * env.cam_distance; }
    env.Rpose0 = Mat::eye(3); env.Tpose0 = Mat(3, 1);
    env.Rpose1 = env.R; env.Tpose1 = env.T;
    env.K0 = load_matrix_xml(path_join(env.workdir, "intrinsics_00000000.xml"));
    env.K1 = load_matrix_xml(path_join(env.workdir, "intrinsics_00000001.xml"));
    if (env.K0.rows != 3 || env.K0.cols != 3 || env.K1.rows != 3 || env.K1.cols != 3) { WLOGE << "invalid intrinsics"; return false; }
    env.K_left = env.K0; env.K_right = env.K1;
    computeP(env);
    return true;
}

// env.left / env.right are in place (cam0 / cam1): the lines and the check of :393-399
bool images_loaded(Env& env)
{
    WLOG_SCOPE("load_data");
    env.left_index = 0;
    WLOGI << "image 0 loaded, Size: " << env.left.w << "x" << env.left.h;
    env.right_index = 1;
    WLOGI << "image 1 loaded, Size: " << env.right.w << "x" << env.right.h;
    if (env.left.w != env.right.w || env.left.h != env.right.h) { WLOGE << "left and right images differ in size"; return false; }
    return true;
}

inline void write_previews(const std::string& wd, const Image& cam0, const Image& cam1, int nw, int nh)   // :413-418
{
    write_png_gray(path_join(wd, "00000000_s.png"), resize_cubic(cam0, nw, nh), prepared_png_level());
    write_png_gray(path_join(wd, "00000001_s.png"), resize_cubic(cam1, nw, nh), prepared_png_level());
}

// :401-434.  previews = false: everything but the two scaled pictures (the caller writes them later: *nw, *nh say at which size)
void input_scale_outputs(Env& env, const Config& cfg, bool previews, BgTask* bg, int* pnw = nullptr, int* pnh = nullptr)
{
    WLOG_SCOPE("load_data");
    if (pnw) *pnw = 0;
    if (pnh) *pnh = 0;
    const double sis = cfg.get_double("SAVE_INPUT_SCALE");
    if (sis < 1.0) {
        const size_t nw = (size_t)(env.left.w * sis), nh = (size_t)(env.left.h * sis);
        const double scale = (double)nw / (double)env.left.w;
        WLOGI << "original size: " << env.left.w << "x" << env.left.h;
        WLOGI << "  scaled size: " << nw << "x" << nh;
        WLOGI << "        scale: " << scale;
        if (nw > 0 && nh > 0) {
            if (pnw) *pnw = (int)nw;
            if (pnh) *pnh = (int)nh;
            if (previews) {
                // resize + deflate of the two previews (~20 ms) on a host thread while the frame goes to the GPU.  The task owns
                // copies of the pictures: rectify() may swap env.left / env.right (swapLeftRight) while it runs.
                auto l = std::make_shared<Image>(env.left), r = std::make_shared<Image>(env.right);
                const std::string wd = env.workdir;
                auto work = [l, r, wd, nw, nh]() { write_previews(wd, *l, *r, (int)nw, (int)nh); };
                if (bg) bg->run(work); else work();
            }
        }
        Mat k0 = scaled(env.K_left, scale), k1 = scaled(env.K_right, scale);
        k0(2, 2) = 1; k1(2, 2) = 1;
        save_matrix_txt(path_join(env.workdir, "K0_small.txt"), k0);
        save_matrix_txt(path_join(env.workdir, "K1_small.txt"), k1);
        std::ostringstream ofs;
        ofs.precision(16); ofs << std::scientific << scale;
        commit_text_file(path_join(env.workdir, "scale.txt"), ofs.str());
    }
}

// previews = false: the two scaled pictures are left to the caller (the pipelined chain resizes them on the GPU); *pnw, *pnh: their size
bool load_data(Env& env, const Config& cfg, Preload* pre = nullptr, BgTask* bg = nullptr, bool previews = true, int* pnw = nullptr, int* pnh = nullptr)    // :337-445
{
    WLOG_SCOPE("load_data");
    if (!load_calibration(env)) return false;
    {
        Preload local;
        if (!pre || pre->workdir != env.workdir) { preload_images(env.workdir, local); pre = &local; }   // the two PNGs are inflated side by side
        if (!pre->error.empty()) { WLOGE << "unable to load input images: " << pre->error; return false; }
        env.left = std::move(pre->left); env.right = std::move(pre->right);
    }
    if (!images_loaded(env)) return false;
    input_scale_outputs(env, cfg, previews, bg, pnw, pnh);
    return true;
}

// rectify() in two steps: what it DECIDES (left/right swap, homographies or cv::stereoRectify's matrices, the ROIs: host
// math on the calibration, no pixel touched) and the resampling of the two pictures.  The stage-by-stage executable runs one
// after the other; the pipelined sequence driver takes the decisions on a decode thread and resamples on the GPU from
// device-resident inputs (frame_pipeline.hpp).
bool rectify_plan(Env& env, const Config& cfg)                                            // :447-599
{
    WLOG_SCOPE("rectify");
    WLOGI << "rectifying...";
    bool auto_swap = true, do_swap = false;
    if (std::fabs(env.T(1, 0)) > std::fabs(env.T(0, 0))) { WLOGE << "Vertical stereo not supported"; return false; }
    WLOGI << "Detected stereo setup:";
    WLOGI << (env.T(0, 0) > 0 ? "CAM1 (L) ---------  CAM0 (R)" : "CAM0 (L) ---------  CAM1 (R)");
    if (cfg.get_bool("DISABLE_AUTO_LEFT_RIGHT")) {
        auto_swap = false;
        do_swap = cfg.get_bool("SWAP_LEFT_RIGHT");
        WLOGI << "auto left-right detection disabled. Swap left-right? " << (do_swap ? "YES" : "NO");
        if (do_swap) { WLOGI << "swapping left-right images as requested"; swapLeftRight(env); }
    } else if (env.T(0, 0) < 0) {
        WLOGI << "auto-swapping left-right images";
        swapLeftRight(env);
    }
    const int W = env.left.w, H = env.left.h;
    auto roi_ok = [&](const Rect& r) { return r.x >= 0 && r.y >= 0 && r.width > 0 && r.height > 0 && r.x + r.width <= W && r.y + r.height <= H; };
    env.use_custom = cfg.get_bool("USE_CUSTOM_STEREORECTIFY");
    if (env.use_custom) {
        const double ang = cfg.get_double("RECTIFY_ANGLE");
        WLOGI << "Using WASS custom stereorectify, baseline angle delta=" << ang;
        const double Tinv[3] = { env.Tinv(0, 0), env.Tinv(1, 0), env.Tinv(2, 0) };
        Rect roi;
        stereoRectifyUndistorted(env.K_left, env.K_right, env.Rinv, Tinv, ang, W, H, env.HL, env.HR, roi);
        env.HLi = inv3(env.HL); env.HRi = inv3(env.HR);
        save_matrix_txt(path_join(env.workdir, env.left_index == 0 ? "H0_rect.txt" : "H1_rect.txt"), env.HL);
        save_matrix_txt(path_join(env.workdir, env.left_index == 0 ? "H1_rect.txt" : "H0_rect.txt"), env.HR);
        env.roi_l = env.roi_r = roi;
        if (cfg.get_bool("DISABLE_RECTIFY_ROI")) { env.roi_l = env.roi_r = Rect{ 0, 0, W, H }; }
        if (!roi_ok(env.roi_l)) { WLOGE << "rectification ROI is empty or outside the image"; return false; }
    } else {
        WLOGI << "Rectifying via cv::stereoRectify";
        int roi_left[4], roi_right[4];
        bool rectification_ok = false;
        do {                                                                               // :539-582
            const double T3[3] = { env.T(0, 0), env.T(1, 0), env.T(2, 0) };
            if (wass_stereo_rectify(env.K_left.d.data(), env.K_right.d.data(), W, H, env.R.d.data(), T3, 1.0, env.rec_R1, env.rec_R2, env.rec_P1,
                                    env.rec_P2, roi_left, roi_right) != WASS_OK) { WLOGE << "stereoRectify failed (zero baseline)"; return false; }
            if (std::fabs(env.rec_P2[3]) < std::fabs(env.rec_P2[7])) { WLOGE << "vertical stereo not supported"; return false; }
            if (roi_left[2] == 0 || roi_right[2] == 0 || roi_left[3] == 0 || roi_right[3] == 0) { WLOGE << "the epipole lies inside the image plane"; return false; }
            if (auto_swap) {
                if (env.rec_P2[3] < 0) { WLOGI << "auto-swapping left-right images"; swapLeftRight(env); }
                else rectification_ok = true;
            } else if (do_swap) {          // sic (:570-575): a requested swap is applied a second time here, i.e. undone
                WLOGI << "swapping left-right images as requested";
                swapLeftRight(env);
                do_swap = false;
            } else rectification_ok = true;
        } while (!rectification_ok);
        const int ymin = std::max(roi_left[1], roi_right[1]);
        const int ymax = std::min(roi_left[1] + roi_left[3], roi_right[1] + roi_right[3]);
        env.roi_l = Rect{ roi_left[0], ymin, roi_left[2], ymax - ymin };
        env.roi_r = Rect{ roi_right[0], ymin, roi_right[2], ymax - ymin };
        if (env.roi_l.width > env.roi_r.width) env.roi_l.width = env.roi_r.width; else env.roi_r.width = env.roi_l.width;
        if (!roi_ok(env.roi_l) || !roi_ok(env.roi_r)) { WLOGE << "rectification ROI is empty or outside the image"; return false; }
    }
    return true;
}

bool rectify_resample(Env& env, wass_ctx* ctx)                                            // :515-528, 600-613
{
    WLOG_SCOPE("rectify");
    const int W = env.left.w, H = env.left.h;
    auto gpu = [&](int rc, const char* what) { if (rc != WASS_OK) throw std::runtime_error(std::string(what) + ": " + wass_last_error(ctx)); };
    if (env.use_custom) {
        // cv::warpPerspective (:515-516) and the ROI .clone() (:526-528) in one GPU pass per camera
        const int rl[4] = { env.roi_l.x, env.roi_l.y, env.roi_l.width, env.roi_l.height };
        env.left_crop = Image(rl[2], rl[3]); env.right_crop = Image(rl[2], rl[3]);
        gpu(wass_warp_perspective(ctx, env.left.px.data(), W, H, (size_t)W, env.HL.d.data(), W, H, rl, env.left_crop.px.data()), "wass_warp_perspective");
        gpu(wass_warp_perspective(ctx, env.right.px.data(), W, H, (size_t)W, env.HR.d.data(), W, H, rl, env.right_crop.px.data()), "wass_warp_perspective");
    } else {
        // cv::initUndistortRectifyMap (:600-601), cv::remap INTER_CUBIC (:603-604), ROI .clone() (:606-607)
        std::vector<float> mx((size_t)W * H), my((size_t)W * H);
        const int rl[4] = { env.roi_l.x, env.roi_l.y, env.roi_l.width, env.roi_l.height }, rr[4] = { env.roi_r.x, env.roi_r.y, env.roi_r.width, env.roi_r.height };
        env.left_crop = Image(rl[2], rl[3]); env.right_crop = Image(rr[2], rr[3]);
        if (wass_init_rectify_map(env.K_left.d.data(), env.rec_R1, env.rec_P1, W, H, mx.data(), my.data()) != WASS_OK) { WLOGE << "singular rectification"; return false; }
        gpu(wass_remap_cubic(ctx, env.left.px.data(), W, H, (size_t)W, mx.data(), my.data(), W, H, rl, env.left_crop.px.data()), "wass_remap_cubic");
        if (wass_init_rectify_map(env.K_right.d.data(), env.rec_R2, env.rec_P2, W, H, mx.data(), my.data()) != WASS_OK) { WLOGE << "singular rectification"; return false; }
        gpu(wass_remap_cubic(ctx, env.right.px.data(), W, H, (size_t)W, mx.data(), my.data(), W, H, rr, env.right_crop.px.data()), "wass_remap_cubic");
    }
    WLOGI << "rectification map generated. Size: " << env.left_crop.w << "x" << env.left_crop.h;
    return true;
}

bool rectify(Env& env, const Config& cfg, wass_ctx* ctx) { return rectify_plan(env, cfg) && rectify_resample(env, ctx); }   // :447-613

void show_time_stats(const Timer& t)                                                       // render.hpp:175-191
{
    WLOGI << "+----------------------------+-------------------+";
    WLOGI << "|   Task                     |   Time (seconds)  |";
    WLOGI << "+----------------------------+-------------------+";
    double last = 0.0;
    for (const auto& e : t.events) {
        std::ostringstream os; os << "| " << std::setw(25) << e.second << "  |" << std::setw(18) << (e.first - last) << " |";
        WLOGI << os.str();
        last = e.first;
    }
    WLOGI << "+----------------------------+-------------------+";
    { std::ostringstream os; os << "| " << std::setw(25) << "TOTAL" << "  |" << std::setw(18) << t.elapsed() << " |"; WLOGI << os.str(); }
    WLOGI << "+----------------------------+-------------------+";
}

int save_configuration(const Config& cfg, const std::string& filename)                    // :1776-1794
{
    WLOG_SCOPE("wass_stereo");
    WLOGI << "Writing " << filename;
    if (!commit_text_file(filename, cfg.to_config_string())) { WLOGE << "Unable to open " << filename << " for write"; return -1; }
    WLOGI << "Done!";
    return 0;
}

struct GpuError : std::runtime_error { using std::runtime_error::runtime_error; };
void gpu_check(wass_ctx* ctx, int rc, const char* what, bool allow_overflow = false)
{
    if (rc == WASS_OK || (allow_overflow && rc == WASS_ERR_COST_OVERFLOW)) return;
    throw GpuError(std::string(what) + ": " + wass_last_error(ctx));
}



// ---------------------------------------------------------------- the reference's debug pictures
// Drawn from host copies of a frame's intermediate maps.  The stage-by-stage path has them anyway; the pipelined chain
// downloads them once the frame is complete (frame_pipeline.hpp) -- same functions, same pictures.
struct DebugMaps {
    std::vector<int16_t> disp16;            // ws x hs: the SGBM map (:837-839)
    int ws = 0, hs = 0;
    std::vector<float> dispf;               // cw x ch: after clean_and_convert / dilate / erode / resize / mask / median / component (:853-986)
    std::vector<uint8_t> large_gradient;    // cw x ch, non-zero where the squared Sobel magnitude exceeded the threshold (cc_threshold > 0 only)
    std::vector<uint8_t> codes;             // roi_r: why triangulate kept or rejected each pixel (wass_mesh_reject_codes)
    std::vector<uint8_t> valid_before, valid_after;   // roi_r: the mesh before / after cluster_biggest_connected_component
};

inline void debug_stereo_picture(const Env& env)                                          // stereo.jpg (:1910-1925)
{
    const int W0 = env.left.w, H0 = env.left.h;
    ImageRGB l = gray_to_rgb(paste(env.left_crop, env.roi_l.x, env.roi_l.y, W0, H0)), r = gray_to_rgb(paste(env.right_crop, env.roi_r.x, env.roi_r.y, W0, H0));
    rectangle_red(l, env.roi_l.x, env.roi_l.y, env.roi_l.width, env.roi_l.height);
    rectangle_red(r, env.roi_r.x, env.roi_r.y, env.roi_r.width, env.roi_r.height);
    ImageRGB st(2 * W0, H0);
    for (int y = 0; y < H0; ++y) {
        memcpy(&st.px[(size_t)y * st.w * 3], &l.px[(size_t)y * W0 * 3], (size_t)W0 * 3);
        memcpy(&st.px[((size_t)y * st.w + W0) * 3], &r.px[(size_t)y * W0 * 3], (size_t)W0 * 3);
        if (y % 20 == 0) for (int x = 0; x < st.w; ++x) st.set(y, x, 255, 0, 0);
    }
    write_debug_rgb(path_join(env.workdir, "stereo"), st);
}

inline void debug_dense_pictures(const Env& env, const wass_sgm_params& sp, int cc_threshold, const DebugMaps& dm)
{
    const int cw = env.right_crop.w, ch = env.right_crop.h;
    if (cc_threshold > 0 && !dm.large_gradient.empty()) {        // :958-960, 981-983
        Image lg(cw, ch), nb(cw, ch);
        for (size_t i = 0; i < lg.px.size(); ++i) {
            lg.px[i] = dm.large_gradient[i] ? 255 : 0;
            nb.px[i] = dm.dispf[i] == 0.0f ? 255 : 0;               // 255 outside the biggest component: the map is non-zero exactly on it
        }
        write_debug_gray(path_join(env.workdir, "disparity_large_gradient"), lg);
        write_debug_gray(path_join(env.workdir, "disparity_biggest_component"), nb);
    }
    const int D = sp.num_disp, offp = sp.disp_offset > 0 ? sp.disp_offset : 0, comp = sp.disp_offset > 0 ? 0 : -sp.disp_offset;
    const int Wp = cw + D + offp;
    Image in2(Wp, 2 * ch);                                       // stereo_input.jpg (:820-833): padded left above padded right
    for (int y = 0; y < ch; ++y) {
        memcpy(&in2.px[(size_t)y * Wp + (D + offp - comp)], &env.left_crop.px[(size_t)y * cw], cw);
        memcpy(&in2.px[(size_t)(ch + y) * Wp + D], &env.right_crop.px[(size_t)y * cw], cw);
    }
    if (sp.dense_scale == 1.0) write_debug_gray(path_join(env.workdir, "stereo_input"), in2);   // (the resized inputs stay on the GPU)
    std::vector<float> conv((size_t)dm.ws * dm.hs);              // clean_and_convert_disparity (:714-733) of the raw map
    const double scl = 1.0 / sp.dense_scale;
    for (size_t i = 0; i < conv.size(); ++i) {
        float dval = ((float)dm.disp16[i]) / 16.0f;
        conv[i] = (dval <= (float)sp.min_disp || dval > (float)sp.num_disp) ? 0.0f : (float)((double)(dval + (float)sp.disp_offset) * scl);
    }
    write_debug_gray(path_join(env.workdir, "disparity_stereo_ouput"), render_disparity_float(conv.data(), dm.ws, dm.hs));
    write_debug_gray(path_join(env.workdir, "disparity_final_scaled"), render_disparity_float(dm.dispf.data(), cw, ch));
    const int W0 = env.right.w, H0 = env.right.h;               // disparity_coverage.jpg (:1002-1017)
    ImageRGB cov = gray_to_rgb(paste(env.right_crop, env.roi_r.x, env.roi_r.y, W0, H0));
    for (int y = 0; y < ch; ++y)
        for (int x = 0; x < cw; ++x)
            if (dm.dispf[(size_t)y * cw + x] > 1.0f && env.roi_r.y + y < H0 && env.roi_r.x + x < W0)
                cov.px[((size_t)(env.roi_r.y + y) * W0 + env.roi_r.x + x) * 3 + 1] = 100;
    rectangle_red(cov, env.roi_r.x, env.roi_r.y, env.roi_r.width, env.roi_r.height);
    write_debug_rgb(path_join(env.workdir, "disparity_coverage"), half_size(cov));
}

// undistorted/R0.jpg, R1.jpg (:1111-1119, 1216-1338, 1381-1382): per processed pixel of the right ROI the rectified grey value,
// overpainted with the colour of the test that rejected it (the codes come from the triangulation kernel); R1's grey is the
// LEFT rectified image at the match
inline void debug_triangulation_pictures(const Env& env, double disparity_compensation, double dense_scale, const DebugMaps& dm)
{
    const int roi_l[4] = { env.roi_l.x, env.roi_l.y, env.roi_l.width, env.roi_l.height };
    const int roi_r[4] = { env.roi_r.x, env.roi_r.y, env.roi_r.width, env.roi_r.height };
    const int gw = roi_r[2], gh = roi_r[3], W0 = env.left.w, H0 = env.left.h;
    ImageRGB R0(W0, H0), R1(W0, H0);
    static const uint8_t rgb[7][3] = { { 0, 0, 0 }, { 0, 0, 0 }, { 0, 255, 255 }, { 255, 255, 0 }, { 0, 255, 0 }, { 0, 0, 255 }, { 255, 0, 0 } };
    const float comp = (float)(disparity_compensation / dense_scale);
    for (int v = 0; v < gh; ++v)
        for (int u = 0; u < gw; ++u) {
            const uint8_t cd = dm.codes[(size_t)v * gw + u];
            const int c0 = cd & 15, c1 = cd >> 4, xr = roi_r[0] + u, yr = roi_r[1] + v;
            if (xr < 0 || xr >= W0 || yr < 0 || yr >= H0) continue;
            if (c0 == WASS_CODE_GREY) { const uint8_t gv = env.right_crop.at(v, u); R0.set(yr, xr, gv, gv, gv); }
            else if (c0 != WASS_CODE_NONE) R0.set(yr, xr, rgb[c0][0], rgb[c0][1], rgb[c0][2]);
            if (c1 == WASS_CODE_GREY) {
                const float xl = (float)((float)(u + roi_l[0]) - dm.dispf[(size_t)v * gw + u] + comp);
                const int lx = (int)std::floor(xl + 0.5f) - roi_l[0], ly = yr - roi_l[1];
                const uint8_t gv = (lx >= 0 && lx < env.left_crop.w && ly >= 0 && ly < env.left_crop.h) ? env.left_crop.at(ly, lx) : 0;
                R1.set(yr, xr, gv, gv, gv);
            } else if (c1 != WASS_CODE_NONE) R1.set(yr, xr, rgb[c1][0], rgb[c1][1], rgb[c1][2]);
        }
    write_debug_rgb(path_join(path_join(env.workdir, "undistorted"), "R0"), R0);
    write_debug_rgb(path_join(path_join(env.workdir, "undistorted"), "R1"), R1);
}

inline void debug_components_picture(const Env& env, const DebugMaps& dm)                  // graph_components.jpg (PovMesh.cpp:222-250, 982-984)
{
    ImageRGB gc(env.roi_r.width, env.roi_r.height);
    for (size_t i = 0; i < dm.valid_after.size(); ++i)
        if (dm.valid_after[i]) { gc.px[3 * i + 1] = 255; }                  // biggest component: palette.back() = (0,255,0)
        else if (dm.valid_before[i]) { gc.px[3 * i + 2] = 255; }           // every other component: palette[0] = BGR (255,0,0)
    write_debug_rgb(path_join(env.workdir, "graph_components"), half_size(gc));
}

// plane_refinement_inliers.xyz (:2077-2085): "x y z" per line in the stream's default format
inline void write_inliers_xyz(const std::string& path, const double* xyz, size_t n)
{
    // "x y z" per line in the stream's default format (%g): fmt_g6 gives printf's characters (tests/test_hostio.py)
    std::string text;
    text.resize(n * 48 + 64);
    char* q = &text[0];
    char* const end = q + text.size();
    for (size_t j = 0; j < n; ++j) {
        if (end - q < 128) {                             // (never with 48 bytes per line and |numbers| that fit %g's 13 characters)
            const size_t used = (size_t)(q - &text[0]);
            text.resize(text.size() * 2);
            q = &text[0] + used;
        }
        char* const lim = &text[0] + text.size();
        for (int k = 0; k < 3; ++k) {
            q = fmt_g6(q, lim - 2, xyz[3 * j + k]);
            *q++ = k < 2 ? ' ' : '\n';
        }
    }
    std::ofstream ofs(path.c_str(), std::ios::binary);
    ofs.write(text.data(), (std::streamsize)(q - &text[0]));
}


struct FrameSummary {
    int have_plane = 0;           // 1: plane.txt holds a refined plane, 0: "nan nan nan nan" (RANSAC failed) or the frame failed
    double plane[4] = { 0, 0, 0, 0 };
    unsigned long long n_points = 0;
};

// Everything wass_stereo does with one workdir.  *ctxp may hold a live context (sequence driver); if it is null one is
// created on `device` at the point where the reference would first need the GPU and handed back to the caller, who
// destroys it.  mode: nullptr, "--rectify-only" or "--measure".  Returns the process exit code of wass_stereo (0 / -1).
inline int wass_run_frame(const char* config_path, const std::string& workdir, const char* mode, int device, wass_ctx** ctxp,
                          FrameSummary* summary, bool debug_images = true, Preload* preloaded = nullptr, AsyncWriter* writer = nullptr)
{
    if (const char* e = getenv("WASS_DEBUG_IMAGES")) debug_images = atoi(e) != 0;
    Env env;
    env.workdir = workdir;
    Config cfg;
    register_wass_stereo_options(cfg);
    setup_logger(path_join(env.workdir, "wass_stereo_log.txt"));
    WLOG_SCOPE("wass_stereo");
    {
        WLOGI << "Loading configuration file " << config_path;
        std::ifstream ifs(config_path);
        if (!ifs.is_open()) { WLOGE << "Unable to load " << config_path; return -1; }
        try { cfg.load(ifs); } catch (const std::runtime_error& er) { WLOGE << er.what(); return -1; }
        if (save_configuration(cfg, path_join(env.workdir, "stereo_config.txt")) != 0) WLOGE << "Unable to save stereo configuration file";
    }
    // :1864-1872.  The reference srand()s here and RANSAC (PovMesh.cpp:680-682) is its only rand() consumer; this process also
    // hosts the HIP runtime, whose threads draw from libc rand(), so the seed goes to the library's private generator instead.
    unsigned int ransac_seed = (unsigned int)time(0);
    if (cfg.get_int("RANDOM_SEED") != -1) { ransac_seed = (unsigned int)cfg.get_int("RANDOM_SEED"); WLOGI << "random seed set to: " << cfg.get_int("RANDOM_SEED"); }

    wass_ctx* ctx = *ctxp;
    wass_mesh* mesh = nullptr;
    int ret = 0;
    BgTask previews;                                             // declared before the try block: joined after env's last use
    try {
        WLOGI << "Reconstructing " << env.workdir;
        env.timer.start();
        env.cam_distance = 1.0;
        if (!load_data(env, cfg, preloaded, &previews)) return -1;
        env.timer << "Data load";
        std::cout << "[P|10|100]" << std::endl;
        auto save_cams = [&]() {
            save_matrix_txt(path_join(env.workdir, "P0cam.txt"), env.P0);
            save_matrix_txt(path_join(env.workdir, "P1cam.txt"), env.P1);
            save_matrix_txt(path_join(env.workdir, "Cam0_poseR.txt"), env.Rpose0);
            save_matrix_txt(path_join(env.workdir, "Cam0_poseT.txt"), env.Tpose0);
            save_matrix_txt(path_join(env.workdir, "Cam1_poseR.txt"), env.Rpose1);
            save_matrix_txt(path_join(env.workdir, "Cam1_poseT.txt"), env.Tpose1);
        };
        save_cams();
        const char* dev_env = getenv("WASS_GPU_DEVICE");
        if (!ctx) {
            if (wass_ctx_create(dev_env ? atoi(dev_env) : device, &ctx) != WASS_OK) { WLOGE << "no usable MI355X GPU / HIP runtime (libwassgpu has no CPU fallback)"; return -1; }
            *ctxp = ctx;
        }
        if (!rectify(env, cfg, ctx)) return -1;   // the reference ignores this result (:1897); stricter here
        env.timer << "Rectification";
        std::cout << "[P|20|100]" << std::endl;
        save_cams();
        if (debug_images) debug_stereo_picture(env);                 // stereo.jpg (:1910-1925)
        WLOG_SCOPE("wass_stereo");
        if (mode && std::string("--rectify-only") == mode) { WLOGI << "All done."; return 0; }
        if (mode && std::string("--measure") == mode) { WLOGE << "--measure needs the interactive GUI, which this build does not have"; return -1; }

        // ---- sgbm_dense_stereo (:764-1020)
        WLOG_SCOPE("sgbm_dense_stereo");
        wass_sgm_params sp;
        sp.min_disp = cfg.get_int("MIN_DISPARITY");
        sp.num_disp = cfg.get_int("MAX_DISPARITY");
        sp.win = cfg.get_int("WINSIZE");
        sp.P1 = cfg.get_int("DENSE_P1_MULT") * sp.win * sp.win;
        sp.P2 = cfg.get_int("DENSE_P2_MULT") * sp.win * sp.win;
        sp.uniq_ratio = cfg.get_int("DENSE_UNIQUENESS_RATIO");
        sp.disp12_max_diff = cfg.get_int("DENSE_DISP12MAXDIFF");
        sp.prefilter_cap = cfg.get_int("DENSE_PREFILTER_CAP");
        sp.speckle_win = cfg.get_int("DENSE_SPECKLE_WINDOW_SIZE");
        sp.speckle_range = cfg.get_int("DENSE_SPECKLE_RANGE");
        sp.ndirs = cfg.get_int("DENSE_PATHS");
        sp.disp_offset = cfg.get_int("DISPARITY_OFFSET");
        sp.dense_scale = cfg.get_double("DENSE_SCALE");
        const int cc_threshold = cfg.get_int("DENSE_DISPARITY_BIGGEST_COMPONENT_THRESHOLD");
        WLOGI << "Disparity offset: " << sp.disp_offset << " px";
        env.disparity_compensation = sp.disp_offset > 0 ? 0 : -sp.disp_offset;
        const int cw = env.right_crop.w, ch = env.right_crop.h;
        int ws = cw, hs = ch;                                        // DENSE_SCALE != 1: SGBM runs on resized crops (:788-796)
        if (wass_dense_input_size(cw, ch, sp.dense_scale, &ws, &hs) != WASS_OK) throw std::runtime_error("invalid DENSE_SCALE");
        WLOGI << "Dense-stereo input resize: [" << cw << " x " << ch << "] -> [" << ws << " x " << hs << "]";
        std::vector<int16_t> disp16((size_t)ws * hs);
        WLOGI << "computing dense disparity map... (may take a while)";
        const int rc = wass_sgm_disparity(ctx, env.right_crop.px.data(), env.left_crop.px.data(), cw, ch, (size_t)cw, &sp, disp16.data());
        gpu_check(ctx, rc, "wass_sgm_disparity", true);
        if (rc == WASS_ERR_COST_OVERFLOW) WLOGE << "matching costs exceeded the int16 range; the disparity is outside the reference's defined behaviour";
        std::vector<float> dispf((size_t)cw * ch);
        const int dil = cfg.get_int("DISP_DILATE_STEPS"), ero = cfg.get_int("DISP_EROSION_STEPS");
        if (dil > 0) WLOGI << "applying dilate filter (" << dil << " steps)"; else WLOGI << "dilate filter skipped.";
        if (ero > 0) WLOGI << "applying erode filter (" << ero << " steps)"; else WLOGI << "erode filter skipped.";
        if (cfg.get_int("MEDIAN_FILTER_WSIZE") >= 3) WLOGI << "applying median filter (window size " << cfg.get_int("MEDIAN_FILTER_WSIZE") << " px.)";
        if (cc_threshold > 0) {
            WLOGI << "extracting the biggest connected component from the disparity map";
            WLOGI << "assuming a sq gradient magnitude of " << cc_threshold;
        }
        gpu_check(ctx, wass_disparity_postprocess_ex(ctx, disp16.data(), ws, hs, &sp, dil, ero, cfg.get_int("MEDIAN_FILTER_WSIZE"), cc_threshold,
                                                     cw, ch, dispf.data()), "wass_disparity_postprocess");
        if (const char* dd = getenv("WASS_PIPE_DUMP")) {              // debugging aid, see frame_pipeline.hpp
            auto dump = [&](const char* name, const void* d, size_t nb) { std::ofstream f(path_join(dd, name).c_str(), std::ios::binary); f.write((const char*)d, (std::streamsize)nb); };
            dump("left_crop.bin", env.left_crop.px.data(), env.left_crop.px.size());
            dump("right_crop.bin", env.right_crop.px.data(), env.right_crop.px.size());
            dump("disp16.bin", disp16.data(), disp16.size() * 2);
            dump("dispf.bin", dispf.data(), dispf.size() * 4);
        }
        DebugMaps dm;
        if (debug_images) {
            if (cc_threshold > 0) {                                  // :958-960, 981-983
                dm.large_gradient.resize((size_t)cw * ch);
                gpu_check(ctx, wass_large_gradient_mask(ctx, cw, ch, dm.large_gradient.data()), "wass_large_gradient_mask");
            }
            dm.disp16 = disp16; dm.ws = ws; dm.hs = hs; dm.dispf = dispf;
            debug_dense_pictures(env, sp, cc_threshold, dm);
        }
        WLOGI << "dense stereo completed successfully";
        env.timer << "Dense Stereo";
        std::cout << "[P|40|100]" << std::endl;

        // ---- triangulate (:1039-1386)
        WLOG_SCOPE("triangulate");
        const int W = env.left.w, H = env.left.h, iw = env.left.w, ih = env.left.h;     // rectified images keep the input size
        auto make_mask = [&](const Image& img, const std::string& key, const char* which) {
            std::vector<uint8_t> m((size_t)iw * ih, 1);
            const std::string name = cfg.get_string(key);
            if (name != "none") {
                const std::string fn = path_join(env.workdir, name);
                WLOGI << "Loading " << fn << " as " << which << " camera mask";
                try {
                    const Image aux = read_image_gray(fn);              // PNG or TIFF
                    if (aux.w == iw && aux.h == ih) for (size_t i = 0; i < m.size(); ++i) m[i] = aux.px[i] > 0 ? 1 : 0;   // threshold(0.5)
                    else WLOGE << "not found or invalid image.";
                } catch (const std::exception&) { WLOGE << "not found or invalid image."; }
            }
            if (cfg.get_bool("DISCARD_BURNED_AREAS")) for (size_t i = 0; i < m.size(); ++i) if (img.px[i] > 254) m[i] = 0;
            return m;
        };
        const std::vector<uint8_t> lmask = make_mask(env.left, "LEFT_MASK_IMAGE", "left"), rmask = make_mask(env.right, "RIGHT_MASK_IMAGE", "right");
        wass_geom g;
        memset(&g, 0, sizeof g);
        auto put = [](double* dst, const Mat& m, int n) { for (int i = 0; i < n; ++i) dst[i] = m.d[i]; };
        put(g.K_left, env.K_left, 9); put(g.K_right, env.K_right, 9); put(g.R, env.R, 9); put(g.T, env.T, 3);
        g.use_custom = env.use_custom ? 1 : 0;
        if (env.use_custom) { put(g.HLi, env.HLi, 9); put(g.HRi, env.HRi, 9); }
        else { memcpy(g.R1, env.rec_R1, sizeof g.R1); memcpy(g.R2, env.rec_R2, sizeof g.R2); memcpy(g.P1, env.rec_P1, sizeof g.P1); memcpy(g.P2, env.rec_P2, sizeof g.P2); }
        g.disparity_compensation = env.disparity_compensation;
        g.dense_scale = sp.dense_scale;
        wass_tri_params tp;
        tp.min_angle_deg = cfg.get_double("TRIANG_MIN_ANGLE");
        tp.bbox[0] = 0; tp.bbox[1] = 0; tp.bbox[2] = iw; tp.bbox[3] = ih;
        if (cfg.get_double("TRIANG_BBOX_TOP") >= 0 && cfg.get_double("TRIANG_BBOX_LEFT") >= 0 && cfg.get_double("TRIANG_BBOX_BOTTOM") >= 0 &&
            cfg.get_double("TRIANG_BBOX_RIGHT") >= 0) {
            tp.bbox[0] = cfg.get_double("TRIANG_BBOX_LEFT"); tp.bbox[1] = cfg.get_double("TRIANG_BBOX_TOP");
            tp.bbox[2] = cfg.get_double("TRIANG_BBOX_RIGHT"); tp.bbox[3] = cfg.get_double("TRIANG_BBOX_BOTTOM");
        }
        tp.cam_distance = env.cam_distance;
        const int roi_l[4] = { env.roi_l.x, env.roi_l.y, env.roi_l.width, env.roi_l.height };
        const int roi_r[4] = { env.roi_r.x, env.roi_r.y, env.roi_r.width, env.roi_r.height };
        WLOGI << "triangulating disparity map";
        uint64_t n_pts = 0;
        gpu_check(ctx, wass_triangulate(ctx, dispf.data(), W, H, roi_l, roi_r, &g, env.right.px.data(), iw, ih, lmask.data(), rmask.data(), &tp, &mesh, &n_pts),
                  "wass_triangulate");
        WLOGI << "... 100%";
        WLOGI << n_pts << " valid points found";
        if (debug_images) {                                          // undistorted/R0.jpg, R1.jpg (:1111-1119, 1216-1338, 1381-1382)
            dm.codes.resize((size_t)roi_r[2] * roi_r[3]);
            gpu_check(ctx, wass_mesh_reject_codes(ctx, mesh, dm.codes.data()), "wass_mesh_reject_codes");
            debug_triangulation_pictures(env, g.disparity_compensation, g.dense_scale, dm);
        }
        if (summary) summary->n_points = n_pts;
        env.timer << "Triangulation";
        std::cout << "[P|60|100]" << std::endl;
        WLOG_SCOPE("wass_stereo");
        if ((long long)n_pts < cfg.get_int("MIN_TRIANGULATED_POINTS")) { WLOGE << "Too few points triangulated, aborting"; throw GpuError("too few points"); }

        // ---- outlier removal (:2046-2050)
        double pct = 0; uint64_t ngaps = 0, csize = 0;
        gpu_check(ctx, wass_mesh_zgap_percentile(ctx, mesh, cfg.get_double("ZGAP_PERCENTILE"), &pct, &ngaps), "wass_mesh_zgap_percentile");
        env.timer << "Z-gap stats";
        if (debug_images) { dm.valid_before.resize((size_t)roi_r[2] * roi_r[3]); gpu_check(ctx, wass_mesh_download(ctx, mesh, dm.valid_before.data(), nullptr, nullptr), "wass_mesh_download"); }
        gpu_check(ctx, wass_mesh_keep_biggest_component(ctx, mesh, pct, &csize), "wass_mesh_keep_biggest_component");
        if (debug_images) {                                          // graph_components.jpg (PovMesh.cpp:222-250, 982-984)
            dm.valid_after.resize(dm.valid_before.size());
            gpu_check(ctx, wass_mesh_download(ctx, mesh, dm.valid_after.data(), nullptr, nullptr), "wass_mesh_download");
            debug_components_picture(env, dm);
        }
        WLOG_SCOPE("cluster");
        WLOGI << "biggest component size: " << csize << " (px)";
        env.timer << "Outlier removal";
        std::cout << "[P|80|100]" << std::endl;
        WLOG_SCOPE("wass_stereo");

        const int mw = roi_r[2], mh = roi_r[3];
        std::vector<uint8_t> hv, hg; std::vector<double> hp;
        auto download = [&]() { hv.resize((size_t)mw * mh); hg.resize(hv.size()); hp.resize(hv.size() * 3); gpu_check(ctx, wass_mesh_download(ctx, mesh, hv.data(), hp.data(), hg.data()), "wass_mesh_download"); };
        if (cfg.get_bool("SAVE_FULL_MESH")) {
            download();
            if (!save_ply_points(path_join(env.workdir, "mesh_full.ply"), hv, hp, hg)) { WLOGE << "unable to save mesh data."; throw GpuError("write failed"); }
        }

        // ---- plane (:2062-2107)
        WLOGI << "estimating best fitting plane...";
        const int rounds = cfg.get_int("PLANE_RANSAC_ROUNDS");
        std::vector<int32_t> uv((size_t)std::max(rounds, 1) * 6);
        if (rounds <= 0 || wass_ransac_sample_seeded(ransac_seed, mw, mh, rounds, uv.data()) != WASS_OK) throw std::runtime_error("invalid PLANE_RANSAC_ROUNDS / mesh size");
        double plane[4] = { 0, 0, 0, 0 };
        uint64_t best = 0; int found = 0;
        gpu_check(ctx, wass_mesh_ransac_plane(ctx, mesh, uv.data(), rounds, cfg.get_double("PLANE_RANSAC_THRESHOLD"), plane, &best, &found), "wass_mesh_ransac_plane");
        WLOG_SCOPE("ransac_find_plane");
        WLOGI << rounds << " ransac rounds, " << best << " best inliers";
        WLOGI << "ransac plane coeffs: " << plane[0] << " " << plane[1] << " " << plane[2] << " " << plane[3];
        WLOG_SCOPE("wass_stereo");
        bool have_plane = false;
        if (found) {
            env.timer << "Plane fitting";
            std::cout << "[P|90|100]" << std::endl;
            WLOGI << "refining plane";
            uint64_t kept = 0, ninl = 0;
            gpu_check(ctx, wass_mesh_crop_plane(ctx, mesh, plane, cfg.get_double("PLANE_RANSAC_THRESHOLD"), &kept), "wass_mesh_crop_plane");
            wass_refine_params rp;
            rp.xmin = cfg.get_double("PLANE_REFINE_XMIN"); rp.xmax = cfg.get_double("PLANE_REFINE_XMAX");
            rp.ymin = cfg.get_double("PLANE_REFINE_YMIN"); rp.ymax = cfg.get_double("PLANE_REFINE_YMAX");
            rp.max_distance = cfg.get_double("PLANE_REFINEMENT_MAX_DISTANCE");
            rp.weight_by_distance = cfg.get_bool("PLANE_WEIGHT_PROPORTIONAL_TO_DISTANCE");
            rp.central_third_only = cfg.get_bool("PLANE_USE_CENTRAL_THIRD_ONLY");
            gpu_check(ctx, wass_mesh_refine_plane(ctx, mesh, &rp, plane, &ninl), "wass_mesh_refine_plane");
            WLOG_SCOPE("refine_plane");
            WLOGI << "refinement inliers (after cropping): " << ninl;
            WLOGI << "estimated plane coeffs: " << plane[0] << " " << plane[1] << " " << plane[2] << " " << plane[3];
            WLOG_SCOPE("wass_stereo");
            {   // plane_refinement_inliers.xyz: every 10th refinement inlier in raster order (:2077-2085), selected on the device:
                // ~150 000 points come back instead of the whole mesh (126 MB at 2456 x 2058, 60-100 ms per frame)
                double* sel = nullptr;
                uint64_t nsel = 0;
                gpu_check(ctx, wass_mesh_refinement_inliers(ctx, mesh, &rp, 10, &sel, &nsel), "wass_mesh_refinement_inliers");
                write_inliers_xyz(path_join(env.workdir, "plane_refinement_inliers.xyz"), sel, (size_t)nsel);
                wass_free(sel);
            }
            gpu_check(ctx, wass_mesh_crop_plane(ctx, mesh, plane, cfg.get_double("PLANE_MAX_DISTANCE"), &kept), "wass_mesh_crop_plane");
            WLOG_SCOPE("crop_plane");
            WLOGI << "number of points after plane cropping: " << kept;
            WLOG_SCOPE("wass_stereo");
            env.timer << "Plane refinement";
            std::ofstream ofs(path_join(env.workdir, "plane.txt").c_str());
            ofs << std::setprecision(20);
            for (int i = 0; i < 4; ++i) ofs << plane[i] << std::endl;
            have_plane = true;
            if (summary) { summary->have_plane = 1; for (int i = 0; i < 4; ++i) summary->plane[i] = plane[i]; }
        } else {
            WLOGE << "ransac failed. I'll continue anyway but plane data won't be available!";
            std::ofstream ofs(path_join(env.workdir, "plane.txt").c_str());
            ofs << "nan nan nan nan" << std::endl;
        }

        // ---- export (:2110-2135)
        WLOGI << "Exporting point cloud data";
        if (cfg.get_bool("SAVE_AS_PLY")) {
            download();
            if (!save_ply_points(path_join(env.workdir, "mesh.ply"), hv, hp, hg)) { WLOGE << "unable to save mesh data."; throw GpuError("write failed"); }
        }
        if (cfg.get_bool("SAVE_COMPRESSED")) {
            WLOG_SCOPE("save_as_xyz_compressed");
            WLOGI << "saving mesh as compressed xyz file...";
            void* bytes = nullptr; size_t nb = 0;
            gpu_check(ctx, wass_mesh_encode_xyzc(ctx, mesh, have_plane ? plane : nullptr, &bytes, &nb), "wass_mesh_encode_xyzc");
            const std::string xyzc_path = path_join(env.workdir, "mesh_cam.xyzC");
            // written under a temporary name and renamed: a worker killed in the middle of the (possibly asynchronous) write
            // must not leave a truncated mesh_cam.xyzC behind that --skip-existing would take for a finished frame
            auto write_xyzc = [xyzc_path, bytes, nb]() {
                const std::string tmp = xyzc_path + ".tmp";
                bool ok;
                {
                    std::ofstream ofs(tmp.c_str(), std::ios::binary);
                    ok = !ofs.fail() && ofs.write((const char*)bytes, (std::streamsize)nb).good();
                    ofs.close();
                    ok = ok && !ofs.fail();
                }
                ok = ok && rename(tmp.c_str(), xyzc_path.c_str()) == 0;
                wass_free(bytes);
                if (!ok) fprintf(stderr, "wass_stereo [error] unable to save %s\n", xyzc_path.c_str());
                return ok;
            };
            if (writer) writer->submit(write_xyzc);              // sequence driver: the 28 MB write overlaps the next frame
            else if (!write_xyzc()) { WLOGE << "unable to save mesh data"; throw GpuError("write failed"); }
            WLOGI << "total data size: " << ((double)nb / 1E6) << " MB";
            WLOG_SCOPE("wass_stereo");
        } else {
            download();
            if (!save_xyz_binary(path_join(env.workdir, "mesh_cam.xyzbin"), hv, hp)) { WLOGE << "unable to save mesh data"; throw GpuError("write failed"); }
        }
        env.timer.stop();
        std::cout << "[P|100|100]" << std::endl;
        show_time_stats(env.timer);
        WLOGI << "All done.";
    } catch (const GpuError& e) {
        WLOGE << e.what();
        ret = -1;
    } catch (const std::runtime_error& e) {
        WLOGE << e.what();
        ret = -1;
    }
    if (mesh) wass_mesh_destroy(mesh);
    return ret;
}

}  // namespace wassframe
