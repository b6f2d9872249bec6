// frame_pipeline.hpp -- the device-resident, host-sync-free frame chain of the C++ drop-in: what wass_stereo's main()
// (/root/reference/src/wass_stereo/wass_stereo.cpp:1833-2147) does with one workdir, cut into three phases that run on
// different threads for different frames at the same time:
//
//   prepare(job)   any thread   configuration echo, calibration, PNG inflation, previews, camera files, the rectification's
//                               decisions (rectify_plan) -- host only, nothing of it touches the GPU
//   submit(job)    owner thread both pictures to HBM (wass_upload_async, one frame ahead), rectification resampling
//                               (wass_warp_perspective_dev / wass_remap_cubic_dev), wass_sgm_disparity_dev, disparity clean-up,
//                               wass_triangulate_dev and the whole mesh tail (wass_mesh_finish_frame_async_ex) enqueued on the
//                               context's streams WITHOUT a host synchronisation; hands back the PREVIOUS frame, whose one
//                               result record and file image have arrived meanwhile
//   finish(job)    any thread   the log lines that carry numbers (from the result record), plane.txt,
//                               plane_refinement_inliers.xyz, mesh_cam.xyzC, the time table, wass_stereo_log.txt
//
// Same files, byte for byte, as the stage-by-stage wass_run_frame (tests/test_batch_driver.py, tests/test_cli.py); used by
// wass_stereo_batch for every frame and by wass_stereo for its single frame whenever the configuration allows it
// (pipeline_eligible: the options that need an intermediate mesh or map on the host keep the stage-by-stage calls).
#pragma once

#include <atomic>
#include <condition_variable>
#include <deque>
#include <mutex>

#include "prepare_common.hpp"
#include "wass_frame.hpp"

namespace wassframe {

struct FrameJob {
    size_t index = 0;
    std::string workdir;
    std::string config_path;         // as the caller named it, for the log (empty: the pipeline's)
    Env env;
    std::string log;                 // the frame's wass_stereo_log.txt, in the order the reference writes it
    int rc = 0;                      // -1: the frame failed (the log says why)
    bool skipped = false;            // --skip-existing: nothing to do, summary read back from plane.txt
    bool staged = false;
    // prepare-less mode: the frame starts from the cameras' RAW pictures (c0, c1) and a calibration directory -- what
    // wass_prepare would have turned into <workdir>/undistorted/*.png first
    bool raw = false;
    std::string c0, c1;
    DeferredFiles deferred;          // the files prepare() would have written, when it ran under a DeferredFilesScope (a frame computed before it was
                                     // asked for: stereo_server.hpp); finish() writes them first
    Preload* pre = nullptr;          // the workdir's two pictures, decoded ahead of the request (the resident worker's read-ahead); used and emptied by prepare
    int prev_w = 0, prev_h = 0;      // size of the scaled previews, written once the undistorted pictures are back (0: none)
    int img_w = 0, img_h = 0;        // size of the cameras' pictures (the driver drops the decoded pictures once they are staged)
    // LEFT_MASK_IMAGE / RIGHT_MASK_IMAGE (wass_stereo.cpp:1059-1087): thresholded to 0/1 on the decode thread (empty: none);
    // mask_log: the lines triangulate() would have logged while loading them, replayed at their place
    std::vector<uint8_t> fmask[2];
    std::string mask_log;
    // debug pictures: the mesh is kept until its rejection codes have been fetched; the maps the pictures are drawn from
    wass_mesh* mesh = nullptr;
    int disp_slot = 0;
    DebugMaps dbg;
    bool have_dbg = false;
    uint64_t dbg_ticket = 0;         // the pictures were drawn and coded on the device (wass_debug_pictures_async): dbg_bytes[k] of file k in the output set
    size_t dbg_bytes[WASS_DEBUG_PICTURES] = {};
    unsigned int ransac_seed = 0;
    int in_slot = -1, out_slot = -1;
    long long sgm_call = -1;         // which wass_sgm_disparity_dev call of the pipeline's context produced the frame's disparity
    double t_prepare0 = 0, t_loaded = 0, t_planned = 0, t_submit0 = 0, t_submitted = 0, t_result = 0;
    wass_frame_result res{};
    wass_sgm_timings sgm{};
    bool have_sgm = false;
    FrameSummary summary;
};

class FramePipeline {
public:
    // input sets: frames n+1 and n+2 may be staged (uploaded) while frame n runs and frame n-1's tail still reads its picture.  Two
    // frames ahead, not one: the copy stream carries uploads AND the results' downloads in the order they were enqueued, so the
    // upload staged right before frame n is submitted queues behind frame n-1's downloads, which wait for frame n-1's tail -- and
    // that tail ends late in frame n.  One frame ahead that was the upload frame n+1 was waiting for (a 0.5 ms stall per frame once
    // the downloads grew by the inlier text); two ahead it is frame n+2's, with a whole frame of slack.
    static constexpr int NIN = 6;    // (two staged ahead, the one being submitted, two pending, one whose tail may still read its picture)
    struct Options {
        int out_slots = 4;           // pinned output sets (file image + inlier points) that writer threads may hold at once
        bool inliers_file = true;    // plane_refinement_inliers.xyz (a debug artefact of the reference; 14 MB of text per 5-megapixel frame)
        bool live = false;           // single-frame executable: echo log and progress markers to stdout as the phases end
        bool echo = true;            // live && !echo: the markers are recorded in the frame's log but nothing is printed here (the resident
                                     // worker relays the log to the wass_stereo process that sent the frame, stereo_server.hpp)
        bool debug_pictures = false; // the reference's debug pictures (stereo.jpg ... graph_components.jpg): rendered and JPEG-coded on the device
                                     // behind the frame's tail (dbg_dev_); in the host form (dbg_host_: PNG format, component-option masks) every
                                     // frame's intermediate maps come back once it is complete, which takes the pipeline down to one frame
        int max_pending = 2;         // frames whose record has not been read when the next tail is enqueued: 2 keeps a sequence driver two frames
                                     // ahead of the GPU (throughput); 1 hands a frame out one submission earlier (latency: the resident worker,
                                     // whose callers each wait for ONE frame)
        bool device_previews = true; // the scaled previews 0000000X_s.png are resized on the GPU and written by finish() (sequence drivers);
                                     // false: load_data writes them from the decoded pictures before the GPU is needed, like the reference
        // -1: as this process's environment says (WASS_DEBUG_FORMAT=png, WASS_HOST_INLIER_TEXT, WASS_HOST_DEBUG_PICTURES); 0 / 1: as the
        // CALLER's environment says -- the resident worker serves callers whose environments differ from the one it was started in
        int debug_png = -1, host_inlier_text = -1, host_debug_pictures = -1;
        const PrepareSetup* prep = nullptr;   // prepare-less mode: the calibration directory (jobs with raw = true need it)
        bool save_undistorted = false;        // ... and whether undistorted/0000000X.png are written all the same
    };

    // The context is created on `device` when the first frame needs the GPU (as wass_stereo does: a sequence whose frames
    // are all finished, or all unreadable, never initialises HIP); a frame that finds no GPU fails loudly.
    FramePipeline(int device, const Config& cfg, const std::string& config_path, const Options& opt)
        : device_(device), cfg_(cfg), config_path_(config_path), opt_(opt), max_pending_(opt.max_pending)
    {
        if (opt_.host_inlier_text >= 0) device_text_ = opt_.host_inlier_text == 0;
        if (opt_.debug_pictures) {
            const char* e = getenv("WASS_HOST_DEBUG_PICTURES");
            const bool png = opt_.debug_png >= 0 ? opt_.debug_png != 0 : debug_png();
            const bool host = opt_.host_debug_pictures >= 0 ? opt_.host_debug_pictures != 0 : (e && atoi(e) != 0);
            dbg_host_ = png || cfg.get_int("DENSE_DISPARITY_BIGGEST_COMPONENT_THRESHOLD") > 0 || host;
            dbg_dev_ = !dbg_host_;
        }
        sp_.min_disp = cfg.get_int("MIN_DISPARITY");
        sp_.num_disp = cfg.get_int("MAX_DISPARITY");
        sp_.win = cfg.get_int("WINSIZE");
        sp_.P1 = cfg.get_int("DENSE_P1_MULT") * sp_.win * sp_.win;
        sp_.P2 = cfg.get_int("DENSE_P2_MULT") * sp_.win * sp_.win;
        sp_.uniq_ratio = cfg.get_int("DENSE_UNIQUENESS_RATIO");
        sp_.disp12_max_diff = cfg.get_int("DENSE_DISP12MAXDIFF");
        sp_.prefilter_cap = cfg.get_int("DENSE_PREFILTER_CAP");
        sp_.speckle_win = cfg.get_int("DENSE_SPECKLE_WINDOW_SIZE");
        sp_.speckle_range = cfg.get_int("DENSE_SPECKLE_RANGE");
        sp_.ndirs = cfg.get_int("DENSE_PATHS");
        sp_.disp_offset = cfg.get_int("DISPARITY_OFFSET");
        sp_.dense_scale = cfg.get_double("DENSE_SCALE");
        rp_.xmin = cfg.get_double("PLANE_REFINE_XMIN"); rp_.xmax = cfg.get_double("PLANE_REFINE_XMAX");
        rp_.ymin = cfg.get_double("PLANE_REFINE_YMIN"); rp_.ymax = cfg.get_double("PLANE_REFINE_YMAX");
        rp_.max_distance = cfg.get_double("PLANE_REFINEMENT_MAX_DISTANCE");
        rp_.weight_by_distance = cfg.get_bool("PLANE_WEIGHT_PROPORTIONAL_TO_DISTANCE");
        rp_.central_third_only = cfg.get_bool("PLANE_USE_CENTRAL_THIRD_ONLY");
        out_free_.assign((size_t)std::max(1, opt_.out_slots), true);
        out_.resize(out_free_.size());
    }
    ~FramePipeline()
    {
        if (timing_ && nsub_ > 0) {
            static const char* nm[7] = { "out slot + raw + previews", "rectify + masks", "wass_sgm_disparity_dev", "postprocess", "triangulate", "collect the frame before last", "finish_frame_async" };
            fprintf(stderr, "FramePipeline: submit() per frame over %d frames:", nsub_);
            for (int k = 0; k < 7; ++k) fprintf(stderr, "  %s %.2f ms", nm[k], 1e3 * lap_[k] / nsub_);
            fprintf(stderr, "\n");
        }
        if (!ctx_) return;
        (void)wass_ctx_synchronize(ctx_);
        release_buffers();
        for (auto& o : out_) { if (o.xyzc) wass_pinned_free(ctx_, o.xyzc); if (o.inl) wass_pinned_free(ctx_, o.inl); if (o.inl_text) wass_pinned_free(ctx_, o.inl_text); for (auto& u : o.prev) if (u) wass_pinned_free(ctx_, u); for (auto& u : o.und) if (u) wass_pinned_free(ctx_, u); if (o.ccmask) wass_pinned_free(ctx_, o.ccmask); if (o.dbg) wass_pinned_free(ctx_, o.dbg); }
        wass_ctx_destroy(ctx_);
    }
    // the pipeline's context, created if need be (owner thread); nullptr when there is no usable GPU
    wass_ctx* context()
    {
        if (!ctx_ && !ctx_failed_) {
            if (wass_ctx_create(device_, &ctx_) != WASS_OK) { ctx_ = nullptr; ctx_failed_ = true; }
            else if (wass_ctx_set_tail_overlap(ctx_, 1) != WASS_OK) { wass_ctx_destroy(ctx_); ctx_ = nullptr; ctx_failed_ = true; }
        }
        return ctx_;
    }
    FramePipeline(const FramePipeline&) = delete;
    FramePipeline& operator=(const FramePipeline&) = delete;

    // ---- phase 1 (any thread, host only): wass_stereo.cpp:1840-1908 without the resampling
    void prepare(FrameJob& job) const
    {
        LogSinkScope sink(&job.log);
        job.t_prepare0 = Timer::now();
        job.env.workdir = job.workdir;
        Env& env = job.env;
        WLOG_SCOPE("wass_stereo");
        try {
            if (job.raw) {
                if (!opt_.prep) throw std::runtime_error("prepare-less frame without a calibration directory");
                create_directories(env.workdir);
                if (opt_.save_undistorted) create_directories(path_join(env.workdir, "undistorted"));
                write_prepared_calibration(env.workdir, *opt_.prep);
            }
            WLOGI << "Loading configuration file " << (job.config_path.empty() ? config_path_ : job.config_path);
            if (save_configuration(cfg_, path_join(env.workdir, "stereo_config.txt")) != 0) WLOGE << "Unable to save stereo configuration file";
            job.ransac_seed = (unsigned int)time(0);
            if (cfg_.get_int("RANDOM_SEED") != -1) { job.ransac_seed = (unsigned int)cfg_.get_int("RANDOM_SEED"); WLOGI << "random seed set to: " << cfg_.get_int("RANDOM_SEED"); }
            WLOGI << "Reconstructing " << env.workdir;
            env.timer.start();
            env.cam_distance = 1.0;
            if (job.raw) {
                // wass_prepare's part of the workdir (wass_prepare.cpp:505-533), then wass_stereo's load_data on it -- with the
                // raw pictures in place of undistorted/*.png: the GPU undistorts them inside the frame chain (submit)
                if (!load_calibration(env)) { job.rc = -1; return; }
                try { env.left = read_image_gray(job.c0); env.right = read_image_gray(job.c1); }
                catch (const std::exception& e) { WLOG_SCOPE("load_data"); WLOGE << "unable to load input images: " << e.what(); job.rc = -1; return; }
                if (!images_loaded(env)) { job.rc = -1; return; }
                job.img_w = env.left.w; job.img_h = env.left.h;
                input_scale_outputs(env, cfg_, false, nullptr, &job.prev_w, &job.prev_h);
            } else if (opt_.device_previews) {
                if (!load_data(env, cfg_, job.pre, nullptr, false, &job.prev_w, &job.prev_h)) { job.rc = -1; return; }   // previews: on the GPU (submit)
            } else if (!load_data(env, cfg_, job.pre, nullptr)) { job.rc = -1; return; }
            job.t_loaded = Timer::now();
            marker(job, 10);
            auto save_cams = [&]() {
                save_matrix_txt(path_join(env.workdir, "P0cam.txt"), env.P0);
                save_matrix_txt(path_join(env.workdir, "P1cam.txt"), env.P1);
                save_matrix_txt(path_join(env.workdir, "Cam0_poseR.txt"), env.Rpose0);
                save_matrix_txt(path_join(env.workdir, "Cam0_poseT.txt"), env.Tpose0);
                save_matrix_txt(path_join(env.workdir, "Cam1_poseR.txt"), env.Rpose1);
                save_matrix_txt(path_join(env.workdir, "Cam1_poseT.txt"), env.Tpose1);
            };
            save_cams();
            if (!rectify_plan(env, cfg_)) { job.rc = -1; return; }
            save_cams();
            for (int side = 0; side < 2; ++side) {              // the masks belong to the pictures that are left / right NOW (after a swap)
                const std::string name = cfg_.get_string(side == 0 ? "LEFT_MASK_IMAGE" : "RIGHT_MASK_IMAGE");
                if (name == "none") continue;
                const std::string fn = path_join(env.workdir, name);
                LogSinkScope ms(&job.mask_log);
                WLOG_SCOPE("triangulate");
                WLOGI << "Loading " << fn << " as " << (side == 0 ? "left" : "right") << " camera mask";
                try {
                    const Image aux = read_image_gray(fn);
                    if (aux.w == env.left.w && aux.h == env.left.h) {
                        job.fmask[side].resize(aux.px.size());
                        for (size_t i = 0; i < aux.px.size(); ++i) job.fmask[side][i] = aux.px[i] > 0 ? 1 : 0;    // threshold(0.5)
                    } else WLOGE << "not found or invalid image.";
                } catch (const std::exception&) { WLOGE << "not found or invalid image."; }
            }
            job.t_planned = Timer::now();
        } catch (const std::exception& e) {
            WLOG_SCOPE("wass_stereo");
            WLOGE << e.what();
            job.rc = -1;
        }
    }

    // May the frame be staged AHEAD of its turn?  Only with the buffers as they are: a change of picture or ROI size
    // re-allocates them, which must wait until the frames staged before it have been submitted.
    bool same_geometry(const FrameJob& job) const
    {
        const Env& e = job.env;
        return e.left.w == W_ && e.left.h == H_ && e.roi_l.width == cwl_ && e.roi_l.height == chl_ && e.roi_r.width == cwr_ && e.roi_r.height == chr_;
    }
    // ---- phase 2 (owner thread): the frame's pictures into the pinned ring and on their way to HBM.  Called for frame n+1
    // right before frame n is submitted (when frame n+1 is ready by then), so that the transfer runs underneath frame n.
    // A frame whose pictures or ROIs differ in size from the buffers as they are is NOT staged ahead of its turn while frames staged
    // before it wait for theirs: ensure_buffers would free the very buffers their pictures are in (a server fed by two sequences of
    // different cameras with one configuration text did exactly that).  Its turn comes in submit() (its_turn), by which time they have
    // been submitted; a frame staged AFTER it in the meantime goes back to "not staged" and is staged again when it is submitted.
    void stage(FrameJob& job, bool its_turn = false)
    {
        if (job.staged || job.rc != 0 || job.skipped) return;
        if (!staged_.empty() && !same_geometry(job)) {
            if (!its_turn) return;
            for (FrameJob* s : staged_) { s->staged = false; s->in_slot = -1; }
            staged_.clear();
        }
        LogSinkScope sink(&job.log);
        try {
            const Env& env = job.env;
            if (!context()) throw std::runtime_error("no usable MI355X GPU / HIP runtime (libwassgpu has no CPU fallback)");
            ensure_buffers(env.left.w, env.left.h, env.roi_l, env.roi_r);
            const int k = next_in_;
            next_in_ = (next_in_ + 1) % NIN;
            const size_t n = (size_t)W_ * H_;
            memcpy(in_[k].h_l, env.left.px.data(), n);
            memcpy(in_[k].h_r, env.right.px.data(), n);
            check(wass_upload_async(ctx_, job.raw ? in_[k].d_rawl : in_[k].d_l, in_[k].h_l, n), "wass_upload_async");
            check(wass_upload_async(ctx_, job.raw ? in_[k].d_rawr : in_[k].d_r, in_[k].h_r, n), "wass_upload_async");
            for (int side = 0; side < 2; ++side)
                if (!job.fmask[side].empty()) {
                    uint8_t*& hp = side == 0 ? in_[k].h_fl : in_[k].h_fr;
                    uint8_t*& dp = side == 0 ? in_[k].d_fl : in_[k].d_fr;
                    if (!hp) { void* p = nullptr; check(wass_pinned_alloc(ctx_, n + 4, &p), "wass_pinned_alloc"); hp = (uint8_t*)p; }
                    if (!dp) { void* p = nullptr; check(wass_device_alloc(ctx_, n + 4, &p), "wass_device_alloc"); dp = (uint8_t*)p; }
                    memcpy(hp, job.fmask[side].data(), n);
                    check(wass_upload_async(ctx_, dp, hp, n), "wass_upload_async");
                }
            job.in_slot = k;
            job.staged = true;
            staged_.push_back(&job);
        } catch (const std::exception& e) {
            WLOG_SCOPE("wass_stereo");
            WLOGE << e.what();
            job.rc = -1;
        }
    }

    // Enqueues the frame and appends to `done` the frames that are complete as far as the GPU is concerned (at most the
    // previous frame, and `job` itself if it cannot run), in submission order.  They go to finish().
    void submit(FrameJob& job, std::vector<FrameJob*>& done)
    {
        struct Early { FramePipeline* p; std::vector<FrameJob*>& d; size_t at; ~Early() { d.insert(d.begin() + (long)at, p->early_.begin(), p->early_.end()); p->early_.clear(); } } early{ this, done, done.size() };
        if (job.rc != 0 || job.skipped) {                 // nothing to enqueue: keep the order
            while (FrameJob* p = collect()) done.push_back(p);
            done.push_back(&job);
            return;
        }
        stage(job, true);
        for (auto it = staged_.begin(); it != staged_.end(); ++it) if (*it == &job) { staged_.erase(it); break; }
        if (job.rc != 0) { while (FrameJob* p = collect()) done.push_back(p); done.push_back(&job); return; }
        // debug pictures: the previous frame's maps are fetched from buffers this frame is about to overwrite
        if (dbg_host_) while (FrameJob* p = collect()) done.push_back(p);
        LogSinkScope sink(&job.log);
        job.t_submit0 = Timer::now();
        Env& env = job.env;
        wass_mesh* mesh = nullptr;
        double tp = job.t_submit0;                        // WASS_PIPE_TIMING=1: where the submitting thread's time goes (printed when the pipeline is destroyed)
        auto lap = [&](int k) { if (timing_) { const double t = Timer::now(); lap_[k] += t - tp; tp = t; } };
        try {
            const int k = job.in_slot;
            const int rl[4] = { env.roi_l.x, env.roi_l.y, env.roi_l.width, env.roi_l.height };
            const int rr[4] = { env.roi_r.x, env.roi_r.y, env.roi_r.width, env.roi_r.height };
            job.out_slot = acquire_out((size_t)rr[2] * rr[3], job.raw && opt_.save_undistorted ? (size_t)W_ * H_ : 0, opt_.debug_pictures);
            if (job.raw) {
                // ---- wass_prepare's process_image (wass_prepare.cpp:257-275) on the device: optional CLAHE, cv::undistort.  The
                // camera a picture came from decides its parameters -- rectify_plan may have swapped left and right since.
                const PrepareSetup& ps = *opt_.prep;
                for (int side = 0; side < 2; ++side) {
                    const int cam = side == 0 ? env.left_index : env.right_index;
                    const uint8_t* src = side == 0 ? in_[k].d_rawl : in_[k].d_rawr;
                    uint8_t* dst = side == 0 ? in_[k].d_l : in_[k].d_r;
                    if (ps.clahe_tiles[cam] > 0) {
                        check(wass_clahe_dev(ctx_, src, W_, H_, (size_t)W_, ps.clahe_clip[cam], ps.clahe_tiles[cam], ps.clahe_tiles[cam], in_[k].d_tmp), "wass_clahe");
                        src = in_[k].d_tmp;
                    }
                    check(wass_undistort_dev(ctx_, src, W_, H_, (size_t)W_, ps.intr[cam].d.data(), ps.dist[cam].d.data(), (int)ps.dist[cam].d.size(), dst), "wass_undistort");
                    // back to the host only when undistorted/*.png are asked for (the previews are resized on the device, below)
                    if (opt_.save_undistorted) check(wass_download_async(ctx_, out_[job.out_slot].und[side], dst, (size_t)W_ * H_), "wass_download_async");
                }
            }
            // ---- the scaled previews 0000000X_s.png of load_data (:401-417): cv::resize(INTER_CUBIC) of the two (undistorted) pictures,
            // on the device; they come back with the frame's result and are deflated by the writer thread (finish)
            if (job.prev_w > 0 && job.prev_h > 0) {
                const size_t pn = (size_t)job.prev_w * job.prev_h;
                OutSet& o = out_[job.out_slot];
                for (int side = 0; side < 2; ++side) {
                    uint8_t*& dp = side == 0 ? in_[k].d_pl : in_[k].d_pr;
                    if (in_[k].prev_cap < pn) { if (dp) wass_device_free(ctx_, dp); dp = nullptr; }
                    if (!dp) { void* p = nullptr; check(wass_device_alloc(ctx_, pn + 4, &p), "wass_device_alloc"); dp = (uint8_t*)p; }
                }
                in_[k].prev_cap = std::max(in_[k].prev_cap, pn);
                for (int side = 0; side < 2; ++side) {
                    if (o.prev_cap[side] < pn) {
                        if (o.prev[side]) wass_pinned_free(ctx_, o.prev[side]);
                        o.prev[side] = nullptr; o.prev_cap[side] = 0;
                        void* p = nullptr;
                        check(wass_pinned_alloc(ctx_, pn, &p), "wass_pinned_alloc");
                        o.prev[side] = (uint8_t*)p; o.prev_cap[side] = pn;
                    }
                    uint8_t* dp = side == 0 ? in_[k].d_pl : in_[k].d_pr;
                    check(wass_resize_cubic_u8_dev(ctx_, side == 0 ? in_[k].d_l : in_[k].d_r, W_, H_, (size_t)W_, dp, job.prev_w, job.prev_h), "wass_resize_cubic_u8");
                    check(wass_download_async(ctx_, o.prev[side], dp, pn), "wass_download_async");
                }
            }
            lap(0);
            // ---- rectify(): the resampling (:515-528, 600-607), ROI crop fused, from the device-resident pictures
            WLOG_SCOPE("rectify");
            if (env.use_custom) {
                check(wass_warp_perspective_dev(ctx_, in_[k].d_l, W_, H_, (size_t)W_, env.HL.d.data(), W_, H_, rl, in_[k].d_cl), "wass_warp_perspective");
                check(wass_warp_perspective_dev(ctx_, in_[k].d_r, W_, H_, (size_t)W_, env.HR.d.data(), W_, H_, rl, in_[k].d_cr), "wass_warp_perspective");
            } else {
                ensure_maps(env);
                check(wass_remap_cubic_dev(ctx_, in_[k].d_l, W_, H_, (size_t)W_, d_map_[0], d_map_[1], W_, H_, rl, in_[k].d_cl), "wass_remap_cubic");
                check(wass_remap_cubic_dev(ctx_, in_[k].d_r, W_, H_, (size_t)W_, d_map_[2], d_map_[3], W_, H_, rr, in_[k].d_cr), "wass_remap_cubic");
            }
            WLOGI << "rectification map generated. Size: " << rl[2] << "x" << rl[3];
            marker(job, 20);
            const bool burned = cfg_.get_bool("DISCARD_BURNED_AREAS");
            const bool mask_l = burned || !job.fmask[0].empty(), mask_r = burned || !job.fmask[1].empty();
            {                                               // :1057-1093 -- the masks of the ORIGINAL (undistorted) pictures
                const size_t n = (size_t)W_ * H_;
                if (mask_l) check(wass_camera_mask_dev(ctx_, burned ? in_[k].d_l : nullptr, job.fmask[0].empty() ? nullptr : in_[k].d_fl, n, in_[k].d_ml), "wass_camera_mask");
                if (mask_r) check(wass_camera_mask_dev(ctx_, burned ? in_[k].d_r : nullptr, job.fmask[1].empty() ? nullptr : in_[k].d_fr, n, in_[k].d_mr), "wass_camera_mask");
            }
            lap(1);
            // ---- sgbm_dense_stereo (:764-1020)
            WLOG_SCOPE("sgbm_dense_stereo");
            const int cw = rr[2], ch = rr[3];
            const int cc_threshold = cfg_.get_int("DENSE_DISPARITY_BIGGEST_COMPONENT_THRESHOLD");
            WLOGI << "Disparity offset: " << sp_.disp_offset << " px";
            env.disparity_compensation = sp_.disp_offset > 0 ? 0 : -sp_.disp_offset;
            WLOGI << "Dense-stereo input resize: [" << cw << " x " << ch << "] -> [" << cw << " x " << ch << "]";
            WLOGI << "computing dense disparity map... (may take a while)";
            job.disp_slot = nsub_ & 1;
            int16_t* d16 = d_disp16_[job.disp_slot];
            check(wass_sgm_disparity_dev(ctx_, in_[k].d_cr, in_[k].d_cl, cw, ch, (size_t)cw, &sp_, d16), "wass_sgm_disparity");
            job.sgm_call = (long long)sgm_calls();         // the library's own count: no shadow counter to fall out of step after a failed frame
            lap(2);
            const int dil = cfg_.get_int("DISP_DILATE_STEPS"), ero = cfg_.get_int("DISP_EROSION_STEPS"), med = cfg_.get_int("MEDIAN_FILTER_WSIZE");
            if (dil > 0) WLOGI << "applying dilate filter (" << dil << " steps)"; else WLOGI << "dilate filter skipped.";
            if (ero > 0) WLOGI << "applying erode filter (" << ero << " steps)"; else WLOGI << "erode filter skipped.";
            if (med >= 3) WLOGI << "applying median filter (window size " << med << " px.)";
            if (cc_threshold > 0) {
                WLOGI << "extracting the biggest connected component from the disparity map";
                WLOGI << "assuming a sq gradient magnitude of " << cc_threshold;
            }
            check(wass_disparity_postprocess_ex_dev(ctx_, d16, cw, ch, &sp_, dil, ero, med, cc_threshold, cw, ch, d_dispf_), "wass_disparity_postprocess");
            WLOGI << "dense stereo completed successfully";
            marker(job, 40);
            lap(3);
            // ---- triangulate (:1039-1386)
            WLOG_SCOPE("triangulate");
            wass_geom g;
            memset(&g, 0, sizeof g);
            auto put = [](double* dst, const Mat& m, int n) { for (int i = 0; i < n; ++i) dst[i] = m.d[i]; };
            put(g.K_left, env.K_left, 9); put(g.K_right, env.K_right, 9); put(g.R, env.R, 9); put(g.T, env.T, 3);
            g.use_custom = env.use_custom ? 1 : 0;
            if (env.use_custom) { put(g.HLi, env.HLi, 9); put(g.HRi, env.HRi, 9); }
            else { memcpy(g.R1, env.rec_R1, sizeof g.R1); memcpy(g.R2, env.rec_R2, sizeof g.R2); memcpy(g.P1, env.rec_P1, sizeof g.P1); memcpy(g.P2, env.rec_P2, sizeof g.P2); }
            g.disparity_compensation = env.disparity_compensation;
            g.dense_scale = sp_.dense_scale;
            wass_tri_params tp;
            tp.min_angle_deg = cfg_.get_double("TRIANG_MIN_ANGLE");
            tp.bbox[0] = 0; tp.bbox[1] = 0; tp.bbox[2] = W_; tp.bbox[3] = H_;
            if (cfg_.get_double("TRIANG_BBOX_TOP") >= 0 && cfg_.get_double("TRIANG_BBOX_LEFT") >= 0 && cfg_.get_double("TRIANG_BBOX_BOTTOM") >= 0 &&
                cfg_.get_double("TRIANG_BBOX_RIGHT") >= 0) {
                tp.bbox[0] = cfg_.get_double("TRIANG_BBOX_LEFT"); tp.bbox[1] = cfg_.get_double("TRIANG_BBOX_TOP");
                tp.bbox[2] = cfg_.get_double("TRIANG_BBOX_RIGHT"); tp.bbox[3] = cfg_.get_double("TRIANG_BBOX_BOTTOM");
            }
            tp.cam_distance = env.cam_distance;
            job.log += job.mask_log;                        // "Loading ... as left camera mask" (:1062,1079), where triangulate() logs it
            WLOGI << "triangulating disparity map";
            check(wass_triangulate_dev(ctx_, d_dispf_, W_, H_, rl, rr, &g, in_[k].d_r, W_, H_, mask_l ? in_[k].d_ml : nullptr,
                                       mask_r ? in_[k].d_mr : nullptr, &tp, &mesh, nullptr), "wass_triangulate");
            WLOGI << "... 100%";
            lap(4);
            // ---- the frame before last: TWO frames stay pending (the library's limit), so the record read here belongs to a tail that
            // ended a whole frame ago -- this thread does not wait, and the next frame's SGM stage is in the queue before the current
            // one has finished.  (Reading the previous frame's record here instead -- one frame pending -- put tail + downloads + this
            // thread's enqueue work, 8.8 ms, on the path between two SGM stages of 7.6 ms: 114 instead of 125 frames/s.)
            while ((int)pend_.size() >= std::max(1, std::min(2, max_pending_))) if (FrameJob* p = collect()) done.push_back(p);
            lap(5);
            // ---- the mesh tail (:2046-2123), decided on the device
            const int rounds = cfg_.get_int("PLANE_RANSAC_ROUNDS");
            if (uv_.empty() || uv_seed_ != job.ransac_seed || uv_w_ != rr[2] || uv_h_ != rr[3]) {
                uv_.assign((size_t)rounds * 6, 0);
                if (wass_ransac_sample_seeded(job.ransac_seed, rr[2], rr[3], rounds, uv_.data()) != WASS_OK) throw std::runtime_error("invalid PLANE_RANSAC_ROUNDS / mesh size");
                uv_seed_ = job.ransac_seed; uv_w_ = rr[2]; uv_h_ = rr[3];
            }
            const int slot = job.out_slot;
            // plane_refinement_inliers.xyz comes back as TEXT, formatted on the device (csrc/fmt_g6.h); the points come along for the
            // one case in which the host still has to format them (a number outside the device formatter's domain)
            check(wass_mesh_finish_frame_async_ex2(ctx_, mesh, cfg_.get_double("ZGAP_PERCENTILE"), uv_.data(), rounds, cfg_.get_double("PLANE_RANSAC_THRESHOLD"),
                                                   &rp_, cfg_.get_double("PLANE_MAX_DISTANCE"), out_[slot].xyzc, out_[slot].xyzc_cap,
                                                   opt_.inliers_file && !device_text_ ? out_[slot].inl : nullptr, opt_.inliers_file ? out_[slot].inl_cap : 0, 10,
                                                   opt_.debug_pictures ? out_[slot].ccmask : nullptr,
                                                   opt_.inliers_file && device_text_ ? out_[slot].inl_text : nullptr, opt_.inliers_file && device_text_ ? out_[slot].inl_text_cap : 0),
                  "wass_mesh_finish_frame_async");
            if (dbg_dev_) {
                // the eight pictures, rendered from the maps in HBM and coded behind the tail; the mesh's rejection codes are read in stream order
                wass_debug_desc dd{};
                dd.W0 = W_; dd.H0 = H_;
                for (int i = 0; i < 4; ++i) { dd.roi_l[i] = rl[i]; dd.roi_r[i] = rr[i]; }
                dd.d_left_crop = in_[k].d_cl; dd.d_right_crop = in_[k].d_cr;
                dd.d_disp16 = d_disp16_[job.disp_slot]; dd.d_dispf = d_dispf_;
                dd.num_disp = sp_.num_disp; dd.min_disp = sp_.min_disp; dd.disp_offset = sp_.disp_offset;
                dd.disparity_compensation = env.disparity_compensation;
                dd.quality = 95;
                ensure_dbg(out_[slot], dd);
                check(wass_debug_pictures_async(ctx_, mesh, &dd, out_[slot].dbg, out_[slot].dbg_cap, &job.dbg_ticket), "wass_debug_pictures_async");
            }
            if (dbg_host_) job.mesh = mesh;                 // its rejection codes are fetched when the frame is collected
            else wass_mesh_destroy(mesh);                   // back to the context's pool; the kernels enqueued on it run in stream order
            mesh = nullptr;
            lap(6);
            pend_.push_back(&job);
            ++nsub_;
            if (const char* dd = getenv("WASS_PIPE_DUMP")) {          // debugging aid: the frame's intermediate maps, as raw bytes
                auto dump = [&](const char* name, const void* d, size_t nb) {
                    std::vector<char> hbuf(nb);
                    if (wass_download(ctx_, hbuf.data(), d, nb) == WASS_OK) { std::ofstream f(path_join(dd, name).c_str(), std::ios::binary); f.write(hbuf.data(), (std::streamsize)nb); }
                };
                dump("left_crop.bin", in_[k].d_cl, (size_t)rl[2] * rl[3]);
                dump("right_crop.bin", in_[k].d_cr, (size_t)rr[2] * rr[3]);
                dump("disp16.bin", d16, (size_t)cw * ch * 2);
                dump("dispf.bin", d_dispf_, (size_t)cw * ch * 4);
            }
            job.t_submitted = Timer::now();
        } catch (const std::exception& e) {
            if (mesh) wass_mesh_destroy(mesh);
            WLOG_SCOPE("wass_stereo");
            WLOGE << e.what();
            job.rc = -1;
            if (job.out_slot >= 0) { release_out(job.out_slot); job.out_slot = -1; }
            while (FrameJob* p = collect()) done.push_back(p);
            done.push_back(&job);
        }
    }

    // the last submitted frame (waits for it)
    void flush(std::vector<FrameJob*>& done)
    {
        done.insert(done.end(), early_.begin(), early_.end());
        early_.clear();
        while (FrameJob* p = collect()) done.push_back(p);
    }

    // ---- phase 3 (any thread): everything that is written from the result record (:1374, 1993, 2046-2139)
    // A frame that was enqueued and will never be finished (computed ahead of a request that did not come, or whose inputs changed):
    // gives its output set back.  Only for frames that have been collected.
    void abandon(FrameJob& job)                 // (any thread; never a frame that is staged and not yet submitted)
    {
        if (job.out_slot >= 0) { release_out(job.out_slot); job.out_slot = -1; }
        if (job.mesh) { wass_mesh_destroy(job.mesh); job.mesh = nullptr; }
    }

    void finish(FrameJob& job)
    {
        DebugFormatScope fmt(opt_.debug_png);       // the host renderers' file format: the caller's, not this process's
        LogSinkScope sink(&job.log);
        WLOG_SCOPE("wass_stereo");
        Env& env = job.env;
        const double t0 = Timer::now();
        if (!job.deferred.files.empty()) { if (!job.deferred.write_all()) WLOGE << "Unable to write the files of the frame's preparation"; job.deferred.files.clear(); }
        if (job.rc == 0 && !job.skipped) {
            try {
                const wass_frame_result& r = job.res;
                static const char* const kPic[WASS_DEBUG_PICTURES] = { "stereo.jpg", "stereo_input.jpg", "disparity_stereo_ouput.jpg", "disparity_final_scaled.jpg",
                                                                       "disparity_coverage.jpg", "undistorted/R0.jpg", "undistorted/R1.jpg", "graph_components.jpg" };
                auto write_pic = [&](int k) {
                    const OutSet& o = out_[job.out_slot];
                    size_t off = 0;
                    for (int i = 0; i < k; ++i) off += o.dbg_cap[i];
                    if (job.dbg_bytes[k] == 0) { WLOGE << kPic[k] << ": the coded picture did not fit into its buffer, not written"; return; }
                    if (!write_whole_file(path_join(env.workdir, kPic[k]), o.dbg + off, job.dbg_bytes[k])) WLOGE << "unable to write " << kPic[k];
                };
                if (job.dbg_ticket) for (int k = 0; k < WASS_PIC_COMPONENTS; ++k) write_pic(k);      // :1910-1925, 833-1017, 1381-1382
                if (job.have_dbg) {                          // (the host's renderers: PNG form, component-option masks)
                    debug_stereo_picture(env);
                    debug_dense_pictures(env, sp_, cfg_.get_int("DENSE_DISPARITY_BIGGEST_COMPONENT_THRESHOLD"), job.dbg);
                    debug_triangulation_pictures(env, env.disparity_compensation, sp_.dense_scale, job.dbg);
                }
                WLOG_SCOPE("triangulate");
                WLOGI << r.n_triangulated << " valid points found";
                job.summary.n_points = r.n_triangulated;
                marker(job, 60);
                WLOG_SCOPE("wass_stereo");
                if (r.sgm_cost_overflow == 1) WLOGE << "matching costs exceeded the int16 range; the disparity is outside the reference's defined behaviour";
                if ((long long)r.n_triangulated < cfg_.get_int("MIN_TRIANGULATED_POINTS")) { WLOGE << "Too few points triangulated, aborting"; throw GpuError("too few points"); }
                if (job.dbg_ticket) write_pic(WASS_PIC_COMPONENTS);
                if (job.have_dbg) debug_components_picture(env, job.dbg);
                WLOG_SCOPE("cluster");
                WLOGI << "biggest component size: " << r.component_size << " (px)";
                marker(job, 80);
                WLOG_SCOPE("wass_stereo");
                WLOGI << "estimating best fitting plane...";
                const int rounds = cfg_.get_int("PLANE_RANSAC_ROUNDS");
                WLOG_SCOPE("ransac_find_plane");
                WLOGI << rounds << " ransac rounds, " << r.ransac_inliers << " best inliers";
                WLOGI << "ransac plane coeffs: " << r.ransac_plane[0] << " " << r.ransac_plane[1] << " " << r.ransac_plane[2] << " " << r.ransac_plane[3];
                WLOG_SCOPE("wass_stereo");
                bool have_plane = false;
                if (r.found) {
                    marker(job, 90);
                    WLOGI << "refining plane";
                    if (!r.refine_ok) {
                        char msg[128];
                        snprintf(msg, sizeof msg, "wass_mesh_refine_plane: plane refinement has %g inliers", (double)r.refine_inliers);
                        throw GpuError(msg);
                    }
                    WLOG_SCOPE("refine_plane");
                    WLOGI << "refinement inliers (after cropping): " << r.refine_inliers;
                    WLOGI << "estimated plane coeffs: " << r.plane[0] << " " << r.plane[1] << " " << r.plane[2] << " " << r.plane[3];
                    WLOG_SCOPE("wass_stereo");
                    if (opt_.inliers_file) {
                        const std::string ip = path_join(env.workdir, "plane_refinement_inliers.xyz");
                        if (device_text_ && r.inliers_text_unsupported == 0 && (r.inliers_text_bytes > 0 || r.n_inliers_out == 0))
                            (void)write_whole_file(ip, out_[job.out_slot].inl_text, (size_t)r.inliers_text_bytes);
                        else write_inliers_xyz(ip, out_[job.out_slot].inl, (size_t)r.n_inliers_out);      // the host's formatter (hostio.hpp fmt_g6)
                    }
                    WLOG_SCOPE("crop_plane");
                    WLOGI << "number of points after plane cropping: " << r.kept_final;
                    WLOG_SCOPE("wass_stereo");
                    std::ofstream ofs(path_join(env.workdir, "plane.txt").c_str());
                    ofs << std::setprecision(20);
                    for (int i = 0; i < 4; ++i) ofs << r.plane[i] << std::endl;
                    have_plane = true;
                    job.summary.have_plane = 1;
                    for (int i = 0; i < 4; ++i) job.summary.plane[i] = r.plane[i];
                } else {
                    WLOGE << "ransac failed. I'll continue anyway but plane data won't be available!";
                    std::ofstream ofs(path_join(env.workdir, "plane.txt").c_str());
                    ofs << "nan nan nan nan" << std::endl;
                }
                (void)have_plane;
                WLOGI << "Exporting point cloud data";
                {
                    WLOG_SCOPE("save_as_xyz_compressed");
                    WLOGI << "saving mesh as compressed xyz file...";
                    const std::string xyzc_path = path_join(env.workdir, "mesh_cam.xyzC"), tmp = xyzc_path + ".tmp";
                    bool ok = write_whole_file(tmp, out_[job.out_slot].xyzc, (size_t)r.xyzc_bytes);
                    ok = ok && rename(tmp.c_str(), xyzc_path.c_str()) == 0;
                    if (!ok) { WLOGE << "unable to save mesh data"; throw GpuError("write failed"); }
                    WLOGI << "total data size: " << ((double)r.xyzc_bytes / 1E6) << " MB";
                    WLOG_SCOPE("wass_stereo");
                }
                if (job.prev_w > 0 && job.prev_h > 0) {
                    // the scaled previews of load_data (:413-418), resized on the device: prev[0] = the picture that ended up LEFT, prev[1] = right
                    const OutSet& o = out_[job.out_slot];
                    const size_t pn = (size_t)job.prev_w * job.prev_h;
                    Image cam[2] = { Image(job.prev_w, job.prev_h), Image(job.prev_w, job.prev_h) };
                    memcpy(cam[env.left_index].px.data(), o.prev[0], pn);
                    memcpy(cam[env.right_index].px.data(), o.prev[1], pn);
                    write_png_gray(path_join(env.workdir, "00000000_s.png"), cam[0], prepared_png_level());
                    write_png_gray(path_join(env.workdir, "00000001_s.png"), cam[1], prepared_png_level());
                }
                if (job.raw && opt_.save_undistorted) {
                    // wass_prepare's own output, on request: the undistorted pictures came back with the result (und[0] = LEFT, und[1] = right)
                    const OutSet& o = out_[job.out_slot];
                    const size_t n = (size_t)job.img_w * job.img_h;
                    Image cam[2] = { Image(job.img_w, job.img_h), Image(job.img_w, job.img_h) };
                    memcpy(cam[env.left_index].px.data(), o.und[0], n);
                    memcpy(cam[env.right_index].px.data(), o.und[1], n);
                    write_png_gray(path_join(path_join(env.workdir, "undistorted"), "00000000.png"), cam[0], prepared_png_level());
                    write_png_gray(path_join(path_join(env.workdir, "undistorted"), "00000001.png"), cam[1], prepared_png_level());
                }
                marker(job, 100);
                // the time table (render.hpp:175-191) with the reference's rows.  A pipelined frame has no per-stage WALL times (its
                // GPU stages run underneath its neighbours'): the two host rows are wall times, the others GPU times from hipEvents
                // (wass_sgm_timings.total_ms, wass_frame_result.stage_ms); the line after the table says what the frame took end to end.
                Timer t;
                double at = 0;
                t.events.emplace_back(at += job.t_loaded - job.t_prepare0, "Data load");
                t.events.emplace_back(at += (job.t_planned - job.t_loaded) + (job.t_submitted - job.t_submit0), "Rectification");
                t.events.emplace_back(at += (job.have_sgm ? job.sgm.total_ms : 0.0f) / 1e3, "Dense Stereo");
                static const char* rows[5] = { "Triangulation", "Z-gap stats", "Outlier removal", "Plane fitting", "Plane refinement" };
                for (int k = 0; k < 5; ++k)
                    if (k < 3 || r.found) t.events.emplace_back(at += r.stage_ms[k] / 1e3, rows[k]);
                t.t0 = 0; t.tend = at;
                show_time_stats(t);
                WLOGI << "pipelined chain: GPU stage times above; submission to result " << (job.t_result - job.t_submit0) << " s, output " << (Timer::now() - t0) << " s";
                WLOGI << "All done.";
            } catch (const GpuError& e) {
                WLOG_SCOPE("wass_stereo");
                WLOGE << e.what();
                job.rc = -1;
            } catch (const std::exception& e) {
                WLOG_SCOPE("wass_stereo");
                WLOGE << e.what();
                job.rc = -1;
            }
        }
        if (job.out_slot >= 0) { release_out(job.out_slot); job.out_slot = -1; }
        if (!job.skipped) {
            flush_live(job);
            std::ofstream lf(path_join(job.workdir, "wass_stereo_log.txt").c_str(), std::ios::binary);
            write_log(lf, job.log);
        }
    }

    // log text with the progress markers (lines starting with \x01) removed / kept
    static void write_log(std::ostream& os, const std::string& log)
    {
        size_t p = 0;
        while (p < log.size()) {
            size_t e = log.find('\n', p);
            if (e == std::string::npos) e = log.size(); else ++e;
            if (log[p] != '\x01') os.write(log.data() + p, (std::streamsize)(e - p));
            p = e;
        }
    }
    // single-frame executable: what the phases have logged since the last call goes to stdout, markers included
    void flush_live(FrameJob& job)
    {
        if (!opt_.live || !opt_.echo) return;
        size_t p = live_pos_;
        const std::string& log = job.log;
        while (p < log.size()) {
            size_t e = log.find('\n', p);
            if (e == std::string::npos) e = log.size(); else ++e;
            if (log[p] == '\x01') std::cout.write(log.data() + p + 1, (std::streamsize)(e - p - 1));
            else std::cout.write(log.data() + p, (std::streamsize)(e - p));
            p = e;
        }
        std::cout.flush();
        live_pos_ = log.size();
    }

    int frames_submitted() const { return nsub_; }
    // how many frames may be pending when the next tail is enqueued (Options::max_pending), changed between submissions: the resident
    // worker goes two deep only while callers queue up behind the GPU
    void set_max_pending(int n) { max_pending_ = n; }
    bool host_debug_pictures() const { return dbg_host_; }          // the host renderers are in use: every frame is collected with its maps
    bool pending() const { return !pend_.empty() || !early_.empty(); }   // a submitted frame has not been handed out yet (flush() returns it)

private:
    struct InSet { uint8_t *h_l = nullptr, *h_r = nullptr, *d_l = nullptr, *d_r = nullptr, *d_cl = nullptr, *d_cr = nullptr, *d_ml = nullptr, *d_mr = nullptr,
                           *d_rawl = nullptr, *d_rawr = nullptr, *d_tmp = nullptr, *h_fl = nullptr, *h_fr = nullptr, *d_fl = nullptr, *d_fr = nullptr,
                           *d_pl = nullptr, *d_pr = nullptr; size_t prev_cap = 0; };
    struct OutSet { void* xyzc = nullptr; size_t xyzc_cap = 0; double* inl = nullptr; size_t inl_cap = 0; uint8_t* und[2] = { nullptr, nullptr }; size_t und_cap = 0;
                    uint8_t* ccmask = nullptr; size_t cc_cap = 0; char* inl_text = nullptr; size_t inl_text_cap = 0;
                    uint8_t* prev[2] = { nullptr, nullptr }; size_t prev_cap[2] = { 0, 0 };
                    uint8_t* dbg = nullptr; size_t dbg_total = 0; size_t dbg_cap[WASS_DEBUG_PICTURES] = {}; };
    // pinned room for a frame's eight coded pictures: 1 byte per pixel for grey ones, 1.5 for colour ones (quality 95 needs 0.3-0.6 on
    // pictures of the sea; a picture that does not fit is not written and the log says so)
    void ensure_dbg(OutSet& o, const wass_debug_desc& dd)
    {
        size_t cap[WASS_DEBUG_PICTURES], total = 0;
        for (int k = 0; k < WASS_DEBUG_PICTURES; ++k) {
            int w = 0, h = 0, c = 0;
            check(wass_debug_picture_size(&dd, k, &w, &h, &c), "wass_debug_picture_size");
            cap[k] = ((size_t)w * h * (c == 3 ? 3 : 2) / 2 + 65536 + 63) & ~(size_t)63;
            total += cap[k];
        }
        if (o.dbg && o.dbg_total >= total && !memcmp(cap, o.dbg_cap, sizeof cap)) return;
        if (o.dbg) wass_pinned_free(ctx_, o.dbg);
        o.dbg = nullptr; o.dbg_total = 0;
        void* p = nullptr;
        check(wass_pinned_alloc(ctx_, total, &p), "wass_pinned_alloc");
        o.dbg = (uint8_t*)p; o.dbg_total = total;
        memcpy(o.dbg_cap, cap, sizeof cap);
    }

    void check(int rc, const char* what) const { if (rc != WASS_OK) throw GpuError(std::string(what) + ": " + wass_last_error(ctx_)); }
    uint64_t sgm_calls() const { uint64_t n = 0; (void)wass_sgm_call_count(ctx_, &n); return n; }
    void marker(FrameJob& job, int pct) const
    {
        if (!opt_.live) return;
        char b[32];
        snprintf(b, sizeof b, "\x01[P|%d|100]\n", pct);
        job.log += b;
    }

    FrameJob* collect()
    {
        if (pend_.empty()) return nullptr;
        FrameJob* j = pend_.front();                      // the oldest pending frame: the library hands the records out in order
        pend_.pop_front();
        LogSinkScope sink(&j->log);
        WLOG_SCOPE("wass_stereo");
        if (wass_ctx_frame_result(ctx_, &j->res) != WASS_OK) { WLOGE << "wass_ctx_frame_result: " << wass_last_error(ctx_); j->rc = -1; }
        j->t_result = Timer::now();
        // a number the device formatter does not cover (inf, nan, |v| >= 1e6 or < 1e-22): the points come over for the host's formatter --
        // now, before the next frame's tail reuses the device buffer
        if (j->rc == 0 && opt_.inliers_file && device_text_ && j->res.inliers_text_unsupported != 0 && j->out_slot >= 0) {
            uint64_t got = 0;
            if (wass_ctx_frame_inliers(ctx_, out_[j->out_slot].inl, out_[j->out_slot].inl_cap, &got) != WASS_OK) { WLOGE << "wass_ctx_frame_inliers: " << wass_last_error(ctx_); j->rc = -1; }
        }
        // stage times of the frame's own SGM call (the library keeps the last four)
        j->have_sgm = j->sgm_call > 0 && wass_sgm_call_timings(ctx_, (uint64_t)j->sgm_call, &j->sgm) == WASS_OK;
        if (dbg_dev_ && j->dbg_ticket) {
            if (wass_debug_pictures_result(ctx_, j->dbg_ticket, j->dbg_bytes) != WASS_OK) { WLOGE << "wass_debug_pictures_result: " << wass_last_error(ctx_); j->dbg_ticket = 0; }
        }
        if (dbg_host_ && j->mesh) {
            // the maps the debug pictures are drawn from (nothing else has been enqueued since this frame: see submit)
            try {
                Env& env = j->env;
                const int k = j->in_slot, cw = env.roi_r.width, ch = env.roi_r.height;
                DebugMaps& dm = j->dbg;
                env.left_crop = Image(env.roi_l.width, env.roi_l.height); env.right_crop = Image(cw, ch);
                check(wass_download(ctx_, env.left_crop.px.data(), in_[k].d_cl, env.left_crop.px.size()), "wass_download");
                check(wass_download(ctx_, env.right_crop.px.data(), in_[k].d_cr, env.right_crop.px.size()), "wass_download");
                dm.ws = cw; dm.hs = ch;
                dm.disp16.resize((size_t)cw * ch); dm.dispf.resize((size_t)cw * ch); dm.codes.resize((size_t)cw * ch);
                check(wass_download(ctx_, dm.disp16.data(), d_disp16_[j->disp_slot], dm.disp16.size() * 2), "wass_download");
                check(wass_download(ctx_, dm.dispf.data(), d_dispf_, dm.dispf.size() * 4), "wass_download");
                if (cfg_.get_int("DENSE_DISPARITY_BIGGEST_COMPONENT_THRESHOLD") > 0) {
                    dm.large_gradient.resize((size_t)cw * ch);
                    check(wass_large_gradient_mask(ctx_, cw, ch, dm.large_gradient.data()), "wass_large_gradient_mask");
                }
                check(wass_mesh_reject_codes(ctx_, j->mesh, dm.codes.data()), "wass_mesh_reject_codes");
                dm.valid_before.resize(dm.codes.size());
                for (size_t i = 0; i < dm.codes.size(); ++i) dm.valid_before[i] = dm.codes[i] == (WASS_CODE_GREY | (WASS_CODE_GREY << 4));   // triangulated
                dm.valid_after.assign(out_[j->out_slot].ccmask, out_[j->out_slot].ccmask + dm.codes.size());
                j->have_dbg = true;
            } catch (const std::exception& e) { WLOGE << e.what(); }
            wass_mesh_destroy(j->mesh);
            j->mesh = nullptr;
        }
        return j;
    }

    void release_buffers()
    {
        for (auto& s : in_) {
            for (uint8_t** p : { &s.d_l, &s.d_r, &s.d_cl, &s.d_cr, &s.d_ml, &s.d_mr, &s.d_rawl, &s.d_rawr, &s.d_tmp, &s.d_fl, &s.d_fr, &s.d_pl, &s.d_pr }) { if (*p) wass_device_free(ctx_, *p); *p = nullptr; }
            s.prev_cap = 0;
            for (uint8_t** p : { &s.h_l, &s.h_r, &s.h_fl, &s.h_fr }) { if (*p) wass_pinned_free(ctx_, *p); *p = nullptr; }
        }
        for (auto& p : d_disp16_) { if (p) wass_device_free(ctx_, p); p = nullptr; }
        if (d_dispf_) wass_device_free(ctx_, d_dispf_);
        d_dispf_ = nullptr;
        for (auto& p : d_map_) { if (p) wass_device_free(ctx_, p); p = nullptr; }
        map_valid_ = false;
    }
    // (re)allocated when the pictures or the ROIs change size -- once per sequence in practice
    void ensure_buffers(int W, int H, const Rect& roi_l, const Rect& roi_r)
    {
        if (W == W_ && H == H_ && roi_l.width == cwl_ && roi_l.height == chl_ && roi_r.width == cwr_ && roi_r.height == chr_) return;
        check(wass_ctx_synchronize(ctx_), "wass_ctx_synchronize");
        // debug pictures: the pending frame's maps are fetched from these buffers when it is collected -- do that first
        if (dbg_host_) while (FrameJob* p = collect()) early_.push_back(p);
        release_buffers();
        W_ = W; H_ = H; cwl_ = roi_l.width; chl_ = roi_l.height; cwr_ = roi_r.width; chr_ = roi_r.height;
        const size_t n = (size_t)W * H + 4;
        auto dev = [&](size_t bytes) { void* p = nullptr; check(wass_device_alloc(ctx_, bytes, &p), "wass_device_alloc"); return p; };
        auto pin = [&](size_t bytes) { void* p = nullptr; check(wass_pinned_alloc(ctx_, bytes, &p), "wass_pinned_alloc"); return p; };
        for (auto& s : in_) {
            s.h_l = (uint8_t*)pin(n); s.h_r = (uint8_t*)pin(n);
            s.d_l = (uint8_t*)dev(n); s.d_r = (uint8_t*)dev(n); s.d_ml = (uint8_t*)dev(n); s.d_mr = (uint8_t*)dev(n);
            s.d_cl = (uint8_t*)dev((size_t)cwl_ * chl_ + 4); s.d_cr = (uint8_t*)dev((size_t)cwr_ * chr_ + 4);
            if (opt_.prep) { s.d_rawl = (uint8_t*)dev(n); s.d_rawr = (uint8_t*)dev(n); s.d_tmp = (uint8_t*)dev(n); }
        }
        for (auto& p : d_disp16_) p = (int16_t*)dev((size_t)cwr_ * chr_ * 2);
        d_dispf_ = (float*)dev((size_t)cwr_ * chr_ * 4);
    }
    // cv::initUndistortRectifyMap (:600-601): rig constants, computed and uploaded when the calibration changes
    void ensure_maps(const Env& env)
    {
        std::vector<double> key;
        auto add = [&](const double* p, int n) { key.insert(key.end(), p, p + n); };
        add(env.K_left.d.data(), 9); add(env.rec_R1, 9); add(env.rec_P1, 12); add(env.K_right.d.data(), 9); add(env.rec_R2, 9); add(env.rec_P2, 12);
        key.push_back(W_); key.push_back(H_);
        if (map_valid_ && key == map_key_) return;
        check(wass_ctx_synchronize(ctx_), "wass_ctx_synchronize");          // a frame in flight may still read the old maps
        const size_t n = (size_t)W_ * H_;
        void* h = nullptr;
        check(wass_pinned_alloc(ctx_, n * 4 * 4, &h), "wass_pinned_alloc");
        // freed on every way out; a throw inside the upload loop first waits for the copies that may still read it
        struct Staging { wass_ctx* c; void* h; ~Staging() { (void)wass_ctx_synchronize(c); wass_pinned_free(c, h); } } staging{ ctx_, h };
        float* hp = (float*)h;
        for (int cam = 0; cam < 2; ++cam) {
            const int rc = cam == 0 ? wass_init_rectify_map(env.K_left.d.data(), env.rec_R1, env.rec_P1, W_, H_, hp, hp + n)
                                    : wass_init_rectify_map(env.K_right.d.data(), env.rec_R2, env.rec_P2, W_, H_, hp + 2 * n, hp + 3 * n);
            if (rc != WASS_OK) throw std::runtime_error("singular rectification");
        }
        for (int i = 0; i < 4; ++i) {
            if (!d_map_[i]) { void* p = nullptr; check(wass_device_alloc(ctx_, n * 4, &p), "wass_device_alloc"); d_map_[i] = (float*)p; }
            check(wass_upload_async(ctx_, d_map_[i], hp + (size_t)i * n, n * 4), "wass_upload_async");
        }
        check(wass_ctx_synchronize(ctx_), "wass_ctx_synchronize");          // the staging buffer goes away (Staging)
        map_key_ = key;
        map_valid_ = true;
    }

    int acquire_out(size_t npts, size_t und_bytes, bool ccmask)
    {
        std::unique_lock<std::mutex> lk(out_mu_);
        out_cv_.wait(lk, [&]() { for (bool f : out_free_) if (f) return true; return false; });
        int slot = 0;
        while (!out_free_[(size_t)slot]) ++slot;
        out_free_[(size_t)slot] = false;
        lk.unlock();
        struct Guard { FramePipeline* p; int slot; bool armed; ~Guard() { if (armed) p->release_out(slot); } } guard{ this, slot, true };   // an allocation below may throw
        OutSet& o = out_[(size_t)slot];
        const size_t need = 148 + 6 * npts, icap = (npts + 9) / 10;
        if (o.xyzc_cap < need) {
            if (o.xyzc) wass_pinned_free(ctx_, o.xyzc);
            o.xyzc = nullptr; o.xyzc_cap = 0;
            check(wass_pinned_alloc(ctx_, need, &o.xyzc), "wass_pinned_alloc");
            o.xyzc_cap = need;
        }
        if (opt_.inliers_file && o.inl_cap < icap) {
            if (o.inl) wass_pinned_free(ctx_, o.inl);
            o.inl = nullptr; o.inl_cap = 0;
            void* p = nullptr;
            check(wass_pinned_alloc(ctx_, icap * 24, &p), "wass_pinned_alloc");
            o.inl = (double*)p; o.inl_cap = icap;
        }
        if (opt_.inliers_file && device_text_ && o.inl_text_cap < icap * 40) {
            if (o.inl_text) wass_pinned_free(ctx_, o.inl_text);
            o.inl_text = nullptr; o.inl_text_cap = 0;
            void* p = nullptr;
            check(wass_pinned_alloc(ctx_, icap * 40, &p), "wass_pinned_alloc");
            o.inl_text = (char*)p; o.inl_text_cap = icap * 40;
        }
        if (und_bytes > o.und_cap) {
            for (auto& u : o.und) {
                if (u) wass_pinned_free(ctx_, u);
                u = nullptr;
                void* p = nullptr;
                check(wass_pinned_alloc(ctx_, und_bytes, &p), "wass_pinned_alloc");
                u = (uint8_t*)p;
            }
            o.und_cap = und_bytes;
        }
        if (ccmask && o.cc_cap < npts) {
            if (o.ccmask) wass_pinned_free(ctx_, o.ccmask);
            o.ccmask = nullptr; o.cc_cap = 0;
            void* p = nullptr;
            check(wass_pinned_alloc(ctx_, npts, &p), "wass_pinned_alloc");
            o.ccmask = (uint8_t*)p; o.cc_cap = npts;
        }
        guard.armed = false;
        return slot;
    }
    void release_out(int slot)
    {
        { std::lock_guard<std::mutex> lk(out_mu_); out_free_[(size_t)slot] = true; }
        out_cv_.notify_one();
    }

    int device_ = 0;
    wass_ctx* ctx_ = nullptr;
    bool ctx_failed_ = false;
    const Config& cfg_;
    std::string config_path_;
    Options opt_;
    int max_pending_;
    wass_sgm_params sp_{};
    wass_refine_params rp_{};
    int W_ = 0, H_ = 0, cwl_ = 0, chl_ = 0, cwr_ = 0, chr_ = 0;
    InSet in_[NIN];
    int next_in_ = 0;
    int16_t* d_disp16_[2] = { nullptr, nullptr };
    float* d_dispf_ = nullptr;
    float* d_map_[4] = { nullptr, nullptr, nullptr, nullptr };
    std::vector<double> map_key_;
    bool map_valid_ = false;
    std::vector<int32_t> uv_;
    unsigned int uv_seed_ = 0;
    int uv_w_ = 0, uv_h_ = 0;
    std::vector<OutSet> out_;
    std::vector<bool> out_free_;
    std::mutex out_mu_;
    std::condition_variable out_cv_;
    std::deque<FrameJob*> pend_;     // submitted, record not read yet: at most two
    int nsub_ = 0;
    std::deque<FrameJob*> staged_;   // staged (pictures in the input ring), not yet submitted: oldest first
    std::vector<FrameJob*> early_;   // frames collected before their turn (ensure_buffers); handed out by the next submit / flush
    size_t live_pos_ = 0;
    bool timing_ = getenv("WASS_PIPE_TIMING") && atoi(getenv("WASS_PIPE_TIMING")) != 0;
    double lap_[7] = {};
    // WASS_HOST_INLIER_TEXT=1: the round-4 form (the host formats the inlier file from the downloaded points); same bytes either way
    bool device_text_ = !(getenv("WASS_HOST_INLIER_TEXT") && atoi(getenv("WASS_HOST_INLIER_TEXT")) != 0);
    // The debug pictures: drawn AND JPEG-coded on the device behind the frame's tail (wass_debug_pictures_async, csrc/jpeg.hip) -- the files'
    // bytes are all that comes back, the chain keeps its depth.  The host's renderers (wass_frame.hpp) remain for the lossless PNG form the
    // tests read pixels from (WASS_DEBUG_FORMAT=png), for the two extra masks of the component option and on request
    // (WASS_HOST_DEBUG_PICTURES=1): those fetch every intermediate map when a frame is collected, one frame at a time.
    bool dbg_dev_ = false, dbg_host_ = false;
};

}  // namespace wassframe
