// wass_stereo_batch.cpp -- sequence driver: the MI355X replacement for wasscli's process fan-out
// (/root/reference/cli/wasscli/wasscli.py:305-364).
//
//   wass_stereo_batch <config_file> <workdir_0> <workdir_1> ... [--gpus G] [--procs-per-gpu P] [--out <dir>] [--verbose]
//                     [--skip-existing]
//   wass_stereo_batch <config_file> --sequence <output_dir> [...]      (every <output_dir>/NNNNNN_wd, in order)
//
// The reference starts one wass_stereo process per frame, NUM_PARALLEL_PROCESSES at a time, and appends each
// successful frame's plane.txt to output/planes.txt (:341-343); the gridding stage later takes np.nanmean of that file
// (gridding/wassgridsurface/wassgridsurface.py:672-678).  Here: one worker PROCESS per GPU (frames are independent, so
// frame i goes to worker i mod G and no image data ever crosses GPUs), each with ONE persistent libwassgpu context
// -- HIP start-up and the 5-8 GB of scratch HBM are paid once per worker, not once per frame -- running exactly the
// per-frame code of the drop-in wass_stereo (wass_frame.hpp), so every workdir receives the same files.  The only
// exchange is Coll-1: [sum a, sum b, sum c, sum d, n_valid] over the workers' planes, all-reduced over RCCL/xGMI when
// every worker owns a GPU of its own (wass_coll_*), and the per-frame records that the parent collects through pipes to
// write planes.txt in FRAME order (wasscli writes it in completion order) and planes_mean.txt.
// A worker runs its frames through the PIPELINED chain of frame_pipeline.hpp: decode threads (calibration, PNG inflation,
// previews, the rectification's decisions) -> one submitting thread (uploads one frame ahead, every GPU stage enqueued without
// a host synchronisation, one result record per frame read one frame late) -> writer threads (the log lines that carry
// numbers, plane.txt, plane_refinement_inliers.xyz, mesh_cam.xyzC).  Configurations that need an intermediate map or mesh on
// the host (pipeline_eligible), --threads-per-proc > 1 and --stage-by-stage use the synchronous stage-by-stage calls of
// wass_run_frame instead; the files are the same either way.  --debug-images (the reference's eight pictures per frame) stays
// in the pipelined chain: they are rendered and JPEG-coded on the device behind the frame's tail (csrc/jpeg.hip).
// --skip-existing: the workdir is the checkpoint: a frame whose plane.txt and mesh_cam.xyzC /
// mesh_cam.xyzbin exist is not recomputed, its plane is read back from plane.txt.
#include <dirent.h>
#include <fcntl.h>
#include <sys/resource.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <map>
#include <poll.h>
#include <signal.h>
#include <unistd.h>

#include <algorithm>

#include <mutex>
#include <thread>

#include "frame_pipeline.hpp"

using namespace wassframe;

namespace {

struct Record {            // worker -> parent, one per frame
    int index, rc, have_plane;
    double plane[4];
    double seconds;
    unsigned long long n_points;
};
struct Tail {              // worker -> parent, once: what the worker's all-reduce returned
    int magic, used_rccl, n_valid;
    double mean[4];
    // when the worker's first and last computed frames were complete (files written), and how many it computed: the rate
    // between the two is the worker's steady state, without its start-up (HIP initialisation, first allocations)
    double first_done, last_done;
    int computed;
};

bool write_all(int fd, const void* p, size_t n)
{
    const char* c = (const char*)p;
    while (n) { const ssize_t k = write(fd, c, n); if (k <= 0) return false; c += k; n -= (size_t)k; }
    return true;
}
bool read_all(int fd, void* p, size_t n)
{
    char* c = (char*)p;
    while (n) { const ssize_t k = read(fd, c, n); if (k <= 0) return false; c += k; n -= (size_t)k; }
    return true;
}
double now() { timeval tv; gettimeofday(&tv, nullptr); return (double)tv.tv_sec + (double)tv.tv_usec / 1e6; }

// plane.txt of a finished frame: four lines, or the single line "nan nan nan nan" (wass_stereo.cpp:2092-2107)
bool read_plane_txt(const std::string& wd, FrameSummary& fs)
{
    std::ifstream f(path_join(wd, "plane.txt").c_str());
    if (!f.is_open()) return false;
    std::string tok[4];
    for (auto& t : tok) if (!(f >> t)) return false;
    fs.have_plane = 1;
    for (int k = 0; k < 4; ++k) {
        if (tok[k] == "nan" || tok[k] == "-nan") { fs.have_plane = 0; break; }
        fs.plane[k] = atof(tok[k].c_str());
    }
    return true;
}

// Coll-1's unique id.  It is taken by WORKER 0, not by the parent: ncclGetUniqueId initialises the RCCL / HIP runtimes of the calling
// process, and a process that has done that must not fork workers which then use the GPU (round 6: the first torch-less worker forked
// behind the parent's ncclGetUniqueId died with SIGSEGV in its first HIP call -- on a 1-GPU box, with --rccl-always; every N-GPU run
// would have).  Worker 0 sends [ok, id] up its own pipe right after it starts; the parent relays it down one pipe per other worker;
// workers read it when they get to the all-reduce.  ok = 0 (no loadable librccl): nobody calls RCCL, the parent reduces the planes.
struct CollSetup {
    bool use = false;
    int up_fd = -1;        // worker 0 -> parent
    int down_fd = -1;      // parent -> this worker (ranks >= 1)
    unsigned char uid[128] = {};
    bool ok = false;
    void at_start(int rank)
    {
        if (!use || rank != 0) return;
        ok = wass_coll_unique_id(uid) == WASS_OK;
        unsigned char msg[129];
        msg[0] = ok ? 1 : 0;
        memcpy(msg + 1, uid, 128);
        (void)write_all(up_fd, msg, sizeof msg);
    }
    bool before_allreduce(int rank)       // true: go through RCCL
    {
        if (!use) return false;
        if (rank != 0) {
            unsigned char msg[129] = {};
            ok = read_all(down_fd, msg, sizeof msg) && msg[0] == 1;
            memcpy(uid, msg + 1, 128);
        }
        return ok;
    }
};

int worker(int rank, int world, int device, CollSetup coll, const char* cfg,
           const std::vector<std::string>& wds, bool verbose, bool skip_existing, bool debug_images, int fd, int threads)
{
    coll.at_start(rank);
    if (!verbose) {                       // per-frame logs still go to <workdir>/wass_stereo_log.txt
        const int nul = open("/dev/null", O_WRONLY);
        if (nul >= 0) { dup2(nul, 1); close(nul); }
    }
    wass_ctx* ctx = nullptr;                                     // thread 0's context: also carries the RCCL collective below
    double acc[5] = { 0, 0, 0, 0, 0 };
    std::mutex mu;                                               // the pipe and the running plane sum
    int status = 0;
    // Frames rank, rank + world, ... of this worker are dealt round-robin to `threads` host threads, each with a context of
    // its own (own streams and buffers): kernels of different frames then overlap on the GPU inside ONE process, which
    // processes sharing a GPU do not do nearly as well.
    auto run = [&](int tid) {
        wass_ctx* my = nullptr;
        // the next frame's PNGs are inflated on a host thread while this frame is being processed
        Preload pre[2];
        AsyncWriter writer;                                      // a frame's mesh_cam.xyzC is written while the next frame runs
        std::thread loader;
        struct Joiner { std::thread& t; ~Joiner() { if (t.joinable()) t.join(); } } joiner{ loader };     // also on the early returns
        const size_t stride = (size_t)world * (size_t)threads, first = (size_t)rank + (size_t)tid * (size_t)world;
        // finished = the point cloud file is there and at least as long as its header (148 bytes, PovMesh.cpp:377-460; the file
        // only gets its name once it is complete) -- plane.txt is checked by the caller
        auto done = [&](size_t k) {
            if (!skip_existing) return false;
            struct stat sb;
            const std::string a = path_join(wds[k], "mesh_cam.xyzC");
            if (stat(a.c_str(), &sb) == 0) return sb.st_size >= 148;
            return exists(path_join(wds[k], "mesh_cam.xyzbin"));
        };
        auto start_load = [&](size_t k, int slot) {
            if (k < wds.size() && exists(wds[k]) && !done(k)) loader = std::thread([&, k, slot]() { preload_images(wds[k], pre[slot]); });
        };
        start_load(first, 0);
        int slot = 0;
        for (size_t i = first; i < wds.size(); i += stride, slot ^= 1) {
            Record r = {};
            r.index = (int)i;
            const double t0 = now();
            FrameSummary fs;
            if (loader.joinable()) loader.join();                 // this frame's pictures (slot) are in memory now
            start_load(i + stride, slot ^ 1);
            if (done(i) && read_plane_txt(wds[i], fs))
                r.rc = 0;
            else
                r.rc = exists(wds[i]) ? wass_run_frame(cfg, wds[i], nullptr, device, &my, &fs, debug_images, &pre[slot], &writer) : -1;
            r.seconds = now() - t0;
            r.have_plane = r.rc == 0 && fs.have_plane;
            r.n_points = fs.n_points;
            for (int k = 0; k < 4; ++k) r.plane[k] = r.have_plane ? fs.plane[k] : std::nan("");
            std::lock_guard<std::mutex> lk(mu);
            if (r.rc == 0) wass_planes_mean_accumulate(r.plane, 1, acc);     // NaN planes are skipped (nanmean)
            if (!write_all(fd, &r, sizeof r)) { status = 2; break; }
        }
        if (loader.joinable()) loader.join();
        writer.wait();
        std::lock_guard<std::mutex> lk(mu);
        if (writer.failed && !status) status = 4;
        if (tid == 0) ctx = my; else if (my) wass_ctx_destroy(my);
    };
    {
        std::vector<std::thread> pool;
        for (int t = 1; t < threads; ++t) pool.emplace_back(run, t);
        run(0);
        for (auto& th : pool) th.join();
    }
    if (status) return status;
    Tail t = {};
    t.magic = 0x57415353;
    if (coll.before_allreduce(rank)) {
        // Coll-1 over RCCL: needs a context (a worker with no frames creates one just for the collective)
        if (!ctx && wass_ctx_create(device, &ctx) != WASS_OK) return 3;
        if (wass_coll_init(ctx, rank, world, coll.uid) != WASS_OK || wass_coll_allreduce_sum_f64(ctx, acc, 5) != WASS_OK) {
            std::cerr << "worker " << rank << ": RCCL all-reduce failed: " << wass_last_error(ctx) << std::endl;
            return 4;
        }
        t.used_rccl = 1;
        wass_planes_mean_finish(acc, t.mean, &t.n_valid);
    }
    if (!write_all(fd, &t, sizeof t)) return 2;
    if (ctx) wass_ctx_destroy(ctx);
    return 0;
}

// finished = the point cloud file is there and at least as long as its header (148 bytes, PovMesh.cpp:377-460; the file only
// gets its name once it is complete) -- plane.txt is checked by the caller
bool frame_done(const std::string& wd)
{
    struct stat sb;
    const std::string a = path_join(wd, "mesh_cam.xyzC");
    if (stat(a.c_str(), &sb) == 0) return sb.st_size >= 148;
    return exists(path_join(wd, "mesh_cam.xyzbin"));
}

struct PipeOptions {
    int decode_threads = 8, writer_threads = 4;
    bool inliers_file = true;
    bool debug_pictures = false;
    // prepare-less mode (--raw): frame i starts from cam0[i] / cam1[i] and the calibration directory
    const PrepareSetup* prep = nullptr;
    const std::vector<std::string>* cam0 = nullptr;
    const std::vector<std::string>* cam1 = nullptr;
    bool save_undistorted = false;
};

// One worker process, one context, frames rank, rank + world, ... through FramePipeline.
int worker_pipelined(int rank, int world, int device, CollSetup coll, const char* cfgpath, const Config& cfg,
                     const std::vector<std::string>& wds, bool verbose, bool skip_existing, int fd, const PipeOptions& po)
{
    coll.at_start(rank);
    if (!verbose) {
        const int nul = open("/dev/null", O_WRONLY);
        if (nul >= 0) { dup2(nul, 1); close(nul); }
    }
    const char* dev_env = getenv("WASS_GPU_DEVICE");
    std::vector<size_t> mine;
    for (size_t i = (size_t)rank; i < wds.size(); i += (size_t)world) mine.push_back(i);
    const size_t n = mine.size();
    double acc[5] = { 0, 0, 0, 0, 0 };
    int status = 0;
    double first_done = 0, last_done = 0;
    int computed = 0;
    FramePipeline::Options fo;
    fo.out_slots = po.writer_threads + 2;
    fo.inliers_file = po.inliers_file;
    fo.prep = po.prep;
    fo.save_undistorted = po.save_undistorted;
    fo.debug_pictures = po.debug_pictures;
    FramePipeline pl(dev_env ? atoi(dev_env) : device, cfg, cfgpath, fo);
    {
        std::vector<std::unique_ptr<FrameJob>> jobs(n);
        std::vector<double> t_begin(n, 0.0);
        std::mutex mu;                                             // ready flags, the load window, the writer queue
        std::condition_variable cv_ready, cv_window, cv_write;
        std::vector<char> ready(n, 0);
        size_t next_load = 0, submitted = 0;
        const size_t look = (size_t)po.decode_threads + 2;         // decoded frames waiting for the GPU: 10 MB each at 5 megapixels
        std::deque<size_t> wq;                                     // positions whose GPU work is complete
        bool wq_closed = false;
        std::mutex out_mu;                                         // the pipe, the plane sum, stdout

        auto loader = [&]() {
            pthread_setname_np(pthread_self(), "wass-decode");
            for (;;) {
                size_t pos;
                {
                    std::unique_lock<std::mutex> lk(mu);
                    cv_window.wait(lk, [&]() { return next_load >= n || next_load < submitted + look; });
                    if (next_load >= n) return;
                    pos = next_load++;
                }
                std::unique_ptr<FrameJob> j(new FrameJob());
                j->index = mine[pos];
                j->workdir = wds[mine[pos]];
                t_begin[pos] = now();
                if (po.prep) { j->raw = true; j->c0 = (*po.cam0)[mine[pos]]; j->c1 = (*po.cam1)[mine[pos]]; }
                if (!po.prep && !exists(j->workdir)) j->rc = -1;
                else if (skip_existing && frame_done(j->workdir) && read_plane_txt(j->workdir, j->summary)) j->skipped = true;
                else pl.prepare(*j);
                std::lock_guard<std::mutex> lk(mu);
                jobs[pos] = std::move(j);
                ready[pos] = 1;
                cv_ready.notify_all();
            }
        };
        auto writer = [&]() {
            pthread_setname_np(pthread_self(), "wass-write");
            for (;;) {
                size_t pos;
                {
                    std::unique_lock<std::mutex> lk(mu);
                    cv_write.wait(lk, [&]() { return !wq.empty() || wq_closed; });
                    if (wq.empty()) return;
                    pos = wq.front();
                    wq.pop_front();
                }
                FrameJob& j = *jobs[pos];
                pl.finish(j);
                Record r = {};
                r.index = (int)j.index;
                r.rc = j.rc;
                r.seconds = now() - t_begin[pos];
                r.have_plane = j.rc == 0 && j.summary.have_plane;
                r.n_points = j.summary.n_points;
                for (int k = 0; k < 4; ++k) r.plane[k] = r.have_plane ? j.summary.plane[k] : std::nan("");
                {
                    std::lock_guard<std::mutex> lk(out_mu);
                    if (verbose && !j.skipped) { FramePipeline::write_log(std::cout, j.log); std::cout.flush(); }
                    if (r.rc == 0) wass_planes_mean_accumulate(r.plane, 1, acc);
                    if (!write_all(fd, &r, sizeof r)) status = 2;
                    if (!j.skipped && r.rc == 0) { const double t = now(); if (!computed++) first_done = t; last_done = t; }
                }
                std::lock_guard<std::mutex> lk(mu);
                jobs[pos].reset();
            }
        };
        pthread_setname_np(pthread_self(), "wass-hip-rt");       // (threads the HIP / ROCr runtime starts from this one inherit the name ...)
        std::vector<std::thread> loaders, writers;
        for (int t = 0; t < po.decode_threads; ++t) loaders.emplace_back(loader);
        for (int t = 0; t < po.writer_threads; ++t) writers.emplace_back(writer);

        std::vector<FrameJob*> done;
        auto hand_over = [&](std::vector<FrameJob*>& d) {
            if (d.empty()) return;
            std::lock_guard<std::mutex> lk(mu);
            for (FrameJob* j : d) wq.push_back((j->index - (size_t)rank) / (size_t)world);
            cv_write.notify_all();
            d.clear();
        };
        for (size_t pos = 0; pos < n; ++pos) {
            FrameJob *cur, *nxt = nullptr, *nxt2 = nullptr;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_ready.wait(lk, [&]() { return ready[pos] != 0; });
                cur = jobs[pos].get();
                if (pos + 1 < n && ready[pos + 1]) nxt = jobs[pos + 1].get();
                if (nxt && pos + 2 < n && ready[pos + 2]) nxt2 = jobs[pos + 2].get();
            }
            pl.stage(*cur);
            if (pos == 0) pthread_setname_np(pthread_self(), "wass-submit");   // (... this thread takes its own once the runtime is up)
            // uploads run two frames ahead, in order (FramePipeline::NIN): underneath this frame, in front of its downloads in the copy stream
            const bool ahead1 = nxt && nxt->rc == 0 && !nxt->skipped && pl.same_geometry(*nxt);
            if (ahead1) pl.stage(*nxt);
            if (ahead1 && nxt2 && nxt2->rc == 0 && !nxt2->skipped && pl.same_geometry(*nxt2)) pl.stage(*nxt2);
            pl.submit(*cur, done);
            // the decoded pictures have gone to the pinned ring (the debug pictures still want their size)
            if (!po.debug_pictures) { cur->env.left = Image(); cur->env.right = Image(); }
            hand_over(done);
            {
                std::lock_guard<std::mutex> lk(mu);
                submitted = pos + 1;
            }
            cv_window.notify_all();
        }
        pl.flush(done);
        hand_over(done);
        {
            std::lock_guard<std::mutex> lk(mu);
            wq_closed = true;
        }
        cv_write.notify_all();
        cv_window.notify_all();
        if (getenv("WASS_THREAD_CPU")) {
            // diagnostic: CPU seconds per thread of this worker, by thread name (the unnamed ones are the HIP / ROCr runtime's)
            std::map<std::string, std::pair<int, double>> by;
            if (DIR* d = opendir("/proc/self/task")) {
                while (dirent* e = readdir(d)) {
                    if (e->d_name[0] == '.') continue;
                    std::ifstream f(std::string("/proc/self/task/") + e->d_name + "/stat");
                    std::string line;
                    std::getline(f, line);
                    const size_t a = line.find('('), b = line.rfind(')');
                    if (a == std::string::npos || b == std::string::npos) continue;
                    std::istringstream is(line.substr(b + 2));
                    std::string tok;
                    unsigned long long ut = 0, st = 0;
                    for (int k = 3; k <= 15 && (is >> tok); ++k) { if (k == 14) ut = strtoull(tok.c_str(), nullptr, 10); if (k == 15) st = strtoull(tok.c_str(), nullptr, 10); }
                    auto& s = by[line.substr(a + 1, b - a - 1)];
                    s.first++; s.second += (double)(ut + st) / (double)sysconf(_SC_CLK_TCK);
                }
                closedir(d);
            }
            for (const auto& kv : by) fprintf(stderr, "thread CPU: %-16s x%-3d %.3f s\n", kv.first.c_str(), kv.second.first, kv.second.second);
        }
        for (auto& t : loaders) t.join();
        for (auto& t : writers) t.join();
    }
    if (status) return status;
    Tail t = {};
    t.magic = 0x57415353;
    t.first_done = first_done; t.last_done = last_done; t.computed = computed;
    if (coll.before_allreduce(rank)) {
        // Coll-1 over RCCL: needs a context (a worker without frames to compute creates one just for the collective)
        wass_ctx* ctx = pl.context();
        if (!ctx) return 3;
        if (wass_coll_init(ctx, rank, world, coll.uid) != WASS_OK || wass_coll_allreduce_sum_f64(ctx, acc, 5) != WASS_OK) {
            std::cerr << "worker " << rank << ": RCCL all-reduce failed: " << wass_last_error(ctx) << std::endl;
            return 4;
        }
        t.used_rccl = 1;
        wass_planes_mean_finish(acc, t.mean, &t.n_valid);
    }
    return write_all(fd, &t, sizeof t) ? 0 : 2;
}

}  // namespace

int main(int argc, char* argv[])
{
    // six hardware queues for the HIP runtime, while this process is still single-threaded (libwassgpu sets the same default before its first
    // HIP call -- wass_amd/csrc/api.hip default_hw_queues has the story -- but setenv() there would race with the getenv() of other threads)
    (void)setenv("GPU_MAX_HW_QUEUES", "6", 0);
    if (argc < 3) {
        std::cout << "Usage:\n  wass_stereo_batch <config_file> <workdir>... [--gpus G] [--procs-per-gpu P] [--out <dir>] [--verbose] [--skip-existing] [--debug-images]\n"
                     "  wass_stereo_batch <config_file> --sequence <output_dir> [--gpus G] ...\n"
                     "  --decode-threads N / --writer-threads N   host threads of a worker's pipeline (default 8 / 4)\n"
                     "  --raw <calibdir> --cam0 <dir> --cam1 <dir> --sequence <output_dir> [--frames N] [--save-undistorted]\n"
                     "                      prepare-less mode: wass_prepare's undistortion (and CLAHE) runs on the GPU inside the frame chain, from\n"
                     "                      the cameras' raw pictures; undistorted/*.png are only written with --save-undistorted\n"
                     "  --rccl-always       the RCCL all-reduce of the mean plane also with one worker (test of the library search on a 1-GPU box)\n"
                     "  --no-inliers-file   do not write plane_refinement_inliers.xyz (14 MB of text per 5-megapixel frame that nothing reads)\n"
                     "  --stage-by-stage    synchronous per-stage calls instead of the pipelined chain (same files)\n"
                     "  --threads-per-proc T  stage-by-stage only: frames in flight per worker process, each on a thread and a context of its own\n"
                     "  --procs-per-gpu P   worker processes per GPU (default 1; RCCL needs one worker per GPU, with more the parent reduces)\n";
        return argc == 1 ? 0 : -1;
    }
    const char* cfg = argv[1];
    std::vector<std::string> wds;
    std::string outdir;
    int gpus = 1, ppg = 1, tpp = 1;
    bool verbose = false, skip_existing = false, debug_images = false, stage_by_stage = false, in_process = false, rccl_always = false;
    PipeOptions po;
    std::string raw_calibdir, cam0_dir, cam1_dir, raw_out;
    long max_frames = -1;
    for (int i = 2; i < argc; ++i) {
        const std::string a = argv[i];
        if (a == "--gpus" && i + 1 < argc) gpus = atoi(argv[++i]);
        else if (a == "--procs-per-gpu" && i + 1 < argc) ppg = atoi(argv[++i]);
        else if (a == "--threads-per-proc" && i + 1 < argc) tpp = atoi(argv[++i]);
        else if (a == "--out" && i + 1 < argc) outdir = argv[++i];
        else if (a == "--verbose") verbose = true;
        else if (a == "--skip-existing") skip_existing = true;
        else if (a == "--debug-images") debug_images = true;       // the reference's per-frame debug pictures (render.hpp); off here
        else if (a == "--stage-by-stage") stage_by_stage = true;   // the synchronous per-stage calls instead of the pipelined chain
        else if (a == "--decode-threads" && i + 1 < argc) po.decode_threads = atoi(argv[++i]);
        else if (a == "--writer-threads" && i + 1 < argc) po.writer_threads = atoi(argv[++i]);
        else if (a == "--no-inliers-file") po.inliers_file = false;
        else if (a == "--rccl-always") rccl_always = true;         // the RCCL leg of the mean plane also with ONE worker (a world of one)
        else if (a == "--in-process") in_process = true;           // one worker, not forked: lets a profiler (rocprofv3) see the GPU work
        else if (a == "--raw" && i + 1 < argc) raw_calibdir = argv[++i];
        else if (a == "--cam0" && i + 1 < argc) cam0_dir = argv[++i];
        else if (a == "--cam1" && i + 1 < argc) cam1_dir = argv[++i];
        else if (a == "--frames" && i + 1 < argc) max_frames = atol(argv[++i]);
        else if (a == "--save-undistorted") po.save_undistorted = true;
        else if (a == "--sequence" && i + 1 < argc && !raw_calibdir.empty()) {
            raw_out = argv[++i];                                  // prepare-less mode: the workdirs are created below
            if (outdir.empty()) outdir = raw_out;
        }
        else if (a == "--sequence" && i + 1 < argc) {
            const std::string root = argv[++i];
            if (outdir.empty()) outdir = root;
            std::vector<std::string> found;
            if (DIR* d = opendir(root.c_str())) {
                while (dirent* e = readdir(d)) {
                    const std::string n = e->d_name;
                    if (n.size() == 9 && n.compare(6, 3, "_wd") == 0 && std::all_of(n.begin(), n.begin() + 6, ::isdigit)) found.push_back(n);
                }
                closedir(d);
            }
            std::sort(found.begin(), found.end());
            for (const auto& n : found) wds.push_back(path_join(root, n));
        } else if (a.rfind("--", 0) == 0) { std::cerr << "unknown option " << a << std::endl; return -1; }
        else wds.push_back(a);
    }
    // prepare-less mode: wass_prepare's work (undistortion, optional CLAHE) is done on the GPU inside the frame chain, from the
    // cameras' raw pictures; frame t -> <output_dir>/%06d_wd as wasscli numbers them (wasscli.py:222-227)
    PrepareSetup prep;
    std::vector<std::string> cam0_files, cam1_files;
    if (!raw_calibdir.empty()) {
        if (raw_out.empty() || cam0_dir.empty() || cam1_dir.empty() || !wds.empty()) {
            std::cerr << "--raw <calibdir> needs --cam0 <dir> --cam1 <dir> and, AFTER it, --sequence <output_dir> (no workdir arguments)" << std::endl;
            return -1;
        }
        std::string err;
        if (!load_prepare_setup(raw_calibdir, prep, &err)) { std::cerr << err << std::endl; return -1; }
        if (!prep.have_ext) { std::cerr << "Extrinsic calibration not found in " << raw_calibdir << " (ext_R.xml, ext_T.xml): wass_stereo cannot run on such workdirs" << std::endl; return -1; }
        cam0_files = list_image_files(cam0_dir);
        cam1_files = list_image_files(cam1_dir);
        if (cam0_files.empty() || cam0_files.size() != cam1_files.size()) {
            std::cerr << "cam0 and cam1 directories are empty or contain a different set of images" << std::endl;
            return -1;
        }
        size_t n = cam0_files.size();
        if (max_frames >= 0 && (size_t)max_frames < n) n = (size_t)max_frames;
        create_directories(raw_out);
        for (size_t t = 0; t < n; ++t) { char nm[32]; snprintf(nm, sizeof nm, "%06zu_wd", t); wds.push_back(path_join(raw_out, nm)); }
        po.prep = &prep; po.cam0 = &cam0_files; po.cam1 = &cam1_files;
    }
    if (gpus < 1 || ppg < 1 || tpp < 1 || po.decode_threads < 1 || po.writer_threads < 1 || wds.empty()) { std::cerr << "Invalid arguments" << std::endl; return -1; }
    // The configuration is read once here to choose the worker form; a file that does not parse is left to the per-frame
    // path, which reports it in every frame's log exactly as wass_stereo does.
    Config config;
    register_wass_stereo_options(config);
    if (const char* e = getenv("WASS_DEBUG_IMAGES")) debug_images = atoi(e) != 0;
    bool pipelined = !stage_by_stage && tpp == 1;
    po.debug_pictures = debug_images;
    {
        std::ifstream ifs(cfg);
        if (!ifs.is_open()) { std::cerr << "Unable to load " << cfg << std::endl; return -1; }
        std::string why;
        try { config.load(ifs); if (pipelined && !pipeline_eligible(config, &why)) { pipelined = false; std::cout << "stage-by-stage calls: " << why << std::endl; } }
        catch (const std::runtime_error&) { pipelined = false; }
    }
    if (po.prep && !pipelined) { std::cerr << "--raw needs the pipelined chain (no --stage-by-stage / --threads-per-proc, an eligible configuration)" << std::endl; return -1; }
    if (outdir.empty()) outdir = ".";
    const int world = gpus * ppg;
    // Coll-1 over RCCL: every worker owns a GPU (one worker per GPU) and there is more than one of them -- or --rccl-always, which runs the
    // same leg with a world of one (how a single-GPU box tests that a worker WITHOUT PyTorch in its process finds librccl, builds a
    // communicator from the parent's unique id and all-reduces).  If this process cannot load librccl the sequence is not lost: the
    // parent gathers every frame's plane anyway and reduces them itself.
    bool distinct = ppg == 1 && (world > 1 || rccl_always);
    int uid_up[2] = { -1, -1 };
    std::vector<int> uid_down_r(world, -1), uid_down_w(world, -1);
    if (distinct) {
        bool pipes_ok = pipe(uid_up) == 0;
        for (int r = 1; r < world && pipes_ok; ++r) { int pfd2[2]; pipes_ok = pipe(pfd2) == 0; if (pipes_ok) { uid_down_r[r] = pfd2[0]; uid_down_w[r] = pfd2[1]; } }
        if (!pipes_ok) { perror("pipe"); return -1; }
    }
    auto coll_of = [&](int r) { CollSetup c; c.use = distinct; c.up_fd = uid_up[1]; c.down_fd = uid_down_r[r]; return c; };

    std::cout << "wass_stereo_batch: " << wds.size() << " frame(s), " << world << " worker process(es) x " << tpp << " thread(s) on " << gpus << " GPU(s)"
              << (pipelined ? ", pipelined (" + std::to_string(po.decode_threads) + " decode / " + std::to_string(po.writer_threads) + " writer threads per worker)" : std::string(", stage by stage")) << std::endl;
    const double t0 = now();
    std::vector<pid_t> pids(world);
    std::vector<int> fds(world);
    std::thread inproc;
    for (int r = 0; r < world; ++r) {          // fork BEFORE any HIP call in this process: every worker initialises its own runtime
        int pfd[2];
        if (pipe(pfd) != 0) { perror("pipe"); return -1; }
        if (in_process && world == 1 && pipelined) {
            // profiling aid: the worker as a thread of this process (records still travel through the pipe; verbose, so that stdout stays ours)
            const int wfd = pfd[1];
            inproc = std::thread([&, wfd]() { (void)worker_pipelined(0, 1, 0, coll_of(0), cfg, config, wds, true, skip_existing, wfd, po); close(wfd); });
            pids[r] = -1; fds[r] = pfd[0];
            continue;
        }
        const pid_t pid = fork();
        if (pid < 0) { perror("fork"); return -1; }
        if (pid == 0) {
            close(pfd[0]);
            for (int q = 0; q < r; ++q) close(fds[q]);
            if (uid_up[0] >= 0) close(uid_up[0]);
            for (int q = 0; q < world; ++q) { if (uid_down_w[q] >= 0) close(uid_down_w[q]); if (q != r && uid_down_r[q] >= 0) close(uid_down_r[q]); }
            _exit(pipelined ? worker_pipelined(r, world, r / ppg, coll_of(r), cfg, config, wds, verbose, skip_existing, pfd[1], po)
                            : worker(r, world, r / ppg, coll_of(r), cfg, wds, verbose, skip_existing, debug_images, pfd[1], tpp));
        }
        close(pfd[1]);
        pids[r] = pid; fds[r] = pfd[0];
    }
    if (distinct) {
        // worker 0's [ok, id] -> every other worker (worker 0 sends it before anything else; a worker 0 that died sends nothing: ok = 0)
        const bool threaded = inproc.joinable();
        if (!threaded) close(uid_up[1]);
        unsigned char msg[129] = {};
        if (!read_all(uid_up[0], msg, sizeof msg)) memset(msg, 0, sizeof msg);
        for (int r = 1; r < world; ++r) { (void)write_all(uid_down_w[r], msg, sizeof msg); close(uid_down_w[r]); close(uid_down_r[r]); }
        close(uid_up[0]);
        if (msg[0] != 1) {
            std::cerr << "RCCL is not available (wass_coll_unique_id failed in worker 0); planes will be reduced by the parent process" << std::endl;
            distinct = false;
        }
    }
    std::vector<Record> recs(wds.size());
    std::vector<char> got(wds.size(), 0);
    std::vector<Tail> tails(world);
    bool ok = true;
    // Every worker's pipe is drained as it fills (poll): a worker blocks in write() once the 64 KiB pipe holds ~1000
    // records, and with one worker per GPU rank 0 only sends its tail after the RCCL all-reduce, which needs EVERY rank --
    // draining the workers one after the other would deadlock sequences of more than ~1000 frames per worker.
    {
        struct Chan { std::vector<unsigned char> buf; size_t nrec = 0, want = 0; bool tail = false, open = true; };
        std::vector<Chan> ch(world);
        for (int r = 0; r < world; ++r) ch[r].want = (wds.size() + world - 1 - r) / world;
        int live = world;
        while (live > 0) {
            std::vector<pollfd> pf;
            std::vector<int> who;
            for (int r = 0; r < world; ++r)
                if (ch[r].open) { pf.push_back({ fds[r], POLLIN, 0 }); who.push_back(r); }
            if (poll(pf.data(), pf.size(), -1) < 0) {
                if (errno == EINTR) continue;
                perror("poll");
                ok = false;
                // nobody will read the pipes any more: close them (a worker blocked in write() gets EPIPE and exits) and end
                // the workers, so that the waitpid() below cannot hang
                for (int r = 0; r < world; ++r)
                    if (ch[r].open) { close(fds[r]); ch[r].open = false; kill(pids[r], SIGTERM); }
                break;
            }
            for (size_t q = 0; q < pf.size(); ++q) {
                if (pf[q].revents & POLLNVAL) {             // not an open descriptor: would spin forever
                    ch[who[q]].open = false; --live; ok = false;
                    continue;
                }
                if (!(pf[q].revents & (POLLIN | POLLHUP | POLLERR))) continue;
                Chan& c = ch[who[q]];
                unsigned char tmp[16384];
                const ssize_t n = read(pf[q].fd, tmp, sizeof tmp);
                if (n < 0 && (errno == EINTR || errno == EAGAIN)) continue;
                if (n > 0) c.buf.insert(c.buf.end(), tmp, tmp + n);
                size_t off = 0;
                while (c.nrec < c.want && c.buf.size() - off >= sizeof(Record)) {
                    Record rec;
                    memcpy(&rec, c.buf.data() + off, sizeof rec);
                    off += sizeof rec;
                    ++c.nrec;
                    if (rec.index < 0 || (size_t)rec.index >= wds.size()) { ok = false; continue; }
                    recs[rec.index] = rec; got[rec.index] = 1;
                    std::cout << "[frame " << rec.index << "] " << wds[rec.index] << "  rc=" << rec.rc << "  " << rec.seconds << " s  "
                              << rec.n_points << " pts" << (rec.have_plane ? "" : "  (no plane)") << std::endl;
                }
                if (c.nrec == c.want && !c.tail && c.buf.size() - off >= sizeof(Tail)) {
                    memcpy(&tails[who[q]], c.buf.data() + off, sizeof(Tail));
                    off += sizeof(Tail);
                    c.tail = true;
                }
                c.buf.erase(c.buf.begin(), c.buf.begin() + off);
                if (n <= 0 || c.tail) {                     // end of stream (worker gone) or everything received
                    if (!c.tail || tails[who[q]].magic != 0x57415353) ok = false;
                    c.open = false;
                    --live;
                    close(pf[q].fd);
                }
            }
        }
        if (inproc.joinable()) inproc.join();
        for (int r = 0; r < world; ++r) {
            int st = 0;
            if (pids[r] < 0) continue;
            waitpid(pids[r], &st, 0);
            if (!WIFEXITED(st) || WEXITSTATUS(st) != 0) { std::cerr << "worker " << r << " failed (status " << st << ")" << std::endl; ok = false; }
        }
    }
    // planes.txt as wasscli builds it: the lines of plane.txt joined by single blanks, successful frames only (:341-343)
    std::ofstream fpl(path_join(outdir, "planes.txt").c_str());
    double acc[5] = { 0, 0, 0, 0, 0 };
    int nfail = 0;
    for (size_t i = 0; i < wds.size(); ++i) {
        if (!got[i] || recs[i].rc != 0) { ++nfail; continue; }
        std::ifstream fin(path_join(wds[i], "plane.txt").c_str());
        std::string line, joined;
        while (std::getline(fin, line)) { if (!joined.empty()) joined += " "; joined += line; }
        fpl << joined << "\n";
        wass_planes_mean_accumulate(recs[i].plane, 1, acc);
    }
    double mean[4]; int nvalid = 0;
    wass_planes_mean_finish(acc, mean, &nvalid);
    if (distinct && ok) {                      // what the workers agreed on over RCCL must be what the parent sees
        for (int r = 0; r < world; ++r)
            if (!tails[r].used_rccl || tails[r].n_valid != nvalid) { std::cerr << "RCCL mean plane disagrees with the gathered planes" << std::endl; ok = false; }
        for (int k = 0; k < 4 && ok; ++k) mean[k] = tails[0].mean[k];
    }
    {
        std::ofstream fm(path_join(outdir, "planes_mean.txt").c_str());
        fm << std::setprecision(20);
        for (int k = 0; k < 4; ++k) fm << mean[k] << std::endl;
    }
    const double dt = now() - t0;
    std::cout << "mean plane over " << nvalid << " frame(s): " << std::setprecision(12) << mean[0] << " " << mean[1] << " " << mean[2] << " " << mean[3]
              << (distinct ? "  (RCCL all-reduce)" : "") << std::endl;
    std::cout << wds.size() - nfail << "/" << wds.size() << " frame(s) ok in " << dt << " s (" << (wds.size() / dt) << " frames/s)" << std::endl;
    {
        double steady = 0;
        for (int r = 0; r < world; ++r)
            if (tails[r].computed > 1 && tails[r].last_done > tails[r].first_done) steady += (tails[r].computed - 1) / (tails[r].last_done - tails[r].first_done);
        if (steady > 0) std::cout << "steady state (first to last finished frame of every worker, start-up excluded): " << steady << " frames/s" << std::endl;
        rusage ru;                                           // the workers have been waited for: their CPU time is the host's share of a frame
        if (getrusage(RUSAGE_CHILDREN, &ru) == 0) {
            const double cpu = (double)ru.ru_utime.tv_sec + ru.ru_utime.tv_usec / 1e6 + (double)ru.ru_stime.tv_sec + ru.ru_stime.tv_usec / 1e6;
            std::cout << "host CPU of the workers: " << cpu << " s (" << (cpu / dt) << " cores busy on average, " << (1e3 * cpu / std::max<size_t>(wds.size(), 1))
                      << " ms per frame)" << std::endl;
        }
    }
    return ok && nfail == 0 ? 0 : -1;
}
