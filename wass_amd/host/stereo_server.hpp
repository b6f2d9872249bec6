// stereo_server.hpp -- a resident worker behind the UNCHANGED command line.
//
// wasscli starts one `wass_stereo <config> <workdir>` process per frame, NUM_PARALLEL_PROCESSES (4) at a time
// (/root/reference/cli/wasscli/wasscli.py:326-346).  For the reference that is free; here every process pays 0.3 s of HIP
// start-up and 5-8 GB of scratch allocation before its 8 ms of GPU work, and no two frames ever share the device-resident
// chain.  With this header the executable stays what wasscli calls -- same argv, same files, same stdout, same exit code --
// but the frame is computed by a per-GPU SERVER process that the first caller starts and later callers find:
//
//   client  (wass_stereo <cfg> <workdir>)      connects to a unix socket (under $XDG_RUNTIME_DIR or /tmp, one per user and GPU);
//                                              nobody there -> takes a lock file, starts `wass_stereo --server <socket> <gpu>`
//                                              detached, waits for the socket; sends configuration text + workdir; prints the
//                                              frame's log and progress markers as they come back; exits with the frame's code
//   server                                     accept thread -> decode threads (FramePipeline::prepare: calibration, PNGs, camera
//                                              files, rectification's decisions) -> ONE thread that owns the GPU context(s) and
//                                              submits frames of all clients back to back through the device-resident chain
//                                              (stage / submit, flush when no other frame is waiting) -> writer threads (finish:
//                                              plane.txt, mesh_cam.xyzC, inlier text, previews, log) -> reply.  One FramePipeline
//                                              (context, scratch, rectification maps) per distinct configuration text, created on
//                                              first use; exits after WASS_SERVER_IDLE seconds (default 20) without a client.
//
// WASS_NO_SERVER=1 keeps everything in the calling process (the round-4 behaviour); so do --rectify-only, --measure,
// WASS_STAGE_BY_STAGE=1 and configurations the chain does not cover (pipeline_eligible).  If the server cannot be reached or
// dies before it answers, the client computes the frame itself: a lost server costs time, never a frame.
#pragma once

#include <fcntl.h>
#include <poll.h>
#include <signal.h>
#include <sys/file.h>
#include <sys/socket.h>
#include <sys/stat.h>
#include <sys/un.h>
#include <sys/wait.h>
#include <unistd.h>

#include <dirent.h>

#include <algorithm>
#include <functional>
#include <map>
#include <thread>

#include "frame_pipeline.hpp"
#include "stereo_client.hpp"

namespace wassserver {

using namespace wassframe;

// ------------------------------------------------------------------ server
struct PipeEntry {
    std::string key;
    Config cfg;
    std::unique_ptr<FramePipeline> pl;
    bool bad = false;                  // the configuration text does not parse or is not eligible: the client computes such frames itself
    bool debug = false;                // the reference's debug pictures are drawn (they want the decoded pictures in finish())
};
struct RaEntry;
struct ServerJob : FrameJob {
    int fd = -1;
    PipeEntry* entry = nullptr;
    bool read_ahead = false;           // its pictures were decoded before it asked
    bool computed_ahead = false;       // ... and the whole frame was computed before it asked (speculation)
    bool counted = false;              // ... and counts among the frames computed ahead that hold an output set
    std::shared_ptr<RaEntry> spec;     // set while the frame is nobody's yet: computed ahead, waiting for its caller
    double t_accept = 0, t_decoded = 0, t_submit = 0, t_collected = 0, t_written = 0;     // WASS_SERVER_TIMING
};

// ---- read-ahead.  wasscli walks a sequence in order (000000_wd, 000001_wd, ...: wasscli.py:308-346), and every caller waits for ITS
// frame: 16 of the 42 ms a call spends in the server are the inflation of its two PNGs.  Once two frames of one sequence directory have
// been asked for, the pictures of the next few workdirs of that directory are decoded before their callers exist.  A decoded pair is
// used only if both files still have the inode, size and modification time they had before AND after they were read; anything else
// (and any workdir nobody predicted) is decoded on demand, as before.  WASS_SERVER_READAHEAD=<n> workdirs (default 6, 0 = off).
//
// ---- speculation (WASS_SERVER_SPECULATE=<n>, default 5, 0 = off).  Those workdirs are not only decoded: a frame job is PREPARED for each
// (calibration, rectification's decisions; the small files prepare() writes are kept in memory, FrameJob::deferred) and, whenever no
// caller's frame is waiting, sent through the GPU chain -- up to n frames may be complete and waiting for their callers.  Nothing touches the workdir until its caller arrives; then the
// inputs are checked (both pictures and the four calibration files: inode, size, mtime as before AND after they were read; same
// configuration text, same configuration path) and the frame's files are written -- the caller waits for its 41 MB, not for its frame.
// A frame whose caller never comes is dropped when it falls out of the cache or the server goes.  With N callers that each spend most
// of a call waiting for files, the GPU would otherwise idle between their requests; this way the unchanged command line runs at
// the sequence driver's rate.
struct FileSig {
    bool ok = false;
    ino_t ino = 0; off_t size = 0; timespec mtime = { 0, 0 };
    static FileSig of(const std::string& path)
    {
        FileSig s;
        struct stat st;
        if (stat(path.c_str(), &st) == 0) { s.ok = true; s.ino = st.st_ino; s.size = st.st_size; s.mtime = st.st_mtim; }
        return s;
    }
    bool operator==(const FileSig& o) const { return ok && o.ok && ino == o.ino && size == o.size && mtime.tv_sec == o.mtime.tv_sec && mtime.tv_nsec == o.mtime.tv_nsec; }
};
enum RaState { RA_QUEUED, RA_RUNNING, RA_DONE, RA_PREPARED, RA_ONGPU, RA_COMPUTED, RA_TAKEN };
struct RaEntry {
    std::string workdir;
    RaState state = RA_QUEUED;
    Preload pre;
    FileSig sig[2];
    bool valid = false;                  // the files did not change while they were read
    // speculation: the frame job built (and possibly computed) for this workdir, owned by the entry until a caller claims it
    PipeEntry* pe = nullptr;
    std::string cfgpath;
    ServerJob* job = nullptr;
    FileSig calib[4];
};
struct ReadAhead {
    typedef RaEntry Entry;
    static constexpr RaState QUEUED = RA_QUEUED, RUNNING = RA_RUNNING, DONE = RA_DONE, PREPARED = RA_PREPARED, ONGPU = RA_ONGPU, COMPUTED = RA_COMPUTED, TAKEN = RA_TAKEN;
    std::mutex mu;
    std::condition_variable cv;
    std::map<std::string, std::shared_ptr<Entry>> by_dir;
    std::deque<std::shared_ptr<Entry>> order;          // oldest first: the cache holds at most `cap` pairs (10 MB each at 5 megapixels)
    std::deque<std::shared_ptr<Entry>> todo;
    std::map<std::string, int> seen;                   // requests per sequence directory
    int depth = 6;
    int spec_depth = 5;                                // > 0: they are also prepared as frame jobs, and this many may be computed and waiting for their callers
    size_t cap = 16;
    bool closed = false;
    std::atomic<uint64_t> hits{ 0 }, misses{ 0 }, stale{ 0 }, computed{ 0 }, prepared{ 0 }, dropped{ 0 };
    std::deque<std::pair<std::string, double>> recent; // the requests of the last seconds: not predicted again (concurrent callers arrive out of order)
    std::function<void(const std::shared_ptr<Entry>&)> on_prepared;      // a frame job is ready for the GPU thread
    std::function<void(ServerJob*)> drop_job;                            // give back what a dropped job holds (output set), delete it
    static std::string calib_file(const std::string& wd, int k)
    {
        static const char* const n[4] = { "intrinsics_00000000.xml", "intrinsics_00000001.xml", "ext_R.xml", "ext_T.xml" };
        return path_join(wd, n[k]);
    }

    static std::string pic(const std::string& wd, int k) { return path_join(wd, k == 0 ? "undistorted/00000000.png" : "undistorted/00000001.png"); }
    // "<parent>/<prefix><digits><suffix>" -> the same name with the number raised by k (same width); empty when the name holds no number
    static std::string sibling(const std::string& wd, unsigned long long k)
    {
        std::string w = wd;
        while (w.size() > 1 && w.back() == '/') w.pop_back();
        unsigned long long v = 0;
        size_t b = 0, e = 0;
        if (!workdir_number(w, &v, &b, &e)) return std::string();
        char num[32];
        snprintf(num, sizeof num, "%0*llu", (int)(e - b), v + k);
        if (strlen(num) != e - b) return std::string();
        return w.substr(0, b) + num + w.substr(e);
    }
    static std::string parent(const std::string& wd)
    {
        std::string w = wd;
        while (w.size() > 1 && w.back() == '/') w.pop_back();
        const size_t slash = w.rfind('/');
        return slash == std::string::npos ? std::string(".") : w.substr(0, slash);
    }
    static void decode(Entry& e)
    {
        for (int k = 0; k < 2; ++k) e.sig[k] = FileSig::of(pic(e.workdir, k));
        preload_images(e.workdir, e.pre);
        e.valid = e.pre.error.empty() && e.sig[0] == FileSig::of(pic(e.workdir, 0)) && e.sig[1] == FileSig::of(pic(e.workdir, 1));
    }
    struct Claim {
        std::shared_ptr<Entry> pictures;     // decoded ahead: the caller's prepare() takes them
        ServerJob* job = nullptr;            // prepared ahead (computed = false: goes to the GPU queue) or computed ahead (goes to the writers)
        bool computed = false;
    };
    // the files an entry was made from are still those files (with_calib: a frame job also depends on the four calibration files)
    bool inputs_unchanged(const Entry& e, const std::string& wd, bool with_calib) const
    {
        if (!e.valid) return false;
        for (int k = 0; k < 2; ++k) if (!(e.sig[k] == FileSig::of(pic(wd, k)))) return false;
        if (with_calib) for (int k = 0; k < 4; ++k) if (!(e.calib[k] == FileSig::of(calib_file(wd, k)))) return false;
        return true;
    }
    // (mu held) an entry leaves the cache for good: what it owns goes back
    void discard(const std::shared_ptr<Entry>& e)
    {
        if (e->state == QUEUED) e->state = TAKEN;
        if ((e->state == PREPARED || e->state == COMPUTED || e->state == DONE) && e->job) { ServerJob* j = e->job; e->job = nullptr; e->state = TAKEN; ++dropped; if (drop_job) drop_job(j); }
    }
    // A request for `wd` has arrived (configuration `pe`, named `cfgpath` by the caller).  Returns what was done for it ahead of time if that
    // is still good for the files as they are now (the entry leaves the cache either way), and queues the workdirs that follow it: every
    // stride-th (a node's callers pick GPU = frame number mod GPUs); the nearest `spec_depth` of them with a frame job (speculate).
    Claim request(const std::string& asked, int stride, PipeEntry* pe, const std::string& cfgpath, bool speculate)
    {
        // (entries are kept under the name without trailing slashes: matlab/run_wass.m:103 calls with "<dir>/000012_wd/", wasscli without)
        std::string wd = asked;
        while (wd.size() > 1 && wd.back() == '/') wd.pop_back();
        std::shared_ptr<Entry> mine;
        Claim c;
        {
            std::unique_lock<std::mutex> lk(mu);
            if (depth <= 0) return c;
            auto it = by_dir.find(wd);
            if (it != by_dir.end()) {
                mine = it->second;
                const std::shared_ptr<Entry> mine0 = mine;
                if (mine->state == QUEUED) { mine->state = TAKEN; mine = nullptr; }          // not started: this thread decodes it itself, now
                else cv.wait(lk, [&]() { return (mine->state != RUNNING && mine->state != ONGPU) || closed; });   // (a frame on the GPU: a few ms)
                // (by identity: while this thread waited, the entry may have been evicted and the workdir predicted afresh)
                auto again = by_dir.find(wd);
                if (again != by_dir.end() && again->second == mine0) by_dir.erase(again);
                for (auto o = order.begin(); o != order.end(); ++o) if (*o == mine0) { order.erase(o); break; }
            }
            const double now = Timer::now();
            recent.emplace_back(wd, now);
            while (recent.size() > 64 || (!recent.empty() && now - recent.front().second > 0.5)) recent.pop_front();
            auto asked_lately = [&](const std::string& d) { for (const auto& r : recent) if (r.first == d) return true; return false; };
            const int n = ++seen[parent(wd)];
            if (seen.size() > 64) { seen.clear(); }
            if (n >= 2 && !closed)
                for (int k = 1; k <= depth; ++k) {
                    const std::string nx = sibling(wd, (unsigned long long)k * (unsigned long long)std::max(1, stride));
                    if (nx.empty() || by_dir.count(nx) || asked_lately(nx)) continue;
                    if (!FileSig::of(pic(nx, 0)).ok) break;                                // the sequence ends here (or is not one)
                    auto e = std::make_shared<Entry>();
                    e->workdir = nx;
                    if (speculate) { e->pe = pe; e->cfgpath = cfgpath; }      // (every predicted workdir gets its frame job; how many are COMPUTED ahead is the GPU thread's business)
                    by_dir[nx] = e;
                    order.push_back(e);
                    todo.push_back(e);
                    for (size_t scan = 0; order.size() > cap && scan < order.size();) {     // predicted and never asked for: oldest out
                        auto old = order[scan];
                        if (old->state == RUNNING || old->state == ONGPU) { ++scan; continue; }     // (busy: it goes the next time)
                        order.erase(order.begin() + (long)scan);
                        by_dir.erase(old->workdir);
                        discard(old);
                    }
                }
            cv.notify_all();
            if (mine && mine->state != DONE && mine->state != PREPARED && mine->state != COMPUTED) mine = nullptr;
            if (mine) {
                // it is this caller's from here on: the GPU thread must not pick a PREPARED job up while the files are being looked at (the
                // entry is still in its queue; taken twice, the frame went to two writers)
                const RaState was = mine->state;
                mine->state = TAKEN;
                ServerJob* job = mine->job;
                mine->job = nullptr;
                const bool same = !job || (mine->pe == pe && mine->cfgpath == cfgpath);
                lk.unlock();
                const bool fresh = same && inputs_unchanged(*mine, wd, job != nullptr);
                lk.lock();
                if (!fresh) { ++stale; if (job) { ++dropped; if (drop_job) drop_job(job); } mine = nullptr; }
                else if (job) { c.job = job; c.computed = was == COMPUTED; }
                else if (was == DONE) c.pictures = mine;
                else mine = nullptr;
            }
        }
        if (c.job) { if (c.computed) ++computed; else ++prepared; c.job->workdir = asked; c.job->env.workdir = asked; return c; }
        if (!c.pictures) { ++misses; return c; }
        ++hits;
        c.pictures->pre.workdir = asked;                             // load_data takes the pair only for the workdir it is loading, as that is spelt
        return c;
    }
    void worker()
    {
        for (;;) {
            std::shared_ptr<Entry> e;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&]() { return closed || !todo.empty(); });
                if (closed) return;
                e = todo.front();
                todo.pop_front();
                if (e->state != QUEUED) continue;                                           // its caller came first, or it fell out of the cache
                e->state = RUNNING;
            }
            decode(*e);
            ServerJob* job = nullptr;
            if (e->valid && e->pe) {
                // the frame job, as a decode thread would build it for a caller -- except that nothing is written (DeferredFilesScope)
                for (int k = 0; k < 4; ++k) e->calib[k] = FileSig::of(calib_file(e->workdir, k));
                job = new ServerJob();
                job->entry = e->pe;
                job->workdir = e->workdir;
                job->config_path = e->cfgpath;
                job->read_ahead = job->computed_ahead = true;
                job->pre = &e->pre;
                { DeferredFilesScope quiet(&job->deferred); e->pe->pl->prepare(*job); }
                job->pre = nullptr;
                bool same = job->rc == 0;
                for (int k = 0; k < 4 && same; ++k) same = e->calib[k] == FileSig::of(calib_file(e->workdir, k));
                if (!same) { delete job; job = nullptr; e->valid = false; }                 // (its caller prepares the frame itself and gets the real message)
            }
            {
                std::lock_guard<std::mutex> lk(mu);
                if (closed && job) { delete job; job = nullptr; }                           // (the server is going: nobody will claim or drop it)
                e->job = job;
                e->state = job ? PREPARED : DONE;
                if (job) job->spec = e;
            }
            cv.notify_all();
            if (job && on_prepared) on_prepared(e);
        }
    }
    // the GPU thread: this prepared frame goes to the GPU now (false: its caller has come meanwhile, or it was dropped)
    ServerJob* to_gpu(const std::shared_ptr<Entry>& e)
    {
        std::lock_guard<std::mutex> lk(mu);
        if (e->state != PREPARED || !e->job) return nullptr;
        e->state = ONGPU;
        return e->job;
    }
    // ... and has come back complete
    void computed_now(const std::shared_ptr<Entry>& e)
    {
        { std::lock_guard<std::mutex> lk(mu); if (e->state == ONGPU) e->state = COMPUTED; }
        cv.notify_all();
    }
    void close()
    {
        {
            std::lock_guard<std::mutex> lk(mu);
            closed = true;
            for (auto& e : order) discard(e);
            order.clear(); by_dir.clear(); todo.clear();
        }
        cv.notify_all();
    }
};

template <typename T> class Queue {
public:
    void push(T v) { { std::lock_guard<std::mutex> lk(mu_); q_.push_back(std::move(v)); } cv_.notify_one(); }
    // false: closed and empty, or timed out (timeout_ms < 0: wait for ever)
    bool pop(T& out, int timeout_ms = -1)
    {
        std::unique_lock<std::mutex> lk(mu_);
        auto ready = [&]() { return !q_.empty() || closed_ || kicked_; };
        if (timeout_ms < 0) cv_.wait(lk, ready);
        else if (!cv_.wait_for(lk, std::chrono::milliseconds(timeout_ms), ready)) return false;
        kicked_ = false;
        if (q_.empty()) return false;
        out = std::move(q_.front());
        q_.pop_front();
        return true;
    }
    void close() { { std::lock_guard<std::mutex> lk(mu_); closed_ = true; } cv_.notify_all(); }
    // wakes a pop() that is waiting: it returns false as if its time were up (there is other work for the waiting thread)
    void kick() { { std::lock_guard<std::mutex> lk(mu_); kicked_ = true; } cv_.notify_all(); }
    bool closed() { std::lock_guard<std::mutex> lk(mu_); return closed_; }
    size_t size() { std::lock_guard<std::mutex> lk(mu_); return q_.size(); }
private:
    std::mutex mu_;
    std::condition_variable cv_;
    std::deque<T> q_;
    bool closed_ = false, kicked_ = false;
};

inline int server_main(const std::string& sock, int device)
{
    signal(SIGPIPE, SIG_IGN);
    int idle_s = 20, ndec = 6, nwr = 4;
    if (const char* e = getenv("WASS_SERVER_IDLE")) idle_s = std::max(1, atoi(e));
    if (const char* e = getenv("WASS_SERVER_DECODE")) ndec = std::max(1, atoi(e));
    if (const char* e = getenv("WASS_SERVER_WRITERS")) nwr = std::max(1, atoi(e));
    int nra = 2;                                                   // read-ahead threads (each decodes the two pictures of a workdir side by side)
    if (const char* e = getenv("WASS_SERVER_RA_THREADS")) nra = std::max(1, std::min(16, atoi(e)));
    ReadAhead readahead;
    if (const char* e = getenv("WASS_SERVER_READAHEAD")) readahead.depth = std::max(0, atoi(e));
    if (const char* e = getenv("WASS_SERVER_SPECULATE")) readahead.spec_depth = std::max(0, atoi(e));
    const int spec_max = std::max(1, std::min(8, readahead.spec_depth));   // frames staged, on the GPU or computed ahead and not yet claimed (the last two kinds hold an output set)
    int spec_stage = 3;                                            // of which so many may be staged (pictures uploaded) in front of the GPU
    if (const char* e = getenv("WASS_SERVER_SPEC_STAGE")) spec_stage = std::max(1, std::min(3, atoi(e)));
    int deep_at = 2;                                               // callers waiting behind the GPU from which the chain runs two frames deep
    if (const char* e = getenv("WASS_SERVER_DEEP_AT")) deep_at = std::max(0, atoi(e));
    // WASS_SERVER_TIMING=<file>: one line per frame -- where a caller's waiting time went (decode, queue, GPU, files)
    FILE* tlog = nullptr;
    std::mutex tlog_mu;
    if (const char* e = getenv("WASS_SERVER_TIMING")) if (*e) tlog = fopen(e, "a");
    // a device this environment does not have (the count honours the runtimes' visibility variables): no server -- the caller that
    // started this process sees it go and computes its frame itself
    if (device < 0 || device >= count_gpus()) return 1;
    int max_configs = 4;                                           // live pipelines (each owns a context and 5-8 GB of scratch once used)
    if (const char* e = getenv("WASS_SERVER_MAX_CONFIGS")) max_configs = std::max(1, atoi(e));
    unlink(sock.c_str());
    const int lfd = socket(AF_UNIX, SOCK_STREAM | SOCK_CLOEXEC, 0);
    sockaddr_un a;
    memset(&a, 0, sizeof a);
    a.sun_family = AF_UNIX;
    if (lfd < 0 || sock.size() >= sizeof a.sun_path) return 1;
    memcpy(a.sun_path, sock.c_str(), sock.size() + 1);
    const mode_t um = umask(0077);
    const bool bound = bind(lfd, (sockaddr*)&a, sizeof a) == 0 && listen(lfd, 64) == 0;
    umask(um);
    if (!bound) return 1;

    std::mutex pipes_mu;
    std::map<std::string, std::unique_ptr<PipeEntry>> pipes;
    Queue<ServerJob*> incoming, ready, towrite;
    Queue<std::shared_ptr<RaEntry>> spec_q;                        // frame jobs prepared ahead of their callers, for the GPU thread's spare time
    std::atomic<int> in_flight{ 0 }, spec_live{ 0 };
    readahead.on_prepared = [&](const std::shared_ptr<RaEntry>& e) { spec_q.push(e); ready.kick(); };
    readahead.drop_job = [&](ServerJob* j) {
        if (j->counted) --spec_live;                               // (it had been through the GPU)
        j->entry->pl->abandon(*j);
        j->spec.reset();
        delete j;
    };
    std::atomic<bool> stopping{ false };

    auto reply = [](ServerJob* j) {
        if (j->fd < 0) return;
        const char o = 'O', x = 'X';
        const int32_t rc = j->rc;
        (void)(send_all(j->fd, &o, 1) && send_str(j->fd, j->log) && send_all(j->fd, &x, 1) && send_all(j->fd, &rc, 4));
        close(j->fd);
        j->fd = -1;
    };
    auto refuse = [](int fd) { const char r = 'R'; (void)send_all(fd, &r, 1); close(fd); };

    // ---- decode threads: request -> pipeline of its configuration -> prepare
    auto decoder = [&]() {
        ServerJob* j;
        while (incoming.pop(j)) {
            std::string magic(6, '\0'), cfgpath, cfgtext, wd, opts;
            uint32_t n = 0;
            const bool ok = recv_all(j->fd, &magic[0], 6) && magic == "WSRV1\n" && recv_all(j->fd, &n, 4) && n == 4 && recv_str(j->fd, cfgpath) &&
                            recv_str(j->fd, cfgtext) && recv_str(j->fd, wd) && recv_str(j->fd, opts);
            if (!ok) { close(j->fd); delete j; --in_flight; continue; }
            const bool debug = opts.find("debug=1") != std::string::npos;
            int stride = 1;
            { const size_t sp = opts.find("stride="); if (sp != std::string::npos) stride = std::max(1, atoi(opts.c_str() + sp + 7)); }
            // what of the caller's environment changes its frame's files is applied per pipeline (the key holds the options)
            auto opt_on = [&](const char* name, const char* value) {
                const std::string k = std::string(name) + "=";
                const size_t at = opts.find(k);
                if (at == std::string::npos) return false;
                const std::string v = opts.substr(at + k.size(), opts.find(';', at) == std::string::npos ? std::string::npos : opts.find(';', at) - at - k.size());
                return value ? v == value : atoi(v.c_str()) != 0;
            };
            const std::string key = opts + "\n" + cfgtext;
            PipeEntry* pe = nullptr;
            {
                std::lock_guard<std::mutex> lk(pipes_mu);
                // a configuration nobody has sent before while the server already holds max_configs pipelines (a parameter sweep): this
                // caller computes its frame itself, the server's HBM stays bounded
                if (!pipes.count(key) && (int)pipes.size() >= max_configs) { refuse(j->fd); delete j; --in_flight; continue; }
                auto& slot = pipes[key];
                if (!slot) {
                    slot.reset(new PipeEntry());
                    slot->key = key;
                    register_wass_stereo_options(slot->cfg);
                    try {
                        std::istringstream is(cfgtext);
                        slot->cfg.load(is);
                        if (!pipeline_eligible(slot->cfg)) slot->bad = true;
                    } catch (const std::runtime_error&) { slot->bad = true; }
                    if (!slot->bad) {
                        FramePipeline::Options fo;
                        fo.out_slots = nwr + 2 + spec_max;
                        fo.live = true;                          // the progress markers go into the frame's log ...
                        fo.echo = false;                         // ... which is relayed to the client, not printed here
                        fo.max_pending = 1;                      // every caller waits for ONE frame: hand it out as early as possible
                        fo.debug_pictures = debug;
                        fo.debug_png = opt_on("WASS_DEBUG_FORMAT", "png") ? 1 : 0;
                        fo.host_inlier_text = opt_on("WASS_HOST_INLIER_TEXT", nullptr) ? 1 : 0;
                        fo.host_debug_pictures = opt_on("WASS_HOST_DEBUG_PICTURES", nullptr) ? 1 : 0;
                        slot->debug = debug;
                        fo.inliers_file = true;
                        slot->pl.reset(new FramePipeline(device, slot->cfg, cfgpath, fo));
                    }
                }
                pe = slot.get();
            }
            if (pe->bad) { refuse(j->fd); delete j; --in_flight; continue; }
            j->entry = pe;
            j->workdir = wd;
            j->config_path = cfgpath;
            if (!exists(wd)) {
                j->log = wd + " does not exists, aborting.\n";      // wass_stereo.cpp:1836 (the reference's wording)
                j->rc = -1;
            } else {
                const bool speculate = readahead.spec_depth > 0 && !pe->pl->host_debug_pictures() && pe->cfg.get_string("LEFT_MASK_IMAGE") == "none" &&
                                       pe->cfg.get_string("RIGHT_MASK_IMAGE") == "none";
                ReadAhead::Claim c = readahead.request(wd, stride, pe, cfgpath, speculate);     // something may have been done for this workdir already
                if (c.job) {
                    // its frame job exists: prepared (it joins the GPU queue as this caller's) or already computed (straight to the writers)
                    ServerJob* s = c.job;
                    s->spec.reset();
                    s->fd = j->fd; s->t_accept = j->t_accept;
                    delete j;
                    s->t_decoded = Timer::now();
                    if (s->counted) { --spec_live; s->counted = false; ready.kick(); }      // (room for another frame ahead)
                    if (c.computed) { s->t_submit = s->t_decoded; towrite.push(s); }
                    else { s->computed_ahead = false; ready.push(s); }
                    continue;
                }
                j->pre = c.pictures ? &c.pictures->pre : nullptr;
                j->read_ahead = c.pictures != nullptr;
                pe->pl->prepare(*j);
                j->pre = nullptr;
            }
            j->t_decoded = Timer::now();
            ready.push(j);
        }
    };
    // ---- writer threads: finish -> reply
    auto writer = [&]() {
        ServerJob* j;
        while (towrite.pop(j)) {
            j->t_collected = Timer::now();
            j->entry->pl->finish(*j);
            j->t_written = Timer::now();
            reply(j);
            if (tlog) {
                std::lock_guard<std::mutex> lk(tlog_mu);
                fprintf(tlog, "%s %s decode %.1f queue %.1f gpu %.1f files %.1f reply %.1f total %.1f ms\n", j->workdir.c_str(), j->computed_ahead ? "computed" : j->read_ahead ? "ahead" : "demand", (j->t_decoded - j->t_accept) * 1e3,
                        (j->t_submit - j->t_decoded) * 1e3, (j->t_collected - j->t_submit) * 1e3, (j->t_written - j->t_collected) * 1e3,
                        (Timer::now() - j->t_written) * 1e3, (Timer::now() - j->t_accept) * 1e3);
                fflush(tlog);
            }
            delete j;
            --in_flight;
        }
    };
    std::vector<std::thread> pool;
    for (int t = 0; t < nra; ++t) pool.emplace_back([&]() { readahead.worker(); });
    for (int t = 0; t < ndec; ++t) pool.emplace_back(decoder);
    for (int t = 0; t < nwr; ++t) pool.emplace_back(writer);

    // ---- accept thread
    std::atomic<long long> last_activity{ (long long)time(nullptr) };
    std::thread acceptor([&]() {
        for (;;) {
            pollfd pf = { lfd, POLLIN, 0 };
            const int r = poll(&pf, 1, 500);
            if (stopping) return;
            if (r <= 0) continue;
            const int fd = accept4(lfd, nullptr, nullptr, SOCK_CLOEXEC);
            if (fd < 0) continue;
            // a request is a few kilobytes sent in one go: a peer that connects and then says nothing must not hold a decode thread
            const timeval tv = { 3, 0 };
            (void)setsockopt(fd, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof tv);
            last_activity = (long long)time(nullptr);
            ServerJob* j = new ServerJob();
            j->fd = fd;
            j->t_accept = Timer::now();
            ++in_flight;
            incoming.push(j);
        }
    });

    // ---- this thread owns every GPU context: frames of all clients, back to back
    PipeEntry* cur = nullptr;
    std::vector<FrameJob*> done;
    std::deque<ServerJob*> ahead;                                  // staged (pictures on their way to the GPU), not yet submitted; all of `cur`
    auto hand_over = [&]() {
        for (FrameJob* f : done) {
            ServerJob* sj = static_cast<ServerJob*>(f);
            if (std::shared_ptr<RaEntry> e = sj->spec) readahead.computed_now(e);      // nobody's yet: it waits, complete, for its caller (ReadAhead owns it)
            else towrite.push(sj);
        }
        done.clear();
    };
    auto submit_oldest = [&]() {
        ServerJob* j = ahead.front();
        ahead.pop_front();
        // few callers (wasscli's four): each frame is handed out as early as possible; callers queueing up behind the GPU (the menu's
        // "number of parallel workers" raised): two frames deep, the GPU never waits for this thread
        // -- and with frames being computed ahead of their callers (speculation) nobody waits for a frame's GPU time at all: two deep, always
        cur->pl->set_max_pending((deep_at > 0 && (int)(ahead.size() + ready.size()) + 1 >= deep_at) || j->computed_ahead || spec_q.size() > 0 ? 2 : 1);
        cur->pl->submit(*j, done);
        if (!cur->debug) { j->env.left = Image(); j->env.right = Image(); }   // the pictures are in the pinned ring now
        hand_over();
    };
    auto drain = [&]() {
        while (!ahead.empty()) submit_oldest();
        if (cur && cur->pl->pending()) { cur->pl->flush(done); hand_over(); }
    };
    bool foreign_device = false;                                   // this server's device index does not exist in this process's HIP runtime
    auto take = [&](ServerJob* j) {
        last_activity = (long long)time(nullptr);
        if (j->entry != cur) { drain(); cur = j->entry; }
        // A GPU that is not there although others are (WASS_NUM_GPUS / the topology said more than the runtime shows): the caller computes
        // its frame itself, on the device an in-process run picks -- slow, not lost.  (No GPU at all stays what it was: the frame fails
        // loudly, with its log, whoever computes it.)
        if (!foreign_device && device > 0 && !cur->pl->context()) { int n = 0; foreign_device = wass_device_count(&n) == WASS_OK && n > 0 && device >= n; }
        if (foreign_device && j->fd >= 0) { refuse(j->fd); j->fd = -1; cur->pl->abandon(*j); delete j; --in_flight; return; }
        // Staged as soon as it is here, up to two frames ahead of the one being submitted: uploads and downloads share one copy queue,
        // and an upload enqueued behind the previous frame's downloads waits for that frame's TAIL -- the SGM stage behind the upload
        // then starts after the tail instead of beside it (8 callers: 11 ms per frame instead of 8.6).
        j->t_submit = Timer::now();
        cur->pl->stage(*j);
        ahead.push_back(j);
    };
    for (;;) {
        ServerJob* j = nullptr;
        const bool pending = cur && cur->pl->pending();
        // a frame in flight with nobody behind it: give a concurrent caller a moment to arrive (its frame's SGM stage then runs
        // while this one's tail does), then complete it
        if (ready.pop(j, !ahead.empty() ? 0 : pending ? 2 : 250)) {
            take(j);
            if (ahead.size() < 3 && ready.size() > 0) continue;       // more callers waiting: their pictures first
        } else if (!foreign_device && (int)ahead.size() < spec_stage && spec_live.load() < spec_max) {
            // no caller's frame is waiting: frames that were prepared ahead of their callers (speculation) -- staged like callers' frames,
            // up to two ahead of the one being submitted (an upload enqueued behind the previous frame's downloads waits for that frame's
            // tail: staged one at a time the speculative chain ran at 96 frames/s where callers' frames run at 114)
            std::shared_ptr<RaEntry> e;
            while ((int)ahead.size() < spec_stage && spec_live.load() < spec_max && spec_q.pop(e, 0))
                if (ServerJob* sj = readahead.to_gpu(e)) { ++spec_live; sj->counted = true; take(sj); }
        }
        if (!ahead.empty()) { submit_oldest(); continue; }
        if (pending) { cur->pl->flush(done); hand_over(); }
        if (in_flight.load() == 0 && (long long)time(nullptr) - last_activity.load() >= idle_s) break;
        if (in_flight.load() > 0) last_activity = (long long)time(nullptr);
    }
    // shutting down: no new clients (the socket goes first: a late caller starts a fresh server), everything in flight completes
    stopping = true;
    unlink(sock.c_str());
    close(lfd);
    acceptor.join();
    incoming.close();
    while (in_flight.load() > 0) {                                 // a request that slipped in between the last check and the unlink
        ServerJob* j = nullptr;
        if (ready.pop(j, 50)) { take(j); if (!ahead.empty()) submit_oldest(); }
        else drain();
    }
    drain();
    readahead.close();                                             // (frames computed ahead and never asked for give their output sets back)
    ready.close();
    towrite.close();
    spec_q.close();
    for (auto& t : pool) t.join();
    pipes.clear();
    if (tlog) {
        fprintf(tlog, "read-ahead: %llu frames decoded before they were asked for, %llu on demand, %llu decoded early and changed since\n",
                (unsigned long long)readahead.hits.load(), (unsigned long long)readahead.misses.load(), (unsigned long long)readahead.stale.load());
        fprintf(tlog, "speculation: %llu frames computed before they were asked for, %llu prepared, %llu dropped unclaimed\n",
                (unsigned long long)readahead.computed.load(), (unsigned long long)readahead.prepared.load(), (unsigned long long)readahead.dropped.load());
        fclose(tlog);
    }
    return 0;
}

}  // namespace wassserver
