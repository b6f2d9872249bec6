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

#include <map>
#include <thread>

#include "frame_pipeline.hpp"

namespace wassserver {

using namespace wassframe;

// ------------------------------------------------------------------ wire format
// request:  "WSRV1\n", u32 n, n x (u32 length, bytes): config path as given, config text, workdir (absolute), options ("k=v;k=v")
// reply:    any number of ('O', u32 length, bytes) stdout chunks, then ('X', i32 exit code)
inline bool send_all(int fd, const void* p, size_t n)
{
    const char* c = (const char*)p;
    while (n) {
        const ssize_t k = send(fd, c, n, MSG_NOSIGNAL);
        if (k < 0 && errno == EINTR) continue;
        if (k <= 0) return false;
        c += k; n -= (size_t)k;
    }
    return true;
}
inline bool recv_all(int fd, void* p, size_t n)
{
    char* c = (char*)p;
    while (n) {
        const ssize_t k = recv(fd, c, n, 0);
        if (k < 0 && errno == EINTR) continue;
        if (k <= 0) return false;
        c += k; n -= (size_t)k;
    }
    return true;
}
inline bool send_str(int fd, const std::string& s) { const uint32_t n = (uint32_t)s.size(); return send_all(fd, &n, 4) && send_all(fd, s.data(), s.size()); }
inline bool recv_str(int fd, std::string& s, size_t limit = 64u << 20)
{
    uint32_t n = 0;
    if (!recv_all(fd, &n, 4) || n > limit) return false;
    s.resize(n);
    return n == 0 || recv_all(fd, &s[0], n);
}

// GPUs of this node without touching HIP (a client must stay cheap): KFD topology nodes with SIMDs
inline int count_gpus()
{
    if (const char* e = getenv("WASS_NUM_GPUS")) { const int n = atoi(e); if (n > 0) return n; }
    int n = 0;
    if (DIR* d = opendir("/sys/class/kfd/kfd/topology/nodes")) {
        while (dirent* e = readdir(d)) {
            if (e->d_name[0] == '.') continue;
            std::ifstream f(std::string("/sys/class/kfd/kfd/topology/nodes/") + e->d_name + "/properties");
            std::string k; long long v;
            while (f >> k >> v) if (k == "simd_count" && v > 0) { ++n; break; }
        }
        closedir(d);
    }
    return n > 0 ? n : 1;
}
inline std::string socket_path(int device)
{
    const char* dir = getenv("WASS_SERVER_DIR");
    if (!dir || !*dir) dir = getenv("XDG_RUNTIME_DIR");
    if (!dir || !*dir || access(dir, W_OK) != 0) dir = "/tmp";
    char b[64];
    snprintf(b, sizeof b, "wass_stereo_%u_gpu%d.sock", (unsigned)getuid(), device);
    return path_join(dir, b);
}
inline int connect_to(const std::string& path)
{
    sockaddr_un a;
    memset(&a, 0, sizeof a);
    a.sun_family = AF_UNIX;
    if (path.size() >= sizeof a.sun_path) return -1;
    memcpy(a.sun_path, path.c_str(), path.size() + 1);
    const int fd = socket(AF_UNIX, SOCK_STREAM | SOCK_CLOEXEC, 0);
    if (fd < 0) return -1;
    if (connect(fd, (sockaddr*)&a, sizeof a) != 0) { close(fd); return -1; }
    return fd;
}

// what the single-frame executable prints from a frame's log: progress markers are lines that start with \x01
inline void print_log(const std::string& log)
{
    size_t p = 0;
    while (p < log.size()) {
        size_t e = log.find('\n', p);
        if (e == std::string::npos) e = log.size(); else ++e;
        if (log[p] == '\x01') std::cout.write(log.data() + p + 1, (std::streamsize)(e - p - 1));
        else std::cout.write(log.data() + p, (std::streamsize)(e - p));
        p = e;
    }
    std::cout.flush();
}

// ------------------------------------------------------------------ client
// Returns the frame's exit code, or -2 when the frame was NOT computed (no server, refused, connection lost before the answer):
// the caller then computes it in-process.
inline int client_run(const char* self_exe, const char* cfg_path, const std::string& cfg_text, const char* workdir, bool debug_images)
{
    int device = 0;
    if (const char* e = getenv("WASS_GPU_DEVICE")) device = atoi(e);
    else { const int g = count_gpus(); if (g > 1) device = (int)((unsigned)getpid() % (unsigned)g); }
    const std::string sock = socket_path(device);
    int fd = connect_to(sock);
    if (fd < 0) {
        // nobody there: one of the callers starts the server, the others wait at the lock and then find it
        const std::string lock = sock + ".lock";
        const int lfd = open(lock.c_str(), O_CREAT | O_RDWR | O_CLOEXEC, 0600);
        if (lfd < 0) return -2;
        if (flock(lfd, LOCK_EX) != 0) { close(lfd); return -2; }
        fd = connect_to(sock);
        if (fd < 0) {
            const pid_t pid = fork();
            if (pid < 0) { close(lfd); return -2; }
            if (pid == 0) {
                // the server must not keep the caller's pipes open (wasscli waits for EOF on them) nor die with its session
                setsid();
                const int nul = open("/dev/null", O_RDWR);
                if (nul >= 0) { dup2(nul, 0); dup2(nul, 1); if (!getenv("WASS_SERVER_STDERR")) dup2(nul, 2); if (nul > 2) close(nul); }
                for (int k = 3; k < 256; ++k) close(k);              // (the lock's descriptor included: the lock belongs to the parent)
                char dev[16];
                snprintf(dev, sizeof dev, "%d", device);
                execl(self_exe, self_exe, "--server", sock.c_str(), dev, (char*)nullptr);
                _exit(127);
            }
            for (int i = 0; i < 3000 && fd < 0; ++i) {              // the socket exists as soon as the server listens: before any HIP call
                usleep(10000);
                fd = connect_to(sock);
                int st;
                if (fd < 0 && waitpid(pid, &st, WNOHANG) == pid) break;     // it died (bad installation): compute here
            }
        }
        flock(lfd, LOCK_UN);
        close(lfd);
        if (fd < 0) return -2;
    }
    char cwd[4096];
    std::string wd = workdir;
    if (!wd.empty() && wd[0] != '/' && getcwd(cwd, sizeof cwd)) wd = path_join(cwd, wd);
    std::string opts = std::string("debug=") + (debug_images ? "1" : "0");
    for (const char* v : { "WASS_DEBUG_FORMAT", "WASS_HOST_INLIER_TEXT" })
        if (const char* e = getenv(v)) opts += std::string(";") + v + "=" + e;
    const uint32_t n = 4;
    bool ok = send_all(fd, "WSRV1\n", 6) && send_all(fd, &n, 4) && send_str(fd, cfg_path) && send_str(fd, cfg_text) && send_str(fd, wd) && send_str(fd, opts);
    bool answered = false;
    int rc = -2;
    while (ok) {
        char t;
        if (!recv_all(fd, &t, 1)) break;
        if (t == 'O') { std::string s; if (!recv_str(fd, s)) break; print_log(s); answered = true; }
        else if (t == 'X') { int32_t v; if (!recv_all(fd, &v, 4)) break; rc = v; answered = true; break; }
        else if (t == 'R') { rc = -2; break; }                      // refused (shutting down): compute here
        else break;
    }
    close(fd);
    if (rc == -2 && answered) return -1;                             // the log was printed and then the server vanished: report a failure, do not print twice
    return rc;
}

// ------------------------------------------------------------------ server
struct PipeEntry {
    std::string key;
    Config cfg;
    std::unique_ptr<FramePipeline> pl;
    bool bad = false;                  // the configuration text does not parse or is not eligible: the client computes such frames itself
    bool debug = false;                // the reference's debug pictures are drawn (they want the decoded pictures in finish())
};
struct ServerJob : FrameJob {
    int fd = -1;
    PipeEntry* entry = nullptr;
};

template <typename T> class Queue {
public:
    void push(T v) { { std::lock_guard<std::mutex> lk(mu_); q_.push_back(std::move(v)); } cv_.notify_one(); }
    // false: closed and empty, or timed out (timeout_ms < 0: wait for ever)
    bool pop(T& out, int timeout_ms = -1)
    {
        std::unique_lock<std::mutex> lk(mu_);
        auto ready = [&]() { return !q_.empty() || closed_; };
        if (timeout_ms < 0) cv_.wait(lk, ready);
        else if (!cv_.wait_for(lk, std::chrono::milliseconds(timeout_ms), ready)) return false;
        if (q_.empty()) return false;
        out = std::move(q_.front());
        q_.pop_front();
        return true;
    }
    void close() { { std::lock_guard<std::mutex> lk(mu_); closed_ = true; } cv_.notify_all(); }
    bool closed() { std::lock_guard<std::mutex> lk(mu_); return closed_; }
    size_t size() { std::lock_guard<std::mutex> lk(mu_); return q_.size(); }
private:
    std::mutex mu_;
    std::condition_variable cv_;
    std::deque<T> q_;
    bool closed_ = false;
};

inline int server_main(const std::string& sock, int device)
{
    signal(SIGPIPE, SIG_IGN);
    int idle_s = 20, ndec = 6, nwr = 4;
    if (const char* e = getenv("WASS_SERVER_IDLE")) idle_s = std::max(1, atoi(e));
    if (const char* e = getenv("WASS_SERVER_DECODE")) ndec = std::max(1, atoi(e));
    if (const char* e = getenv("WASS_SERVER_WRITERS")) nwr = std::max(1, atoi(e));
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
    std::atomic<int> in_flight{ 0 };
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
            const std::string key = opts + "\n" + cfgtext;
            PipeEntry* pe;
            {
                std::lock_guard<std::mutex> lk(pipes_mu);
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
                        fo.out_slots = nwr + 2;
                        fo.live = true;                          // the progress markers go into the frame's log ...
                        fo.echo = false;                         // ... which is relayed to the client, not printed here
                        fo.max_pending = 1;                      // every caller waits for ONE frame: hand it out as early as possible
                        fo.debug_pictures = debug;
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
                LogSinkScope sink(&j->log);
                j->rc = -1;
            } else pe->pl->prepare(*j);
            ready.push(j);
        }
    };
    // ---- writer threads: finish -> reply
    auto writer = [&]() {
        ServerJob* j;
        while (towrite.pop(j)) {
            j->entry->pl->finish(*j);
            reply(j);
            delete j;
            --in_flight;
        }
    };
    std::vector<std::thread> pool;
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
            last_activity = (long long)time(nullptr);
            ServerJob* j = new ServerJob();
            j->fd = fd;
            ++in_flight;
            incoming.push(j);
        }
    });

    // ---- this thread owns every GPU context: frames of all clients, back to back
    PipeEntry* cur = nullptr;
    std::vector<FrameJob*> done;
    auto hand_over = [&]() {
        for (FrameJob* f : done) towrite.push(static_cast<ServerJob*>(f));
        done.clear();
    };
    for (;;) {
        ServerJob* j = nullptr;
        const bool pending = cur && cur->pl->pending();
        // a frame in flight with nobody behind it: give a concurrent caller a moment to arrive (its frame's SGM stage then runs
        // while this one's tail does), then complete it
        if (!ready.pop(j, pending ? 2 : 250)) {
            if (pending) { cur->pl->flush(done); hand_over(); }
            if (in_flight.load() == 0 && (long long)time(nullptr) - last_activity.load() >= idle_s) break;
            if (in_flight.load() > 0) last_activity = (long long)time(nullptr);
            continue;
        }
        last_activity = (long long)time(nullptr);
        if (j->entry != cur) {
            if (cur && cur->pl->pending()) { cur->pl->flush(done); hand_over(); }
            cur = j->entry;
        }
        cur->pl->stage(*j);
        cur->pl->submit(*j, done);
        if (!cur->debug) { j->env.left = Image(); j->env.right = Image(); }   // the pictures are in the pinned ring now
        hand_over();
    }
    // shutting down: no new clients (the socket goes first: a late caller starts a fresh server), everything in flight completes
    stopping = true;
    unlink(sock.c_str());
    close(lfd);
    acceptor.join();
    incoming.close();
    while (in_flight.load() > 0) {                                 // a request that slipped in between the last check and the unlink
        ServerJob* j = nullptr;
        if (ready.pop(j, 50)) {
            if (j->entry != cur) { if (cur && cur->pl->pending()) { cur->pl->flush(done); hand_over(); } cur = j->entry; }
            cur->pl->stage(*j);
            cur->pl->submit(*j, done);
            hand_over();
        } else if (cur && cur->pl->pending()) { cur->pl->flush(done); hand_over(); }
    }
    ready.close();
    towrite.close();
    for (auto& t : pool) t.join();
    pipes.clear();
    return 0;
}

}  // namespace wassserver
