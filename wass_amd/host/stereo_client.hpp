// stereo_client.hpp -- the CLIENT half of the resident worker behind the unchanged command line (stereo_server.hpp has the story
// and the server).  Plain libc / libstdc++ only: the `wass_stereo` that wasscli starts once per frame
// (/root/reference/cli/wasscli/wasscli.py:326-346) is built from this header alone, without libwassgpu or the HIP runtime behind
// it -- loading those costs every call 10 ms before main() -- and hands anything it cannot pass to a server to `wass_stereo_gpu`,
// the full program, by exec.
#pragma once

#include <dirent.h>
#include <fcntl.h>
#include <sys/file.h>
#include <sys/socket.h>
#include <sys/stat.h>
#include <sys/un.h>
#include <sys/wait.h>
#include <unistd.h>

#include <cerrno>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <string>

#ifndef WASS_AMD_VERSION
#define WASS_AMD_VERSION "1.26-mi355x"
#endif
#define WASS_GPU_LIBRARY_VERSION "wass_amd 0.1 (gfx950)"      // = wass_version() of libwassgpu (tests/test_server.py holds the two together)

namespace wassserver {

inline std::string join_path(const std::string& a, const std::string& b) { return (!a.empty() && a.back() == '/') ? a + b : a + "/" + b; }

inline void print_banner()
{
    std::cout << "wass_stereo  v. " << WASS_AMD_VERSION << std::endl;
    std::cout << "----------------------------------------------" << std::endl;
    std::cout << " [Release] MI355X / gfx950 HIP build, " << WASS_GPU_LIBRARY_VERSION << std::endl << std::endl;
}

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

// GPUs of this node that a server process of THIS environment could use, without touching HIP (a client must stay cheap): KFD topology
// nodes with SIMDs whose target is gfx950 (APUs' and other parts' nodes are not counted), at most as many as the visibility variables
// of the HIP / ROCr runtimes leave (HIP_VISIBLE_DEVICES, ROCR_VISIBLE_DEVICES, CUDA_VISIBLE_DEVICES, GPU_DEVICE_ORDINAL: a comma-
// separated list; the runtime numbers what is visible from 0, so only the COUNT matters here).  WASS_NUM_GPUS overrides.
inline int count_gpus()
{
    if (const char* e = getenv("WASS_NUM_GPUS")) { const int n = atoi(e); if (n > 0) return n; }
    int n = 0;
    std::string nodes = "/sys/class/kfd/kfd/topology/nodes";
    if (const char* e = getenv("WASS_KFD_NODES")) nodes = e;            // (tests: a made-up topology)
    if (DIR* d = opendir(nodes.c_str())) {
        while (dirent* e = readdir(d)) {
            if (e->d_name[0] == '.') continue;
            std::ifstream f(nodes + "/" + e->d_name + "/properties");
            std::string k; long long v, simd = 0, target = -1;
            while (f >> k >> v) { if (k == "simd_count") simd = v; else if (k == "gfx_target_version") target = v; }
            if (simd > 0 && (target < 0 || target == 90500)) ++n;
        }
        closedir(d);
    }
    for (const char* var : { "HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "GPU_DEVICE_ORDINAL" })
        if (const char* e = getenv(var)) {
            int listed = 0;
            bool in = false;
            for (const char* c = e; ; ++c) {
                if (*c == ',' || *c == '\0') { if (in) ++listed; in = false; if (!*c) break; }
                else if (*c != ' ') in = true;
            }
            if (n == 0 || listed < n) n = listed;          // (an empty list hides every device: one "GPU", whose server fails loudly)
        }
    return n > 0 ? n : 1;
}
// The number in a workdir's name -- "<parent>/<prefix><digits><suffix>", the LAST run of digits of the last component, as in wasscli's
// 000123_wd -- and where it sits; false when the name holds none.  Frame numbers spread callers over the GPUs of a node and let a
// server predict which workdirs come next (stereo_server.hpp).
inline bool workdir_number(const std::string& wd, unsigned long long* value, size_t* begin = nullptr, size_t* end = nullptr)
{
    size_t n = wd.size();
    while (n > 1 && wd[n - 1] == '/') --n;
    const size_t slash = wd.rfind('/', n ? n - 1 : 0);
    const size_t b0 = (slash == std::string::npos || slash >= n) ? 0 : slash + 1;
    size_t e = n;
    while (e > b0 && !(wd[e - 1] >= '0' && wd[e - 1] <= '9')) --e;
    size_t b = e;
    while (b > b0 && wd[b - 1] >= '0' && wd[b - 1] <= '9') --b;
    if (b == e || e - b > 18) return false;
    if (value) *value = strtoull(wd.substr(b, e - b).c_str(), nullptr, 10);
    if (begin) *begin = b;
    if (end) *end = e;
    return true;
}
// Where the socket and its lock live: WASS_SERVER_DIR, else $XDG_RUNTIME_DIR (per user, mode 0700 by contract), else a directory of
// this user's own under /tmp -- created 0700, and used only if it IS a directory (not a link), owned by this user, closed to everybody
// else: predictable names straight in a world-writable /tmp would let another local user put a socket there first and answer for the
// server.  Empty: no safe place, the caller computes its frame in-process.
inline std::string socket_path(int device)
{
    std::string base;
    const char* dir = getenv("WASS_SERVER_DIR");
    if (!dir || !*dir) dir = getenv("XDG_RUNTIME_DIR");
    if (dir && *dir && access(dir, W_OK) == 0) base = dir;
    else {
        char own[64];
        snprintf(own, sizeof own, "/tmp/wass_stereo_%u", (unsigned)getuid());
        if (mkdir(own, 0700) != 0 && errno != EEXIST) return std::string();
        struct stat st;
        if (lstat(own, &st) != 0 || !S_ISDIR(st.st_mode) || st.st_uid != getuid() || (st.st_mode & 077) != 0) return std::string();
        base = own;
    }
    char b[64];
    snprintf(b, sizeof b, "wass_stereo_%u_gpu%d.sock", (unsigned)getuid(), device);
    return join_path(base, b);
}
inline int connect_to(const std::string& path)
{
    sockaddr_un a;
    memset(&a, 0, sizeof a);
    a.sun_family = AF_UNIX;
    if (path.empty() || path.size() >= sizeof a.sun_path) return -1;
    memcpy(a.sun_path, path.c_str(), path.size() + 1);
    const int fd = socket(AF_UNIX, SOCK_STREAM | SOCK_CLOEXEC, 0);
    if (fd < 0) return -1;
    if (connect(fd, (sockaddr*)&a, sizeof a) != 0) { close(fd); return -1; }
    // whoever listens there must be this user (a server is always started by one of its own callers)
    struct ucred cr;
    socklen_t len = sizeof cr;
    if (getsockopt(fd, SOL_SOCKET, SO_PEERCRED, &cr, &len) != 0 || cr.uid != getuid()) { close(fd); return -1; }
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
    // which GPU: WASS_GPU_DEVICE, else frame number mod GPUs (the sequence driver's rule: frame i -> GPU i mod G, so that wasscli's
    // consecutive frames land on different GPUs and every server sees an arithmetic sequence it can read ahead), else pid mod GPUs
    int device = 0, stride = 1;
    if (const char* e = getenv("WASS_GPU_DEVICE")) device = atoi(e);
    else {
        const int g = count_gpus();
        unsigned long long num = 0;
        if (g > 1 && workdir_number(workdir, &num)) { device = (int)(num % (unsigned long long)g); stride = g; }
        else if (g > 1) device = (int)((unsigned)getpid() % (unsigned)g);
    }
    const std::string sock = socket_path(device);
    if (sock.empty()) return -2;
    int fd = connect_to(sock);
    if (fd < 0) {
        // nobody there: one of the callers starts the server, the others wait at the lock and then find it
        const std::string lock = sock + ".lock";
        const int lfd = open(lock.c_str(), O_CREAT | O_RDWR | O_CLOEXEC | O_NOFOLLOW, 0600);
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
    if (!wd.empty() && wd[0] != '/' && getcwd(cwd, sizeof cwd)) wd = join_path(cwd, wd);
    std::string opts = std::string("debug=") + (debug_images ? "1" : "0") + ";stride=" + std::to_string(stride);
    // what of this caller's environment changes the files of its frame: the server applies these per request (FramePipeline::Options),
    // it does not read its own environment for them
    for (const char* v : { "WASS_DEBUG_FORMAT", "WASS_HOST_INLIER_TEXT", "WASS_HOST_DEBUG_PICTURES" })
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

}  // namespace wassserver
