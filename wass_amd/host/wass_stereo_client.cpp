// wass_stereo_client.cpp -- the `wass_stereo` that wasscli starts once per frame
// (/root/reference/cli/wasscli/wasscli.py:326-346: `wass_stereo <config_file> <workdir>`, four at a time).
//
// Same argv, same files, same stdout, same exit code as the full program (wass_stereo.cpp, installed next to this one as
// `wass_stereo_gpu`) -- but this executable links neither libwassgpu nor the HIP runtime: a frame that a per-GPU resident worker
// can compute (stereo_server.hpp) is handed to it over a unix socket, and a process that only does that should not spend 10 ms
// of every call mapping and relocating libraries it never calls.  Everything else -- no arguments, --genconfig, --measure,
// --rectify-only, WASS_NO_SERVER=1, WASS_STAGE_BY_STAGE=1, a configuration that does not parse or that the device-resident chain does
// not cover (pipeline_eligible, config.hpp), no server and none to be started -- goes to `wass_stereo_gpu` by exec with the arguments untouched.  There is no computation
// here and no CPU fallback: without the full program the call fails.
#include "config.hpp"
#include "stereo_client.hpp"

#include <sstream>

int main(int argc, char* argv[])
{
    char self[4096];
    const ssize_t sl = readlink("/proc/self/exe", self, sizeof self - 1);
    if (sl <= 0) { std::cerr << "wass_stereo: cannot locate the installation (/proc/self/exe)" << std::endl; return -1; }
    self[sl] = 0;
    std::string full(self);
    const size_t slash = full.rfind('/');
    full = (slash == std::string::npos ? std::string(".") : full.substr(0, slash)) + "/wass_stereo_gpu";
    auto hand_over = [&]() {
        std::cout.flush();
        execv(full.c_str(), argv);
        std::cerr << "wass_stereo: cannot start " << full << ": " << strerror(errno) << std::endl;
        return -1;
    };
    auto on = [](const char* name) { const char* e = getenv(name); return e && atoi(e) != 0; };
    if (argc != 3 || argv[1][0] == '-' || on("WASS_NO_SERVER") || on("WASS_STAGE_BY_STAGE")) return hand_over();
    std::string cfg_text;
    {
        std::ifstream ifs(argv[1]);
        if (!ifs.is_open()) return hand_over();
        std::stringstream ss;
        ss << ifs.rdbuf();
        cfg_text = ss.str();
        wasshost::Config cfg;
        wasshost::register_wass_stereo_options(cfg);
        std::istringstream is(cfg_text);
        try { cfg.load(is); } catch (const std::runtime_error&) { return hand_over(); }
        if (!wasshost::pipeline_eligible(cfg)) return hand_over();
    }
    struct stat st;
    if (stat(argv[2], &st) != 0) return hand_over();                       // "<workdir> does not exists, aborting."
    bool debug_images = true;
    if (const char* e = getenv("WASS_DEBUG_IMAGES")) debug_images = atoi(e) != 0;
    wassserver::print_banner();
    const int rc = wassserver::client_run(full.c_str(), argv[1], cfg_text, argv[2], debug_images);
    if (rc != -2) return rc;
    // not computed (nobody to ask, refused, connection lost before an answer): the full program computes it in this process
    setenv("WASS_BANNER_DONE", "1", 1);
    setenv("WASS_CLIENT_TRIED", "1", 1);
    return hand_over();
}
