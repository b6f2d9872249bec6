// wass_stereo.cpp -- drop-in host program for the reference's wass_stereo stage
// (/root/reference/src/wass_stereo/wass_stereo.cpp:1799-2149), calling libwassgpu through its C ABI.
//
//   wass_stereo [--genconfig] <config_file> <workdir> [--measure] [--rectify-only]
//
// Same argv, exit codes, config format, workdir inputs/outputs, stdout progress markers and log format as the
// reference; the per-frame work is wass_frame.hpp.  There is NO CPU implementation of the hot
// path here: without a GPU (or without libwassgpu.so) the program fails with exit code -1.
#include "stereo_server.hpp"

using namespace wassframe;

int main(int argc, char* argv[])
{
    // six hardware queues for the HIP runtime, while this process is still single-threaded (libwassgpu sets the same default before its first
    // HIP call -- wass_amd/csrc/api.hip default_hw_queues has the story -- but setenv() there would race with the getenv() of other threads)
    (void)setenv("GPU_MAX_HW_QUEUES", "6", 0);
    // the resident worker that later wass_stereo processes hand their frames to (stereo_server.hpp); started by the first of them
    if (argc >= 3 && std::string("--server") == argv[1]) return wassserver::server_main(argv[2], argc >= 4 ? atoi(argv[3]) : 0);

    // (installed as `wass_stereo_gpu`; the `wass_stereo` of the same directory is wass_stereo_client.cpp, which printed the banner
    // already when it hands a frame over that no server took)
    if (!getenv("WASS_BANNER_DONE")) {
        std::cout << "wass_stereo  v. " << WASS_AMD_VERSION << std::endl;
        std::cout << "----------------------------------------------" << std::endl;
        std::cout << " [Release] MI355X / gfx950 HIP build, " << wass_version() << std::endl << std::endl;
    }

    if (argc == 1) {
        std::cout << "Usage:" << std::endl;
        std::cout << "wass_stereo [--genconfig] <config_file> <workdir> [--measure] [--rectify-only]" << std::endl << std::endl;
        std::cout << "Not enough arguments, aborting." << std::endl;
        return 0;                                               // wasscli relies on exit 0 here (wasscli.py:66-71)
    }
    if (std::string("--genconfig") == argv[1]) {
        Config cfg;
        register_wass_stereo_options(cfg);
        return save_configuration(cfg, "stereo_config.txt");
    }
    if (argc != 3 && argc != 4) { std::cerr << "Invalid arguments" << std::endl; return -1; }
    if (!exists(argv[2])) { std::cerr << argv[2] << " does not exists, aborting." << std::endl; return -1; }
    wass_ctx* ctx = nullptr;
    // A whole frame goes through the device-resident chain the sequence driver uses (frame_pipeline.hpp), one frame deep: nothing
    // comes back to the host between the stages; the debug pictures (on by default, like the reference: WASS_DEBUG_IMAGES=0
    // switches them off) are rendered and JPEG-coded on the device behind the frame's tail (csrc/jpeg.hip).  The synchronous stage-by-stage calls of
    // wass_frame.hpp remain for --rectify-only, for the options that need an intermediate mesh on the host
    // (pipeline_eligible) and on request (WASS_STAGE_BY_STAGE=1); the files both ways are the same (tests/test_cli.py).
    bool debug_images = true, stage_by_stage = false;
    if (const char* e = getenv("WASS_DEBUG_IMAGES")) debug_images = atoi(e) != 0;
    if (const char* e = getenv("WASS_STAGE_BY_STAGE")) stage_by_stage = atoi(e) != 0;
    if (!stage_by_stage && argc == 3) {
        Config cfg;
        register_wass_stereo_options(cfg);
        bool ok = false;
        std::string cfg_text;
        {
            std::ifstream ifs(argv[1]);
            if (ifs.is_open()) {
                std::stringstream ss;
                ss << ifs.rdbuf();
                cfg_text = ss.str();
                std::istringstream is(cfg_text);
                try { cfg.load(is); ok = pipeline_eligible(cfg); } catch (const std::runtime_error&) {}
            }
        }
        const char* dev_env = getenv("WASS_GPU_DEVICE");
        // A per-GPU server computes the frame when there is one or one can be started: same files, same output, same exit code, without
        // this process ever initialising HIP (WASS_NO_SERVER=1: everything below runs here, as in round 4).
        const char* no_srv = getenv("WASS_NO_SERVER");
        if (ok && !(no_srv && atoi(no_srv) != 0) && !getenv("WASS_CLIENT_TRIED")) {
            char self[4096];
            const ssize_t sl = readlink("/proc/self/exe", self, sizeof self - 1);
            if (sl > 0) {
                self[sl] = 0;
                const int rc = wassserver::client_run(self, argv[1], cfg_text, argv[2], debug_images);
                if (rc != -2) return rc;
            }
        }
        if (ok) {
            FramePipeline::Options fo;
            fo.out_slots = 1;
            fo.live = true;
            fo.device_previews = false;      // one frame: load_data writes the previews before the GPU is needed, like the reference
            fo.debug_pictures = debug_images;
            FramePipeline pl(dev_env ? atoi(dev_env) : 0, cfg, argv[1], fo);
            FrameJob job;
            job.workdir = argv[2];
            std::vector<FrameJob*> done;
            pl.prepare(job);
            pl.flush_live(job);
            pl.submit(job, done);
            pl.flush_live(job);
            pl.flush(done);
            for (FrameJob* j : done) pl.finish(*j);
            return job.rc;
        }
        // a configuration that does not parse or an option the chain does not cover: the stage-by-stage path handles it
    }
    const int ret = wass_run_frame(argv[1], argv[2], argc == 4 ? argv[3] : nullptr, 0, &ctx, nullptr);
    if (ctx) wass_ctx_destroy(ctx);
    return ret;
}
