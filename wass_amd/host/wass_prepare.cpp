// wass_prepare.cpp -- drop-in host program for the reference's wass_prepare stage, the producer of the workdir that
// wass_stereo reads (/root/reference/src/wass_prepare/wass_prepare.cpp:303-540), calling
// libwassgpu through its C ABI: wass_clahe (optional contrast equalisation, :257-262) and wass_undistort (:268).
//
//   wass_prepare --workdir <dir> --calibdir <dir> --c0 <image> --c1 <image> [--continue-if-existing] [--genconfig]
//
// Same options, exit codes, calibdir inputs (intrinsics_0X.xml, distortion_0X.xml, ext_R.xml, ext_T.xml,
// prepare_config.txt), workdir outputs (undistorted/0000000X.png, intrinsics_0000000X.xml, ext_R.xml, ext_T.xml), stdout
// progress markers and log lines as the reference.  Out of scope and rejected loudly: the polarimetric camera branch
// (--demosaic / --hdr / --dolp-aolp / --save-channels / --save-stokes, :100-255).  Inputs: PNG, TIFF and baseline JPEG
// (hostio.hpp, tiff.hpp, jpeg_read.hpp), the formats wasscli accepts (wasscli.py:47).
// There is NO CPU implementation of the image work: without a GPU (or libwassgpu.so) the program fails with exit -1.
#include <sys/stat.h>
#include <sys/types.h>

#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>

#include "prepare_common.hpp"
#include "tiff.hpp"
#include "../../include/wass_gpu.h"

#ifndef WASS_AMD_VERSION
#define WASS_AMD_VERSION "1.26-mi355x"
#endif

using namespace wasshost;

namespace {

bool exists(const std::string& p) { struct stat st; return stat(p.c_str(), &st) == 0; }
bool is_dir(const std::string& p) { return prep_is_dir(p); }
std::string join(const std::string& a, const std::string& b) { return prep_join(a, b); }
void register_options(Config& c) { register_prepare_options(c); }                              // wass_prepare.cpp:36-39

const char* kUsage =
    "wass_prepare arguments:\n"
    "  --workdir arg           Workdir name\n"
    "  --calibdir arg          Calibration data directory\n"
    "  --c0 arg                Cam0 image file\n"
    "  --c1 arg                Cam1 image file\n"
    "  --demosaic              Demosaic polarimetric images\n"
    "  --hdr                   Use light polarization to compute HDR intensity image\n"
    "  --dolp-aolp             Compute DOLP and AOLP and save them as color-mapped images\n"
    "  --save-channels         Output I0,I45,I90,I135 as separate images\n"
    "  --save-stokes           Save the full Stokes vector as 32bit float tiff images\n"
    "  --continue-if-existing  Don't complain if output dir already exists\n"
    "  --genconfig             Generate configuration file\n";

struct Gpu {
    wass_ctx* ctx = nullptr;
    ~Gpu() { if (ctx) wass_ctx_destroy(ctx); }
};

// process_image (:43-282) without the polarimetric branch: load, optional CLAHE, undistort, write
bool process_image(Gpu& gpu, const std::string& filename, const Mat& K, const Mat& dist, int clahe_tiles, double clahe_clip,
                   const std::string& outdir, const std::string& outfile)
{
    WLOGI << "Processing " << filename;
    Image img;
    try { img = read_image_gray(filename); } catch (const std::exception& e) { WLOGE << e.what(); return false; }     // PNG, TIFF or baseline JPEG
    WLOGI << "Input image size: " << img.w << "x" << img.h;
    if (!gpu.ctx) {
        const char* dev = getenv("WASS_GPU_DEVICE");      // the same variable wass_stereo and wass_stereo_batch read
        if (wass_ctx_create(dev ? atoi(dev) : 0, &gpu.ctx) != WASS_OK) {
            WLOGE << "unable to open the GPU: " << wass_last_error(gpu.ctx);
            return false;
        }
    }
    if (clahe_tiles > 0) {
        Image dst(img.w, img.h);
        if (wass_clahe(gpu.ctx, img.px.data(), img.w, img.h, (size_t)img.w, clahe_clip, clahe_tiles, clahe_tiles, dst.px.data()) != WASS_OK) {
            WLOGE << "CLAHE: " << wass_last_error(gpu.ctx);
            return false;
        }
        img = dst;
    }
    if (K.rows != 3 || K.cols != 3) { WLOGE << "invalid intrinsic matrix"; return false; }
    const int nd = (int)dist.d.size();
    Image und(img.w, img.h);
    if (wass_undistort(gpu.ctx, img.px.data(), img.w, img.h, (size_t)img.w, K.d.data(), dist.d.data(), nd, und.px.data()) != WASS_OK) {
        WLOGE << "undistort: " << wass_last_error(gpu.ctx);
        return false;
    }
    if (!write_png_gray(join(outdir, outfile + ".png"), und, prepared_png_level())) { WLOGE << "Unable to write " << join(outdir, outfile + ".png"); return false; }
    WLOGI << "Output image size: " << und.w << "x" << und.h;
    return true;
}

}  // namespace

int main(int argc, char* argv[])
{
    // six hardware queues for the HIP runtime, while this process is still single-threaded (libwassgpu sets the same default before its first
    // HIP call -- wass_amd/csrc/api.hip default_hw_queues has the story -- but setenv() there would race with the getenv() of other threads)
    (void)setenv("GPU_MAX_HW_QUEUES", "6", 0);
    std::cout << "wass_prepare  v. " << WASS_AMD_VERSION << std::endl;
    std::cout << "----------------------------------------------" << std::endl;
    std::cout << " [Release] MI355X / gfx950 HIP build, " << wass_version() << std::endl << std::endl;
    WLOG_SCOPE("wass_prepare");

    if (argc == 1) { std::cout << kUsage << std::endl; return 0; }

    std::string workdir, calibdir, c0, c1;
    bool genconfig = false, cont = false, polarimetric = false;
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i], val;
        // boost::program_options stores no positional token when the program declares none (the reference, :303-346): wasscli relies on it --
        // it passes "%s" % ("--demosaic" if ... else ""), i.e. an EMPTY argument, for every ordinary camera (wasscli.py:226-227)
        if (a.empty() || a[0] != '-') continue;
        bool has_val = false;
        const size_t eq = a.find('=');
        if (a.rfind("--", 0) == 0 && eq != std::string::npos) { val = a.substr(eq + 1); a = a.substr(0, eq); has_val = true; }
        auto value = [&](std::string& dst) -> bool {
            if (has_val) { dst = val; return true; }
            if (i + 1 >= argc) { WLOGE << "the required argument for option '" << a << "' is missing"; return false; }
            dst = argv[++i];
            return true;
        };
        if (a == "--workdir") { if (!value(workdir)) return -1; }
        else if (a == "--calibdir") { if (!value(calibdir)) return -1; }
        else if (a == "--c0") { if (!value(c0)) return -1; }
        else if (a == "--c1") { if (!value(c1)) return -1; }
        else if (a == "--genconfig") genconfig = true;
        else if (a == "--continue-if-existing") cont = true;
        else if (a == "--demosaic" || a == "--hdr" || a == "--dolp-aolp" || a == "--save-channels" || a == "--save-stokes") polarimetric = true;
        else { WLOGE << "unrecognised option '" << a << "'"; return -1; }
    }

    Config cfg;
    register_options(cfg);
    if (genconfig) {                                                                            // :347-351 (return value ignored there too)
        WLOGI << "Writing prepare_config.txt";
        std::ofstream ofs("prepare_config.txt");
        if (!ofs.is_open()) WLOGE << "Unable to open prepare_config.txt for write";
        else { ofs << cfg.to_config_string(); WLOGI << "Done!"; }
        return 0;
    }
    if (workdir.empty()) { WLOGE << "workdir option not specified"; return -1; }
    if (calibdir.empty()) { WLOGE << "calibdir option not specified"; return -1; }
    if (c0.empty() || c1.empty()) { WLOGE << "c0 and c1 options must be both specified"; return -1; }
    if (polarimetric) {
        WLOGE << "the polarimetric camera options (--demosaic, --hdr, --dolp-aolp, --save-channels, --save-stokes) are not part of this build";
        return -1;
    }
    if (!is_dir(calibdir)) { WLOGE << "Invalid calibration directory"; return -1; }
    if (!cont && exists(workdir)) { WLOGE << workdir << " already exists."; return -1; }

    try {
        WLOGI << "Checking if configuration file exists...";
        const std::string config_filename = join(calibdir, "prepare_config.txt");
        std::ifstream ifs(config_filename);
        if (!ifs.is_open()) WLOGE << "Unable to load " << config_filename;
        else { cfg.load(ifs); WLOGI << "Settings loaded"; }
    } catch (const std::runtime_error& er) { WLOGE << er.what(); return -1; }

    WLOGI << "Creating " << workdir;
    create_directories(workdir);
    std::cout << "[P|10|100]" << std::endl;

    WLOGI << "Loading calibration data";
    Mat intr0, dist0, intr1, dist1;
    if ((intr0 = load_matrix_xml(join(calibdir, "intrinsics_00.xml"))).rows == 0) return -1;
    if ((dist0 = load_matrix_xml(join(calibdir, "distortion_00.xml"))).rows == 0) {
        WLOGI << join(calibdir, "distortion_00.xml") << " not found. Assuming no distortion.";
        dist0 = Mat(5, 1);
    }
    if ((intr1 = load_matrix_xml(join(calibdir, "intrinsics_01.xml"))).rows == 0) return -1;
    if ((dist1 = load_matrix_xml(join(calibdir, "distortion_01.xml"))).rows == 0) {
        WLOGI << join(calibdir, "distortion_01.xml") << " not found. Assuming no distortion.";
        dist1 = Mat(5, 1);
    }
    std::cout << "[P|20|100]" << std::endl;

    WLOGI << "Processing images";
    const std::string undist_dir = join(workdir, "undistorted");
    if (!create_directories(undist_dir) && !cont) { WLOGE << "Unable to create " << undist_dir; return -1; }

    Gpu gpu;
    if (!process_image(gpu, c0, intr0, dist0, cfg.get_int("CAM0_CLAHE_TILEGRIDSIZE"), cfg.get_double("CAM0_CLAHE_CLIPLIMIT"), undist_dir, "00000000"))
        return -1;
    std::cout << "[P|50|100]" << std::endl;
    if (!process_image(gpu, c1, intr1, dist1, cfg.get_int("CAM1_CLAHE_TILEGRIDSIZE"), cfg.get_double("CAM1_CLAHE_CLIPLIMIT"), undist_dir, "00000001"))
        return -1;
    std::cout << "[P|70|100]" << std::endl;

    const Mat extR = load_matrix_xml(join(calibdir, "ext_R.xml")), extT = load_matrix_xml(join(calibdir, "ext_T.xml"));
    if (extR.rows == 3 && extR.cols == 3 && extT.rows == 3 && extT.cols == 1) {
        WLOGI << "Extrinsic calibration found, copying to destination workdir";
        save_matrix_xml(join(workdir, "ext_R.xml"), "R", extR);
        save_matrix_xml(join(workdir, "ext_T.xml"), "T", extT);
    } else {
        WLOGE << "Extrinsic calibration not found.";
    }
    WLOGI << "Saving intrinsic calibration data";
    save_matrix_xml(join(workdir, "intrinsics_00000000.xml"), "intr", intr0);
    save_matrix_xml(join(workdir, "intrinsics_00000001.xml"), "intr", intr1);
    WLOGI << "All done, exiting";
    std::cout << "[P|100|100]" << std::endl;
    return 0;
}
