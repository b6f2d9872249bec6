// prepare_common.hpp -- what wass_prepare (/root/reference/src/wass_prepare/wass_prepare.cpp:303-540) reads from its
// calibration directory and writes into a workdir besides the two undistorted pictures.  Shared by the drop-in wass_prepare
// executable and by the sequence driver's prepare-less mode (frame_pipeline.hpp), which undistorts on the GPU inside the
// frame chain and therefore never writes or reads undistorted/*.png unless asked to.
#pragma once

#include <dirent.h>
#include <sys/stat.h>
#include <sys/types.h>

#include <algorithm>

#include "config.hpp"
#include "hostio.hpp"

namespace wasshost {

inline bool prep_is_dir(const std::string& p) { struct stat st; return stat(p.c_str(), &st) == 0 && S_ISDIR(st.st_mode); }
inline std::string prep_join(const std::string& a, const std::string& b) { return (!a.empty() && a.back() == '/') ? a + b : a + "/" + b; }
// boost::filesystem::create_directories: true when something was created
inline bool create_directories(const std::string& p)
{
    if (prep_is_dir(p)) return false;
    bool made = false;
    for (size_t i = 1; i <= p.size(); ++i)
        if (i == p.size() || p[i] == '/') {
            const std::string sub = p.substr(0, i);
            if (!prep_is_dir(sub) && mkdir(sub.c_str(), 0777) == 0) made = true;
        }
    return made && prep_is_dir(p);
}

inline void register_prepare_options(Config& c)                                               // wass_prepare.cpp:36-39
{
    c.add(Config::DOUBLE, "CAM0_CLAHE_CLIPLIMIT", "2.0", "CAM0 CLAHE cliplimit parameter");
    c.add(Config::INT, "CAM0_CLAHE_TILEGRIDSIZE", "0", "CAM0 CLAHE tile grid size (set to 0 to disable CLAHE). 150 is a good value to start");
    c.add(Config::DOUBLE, "CAM1_CLAHE_CLIPLIMIT", "2.0", "CAM1 CLAHE cliplimit parameter");
    c.add(Config::INT, "CAM1_CLAHE_TILEGRIDSIZE", "0", "CAM1 CLAHE tile grid size (set to 0 to disable CLAHE). 150 is a good value to start");
}

// the calibration directory as wass_prepare's main() reads it (:438-472, 505-533)
struct PrepareSetup {
    Mat intr[2], dist[2], extR, extT;
    bool have_ext = false;
    int clahe_tiles[2] = { 0, 0 };
    double clahe_clip[2] = { 2.0, 2.0 };
};
// false (with a message in *err) when the directory cannot be used; a missing prepare_config.txt or distortion file is not an
// error (the reference logs it and goes on with the defaults / zero distortion)
inline bool load_prepare_setup(const std::string& calibdir, PrepareSetup& ps, std::string* err)
{
    if (!prep_is_dir(calibdir)) { if (err) *err = "Invalid calibration directory"; return false; }
    Config cfg;
    register_prepare_options(cfg);
    {
        std::ifstream ifs(prep_join(calibdir, "prepare_config.txt"));
        if (ifs.is_open()) {
            try { cfg.load(ifs); } catch (const std::runtime_error& e) { if (err) *err = e.what(); return false; }
        }
    }
    ps.clahe_tiles[0] = cfg.get_int("CAM0_CLAHE_TILEGRIDSIZE"); ps.clahe_clip[0] = cfg.get_double("CAM0_CLAHE_CLIPLIMIT");
    ps.clahe_tiles[1] = cfg.get_int("CAM1_CLAHE_TILEGRIDSIZE"); ps.clahe_clip[1] = cfg.get_double("CAM1_CLAHE_CLIPLIMIT");
    for (int cam = 0; cam < 2; ++cam) {
        const std::string k = cam == 0 ? "intrinsics_00.xml" : "intrinsics_01.xml", d = cam == 0 ? "distortion_00.xml" : "distortion_01.xml";
        ps.intr[cam] = load_matrix_xml(prep_join(calibdir, k));
        if (ps.intr[cam].rows != 3 || ps.intr[cam].cols != 3) { if (err) *err = "invalid or missing " + k; return false; }
        struct stat st;
        if (stat(prep_join(calibdir, d).c_str(), &st) == 0) ps.dist[cam] = load_matrix_xml(prep_join(calibdir, d));
        if (ps.dist[cam].rows == 0) ps.dist[cam] = Mat(5, 1);               // "not found. Assuming no distortion." (:447-450)
    }
    struct stat st;
    if (stat(prep_join(calibdir, "ext_R.xml").c_str(), &st) == 0 && stat(prep_join(calibdir, "ext_T.xml").c_str(), &st) == 0) {
        ps.extR = load_matrix_xml(prep_join(calibdir, "ext_R.xml"));
        ps.extT = load_matrix_xml(prep_join(calibdir, "ext_T.xml"));
    }
    ps.have_ext = ps.extR.rows == 3 && ps.extR.cols == 3 && ps.extT.rows == 3 && ps.extT.cols == 1;
    return true;
}
// the calibration files wass_prepare leaves in a workdir (:505-533)
inline void write_prepared_calibration(const std::string& workdir, const PrepareSetup& ps)
{
    if (ps.have_ext) {
        save_matrix_xml(prep_join(workdir, "ext_R.xml"), "R", ps.extR);
        save_matrix_xml(prep_join(workdir, "ext_T.xml"), "T", ps.extT);
    }
    save_matrix_xml(prep_join(workdir, "intrinsics_00000000.xml"), "intr", ps.intr[0]);
    save_matrix_xml(prep_join(workdir, "intrinsics_00000001.xml"), "intr", ps.intr[1]);
}

// the pictures of a camera directory as wasscli pairs them (cli/wasscli/wasscli.py:47,118-124): the first of the supported
// extensions (tif, tiff, png, jpg, jpeg -- as written, glob is case-sensitive) that more than one file has, names sorted
inline std::vector<std::string> list_image_files(const std::string& dir)
{
    static const char* exts[] = { "tif", "tiff", "png", "jpg", "jpeg" };
    std::vector<std::string> names;
    if (DIR* d = opendir(dir.c_str())) {
        while (dirent* e = readdir(d)) names.push_back(e->d_name);
        closedir(d);
    }
    for (const char* ext : exts) {
        std::vector<std::string> out;
        const std::string suffix = std::string(".") + ext;
        for (const auto& n : names)
            if (n.size() > suffix.size() && n[0] != '.' && n.compare(n.size() - suffix.size(), suffix.size(), suffix) == 0) out.push_back(prep_join(dir, n));
        if (out.size() > 1) { std::sort(out.begin(), out.end()); return out; }
    }
    return {};
}

}  // namespace wasshost
