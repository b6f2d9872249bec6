// config.hpp -- KEY=value configuration registry with the file format of the reference's `incfg` library.
//
// The reference registers options with INCFG_REQUIRE(type, NAME, default, "description") at file scope and reads
// them with INCFG_GET(NAME) (src/wass_stereo/wass_stereo.cpp:52-74,742-761,1030-1037, PovMesh.cpp:577-579).  incfg
// itself is an un-vendored git submodule (ext/incfg, empty in the snapshot); its grammar is restated from the
// reference's documentation (doc/src/render/documentation/matcher.html.md:36-84):
//   * one `KEY=value` per line, lines starting with '#' are comments, blank lines are ignored
//   * --genconfig writes, per key in alphabetical order:  "# description\n# \n#KEY=default\n\n"
//   * unknown keys and ill-typed values are errors (std::runtime_error -> exit -1, wass_stereo.cpp:1848-1856)
#pragma once

#include <algorithm>
#include <map>
#include <sstream>
#include <stdexcept>
#include <string>

namespace wasshost {

class Config {
public:
    enum Type { INT, DOUBLE, BOOL, STRING };
    struct Opt { Type type; std::string value, def, desc; };

    void add(Type t, const std::string& name, const std::string& def, const std::string& desc) { opts_[name] = { t, def, def, desc }; }

    void load(std::istream& is)
    {
        std::string line;
        int lineno = 0;
        while (std::getline(is, line)) {
            ++lineno;
            if (!line.empty() && line.back() == '\r') line.pop_back();
            const size_t b = line.find_first_not_of(" \t");
            if (b == std::string::npos || line[b] == '#') continue;
            const size_t eq = line.find('=', b);
            if (eq == std::string::npos) throw std::runtime_error("config line " + std::to_string(lineno) + ": expected KEY=value");
            std::string key = trim(line.substr(b, eq - b)), val = trim(line.substr(eq + 1));
            auto it = opts_.find(key);
            if (it == opts_.end()) throw std::runtime_error("config line " + std::to_string(lineno) + ": unknown option " + key);
            check(it->second.type, key, val);
            it->second.value = val;
        }
    }

    std::string to_config_string() const
    {
        std::ostringstream os;
        for (const auto& kv : opts_)      // std::map iterates in alphabetical key order
            os << "# " << kv.second.desc << "\n# \n#" << kv.first << "=" << kv.second.def << "\n\n";
        return os.str();
    }

    int get_int(const std::string& k) const { return (int)std::stol(at(k).value); }
    double get_double(const std::string& k) const { return std::stod(at(k).value); }
    bool get_bool(const std::string& k) const { return parse_bool(at(k).value); }
    const std::string& get_string(const std::string& k) const { return at(k).value; }

private:
    std::map<std::string, Opt> opts_;

    const Opt& at(const std::string& k) const
    {
        auto it = opts_.find(k);
        if (it == opts_.end()) throw std::runtime_error("internal: option " + k + " not registered");
        return it->second;
    }
    static std::string trim(const std::string& s)
    {
        const size_t b = s.find_first_not_of(" \t"), e = s.find_last_not_of(" \t");
        return b == std::string::npos ? std::string() : s.substr(b, e - b + 1);
    }
    static std::string lower(std::string s) { std::transform(s.begin(), s.end(), s.begin(), ::tolower); return s; }
    static bool is_bool(const std::string& v)
    {
        const std::string l = lower(v);
        return l == "true" || l == "false" || l == "1" || l == "0" || l == "yes" || l == "no";
    }
    static bool parse_bool(const std::string& v) { const std::string l = lower(v); return l == "true" || l == "1" || l == "yes"; }
    static void check(Type t, const std::string& key, const std::string& v)
    {
        try {
            size_t pos = 0;
            switch (t) {
                case INT: (void)std::stol(v, &pos); if (pos != v.size()) throw 0; break;
                case DOUBLE: (void)std::stod(v, &pos); if (pos != v.size()) throw 0; break;
                case BOOL: if (!is_bool(v)) throw 0; break;
                case STRING: break;
            }
        } catch (...) {
            throw std::runtime_error("invalid value \"" + v + "\" for option " + key);
        }
    }
};

// every INCFG_REQUIRE of wass_stereo; the optical-flow keys are compiled out in the
// reference (#ifdef WASS_ENABLE_OPTFLOW, wass_stereo.cpp:76-84)
inline void register_wass_stereo_options(Config& c)
{
    using T = Config;
    c.add(T::INT, "RANDOM_SEED", "-1", "Random seed for ransac. -1 to use system timer");
    c.add(T::INT, "MIN_TRIANGULATED_POINTS", "100", "Minimum number of triangulated point to proceed with plane estimation");
    c.add(T::DOUBLE, "SAVE_INPUT_SCALE", "0.3", "Save a scaled version of input images (Set 1 to skip or a value <1 to specify scale ratio)");
    c.add(T::DOUBLE, "ZGAP_PERCENTILE", "99", "Z-gap percentile for outlier filtering");
    c.add(T::BOOL, "DISABLE_AUTO_LEFT_RIGHT", "false", "Disable automatic left-right detection");
    c.add(T::BOOL, "SWAP_LEFT_RIGHT", "false", "Swaps left-right images (only valid if DISABLE_AUTO_LEFT_RIGHT is set)");
    c.add(T::BOOL, "SAVE_FULL_MESH", "false", "Save 3D point cloud before plane outlier removal");
    c.add(T::INT, "PLANE_RANSAC_ROUNDS", "400", "number of RANSAC rounds for plane estimation");
    c.add(T::DOUBLE, "PLANE_RANSAC_THRESHOLD", "1", "RANSAC inlier threshold");
    c.add(T::DOUBLE, "PLANE_REFINE_XMIN", "-9999", "Minimum point x-coordinate for plane refinement");
    c.add(T::DOUBLE, "PLANE_REFINE_XMAX", "9999", "Maximum point x-coordinate for plane refinement");
    c.add(T::DOUBLE, "PLANE_REFINE_YMIN", "-9999", "Minimum point y-coordinate for plane refinement");
    c.add(T::DOUBLE, "PLANE_REFINE_YMAX", "9999", "Maximum point y-coordinate for plane refinement");
    c.add(T::DOUBLE, "PLANE_MAX_DISTANCE", "1.5", "Maximum point-plane distance allowed for the reconstructed point-cloud");
    c.add(T::BOOL, "SAVE_AS_PLY", "false", "Save final reconstructed point cloud also in PLY format");
    c.add(T::BOOL, "SAVE_COMPRESSED", "true", "Save in 16-bit compressed format");
    c.add(T::BOOL, "USE_CUSTOM_STEREORECTIFY", "false", "Use built-in stereorectify algorithm instead of the one provided by OpenCV");
    c.add(T::BOOL, "DISABLE_RECTIFY_ROI", "false", "Disable automatic ROI computation during stereo rectification (only enabled if USE_CUSTOM_STEREORECTIFY=true)");
    c.add(T::DOUBLE, "RECTIFY_ANGLE", "0", "Additional rotation to apply around the baseline (only enabled if USE_CUSTOM_STEREORECTIFY=true");
    c.add(T::INT, "MIN_DISPARITY", "1", "Minimum disparity allowed (in px)");
    c.add(T::INT, "MAX_DISPARITY", "640", "Maximum disparity allowed");
    c.add(T::INT, "WINSIZE", "13", "Stereo match window size");
    c.add(T::DOUBLE, "DENSE_SCALE", "1", "Image resize along epipolar lines before dense stereo");
    c.add(T::INT, "DISPARITY_OFFSET", "0", "Offset in pixel to be applied. Positive: move right image to the right. Negative: move right image to the left");
    c.add(T::INT, "DISP_DILATE_STEPS", "1", "Number of dilate steps to be applied to the disparity map");
    c.add(T::INT, "DISP_EROSION_STEPS", "2", "Number of erosion steps to be applied to the disparity map");
    c.add(T::INT, "MEDIAN_FILTER_WSIZE", "0", "Disparity median filter window size (0 to disable)");
    c.add(T::INT, "DENSE_P1_MULT", "2", "SGBM P1 parameter");
    c.add(T::INT, "DENSE_P2_MULT", "64", "SGBM P2 parameter");
    c.add(T::INT, "DENSE_UNIQUENESS_RATIO", "1", "SGBM Uniqueness ratio");
    c.add(T::INT, "DENSE_DISP12MAXDIFF", "-1", "SGBM Disp12MaxDiff");
    c.add(T::INT, "DENSE_PREFILTER_CAP", "60", "SGBM PreFilterCap");
    c.add(T::INT, "DENSE_SPECKLE_RANGE", "16", "SGBM SpeckleRange");
    c.add(T::INT, "DENSE_SPECKLE_WINDOW_SIZE", "-70", "SGBM SpeckleWindowSize");
    c.add(T::INT, "DENSE_DISPARITY_BIGGEST_COMPONENT_THRESHOLD", "0", "Maximum squared gradient magnitude threshold for biggest connected component extraction (0 to disable)");
    c.add(T::DOUBLE, "TRIANG_MIN_ANGLE", "20", "Minimum ray angle for triangulation (in degrees)");
    c.add(T::DOUBLE, "TRIANG_BBOX_TOP", "-1", "Triangulation bounding box top coordinate in px wrt. the left image (-1 to disable)");
    c.add(T::DOUBLE, "TRIANG_BBOX_LEFT", "-1", "Triangulation bounding box left coordinate in px wrt. the left image (-1 to disable)");
    c.add(T::DOUBLE, "TRIANG_BBOX_RIGHT", "-1", "Triangulation bounding box right coordinate in px wrt. the left image (-1 to disable)");
    c.add(T::DOUBLE, "TRIANG_BBOX_BOTTOM", "-1", "Triangulation bounding box bottom coordinate in px wrt. the left image (-1 to disable)");
    c.add(T::STRING, "LEFT_MASK_IMAGE", "none", "Filename of a (BW) left camera mask image. Note: File path is relative to current workdir. Use \"none\" for no mask");
    c.add(T::STRING, "RIGHT_MASK_IMAGE", "none", "Filename of a (BW) right camera mask image. Note: File path is relative to current workdir. Use \"none\" for no mask");
    c.add(T::BOOL, "DISCARD_BURNED_AREAS", "true", "Discard white pixels (value>254)");
    c.add(T::BOOL, "PLANE_WEIGHT_PROPORTIONAL_TO_DISTANCE", "true", "use point to camera distance as weight during LLS plane fitting");
    c.add(T::BOOL, "PLANE_USE_CENTRAL_THIRD_ONLY", "false", "use only the central third of the image to estimate the mean sea plane");
    c.add(T::DOUBLE, "PLANE_REFINEMENT_MAX_DISTANCE", "70", "max point distance for plane refinement");
    // Extension (not in the reference): aggregation path count.  5 = MODE_SGBM, what the reference runs.
    c.add(T::INT, "DENSE_PATHS", "5", "SGBM aggregation paths: 5 = cv::StereoSGBM MODE_SGBM (reference behaviour), 8 = MODE_HH (full DP)");
}

// Which configurations the device-resident chain covers (frame_pipeline.hpp); the others keep the stage-by-stage calls.  Lives here,
// above nothing but the parser, because the per-frame client (wass_stereo_client.cpp) asks the same question without the library.
inline bool pipeline_eligible(const Config& cfg, std::string* why = nullptr)
{
    auto no = [&](const char* w) { if (why) *why = w; return false; };
    if (cfg.get_double("DENSE_SCALE") != 1.0) return no("DENSE_SCALE != 1 (maps of two sizes)");
    if (cfg.get_bool("SAVE_FULL_MESH")) return no("SAVE_FULL_MESH (the mesh before the plane stages goes to the host)");
    if (cfg.get_bool("SAVE_AS_PLY")) return no("SAVE_AS_PLY (the whole mesh goes to the host)");
    if (!cfg.get_bool("SAVE_COMPRESSED")) return no("SAVE_COMPRESSED=false (the whole mesh goes to the host)");
    const int rounds = cfg.get_int("PLANE_RANSAC_ROUNDS");
    if (rounds <= 0 || rounds > 1800) return no("PLANE_RANSAC_ROUNDS outside 1..1800");
    return true;
}

}  // namespace wasshost
