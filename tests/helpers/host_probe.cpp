// test helper: the host library's file readers (wass_amd/host/hostio.hpp) behind a command line, built and run by tests/test_hostio_*.py.
//   host_probe png <file>   read_png_gray (= cv::imread(IMREAD_GRAYSCALE) of wass_stereo.cpp:393,396): width, height, pixels to stdout
//   host_probe repng <in> <out> <level>   read_png_gray, then write_png_gray at that level (0 = stored blocks, what wass_prepare writes)
//   host_probe xml <file>   load_matrix_xml (= cv::FileStorage >> Mat of wass_stereo.cpp:340-386): "rows cols" and the values, %.17g
#include "../../wass_amd/host/hostio.hpp"

int main(int argc, char** argv)
{
    if (argc == 5 && std::string(argv[1]) == "repng") {
        try { return wasshost::write_png_gray(argv[3], wasshost::read_png_gray(argv[2]), atoi(argv[4])) ? 0 : 1; }
        catch (const std::exception& e) { fprintf(stderr, "%s\n", e.what()); return 1; }
    }
    if (argc != 3) return 2;
    try {
        if (std::string(argv[1]) == "png") {
            const wasshost::Image img = wasshost::read_png_gray(argv[2]);
            fwrite(&img.w, 4, 1, stdout);
            fwrite(&img.h, 4, 1, stdout);
            fwrite(img.px.data(), 1, img.px.size(), stdout);
        } else if (std::string(argv[1]) == "xml") {
            const wasshost::Mat m = wasshost::load_matrix_xml(argv[2]);
            printf("%d %d\n", m.rows, m.cols);
            for (double v : m.d) printf("%.17g\n", v);
        } else return 2;
    } catch (const std::exception& e) {
        fprintf(stderr, "%s\n", e.what());
        return 1;
    }
    return 0;
}
