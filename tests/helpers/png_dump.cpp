// test helper: decode a PNG with the host library's reader (wass_amd/host/hostio.hpp, read_png_gray = cv::imread(IMREAD_GRAYSCALE) of
// wass_stereo.cpp:393,396) and write width, height and the pixels to stdout.  Built and run by tests/test_hostio_png.py.
#include "../../wass_amd/host/hostio.hpp"

int main(int argc, char** argv)
{
    if (argc != 2) return 2;
    try {
        const wasshost::Image img = wasshost::read_png_gray(argv[1]);
        fwrite(&img.w, 4, 1, stdout);
        fwrite(&img.h, 4, 1, stdout);
        fwrite(img.px.data(), 1, img.px.size(), stdout);
    } catch (const std::exception& e) {
        fprintf(stderr, "%s\n", e.what());
        return 1;
    }
    return 0;
}
