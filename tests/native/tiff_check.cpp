// tiff_check.cpp -- test driver for wass_amd/host/tiff.hpp: decodes a picture to 8-bit grey and dumps "w h\n" + pixels.
//   tiff_check <in.tif|in.png> <out.raw>
#include <cstdio>

#include "../../wass_amd/host/hostio.hpp"
#include "../../wass_amd/host/tiff.hpp"

int main(int argc, char** argv)
{
    if (argc < 3) return 2;
    try {
        const wasshost::Image img = wasshost::read_image_gray(argv[1]);
        FILE* f = fopen(argv[2], "wb");
        if (!f) return 3;
        fprintf(f, "%d %d\n", img.w, img.h);
        fwrite(img.px.data(), 1, img.px.size(), f);
        fclose(f);
    } catch (const std::exception& e) {
        fprintf(stderr, "%s\n", e.what());
        return 1;
    }
    return 0;
}
