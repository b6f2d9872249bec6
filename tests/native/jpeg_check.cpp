// jpeg_check.cpp -- test driver for wass_amd/host/jpeg.hpp: reads a raw picture, writes it as JPEG.
//   jpeg_check <raw file> <width> <height> <channels> <out.jpg> [quality]
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../wass_amd/host/jpeg.hpp"

int main(int argc, char** argv)
{
    if (argc < 6) return 2;
    const int w = atoi(argv[2]), h = atoi(argv[3]), ch = atoi(argv[4]);
    std::vector<uint8_t> px((size_t)w * h * ch);
    FILE* f = fopen(argv[1], "rb");
    if (!f || fread(px.data(), 1, px.size(), f) != px.size()) return 3;
    fclose(f);
    return wasshost::write_jpeg_raw(argv[5], w, h, ch, px.data(), argc > 6 ? atoi(argv[6]) : 95) ? 0 : 1;
}
