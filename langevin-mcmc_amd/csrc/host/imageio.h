// Minimal EXR / PNG / JPEG I/O for the drop-in front end (see imageio.cpp).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace lmc {

struct Image3f {
    int width = 0, height = 0;
    std::vector<float> data;  // RGB interleaved, row-major, row 0 = top scanline
    const float *At(int x, int y) const { return &data[((size_t)y * width + x) * 3]; }
};

float HalfToFloat(uint16_t h);
uint16_t FloatToHalf(float f);
Image3f ReadEXR(const std::string &fn);
void WriteEXRHalf(const std::string &fn, const float *rgb, int W, int H);
Image3f ReadPNG(const std::string &fn, bool *is8bit = nullptr);
Image3f ReadJPEG(const std::string &fn);  // jpeg.cpp: baseline + progressive, libjpeg-compatible reconstruction
Image3f ReadImage(const std::string &fn, bool *is8bit = nullptr);

}  // namespace lmc
