// image_loader.h — RGBA8 texture decoding for the headless host layer (see image_loader.cpp).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace prt_host {

struct Image
{
    uint32_t             width = 0, height = 0;
    std::vector<uint8_t> rgba;  // width * height * 4, row-major, top row first
};

// JPEG (baseline / extended / progressive Huffman, 8 bit; grey, YCbCr, RGB, CMYK, YCCK), PNG (all colour types /
// bit depths, tRNS, Adam7), TGA (types 1,2,3,9,10,11), BMP (1/4/8/16/24/32 bit, uncompressed or bit-fields), binary
// PGM/PPM. Returns false for anything else or for a damaged file.
bool load_image(const std::string& path, Image& img);
bool load_image_memory(const std::vector<uint8_t>& file, Image& img);
bool is_jpeg(const std::vector<uint8_t>& file);  // jpeg_loader.cpp
bool load_jpeg(const std::vector<uint8_t>& file, Image& img);

}  // namespace prt_host
