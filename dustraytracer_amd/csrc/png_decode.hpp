// png_decode.hpp -- see png_decode.cpp
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

namespace drt {

struct DecodedImage {
    int width = 0, height = 0, components = 0;
    std::vector<uint8_t> texels;     // width*height*components, row 0 first (no flip, Texture.cu:35-49)
};

bool looks_like_png(const uint8_t *data, size_t size);
DecodedImage decode_png(const uint8_t *data, size_t size);   // throws std::runtime_error

// jpeg_decode.cpp: baseline JPEG, output identical to the reference's decoder (vendored stb_image) byte for byte
bool looks_like_jpeg(const uint8_t *data, size_t size);
DecodedImage decode_jpeg(const uint8_t *data, size_t size);  // throws std::runtime_error

}  // namespace drt
