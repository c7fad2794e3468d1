// png_decode.h — PNG -> 8-bit RGB texels (see png_decode.cpp).
#pragma once

#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

namespace rt {

// Decodes a complete PNG file held in memory into row-major 8-bit RGB (top row first).  Returns false and
// sets `err` for malformed files and for the formats the reference's loader rejects (16-bit samples).
bool decode_png_rgb8(const uint8_t *data, size_t n, uint32_t &width, uint32_t &height, std::vector<uint8_t> &rgb,
                     std::string &err);

} // namespace rt
