// jpeg_decode.h — JPEG -> 8-bit RGB texels (see jpeg_decode.cpp).
#pragma once

#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

namespace rt {

// Decodes a complete JPEG file held in memory (baseline, extended sequential or progressive Huffman, 8-bit samples,
// grey / YCbCr / Adobe RGB, chroma subsampled by 1 or 2 per direction) into row-major 8-bit RGB, top row first.
// Returns false and sets `err` for malformed files and for the kinds not listed.
bool decode_jpeg_rgb8(const uint8_t *data, size_t n, uint32_t &width, uint32_t &height, std::vector<uint8_t> &rgb,
                      std::string &err);

} // namespace rt
