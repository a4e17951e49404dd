// png.cpp — minimal 8-bit RGB PNG encoder on zlib (libpng is not in the image). The reference saves
// through the `image` crate (src/main.rs:86); only the decoded pixels have to agree, not the bytes.
#include <zlib.h>

#include <cstdio>
#include <cstring>
#include <vector>

#include "rbrt.hpp"

namespace rbrt {
namespace {

void put_u32(std::vector<uint8_t>& v, uint32_t x) {
    v.push_back(uint8_t(x >> 24)), v.push_back(uint8_t(x >> 16)), v.push_back(uint8_t(x >> 8)), v.push_back(uint8_t(x));
}

void chunk(std::vector<uint8_t>& out, const char type[4], const uint8_t* data, size_t n) {
    put_u32(out, uint32_t(n));
    size_t start = out.size();
    out.insert(out.end(), type, type + 4);
    if (n) out.insert(out.end(), data, data + n);
    put_u32(out, uint32_t(crc32(0L, out.data() + start, uInt(n + 4))));
}

}  // namespace

void write_png(const std::string& path, const uint8_t* rgb, uint32_t width, uint32_t height) {
    std::vector<uint8_t> raw;
    raw.reserve(size_t(height) * (size_t(width) * 3 + 1));
    for (uint32_t y = 0; y < height; ++y) {
        raw.push_back(0);  // filter: none
        raw.insert(raw.end(), rgb + size_t(y) * width * 3, rgb + size_t(y + 1) * width * 3);
    }
    uLongf clen = compressBound(uLong(raw.size()));
    std::vector<uint8_t> comp(clen);
    if (compress2(comp.data(), &clen, raw.data(), uLong(raw.size()), 6) != Z_OK) throw Error("png: deflate failed");
    std::vector<uint8_t> out = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n'};
    std::vector<uint8_t> ihdr;
    put_u32(ihdr, width), put_u32(ihdr, height);
    ihdr.push_back(8), ihdr.push_back(2), ihdr.push_back(0), ihdr.push_back(0), ihdr.push_back(0);
    chunk(out, "IHDR", ihdr.data(), ihdr.size());
    chunk(out, "IDAT", comp.data(), clen);
    chunk(out, "IEND", nullptr, 0);
    FILE* f = std::fopen(path.c_str(), "wb");
    if (!f) throw Error("Unable to save target img to " + path + "! Maybe the directory does not exist?");
    size_t w = std::fwrite(out.data(), 1, out.size(), f);
    if (std::fclose(f) != 0 || w != out.size())
        throw Error("Unable to save target img to " + path + "! Maybe the directory does not exist?");
}

void ImageBuffer::save(const std::string& path) const {
    auto ends_with = [&](const char* ext) {
        size_t n = std::strlen(ext);
        if (path.size() < n) return false;
        for (size_t i = 0; i < n; ++i)
            if (std::tolower((unsigned char)path[path.size() - n + i]) != ext[i]) return false;
        return true;
    };
    if (ends_with(".ppm")) {
        FILE* f = std::fopen(path.c_str(), "wb");
        if (!f) throw Error("Unable to save target img to " + path + "! Maybe the directory does not exist?");
        std::fprintf(f, "P6\n%u %u\n255\n", width, height);
        std::fwrite(rgb.data(), 1, rgb.size(), f);
        std::fclose(f);
        return;
    }
    if (!ends_with(".png")) throw Error("Unable to save target img to " + path + ": only .png and .ppm are supported");
    write_png(path, rgb.data(), width, height);
}

}  // namespace rbrt
