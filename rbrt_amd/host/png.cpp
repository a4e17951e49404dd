// png.cpp — minimal 8-bit RGB PNG encoder on zlib (libpng is not in the image). The reference saves
// through the `image` crate (src/main.rs:86); only the decoded pixels have to agree, not the bytes.
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
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

// zlib stream of `raw`, deflated by several threads: the data is cut into pieces, every piece is a raw deflate stream of its own
// that ends on a byte boundary without a final block (Z_SYNC_FLUSH; the last piece ends the stream), and the pieces are
// concatenated behind one zlib header and in front of the Adler-32 of the whole (adler32_combine) -- what pigz does. Any
// inflater reads it as one stream. (The 2.4 MB of a 1024x768 image took 55 ms in one thread, more than three times the
// render; a piece restarts with an empty window, which costs a fraction of a percent of the file size.)
std::vector<uint8_t> deflate_parallel(const std::vector<uint8_t>& raw, int level) {
    const size_t kPiece = size_t(128) << 10;
    size_t n_pieces = std::max<size_t>(1, (raw.size() + kPiece - 1) / kPiece);
    unsigned n_threads = std::max(1u, std::min<unsigned>(std::thread::hardware_concurrency(), 16u));
    if (const char* e = std::getenv("RBRT_PNG_THREADS")) n_threads = unsigned(std::max(1, std::atoi(e)));
    n_threads = unsigned(std::min<size_t>(n_threads, n_pieces));
    std::vector<std::vector<uint8_t>> out(n_pieces);
    std::vector<uLong> adler(n_pieces, 1L);
    std::atomic<size_t> next{0};
    std::atomic<bool> failed{false};
    const auto work = [&]() {
        for (size_t k = next.fetch_add(1); k < n_pieces; k = next.fetch_add(1)) {
            const size_t lo = k * kPiece, hi = std::min(raw.size(), lo + kPiece);
            z_stream z;
            std::memset(&z, 0, sizeof(z));
            if (deflateInit2(&z, level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) {
                failed = true;
                return;
            }
            std::vector<uint8_t>& o = out[k];
            o.resize(deflateBound(&z, uLong(hi - lo)) + 16);
            z.next_in = const_cast<Bytef*>(raw.data() + lo), z.avail_in = uInt(hi - lo);
            z.next_out = o.data(), z.avail_out = uInt(o.size());
            const bool last = k + 1 == n_pieces;
            const int rc = deflate(&z, last ? Z_FINISH : Z_SYNC_FLUSH);
            if ((last ? rc != Z_STREAM_END : rc != Z_OK) || z.avail_in != 0) failed = true;
            o.resize(o.size() - z.avail_out);
            deflateEnd(&z);
            adler[k] = adler32(adler32(0L, Z_NULL, 0), raw.data() + lo, uInt(hi - lo));
        }
    };
    std::vector<std::thread> pool;
    for (unsigned i = 1; i < n_threads; ++i) pool.emplace_back(work);
    work();
    for (auto& t : pool) t.join();
    if (failed) throw Error("png: deflate failed");
    std::vector<uint8_t> z = {0x78, 0x9C};  // zlib header: deflate, 32 KiB window, default level, no dictionary
    uLong a = adler32(0L, Z_NULL, 0);
    for (size_t k = 0; k < n_pieces; ++k) {
        z.insert(z.end(), out[k].begin(), out[k].end());
        const size_t lo = k * kPiece, hi = std::min(raw.size(), lo + kPiece);
        a = adler32_combine(a, adler[k], z_off_t(hi - lo));
    }
    put_u32(z, uint32_t(a));
    return z;
}

}  // namespace

void write_png(const std::string& path, const uint8_t* rgb, uint32_t width, uint32_t height) {
    std::vector<uint8_t> raw;
    raw.reserve(size_t(height) * (size_t(width) * 3 + 1));
    for (uint32_t y = 0; y < height; ++y) {
        raw.push_back(0);  // filter: none
        raw.insert(raw.end(), rgb + size_t(y) * width * 3, rgb + size_t(y + 1) * width * 3);
    }
    const std::vector<uint8_t> comp = deflate_parallel(raw, 6);
    const size_t clen = comp.size();
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

namespace {

void write_file(const std::string& path, const std::vector<uint8_t>& bytes) {
    FILE* f = std::fopen(path.c_str(), "wb");
    if (!f) throw Error("Unable to save target img to " + path + "! Maybe the directory does not exist?");
    const size_t w = std::fwrite(bytes.data(), 1, bytes.size(), f);
    if (std::fclose(f) != 0 || w != bytes.size())
        throw Error("Unable to save target img to " + path + "! Maybe the directory does not exist?");
}
void le16(std::vector<uint8_t>& v, uint32_t x) { v.push_back(uint8_t(x)), v.push_back(uint8_t(x >> 8)); }
void le32(std::vector<uint8_t>& v, uint32_t x) { le16(v, x & 0xFFFFu), le16(v, x >> 16); }

// 24-bit uncompressed BMP (BITMAPINFOHEADER, bottom-up, BGR, rows padded to 4 bytes)
std::vector<uint8_t> encode_bmp(const uint8_t* rgb, uint32_t w, uint32_t h) {
    const uint64_t row = (uint64_t(w) * 3u + 3u) & ~uint64_t(3);
    if (54u + row * h > 0xFFFFFFFFull || w > 0x7FFFFFFFu || h > 0x7FFFFFFFu) throw Error("bmp: image too large for the format's 32-bit sizes");
    std::vector<uint8_t> o = {'B', 'M'};
    le32(o, uint32_t(54u + row * h)), le32(o, 0), le32(o, 54);
    le32(o, 40), le32(o, w), le32(o, h), le16(o, 1), le16(o, 24), le32(o, 0), le32(o, uint32_t(row * h)), le32(o, 2835), le32(o, 2835), le32(o, 0), le32(o, 0);
    for (uint32_t y = h; y-- > 0;) {
        for (uint32_t x = 0; x < w; ++x) {
            const uint8_t* p = rgb + (size_t(y) * w + x) * 3;
            o.push_back(p[2]), o.push_back(p[1]), o.push_back(p[0]);
        }
        for (uint64_t k = uint64_t(w) * 3u; k < row; ++k) o.push_back(0);
    }
    return o;
}
// uncompressed true-colour TGA (type 2), top-left origin, BGR
std::vector<uint8_t> encode_tga(const uint8_t* rgb, uint32_t w, uint32_t h) {
    if (w > 65535u || h > 65535u) throw Error("tga: image larger than 65535 pixels on a side");
    std::vector<uint8_t> o = {0, 0, 2, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    le16(o, w), le16(o, h), o.push_back(24), o.push_back(0x20);
    for (size_t i = 0; i < size_t(w) * h; ++i) o.push_back(rgb[i * 3 + 2]), o.push_back(rgb[i * 3 + 1]), o.push_back(rgb[i * 3]);
    return o;
}
// baseline TIFF: one uncompressed RGB strip, little-endian
std::vector<uint8_t> encode_tiff(const uint8_t* rgb, uint32_t w, uint32_t h) {
    const uint64_t n64 = uint64_t(w) * h * 3u;
    if (n64 + 256u > 0xFFFFFFFFull) throw Error("tiff: image too large for the format's 32-bit offsets");
    const uint32_t n = uint32_t(n64);
    std::vector<uint8_t> o = {'I', 'I', 42, 0};
    le32(o, 8u + n + (n & 1u));               // IFD after the pixel data
    o.insert(o.end(), rgb, rgb + n);
    if (n & 1u) o.push_back(0);
    const uint32_t bps_off = uint32_t(o.size()) + 2u + 12u * 10u + 4u;
    const auto entry = [&](uint32_t tag, uint32_t type, uint32_t count, uint32_t value) { le16(o, tag), le16(o, type), le32(o, count), le32(o, value); };
    le16(o, 10);
    entry(256, 4, 1, w), entry(257, 4, 1, h), entry(258, 3, 3, bps_off), entry(259, 3, 1, 1), entry(262, 3, 1, 2), entry(273, 4, 1, 8);
    entry(277, 3, 1, 3), entry(278, 4, 1, h), entry(279, 4, 1, n), entry(284, 3, 1, 1);
    le32(o, 0);
    le16(o, 8), le16(o, 8), le16(o, 8);
    return o;
}
// QOI (qoiformat.org), 3 channels, sRGB
std::vector<uint8_t> encode_qoi(const uint8_t* rgb, uint32_t w, uint32_t h) {
    std::vector<uint8_t> o = {'q', 'o', 'i', 'f'};
    put_u32(o, w), put_u32(o, h), o.push_back(3), o.push_back(0);
    uint8_t index[64][4] = {};
    uint8_t pr = 0, pg = 0, pb = 0;
    uint32_t run = 0;
    const size_t n = size_t(w) * h;
    for (size_t i = 0; i < n; ++i) {
        const uint8_t r = rgb[i * 3], g = rgb[i * 3 + 1], b = rgb[i * 3 + 2];
        if (r == pr && g == pg && b == pb) {
            if (++run == 62 || i + 1 == n) o.push_back(uint8_t(0xC0 | (run - 1))), run = 0;
            continue;
        }
        if (run) o.push_back(uint8_t(0xC0 | (run - 1))), run = 0;
        const uint32_t hsh = (r * 3u + g * 5u + b * 7u + 255u * 11u) % 64u;
        if (index[hsh][3] == 255 && index[hsh][0] == r && index[hsh][1] == g && index[hsh][2] == b) {
            o.push_back(uint8_t(hsh));
        } else {
            index[hsh][0] = r, index[hsh][1] = g, index[hsh][2] = b, index[hsh][3] = 255;
            const int dr = int8_t(r - pr), dg = int8_t(g - pg), db = int8_t(b - pb), dr_dg = dr - dg, db_dg = db - dg;
            if (dr > -3 && dr < 2 && dg > -3 && dg < 2 && db > -3 && db < 2) {
                o.push_back(uint8_t(0x40 | ((dr + 2) << 4) | ((dg + 2) << 2) | (db + 2)));
            } else if (dr_dg > -9 && dr_dg < 8 && dg > -33 && dg < 32 && db_dg > -9 && db_dg < 8) {
                o.push_back(uint8_t(0x80 | (dg + 32))), o.push_back(uint8_t(((dr_dg + 8) << 4) | (db_dg + 8)));
            } else {
                o.push_back(0xFE), o.push_back(r), o.push_back(g), o.push_back(b);
            }
        }
        pr = r, pg = g, pb = b;
    }
    for (int k = 0; k < 7; ++k) o.push_back(0);
    o.push_back(1);
    return o;
}

}  // namespace

// src/main.rs:86 `img_buf.save(target_image_path)`: the `image` crate picks the encoder from the file extension. The
// same here for the formats that hold an 8-bit RGB image LOSSLESSLY (so that the decoded pixels are the reference's):
// png, ppm / pnm (binary P6), pam (P7), bmp, tga, tif / tiff, qoi. The crate's lossy or palette encoders (jpeg, gif,
// webp, avif, ico) are not restated: the error names what is available.
void ImageBuffer::save(const std::string& path) const {
    std::string ext;
    const size_t dot = path.find_last_of('.');
    if (dot != std::string::npos && path.find_first_of("/\\", dot) == std::string::npos)
        for (size_t i = dot + 1; i < path.size(); ++i) ext += char(std::tolower((unsigned char)path[i]));
    if (ext == "png") return write_png(path, rgb.data(), width, height);
    std::vector<uint8_t> bytes;
    if (ext == "ppm" || ext == "pnm") {
        const std::string hd = "P6\n" + std::to_string(width) + " " + std::to_string(height) + "\n255\n";
        bytes.assign(hd.begin(), hd.end());
        bytes.insert(bytes.end(), rgb.begin(), rgb.end());
    } else if (ext == "pam") {
        const std::string hd = "P7\nWIDTH " + std::to_string(width) + "\nHEIGHT " + std::to_string(height) + "\nDEPTH 3\nMAXVAL 255\nTUPLTYPE RGB\nENDHDR\n";
        bytes.assign(hd.begin(), hd.end());
        bytes.insert(bytes.end(), rgb.begin(), rgb.end());
    } else if (ext == "bmp") {
        bytes = encode_bmp(rgb.data(), width, height);
    } else if (ext == "tga") {
        bytes = encode_tga(rgb.data(), width, height);
    } else if (ext == "tif" || ext == "tiff") {
        bytes = encode_tiff(rgb.data(), width, height);
    } else if (ext == "qoi") {
        bytes = encode_qoi(rgb.data(), width, height);
    } else {
        throw Error("Unable to save target img to " + path + ": the extension '" + ext +
                    "' is not one of png, ppm, pnm, pam, bmp, tga, tif, tiff, qoi (the lossless 8-bit RGB formats of the reference's image crate)");
    }
    write_file(path, bytes);
}

}  // namespace rbrt
