// Memory-mapped binary PPM output image, the sink of
// /root/reference/src/lib.rs:64-89 on top of /root/reference/src/mmap.rs:41-122.
//
// The file is opened read+write, created if missing and *never truncated*, then
// sized to header + width*height*3 and mapped MAP_SHARED.  Whatever pixel bytes
// a previous run left are therefore still there: that is the reference's
// checkpoint for `continue(n)` and what makes the image viewable while it is
// being rendered.  Only the header is rewritten on open.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cerrno>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>
#include <new>
#include <string>

#include "rt_host.h"

struct RthImage {
    uint8_t* base = nullptr;
    size_t len = 0;
    size_t header_len = 0;
    uint32_t width = 0, height = 0;
};

static void set_err(char* err, size_t err_len, const std::string& msg) {
    if (err && err_len) std::snprintf(err, err_len, "%s", msg.c_str());
}

// Rows [0, h) of a tile copied by `copy_row`; tiles of tens of megabytes and more are split over a
// few threads (one thread moves about 5-10 GB/s through the page cache, a 16384 x 16384 tile is 805 MB).
template <class CopyRow>
static void copy_rows(uint32_t h, size_t row_bytes, CopyRow copy_row) {
    const size_t total = static_cast<size_t>(h) * row_bytes;
    unsigned n = total >= (size_t{32} << 20) ? 4u : 1u;
    std::vector<std::thread> workers;
    try {
        for (unsigned t = 1; t < n; ++t)
            workers.emplace_back([=] {
                for (uint32_t r = static_cast<uint32_t>(static_cast<uint64_t>(h) * t / n); r < static_cast<uint64_t>(h) * (t + 1) / n; ++r) copy_row(r);
            });
    } catch (...) {
        n = static_cast<unsigned>(workers.size()) + 1;  // threads could not be had: this one does what is left
        for (std::thread& w : workers) w.join();
        workers.clear();
        for (uint32_t r = 0; r < h; ++r) copy_row(r);  // (copies are idempotent)
        return;
    }
    for (uint32_t r = 0; r < static_cast<uint64_t>(h) / n; ++r) copy_row(r);
    for (std::thread& w : workers) w.join();
}

extern "C" {

RthImage* rth_image_open(const char* path, uint32_t width, uint32_t height, char* err,
                         size_t err_len) {
    char header[64];
    const int header_len = std::snprintf(header, sizeof header, "P6\n%u %u 255\n", width, height);
    const size_t len = static_cast<size_t>(header_len) + static_cast<size_t>(width) * height * 3;

    const int fd = ::open(path, O_RDWR | O_CREAT, 0644);
    if (fd < 0) {
        set_err(err, err_len, std::string("open ") + path + ": " + std::strerror(errno));
        return nullptr;
    }
    if (::ftruncate(fd, static_cast<off_t>(len)) != 0) {
        set_err(err, err_len, std::string("set_len ") + path + ": " + std::strerror(errno));
        ::close(fd);
        return nullptr;
    }
    void* p = ::mmap(nullptr, len, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    ::close(fd);  // the mapping keeps the file alive, as in the reference
    if (p == MAP_FAILED) {
        set_err(err, err_len, std::string("mmap ") + path + ": " + std::strerror(errno));
        return nullptr;
    }
    RthImage* img = new (std::nothrow) RthImage;
    if (!img) {
        ::munmap(p, len);
        set_err(err, err_len, "out of memory");
        return nullptr;
    }
    img->base = static_cast<uint8_t*>(p);
    img->len = len;
    img->header_len = static_cast<size_t>(header_len);
    img->width = width;
    img->height = height;
    std::memcpy(img->base, header, img->header_len);
    return img;
}

uint8_t* rth_image_pixels(RthImage* image) { return image ? image->base + image->header_len : nullptr; }

size_t rth_image_header_len(const RthImage* image) { return image ? image->header_len : 0; }

// Rows of a tile <-> a packed tile buffer.  A renderer that works on one tile
// at a time (images larger than a GPU pass, or tiles dealt out to several GPUs)
// reads the stored bytes of its tile for the running average of
// /root/reference/src/lib.rs:286-292 and writes the new bytes back; the
// reference does the same through its per-chunk slices of the mapping
// (/root/reference/src/lib.rs:201-204,248-295).
static bool tile_inside(const RthImage* image, uint32_t x, uint32_t y, uint32_t w, uint32_t h) {
    return image && w && h && static_cast<uint64_t>(x) + w <= image->width && static_cast<uint64_t>(y) + h <= image->height;
}

int rth_image_read_tile(const RthImage* image, uint32_t x, uint32_t y, uint32_t w, uint32_t h, uint8_t* tile) {
    if (!tile || !tile_inside(image, x, y, w, h)) return -1;
    const uint8_t* px = image->base + image->header_len;
    const size_t row = static_cast<size_t>(w) * 3, width = image->width;
    copy_rows(h, row, [=](uint32_t r) { std::memcpy(tile + r * row, px + (static_cast<size_t>(y + r) * width + x) * 3, row); });
    return 0;
}

int rth_image_write_tile(RthImage* image, uint32_t x, uint32_t y, uint32_t w, uint32_t h, const uint8_t* tile) {
    if (!tile || !tile_inside(image, x, y, w, h)) return -1;
    uint8_t* px = image->base + image->header_len;
    const size_t row = static_cast<size_t>(w) * 3, width = image->width;
    copy_rows(h, row, [=](uint32_t r) { std::memcpy(px + (static_cast<size_t>(y + r) * width + x) * 3, tile + r * row, row); });
    return 0;
}

void rth_image_flush(RthImage* image) {
    if (image) ::msync(image->base, image->len, MS_ASYNC);
}

void rth_image_close(RthImage* image) {
    if (!image) return;
    ::munmap(image->base, image->len);
    delete image;
}
}
