// file_image.h -- a whole input file as one read-only memory image: mmap for regular files, read-until-EOF for
// everything else (FIFOs, /dev/stdin, process substitution), which is what the reference's ifstream/getline handles
// too (aligner.cpp:557-558, :412).  Host only.
#ifndef BGREAT_AMD_FILE_IMAGE_H
#define BGREAT_AMD_FILE_IMAGE_H

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <cerrno>
#include <cstring>
#include <cstdint>
#include <string>
#include <vector>

namespace bgr {

struct FileImage {
    const char* data = "";
    uint64_t size = 0;
    bool mapped = false;
    int fd = -1;              // kept open for mapped regular files: copy_out() reads through it (pread) instead of faulting pages in
    std::vector<char> owned;  // the bytes of a stream that cannot be mapped
    std::atomic<uint32_t> irregular_pieces{0};  // pipeline.cpp: pieces of this file the device's text route handed back to the host parser

    FileImage() = default;
    FileImage(const FileImage&) = delete;
    FileImage& operator=(const FileImage&) = delete;

    bool open(const std::string& path, std::string& err) {
        const int fd = ::open(path.c_str(), O_RDONLY);
        if (fd < 0) { err = "cannot open " + path; return false; }
        struct stat st;
        if (fstat(fd, &st) != 0) { ::close(fd); err = "cannot stat " + path; return false; }
        if (S_ISREG(st.st_mode)) {
            size = (uint64_t)st.st_size;
            if (size == 0) { ::close(fd); return true; }
            void* p = mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
            if (p != MAP_FAILED) {
                this->fd = fd;
                data = static_cast<const char*>(p);
                mapped = true;
                madvise(p, size, MADV_SEQUENTIAL);
                return true;
            }
        }
        // not a regular file (st_size says nothing), or it cannot be mapped: read until read() returns 0
        size = 0;
        owned.clear();
        uint64_t cap = 1u << 20;
        for (;;) {
            if (owned.size() < size + cap) owned.resize(size + cap);
            const ssize_t r = ::read(fd, owned.data() + size, owned.size() - size);
            if (r < 0) {
                if (errno == EINTR) continue;
                ::close(fd);
                err = "read error on " + path;
                return false;
            }
            if (r == 0) break;
            size += (uint64_t)r;
            if (cap < (256u << 20)) cap *= 2;
        }
        ::close(fd);
        owned.resize(size);
        owned.shrink_to_fit();
        data = owned.empty() ? "" : owned.data();
        return true;
    }
    // bytes [off, off + len) into dst: pread for a regular file (the kernel copies out of the page cache, ~30 GB/s per thread, no
    // page faults on the mapping), memcpy otherwise
    bool copy_out(char* dst, uint64_t off, uint64_t len) const {
        if (fd < 0) { memcpy(dst, data + off, len); return true; }
        while (len) {
            const ssize_t r = pread(fd, dst, len, (off_t)off);
            if (r < 0) { if (errno == EINTR) continue; return false; }
            if (r == 0) return false;
            dst += r; off += (uint64_t)r; len -= (uint64_t)r;
        }
        return true;
    }
    ~FileImage() {
        if (mapped) munmap(const_cast<char*>(data), size);
        if (fd >= 0) ::close(fd);
    }
};

}  // namespace bgr
#endif
