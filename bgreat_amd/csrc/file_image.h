// file_image.h -- a whole input file as one read-only memory image: mmap for regular files, read-until-EOF for
// everything else (FIFOs, /dev/stdin, process substitution), which is what the reference's ifstream/getline handles
// too (aligner.cpp:557-558, :412).  Host only.
#ifndef BGREAT_AMD_FILE_IMAGE_H
#define BGREAT_AMD_FILE_IMAGE_H

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cerrno>
#include <cstdint>
#include <string>
#include <vector>

namespace bgr {

struct FileImage {
    const char* data = "";
    uint64_t size = 0;
    bool mapped = false;
    std::vector<char> owned;  // the bytes of a stream that cannot be mapped

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
                ::close(fd);
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
    ~FileImage() {
        if (mapped) munmap(const_cast<char*>(data), size);
    }
};

}  // namespace bgr
#endif
