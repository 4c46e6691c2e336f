// fastx.cpp -- FASTA/FASTQ reader reproducing what Aligner::getReads (aligner.cpp:46-117) accepts, over a
// memory image of the file instead of an ifstream.  The reference's behaviour is defined by the iostream
// calls it makes (getline / peek / eof in a fixed order, 10000 records per call); `Cursor` models exactly
// the stream state those calls observe (eofbit, failbit, "getline on a failed stream leaves the string
// untouched"), and the two loops below are the same state machines driven through it.
#include "fastx.h"

#include <cstdio>
#include <cstring>

namespace bgr {

namespace {

struct Slice {
    const char* p = "";
    uint64_t n = 0;
};

struct Cursor {
    const char* d;
    uint64_t n, pos = 0;
    bool eofbit = false, failbit = false;
    Cursor(const char* data, uint64_t size) : d(data), n(size) {}
    bool good() const { return !eofbit && !failbit; }
    // std::getline(stream, s): returns false when s is left untouched (stream was not good()).
    bool getline(Slice& s) {
        if (!good()) { failbit = true; return false; }
        if (pos == n) { eofbit = failbit = true; s.p = d + pos; s.n = 0; return true; }  // s erased, nothing extracted
        const char* b = d + pos;
        const char* q = static_cast<const char*>(memchr(b, '\n', n - pos));
        if (q) { s.p = b; s.n = (uint64_t)(q - b); pos = (uint64_t)(q - d) + 1; }
        else { s.p = b; s.n = n - pos; pos = n; eofbit = true; }
        return true;
    }
    int peek() {
        if (!good()) { failbit = true; return -1; }
        if (pos == n) { eofbit = true; return -1; }
        return (unsigned char)d[pos];
    }
};

inline bool valid_chars(const char* p, uint64_t n) {  // aligner.cpp:56-61
    for (uint64_t i = 0; i < n; ++i) {
        char c = p[i];
        if (c != 'A' && c != 'C' && c != 'T' && c != 'G' && c != 'N') return false;
    }
    return true;
}

inline void push(ReadSet& out, const char* h, uint64_t hn, const char* r, uint64_t rn) {
    out.headers.insert(out.headers.end(), h, h + hn);
    out.header_offs.push_back(out.headers.size());
    out.reads.insert(out.reads.end(), r, r + rn);
    out.read_offs.push_back(out.reads.size());
}

const unsigned kBatch = 10000;  // alignerGreedy.cpp:375

}  // namespace

void parse_reads(const char* data, uint64_t size, bool fastq, uint32_t k, ReadSet& out) {
    if (out.read_offs.empty()) out.clear();
    Cursor cur(data, size);
    std::string joined;  // only for multi-line FASTA records
    while (!cur.eofbit) {  // alignerGreedy.cpp:372  while(!readFile.eof())
        // ---- one getReads(multiread, 10000) call: its locals start empty ----------------------
        Slice header, read, inter;
        if (fastq) {
            for (unsigned i = 0; i < kBatch; ++i) {
                cur.getline(header);
                cur.getline(read);  // untouched (== previous record's sequence) once the stream has failed
                if (read.n > 2 && valid_chars(read.p, read.n)) push(out, header.p, header.n, read.p, read.n);
                cur.getline(header);
                cur.getline(header);
                if (cur.eofbit) break;
            }
        } else {
            bool returned = false;
            for (unsigned i = 0; i < kBatch && !returned; ++i) {
                cur.getline(header);
                cur.getline(read);
                bool multi = false;
                for (;;) {
                    int c = cur.peek();
                    const char* rp = multi ? joined.data() : read.p;
                    uint64_t rn = multi ? joined.size() : read.n;
                    if (c == '>') {
                        if (rn > 2 && valid_chars(rp, rn) && rn > k) push(out, header.p, header.n, rp, rn);
                        read = Slice();  // aligner.cpp:91  read=""
                        break;
                    }
                    if (!cur.eofbit) {
                        cur.getline(inter);
                        if (inter.n) {
                            if (!multi) { joined.assign(read.p, read.n); multi = true; }
                            joined.append(inter.p, inter.n);
                        }
                    } else {
                        if (rn > 2 && valid_chars(rp, rn) && rn > k) push(out, header.p, header.n, rp, rn);
                        returned = true;
                        break;
                    }
                }
            }
        }
    }
}

bool parse_reads_file(const std::string& path, bool fastq, uint32_t k, ReadSet& out, std::string& err) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) { err = "cannot open read file " + path; return false; }
    std::vector<char> buf;
    char tmp[1 << 16];
    size_t got;
    while ((got = fread(tmp, 1, sizeof(tmp), f)) > 0) buf.insert(buf.end(), tmp, tmp + got);
    fclose(f);
    parse_reads(buf.data(), buf.size(), fastq, k, out);
    return true;
}

}  // namespace bgr
