// tools/pipeline_bench.cpp -- host-stage throughput of bgr_align_all (bgreat_amd/csrc/pipeline.cpp) with stand-in devices.
//
// What it answers: does the HOST side of `bgreat --gpus N` keep N devices fed?  The GPU side of the C-ABI is replaced by a stand-in
// that costs the host (almost) nothing -- a text call waits as long as the piece would take to cross PCIe (bytes / 40 GB/s, the
// measured rate of one device's link) and hands back record streams of the real size (51 bytes per read, cut out of the
// piece) -- so the Mreads/s printed are those of the producer, gatherer (pread into the staging buffers), ordered writer and write()
// stages alone.  Two forms per device count: ONE ordered pipeline into one pair of files (the reference's format; one producer, one
// gatherer, one writer per file) and the split run (bgr_run_options.split_output: a pipeline per device, N pairs).  The split run's
// pairs concatenated in device order must equal the single pipeline's bytes (checked with a 64-bit hash of both).
//
//   pipeline_bench <work dir> [reads = 20000000] [threads = 16] [max devices = 8]
//
// TEST/BENCH INFRASTRUCTURE: never linked into the product.  Built by tools/Makefile with plain g++ (no sanitizer).
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../include/bgreat_gpu.h"

struct bgr_graph { uint32_t k; };
struct bgr_aligner { int device = 0; uint64_t counters[5] = {0, 0, 0, 0, 0}; uint64_t pb = 0, nb = 0; const char* text = nullptr; uint64_t n = 0; };
struct bgr_text_stage { int device; const char* text; uint64_t bytes; };
static thread_local std::string tl_err;
namespace bgr { int set_error(int code, const std::string& msg) { tl_err = msg; return code; } }
static double g_link_gbs = 40.0;

extern "C" {
const char* bgr_last_error(void) { return tl_err.c_str(); }
int bgr_graph_info(const bgr_graph* g, bgr_graph_info_t* o) { memset(o, 0, sizeof(*o)); o->k = g->k; return BGR_OK; }
int bgr_graph_unitigs(const bgr_graph*, const char**, const uint64_t**, uint64_t*) { return BGR_E_ARG; }
int bgr_graph_upload(bgr_graph*, int) { return BGR_OK; }
int bgr_aligner_create(bgr_graph*, int device, bgr_aligner** out) { *out = new bgr_aligner(); (*out)->device = device; return BGR_OK; }
void bgr_aligner_destroy(bgr_aligner* a) { delete a; }
int bgr_device_local_cpus(int, char*, uint64_t) { return BGR_E_IO; }
int bgr_host_alloc(uint64_t bytes, void** out) { *out = malloc(bytes ? bytes : 1); return *out ? BGR_OK : BGR_E_HIP; }
int bgr_host_free(void* p) { free(p); return BGR_OK; }
int bgr_aligner_counters(bgr_aligner* a, uint64_t out[5]) { memcpy(out, a->counters, sizeof(a->counters)); return BGR_OK; }
int bgr_aligner_fetch(bgr_aligner*, uint64_t, int32_t*, uint64_t, uint64_t*, uint8_t*) { return BGR_E_INTERNAL; }
int bgr_align_batch_packed(bgr_aligner*, const bgr_params*, const bgr_packed_reads*, uint64_t, int32_t*, uint64_t, uint64_t*, uint8_t*) {
    return bgr::set_error(BGR_E_INTERNAL, "pipeline_bench: the host route is not part of this harness");
}
int bgr_text_stage_create(int device, bgr_text_stage** out) { *out = new bgr_text_stage{device, nullptr, 0}; return BGR_OK; }
void bgr_text_stage_destroy(bgr_text_stage* s) { delete s; }
int bgr_text_stage_device(const bgr_text_stage* s) { return s ? s->device : -1; }
int bgr_text_stage_upload_parts(bgr_text_stage*, uint32_t, const char* const*, const uint64_t*) { return bgr::set_error(BGR_E_INTERNAL, "pipeline_bench: FASTQ is not part of this harness"); }
int bgr_text_stage_upload(bgr_text_stage* s, const char* text, uint64_t n) { s->text = text; s->bytes = n; return BGR_OK; }  // (the copy engine's work: no host CPU)
// Every record is 165 bytes and its header spells its number in the file, so the streams can be defined per record whatever piece it
// travels in: records whose number is a multiple of 6 go to the paths stream whole, those with number % 7 == 3 to the other one --
// 27.5 + 23.6 bytes per read as in a real run, one memcpy per selected record (in the product these bytes arrive by DMA).
static uint64_t first_number(const char* t) { return strtoull(t + 2, nullptr, 10); }   // ">r00000001234\n..."
static uint64_t count_res(uint64_t first, uint64_t n, uint64_t mod, uint64_t res) {     // numbers in [first, first + n) that are = res (mod)
    auto upto = [&](uint64_t x) { return x / mod + (x % mod > res ? 1 : 0); };           // in [0, x)
    return upto(first + n) - upto(first);
}
int bgr_aligner_fetch_text(bgr_aligner* a, bgr_text_batch* b) {
    b->paths_bytes = a->pb;
    b->notaligned_bytes = a->nb;
    if (a->pb > b->paths_cap || a->nb > b->notaligned_cap) return BGR_E_CAPACITY;
    const uint64_t recs = a->n / 165, first = first_number(a->text);
    char* po = b->paths_out;
    char* no = b->notaligned_out;
    for (uint64_t i = (6 - first % 6) % 6; i < recs; i += 6) { memcpy(po, a->text + 165 * i, 165); po += 165; }
    for (uint64_t i = (3 + 7 - first % 7) % 7; i < recs; i += 7) { memcpy(no, a->text + 165 * i, 165); no += 165; }
    return (uint64_t)(po - b->paths_out) == a->pb && (uint64_t)(no - b->notaligned_out) == a->nb ? BGR_OK : bgr::set_error(BGR_E_INTERNAL, "stand-in: stream sizes");
}
int bgr_align_fasta_text(bgr_aligner* a, const bgr_params*, bgr_text_batch* b) {
    b->irregular = 0; b->n_records = b->n_accepted = b->paths_bytes = b->notaligned_bytes = 0;
    if (b->stage && b->stage->device != a->device) return bgr::set_error(BGR_E_ARG, "stand-in: the stage lives on another device");
    const uint64_t n = b->text_bytes;
    if (n == 0) return BGR_OK;
    if (n % 165) return bgr::set_error(BGR_E_INTERNAL, "stand-in: a piece of whole 165-byte records expected");
    // the piece crosses the device's link, the kernels run under the next piece's copy
    std::this_thread::sleep_for(std::chrono::nanoseconds((uint64_t)((double)n / g_link_gbs)));
    a->text = b->text; a->n = n;
    const uint64_t recs = n / 165, first = first_number(b->text);
    a->pb = 165 * count_res(first, recs, 6, 0); a->nb = 165 * count_res(first, recs, 7, 3);
    b->n_records = b->n_accepted = recs;
    a->counters[0] += recs; a->counters[2] += recs * 9 / 10; a->counters[3] += recs - recs * 9 / 10;
    if (!b->want_output) return BGR_OK;
    return bgr_aligner_fetch_text(a, b);
}
}

static uint64_t hash_files(const std::vector<std::string>& files) {  // FNV-1a over every eighth byte of the concatenation (the check must not take longer than the runs)
    uint64_t h = 1469598103934665603ull, pos = 0;
    std::vector<unsigned char> buf(8 << 20);
    for (const std::string& f : files) {
        FILE* fp = fopen(f.c_str(), "rb");
        if (!fp) return 0;
        size_t got;
        while ((got = fread(buf.data(), 1, buf.size(), fp)) > 0) {
            for (size_t i = (size_t)((8 - pos % 8) % 8); i < got; i += 8) { h ^= buf[i]; h *= 1099511628211ull; }
            pos += got;
        }
        fclose(fp);
    }
    return h ^ pos;
}
static uint64_t size_of(const std::vector<std::string>& files) {
    uint64_t t = 0;
    struct stat st;
    for (const std::string& f : files) if (stat(f.c_str(), &st) == 0) t += (uint64_t)st.st_size;
    return t;
}

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: pipeline_bench <work dir> [reads] [threads] [max devices]\n"); return 2; }
    const std::string dir = argv[1];
    const uint64_t reads = argc > 2 ? strtoull(argv[2], nullptr, 10) : 20000000ull;
    const unsigned threads = argc > 3 ? (unsigned)atoi(argv[3]) : 16u;
    const unsigned max_dev = argc > 4 ? (unsigned)atoi(argv[4]) : 8u;
    if (const char* e = getenv("PIPELINE_BENCH_LINK_GBS")) g_link_gbs = atof(e);
    const std::string in = dir + "/reads.fa";
    {   // 150 bp reads, 165 bytes per record, written once (page cache)
        FILE* f = fopen(in.c_str(), "wb");
        if (!f) { fprintf(stderr, "cannot write %s\n", in.c_str()); return 2; }
        std::string block;
        uint64_t x = 88172645463325252ull;
        char hd[32];
        for (uint64_t i = 0; i < reads; ++i) {
            const int hl = snprintf(hd, sizeof(hd), ">r%011llu\n", (unsigned long long)i);  // fixed width: every record 14 + 150 + 1 = 165 bytes
            block.append(hd, hl);
            for (int j = 0; j < 150; ++j) {
                if (j % 32 == 0) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; }
                block.push_back("ACGT"[(x >> (2 * (j % 32))) & 3]);
            }
            block.push_back('\n');
            if (block.size() > (8u << 20)) { fwrite(block.data(), 1, block.size(), f); block.clear(); }
        }
        fwrite(block.data(), 1, block.size(), f);
        fclose(f);
    }
    bgr_graph g{31u};
    bgr_params prm = {BGR_MODE_GREEDY, 2, 2, 0};
    printf("host stages of bgr_align_all, stand-in devices (link %.0f GB/s each), %llu reads x 150 bp, %u host threads\n", g_link_gbs, (unsigned long long)reads, threads);
    printf("%8s %28s %28s %10s\n", "devices", "one ordered pair (Mreads/s)", "split run, N pairs (Mreads/s)", "bytes");
    uint64_t h_ref_p = 0, h_ref_n = 0;
    double split1 = 0;
    int bad = 0;
    for (unsigned n = 1; n <= max_dev; n *= 2) {
        double rate[2] = {0, 0};
        const char* verdict = "";
        for (int split = 0; split < 2; ++split) {
            bgr_run_options opt;
            memset(&opt, 0, sizeof(opt));
    opt.struct_size = sizeof(opt);
            opt.n_gpus = n; opt.threads = threads; opt.split_output = (uint32_t)split; opt.numa = 1;
            const std::string pf = dir + "/paths", nf = dir + "/notAligned.fa";
            uint64_t tot[5]; double secs = 0, best = 0;
            for (int rep = 0; rep < 3; ++rep) {  // (later runs have the staging sets of the first)
                unlink(pf.c_str()); unlink(nf.c_str());
                for (unsigned d = 0; d < n; ++d) { unlink((pf + "." + std::to_string(d)).c_str()); unlink((nf + "." + std::to_string(d)).c_str()); }
                sync();  // not timed: dirty pages of the input file / the previous run's outputs would throttle this run's writes
                const int rc = bgr_align_all(&g, &prm, &opt, in.c_str(), pf.c_str(), nf.c_str(), tot, &secs);
                if (rc != BGR_OK) { fprintf(stderr, "run failed: %s\n", bgr_last_error()); return 1; }
                best = std::max(best, (double)reads / secs / 1e6);
            }
            rate[split] = best;
            std::vector<std::string> ps, ns;
            if (split && n > 1) for (unsigned d = 0; d < n; ++d) { ps.push_back(pf + "." + std::to_string(d)); ns.push_back(nf + "." + std::to_string(d)); }
            else { ps.push_back(pf); ns.push_back(nf); }
            const uint64_t hp = hash_files(ps), hn = hash_files(ns);
            if (n == 1 && !split) { h_ref_p = hp; h_ref_n = hn; }
            if (hp != h_ref_p || hn != h_ref_n || size_of(ps) == 0) { verdict = "DIFFER"; ++bad; }
            else if (!*verdict) verdict = "identical";
            if (!getenv("PIPELINE_BENCH_KEEP")) for (const std::string& f : ps) unlink(f.c_str());
            if (!getenv("PIPELINE_BENCH_KEEP")) for (const std::string& f : ns) unlink(f.c_str());
        }
        if (n == 1) split1 = rate[1];
        printf("%8u %28.1f %21.1f (x%.2f) %10s\n", n, rate[0], rate[1], split1 > 0 ? rate[1] / split1 : 0.0, verdict);
        fflush(stdout);
    }
    bgr_host_cache_release();
    unlink(in.c_str());
    return bad ? 1 : 0;
}
