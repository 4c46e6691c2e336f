// Harness of tests/test_host_sanitizers.py for the host pipeline (bgreat_amd/csrc/pipeline.cpp: producer, gatherer,
// stream workers, formatter, writers, worker pool, page-locked ring) under ThreadSanitizer / AddressSanitizer.
// The GPU side of the C-ABI is replaced by the stand-in below (TEST INFRASTRUCTURE, never linked into the product): a
// read "maps" iff its length is odd, with the path [len, -index_in_batch, 7] -- enough to check that every record
// comes out once, in input order, in the right file, with the right bytes.
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

#include "../include/bgreat_gpu.h"
#include "fastx.h"
#include "options.h"

struct bgr_graph { uint32_t k; };
struct bgr_aligner { int device = 0; uint64_t counters[5] = {0, 0, 0, 0, 0}; std::string ps, ns; /* text form: the last call's streams, for bgr_aligner_fetch_text */ };
static thread_local std::string tl_err;
namespace bgr { int set_error(int code, const std::string& msg) { tl_err = msg; return code; } }
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
int bgr_align_batch_packed(bgr_aligner* a, const bgr_params*, const bgr_packed_reads* pk, uint64_t n, int32_t* paths, uint64_t cap,
                           uint64_t* poffs, uint8_t* status) {
    const uint64_t* offs = pk->read_offsets;
    for (uint64_t i = 0; i < n; ++i) {  // the packed plane must spell the read: check the first base of every read's first word
        const uint64_t len = offs[i + 1] - offs[i];
        if (len && (pk->fw3[(offs[i] >> 5) + i] >> 62) > 3) return BGR_E_INTERNAL;
    }
    std::this_thread::sleep_for(std::chrono::microseconds(200 + (n * 7919) % 900));  // let the stages interleave
    uint64_t w = 0;
    poffs[0] = 0;
    for (uint64_t i = 0; i < n; ++i) {
        const uint64_t len = offs[i + 1] - offs[i];
        if (len & 1) {
            if (w + 3 > cap) return BGR_E_CAPACITY;
            paths[w++] = (int32_t)len; paths[w++] = -(int32_t)i; paths[w++] = 7;
            status[i] = BGR_ST_ALIGNED;
            ++a->counters[2];
        } else {
            status[i] = BGR_ST_FAILED;
            ++a->counters[3];
        }
        poffs[i + 1] = w;
        ++a->counters[0];
    }
    return BGR_OK;
}
}

// Stand-in for the text form (bgr_align_fasta_text): the same contract on the CPU -- a piece of the regular shape (header line,
// one sequence line, ... , ends with a newline) is "mapped" with the rule above and comes back as record bytes; any other piece is
// handed back as irregular, untouched.
static uint32_t g_standin_k = 5;
struct bgr_text_stage { int device; std::string copy; };
extern "C" int bgr_text_stage_create(int device, bgr_text_stage** out) { *out = new bgr_text_stage{device, std::string()}; return BGR_OK; }
extern "C" void bgr_text_stage_destroy(bgr_text_stage* s) { delete s; }
extern "C" int bgr_text_stage_device(const bgr_text_stage* s) { return s ? s->device : -1; }
extern "C" int bgr_text_stage_upload(bgr_text_stage* s, const char* text, uint64_t n) { s->copy.assign(text, n); return BGR_OK; }
extern "C" int bgr_text_stage_upload_parts(bgr_text_stage* s, uint32_t n_parts, const char* const* parts, const uint64_t* bytes) {
    s->copy.clear();
    for (uint32_t i = 0; i < n_parts; ++i) s->copy.append(parts[i], bytes[i]);
    return BGR_OK;
}
extern "C" int bgr_aligner_fetch_text(bgr_aligner* a, bgr_text_batch* b) {
    b->paths_bytes = a->ps.size();
    b->notaligned_bytes = a->ns.size();
    if (a->ps.size() > b->paths_cap || a->ns.size() > b->notaligned_cap) return BGR_E_CAPACITY;
    memcpy(b->paths_out, a->ps.data(), a->ps.size());
    memcpy(b->notaligned_out, a->ns.data(), a->ns.size());
    return BGR_OK;
}
extern "C" int bgr_align_fasta_text(bgr_aligner* a, const bgr_params*, bgr_text_batch* b) {
    b->irregular = 0; b->n_records = b->n_accepted = b->paths_bytes = b->notaligned_bytes = 0;
    if (b->stage && b->stage->device != a->device) return bgr::set_error(BGR_E_ARG, "stand-in: the stage lives on another device");  // (as the real call)
    if (b->stage && (b->stage->copy.size() != b->text_bytes || (b->text && b->text_bytes && memcmp(b->stage->copy.data(), b->text, b->text_bytes) != 0))) return BGR_E_INTERNAL;  // the staged piece is this piece
    if (!b->stage && !b->text) return BGR_E_ARG;
    const char* t = b->stage ? b->stage->copy.data() : b->text;
    const uint64_t n = b->text_bytes;
    if (n == 0) return BGR_OK;
    std::vector<std::pair<uint64_t, uint64_t>> lines;  // [begin, end) without the newline
    uint64_t p = 0;
    while (p < n) {
        const char* q = static_cast<const char*>(memchr(t + p, '\n', n - p));
        if (!q) { b->irregular = 1; return BGR_OK; }   // last line without its newline
        lines.push_back({p, (uint64_t)(q - t)});
        p = (uint64_t)(q - t) + 1;
    }
    const size_t per = b->fastq == 1 ? 4 : 2;  // FASTQ pieces: whole four-line records, whatever the lines hold -- or (2) their header and read lines only
    if (b->fastq && lines.size() % per) return BGR_E_INTERNAL;  // (the pipeline cuts them at record starts)
    if (lines.size() % 2) { b->irregular = 1; return BGR_OK; }
    for (size_t i = 0; i < lines.size() && !b->fastq; ++i) {
        const bool gt = lines[i].second > lines[i].first && t[lines[i].first] == '>';
        if ((i % 2 == 0) != gt) { b->irregular = 1; return BGR_OK; }  // headers start with '>', sequence lines do not
    }
    if (lines.size() / per > n / 24 + 1024) { b->n_records = lines.size() / per; b->irregular = 1; return BGR_OK; }  // (more records than the device's table holds: as the real call)
    std::this_thread::sleep_for(std::chrono::microseconds(200 + (n * 31) % 900));
    std::string& ps = a->ps;
    std::string& ns = a->ns;
    ps.clear(); ns.clear();
    uint64_t acc = 0;
    const uint32_t k = g_standin_k;  // (the FASTA cases of this harness run with k = 5; --files: the k given)
    if (b->record_info_out && b->record_info_cap < n / 24 + 1024) return BGR_E_ARG;
    for (size_t i = 0; i < lines.size(); i += per) {
        ++b->n_records;
        const std::string h(t + lines[i].first, t + lines[i].second), r(t + lines[i + 1].first, t + lines[i + 1].second);
        bool ok = r.size() > 2 && (b->fastq || r.size() > k);
        for (char c : r) if (c != 'A' && c != 'C' && c != 'G' && c != 'T' && c != 'N') ok = false;
        if (b->record_info_out) b->record_info_out[i / per] = ok ? 0x80000000u | ((r.size() & 1) ? 0x40000000u : 0u) | (uint32_t)r.size() : 0u;  // kept | mapped | length
        if (!ok) continue;
        if (r.size() & 1) { ps += h + "\n" + std::to_string(r.size()) + ".-" + std::to_string(acc) + ".7.\n"; ++a->counters[2]; }
        else { ns += h + "\n" + r + "\n"; ++a->counters[3]; }
        ++a->counters[0];
        ++acc;
    }
    b->n_accepted = acc;
    b->paths_bytes = ps.size();
    b->notaligned_bytes = ns.size();
    if (!b->want_output) return BGR_OK;
    return bgr_aligner_fetch_text(a, b);  // (BGR_E_CAPACITY when a stream does not fit: the pipeline grows its buffer and fetches again)
}

static std::string slurp(const std::string& p) { std::ifstream in(p, std::ios::binary); std::stringstream ss; ss << in.rdbuf(); return ss.str(); }

// --files <tmp> <k> <fastq 0|1> file [file ...]: the given files (tools/fuzz_host_sanitizers.py: random files with irregular records) through
// bgr_align_all on both routes, several thread counts, batch sizes and device counts; expected bytes straight from the sequential parser
static int check_files(const std::string& tmp, uint32_t k, bool fastq, const std::vector<std::string>& files) {
    g_standin_k = k;
    std::string list, ep, en;
    uint64_t n_reads = 0, aligned = 0;
    for (size_t i = 0; i < files.size(); ++i) {
        list += (i ? "," : "") + files[i];
        const std::string d = slurp(files[i]);
        bgr::ReadSet rs; bgr::parse_reads(d.data(), d.size(), fastq, k, rs);
        n_reads += rs.count();
        for (uint64_t j = 0; j < rs.count(); ++j) {
            const std::string h(rs.headers.data() + rs.header_offs[j], rs.headers.data() + rs.header_offs[j + 1]);
            const std::string r(rs.reads.data() + rs.read_offs[j], rs.reads.data() + rs.read_offs[j + 1]);
            if (r.size() & 1) { ep += h + "\n" + std::to_string(r.size()) + ".#.7.\n"; ++aligned; }
            else en += h + "\n" + r + "\n";
        }
    }
    const std::string pf = tmp + "/p", nf = tmp + "/n";
    for (unsigned threads : {1u, 6u}) {
        for (uint32_t route : {0u, 1u}) {
            for (uint64_t batch : {37ull, 5000ull, 0ull}) {
                if (batch == 37 && n_reads > 30000) continue;   // (thousands of tiny batches: slow under TSan, nothing new)
                bgr_graph g{k};
                bgr_params prm = {BGR_MODE_GREEDY, 2, 2, 0};
                bgr_run_options opt;
                memset(&opt, 0, sizeof(opt));
    opt.struct_size = sizeof(opt);
                opt.n_gpus = (batch == 5000 && threads == 6) ? 8 : 2;
                opt.threads = threads; opt.batch_reads = batch; opt.chunk_bytes = batch == 0 ? 0 : 4096; opt.fastq = fastq; opt.route = route;
                uint64_t tot[5]; double secs;
                const int rc = bgr_align_all(&g, &prm, &opt, list.c_str(), pf.c_str(), nf.c_str(), tot, &secs);
                if (rc != BGR_OK) { printf("FAIL run threads=%u batch=%llu route=%u: rc %d %s\n", threads, (unsigned long long)batch, route, rc, bgr_last_error()); return 1; }
                const std::string gp = slurp(pf), gn = slurp(nf);
                std::string gp2; gp2.reserve(gp.size());
                { std::istringstream ss(gp); std::string line; bool header = true;
                  while (std::getline(ss, line)) {
                      if (!header) { size_t a = line.find('.'), b = line.find('.', a + 1); if (a == std::string::npos || b == std::string::npos) { printf("FAIL path line\n"); return 1; } line = line.substr(0, a + 1) + "#" + line.substr(b); }
                      gp2 += line + "\n"; header = !header; } }
                if (gp2 != ep || gn != en || tot[0] != n_reads || tot[2] != aligned) {
                    printf("FAIL threads=%u batch=%llu route=%u: paths %zu/%zu notAligned %zu/%zu reads %llu/%llu\n", threads, (unsigned long long)batch, route, gp2.size(), ep.size(), gn.size(), en.size(),
                           (unsigned long long)tot[0], (unsigned long long)n_reads);
                    return 1;
                }
            }
        }
    }
    printf("files ok (%llu reads)\n", (unsigned long long)n_reads);
    return 0;
}

int main(int argc, char** argv) {
    if (argc >= 6 && std::string(argv[1]) == "--files") {
        std::vector<std::string> files;
        for (int i = 5; i < argc; ++i) files.push_back(argv[i]);
        return check_files(argv[2], (uint32_t)atoi(argv[3]), atoi(argv[4]) != 0, files);
    }
    if (argc < 3) return 2;
    const std::string gold = argv[1], tmp = argv[2];
    struct Case { const char* file; bool fastq; };
    {   // a FASTQ file long enough for the chunk-parallel part of the reader (25 003 records: two full getReads()
        // batches of 10 000 in parallel, the rest through the sequential tail with its phantom record)
        std::ofstream big(tmp + "/big.fq", std::ios::binary);
        const char* al = "ACGT";
        for (int i = 0; i < 25003; ++i) {
            std::string r;
            for (int j = 0; j < 33 + i % 7; ++j) r += al[(i * 7 + j * 13 + (j * j) % 5) & 3];
            big << "@q" << i << "\n" << r << "\n+\n" << std::string(r.size(), 'I') << "\n";
        }
    }
    const Case cases[] = {{"syn_r150.fa", false}, {"edge_reads.fa", false}, {"long_r150.fq", true}, {"deg_reads.fa", false}, {"big.fq", true}};
    for (const Case& c : cases) {
        for (unsigned threads : {1u, 6u}) {
          for (uint32_t route : {0u, 1u, 2u}) {   // 2 = the host route with the batches' base cap lowered to 3 000 (a chunk group cut into pieces)
            if (route == 2 && c.fastq) continue;
            bgr::find_option("test.bases_cap")->value.store(route == 2 ? 3000 : 0);  // route 0: through the (stand-in) device as text (FASTA: with the fall-back per irregular piece; FASTQ: whole records up to the last getReads() boundary, host tail), 1: the host route
            for (uint64_t batch : {1ull, 37ull, 100000ull}) {
                if (std::string(c.file) == "big.fq" && batch == 1) continue;  // 50 000 one-read batches: slow under TSan, nothing new
                const std::string in = (std::string(c.file) == "big.fq" ? tmp : gold) + "/" + c.file, pf = tmp + "/p", nf = tmp + "/n";
                bgr_graph g{c.fastq ? 31u : 5u};
                bgr_params prm = {BGR_MODE_GREEDY, 2, 2, 0};
                bgr_run_options opt;
                memset(&opt, 0, sizeof(opt));
    opt.struct_size = sizeof(opt);
                opt.n_gpus = (batch == 37 && threads == 6) ? 8 : 2;  // (eight stand-in devices = 16 stream workers, per-device queues: order and bytes as with two)
                opt.threads = threads; opt.batch_reads = batch; opt.chunk_bytes = 700; opt.fastq = c.fastq; opt.route = route == 2 ? 1u : route;
                opt.first_device = (unsigned)((batch + threads) % 3);  // (the staging sets cached by the previous run carry text stages of ITS devices)
                uint64_t tot[5]; double secs;
                const std::string list = in + "," + in;  // two files: the batches of the second must follow the first's
                { const int rc_ = bgr_align_all(&g, &prm, &opt, list.c_str(), pf.c_str(), nf.c_str(), tot, &secs); if (rc_ != BGR_OK) { printf("FAIL run (%s threads=%u batch=%llu route=%u): rc %d %s\n", c.file, threads, (unsigned long long)batch, route, rc_, bgr_last_error()); return 1; } }
                // expected bytes straight from the sequential parser
                const std::string d = slurp(in);
                bgr::ReadSet rs; bgr::parse_reads(d.data(), d.size(), c.fastq, g.k, rs);
                std::string ep, en;
                // the stand-in numbers reads within a device batch: rebuild the batching (FASTQ: exactly `batch` records per
                // batch and file; FASTA: by chunk groups, so only the batch-independent fields are checked there)
                uint64_t aligned = 0;
                for (int rep = 0; rep < 2; ++rep)
                    for (uint64_t i = 0; i < rs.count(); ++i) {
                        const std::string h(rs.headers.data() + rs.header_offs[i], rs.headers.data() + rs.header_offs[i + 1]);
                        const std::string r(rs.reads.data() + rs.read_offs[i], rs.reads.data() + rs.read_offs[i + 1]);
                        if (r.size() & 1) { ep += h + "\n" + std::to_string(r.size()) + ".#.7.\n"; ++aligned; }
                        else en += h + "\n" + r + "\n";
                    }
                std::string gp = slurp(pf), gn = slurp(nf);
                // blank out the per-batch index (second int of every path) before comparing
                std::string gp2; gp2.reserve(gp.size());
                { std::istringstream ss(gp); std::string line; bool header = true;
                  while (std::getline(ss, line)) {
                      if (!header) { size_t a = line.find('.'), b = line.find('.', a + 1); if (a == std::string::npos || b == std::string::npos) { printf("FAIL path line\n"); return 1; } line = line.substr(0, a + 1) + "#" + line.substr(b); }
                      gp2 += line + "\n"; header = !header; } }
                if (gp2 != ep || gn != en || tot[0] != 2 * rs.count() || tot[2] != aligned) {
                    printf("FAIL %s threads=%u batch=%llu route=%u: paths %zu/%zu notAligned %zu/%zu reads %llu/%llu\n", c.file, threads, (unsigned long long)batch, route,
                           gp2.size(), ep.size(), gn.size(), en.size(), (unsigned long long)tot[0], (unsigned long long)(2 * rs.count()));
                    return 1;
                }
            }
          }
        }
        printf("%s ok\n", c.file);
    }
    // ---- split runs (bgr_run_options.split_output): one pipeline per stand-in device over contiguous shares of the input; the pairs
    // concatenated in device order must be the bytes of the single pipeline, whatever the lane count, route and batch size (the stand-in
    // numbers reads within a device batch: blanked as above; FASTA only -- other runs ignore the flag and write the single pair)
    {
        auto blank = [](const std::string& gp) {
            std::string out; out.reserve(gp.size());
            std::istringstream ss(gp); std::string line; bool header = true;
            while (std::getline(ss, line)) {
                if (!header) { size_t a = line.find('.'), b = line.find('.', a + 1); if (a != std::string::npos && b != std::string::npos) line = line.substr(0, a + 1) + "#" + line.substr(b); }
                out += line + "\n"; header = !header;
            }
            return out;
        };
        const std::string big = tmp + "/lanes.fa";
        {
            std::ofstream o(big, std::ios::binary);
            const char* al = "ACGT";
            for (int i = 0; i < 40000; ++i) {
                std::string r;
                for (int j = 0; j < 7 + (i * 3) % 40; ++j) r += al[(i * 11 + j * 7 + (j * j) % 3) & 3];
                if (i % 211 == 7) r[2] = 'n';
                o << ">L" << i << " x\n" << r << "\n";
            }
        }
        for (const std::string& list : {big, gold + "/syn_r150.fa," + big + "," + gold + "/deg_reads.fa", gold + "/edge_reads.fa"}) {
            for (uint32_t route : {0u, 1u}) {
                bgr_graph g{5u};
                bgr_params prm = {BGR_MODE_GREEDY, 2, 2, 0};
                bgr_run_options opt;
                memset(&opt, 0, sizeof(opt));
    opt.struct_size = sizeof(opt);
                opt.n_gpus = 1; opt.threads = 3; opt.batch_reads = 997; opt.chunk_bytes = 900; opt.route = route;
                uint64_t tot1[5]; double secs;
                if (bgr_align_all(&g, &prm, &opt, list.c_str(), (tmp + "/p1").c_str(), (tmp + "/n1").c_str(), tot1, &secs) != BGR_OK) { printf("FAIL lanes reference run: %s\n", bgr_last_error()); return 1; }
                const std::string want_p = blank(slurp(tmp + "/p1")), want_n = slurp(tmp + "/n1");
                for (unsigned lanes : {2u, 3u, 8u}) {
                    opt.n_gpus = lanes; opt.split_output = 1; opt.threads = 6; opt.first_device = lanes == 3 ? 2 : 0; opt.batch_reads = lanes == 8 ? 64 : 997;
                    uint64_t tot[5];
                    if (bgr_align_all(&g, &prm, &opt, list.c_str(), (tmp + "/ps").c_str(), (tmp + "/ns").c_str(), tot, &secs) != BGR_OK) { printf("FAIL lanes run: %s\n", bgr_last_error()); return 1; }
                    std::string gp, gn;
                    for (unsigned d = 0; d < lanes; ++d) { gp += slurp(tmp + "/ps." + std::to_string(d)); gn += slurp(tmp + "/ns." + std::to_string(d)); }
                    if (blank(gp) != want_p || gn != want_n || tot[0] != tot1[0] || tot[2] != tot1[2] || tot[3] != tot1[3]) {
                        printf("FAIL lanes=%u route=%u list=%s: paths %zu/%zu notAligned %zu/%zu reads %llu/%llu\n", lanes, route, list.c_str(), gp.size(), want_p.size(), gn.size(), want_n.size(),
                               (unsigned long long)tot[0], (unsigned long long)tot1[0]);
                        return 1;
                    }
                }
            }
        }
        {   // a lane that fails (unreadable output directory) stops the others and its error is the one reported
            bgr_graph g{5u};
            bgr_params prm = {BGR_MODE_GREEDY, 2, 2, 0};
            bgr_run_options opt;
            memset(&opt, 0, sizeof(opt));
    opt.struct_size = sizeof(opt);
            opt.n_gpus = 4; opt.split_output = 1; opt.threads = 4;
            uint64_t tot[5]; double secs;
            const int rc = bgr_align_all(&g, &prm, &opt, big.c_str(), (tmp + "/no_such_dir/p").c_str(), (tmp + "/no_such_dir/n").c_str(), tot, &secs);
            if (rc != BGR_E_IO) { printf("FAIL lanes error path: rc %d %s\n", rc, bgr_last_error()); return 1; }
        }
        printf("split runs ok\n");
    }
    // ---- what the reference prints while it maps (file names; in exhaustive mode the block after every tenth getReads()
    // call, alignerExhaustive.cpp:306-316): it must come out of the ordered writer between the right reads, whatever the
    // batch and chunk sizes.  Expected text: the sequential parser's iteration count per record, walked in input order.
    {
        const char* al = "ACGT";
        for (int variant = 0; variant < 2; ++variant) {  // FASTA with dropped records, FASTQ
            const bool fq = variant == 1;
            const std::string f1 = tmp + (fq ? "/prog1.fq" : "/prog1.fa"), f2 = tmp + (fq ? "/prog2.fq" : "/prog2.fa");
            for (int fi = 0; fi < 2; ++fi) {
                std::ofstream o(fi ? f2 : f1, std::ios::binary);
                const int n = fi ? 61000 : 137003;   // calls: 14 + 7 (FASTA), so blocks fall after calls 10 and 20 (the 6th of file 2)
                for (int i = 0; i < n; ++i) {
                    std::string r;
                    for (int j = 0; j < 8 + i % 5; ++j) r += al[(i * 5 + j * 11 + (j * j) % 7) & 3];
                    if (i % 97 == 5) r[3] = 'x';     // dropped: still one getReads() iteration
                    if (fq) o << "@q" << i << "\n" << r << "\n+\n" << std::string(r.size(), 'I') << "\n";
                    else if (fi == 1 && i % 1009 == 3) o << ">r" << i << "\n" << r.substr(0, 4) << "\n" << r.substr(4) << "\n";  // a sequence over two lines: such a piece goes back to the host parser
                    else o << ">r" << i << "\n" << r << "\n";
                }
            }
            std::string expect;
            {
                uint32_t reads = 0, alig = 0, fail_ = 0, ov = 0;
                uint64_t calls_before = 0;
                bgr_graph g{5u};
                auto block = [&]() {
                    std::ostringstream b;
                    const uint32_t got = alig + fail_;
                    b << "Read : " << reads << "\nNo Overlap : 0 Percent : " << (100 * float(0)) / reads << "\nGot Overlap : " << got << " Percent : " << (100 * float(got)) / reads
                      << "\nOverlap and Aligned : " << alig << " Percent : " << (100 * float(alig)) / got << "\nOverlap but no aligne: " << fail_ << " Percent : " << (100 * float(fail_)) / got
                      << "\nReads/seconds : X\nOverlap per reads : " << (got ? ov / got : 0u) << "\n\n";
                    expect += b.str();
                };
                for (const std::string& f : {f1, f2}) {
                    expect += f + "\n";
                    const std::string d = slurp(f);
                    bgr::ParsedChunk pc;
                    pc.track_iters = true;
                    if (fq) bgr::parse_fastq_image(d.data(), d.size(), pc); else bgr::parse_fasta_chunk(d.data(), 0, d.size(), g.k, pc);
                    const uint64_t calls = (pc.iters + 9999) / 10000;
                    size_t r = 0;
                    for (uint64_t c = 1; c <= calls; ++c) {  // call c of this file covers iterations [(c-1)*10000, c*10000)
                        for (; r < pc.recs.size() && pc.rec_iter[r] < c * 10000; ++r) {
                            ++reads;
                            if (pc.recs[r].sl & 1) ++alig; else ++fail_;
                            ov += pc.recs[r].sl >= g.k - 1 ? pc.recs[r].sl - (g.k - 1) + 1 : 1;
                        }
                        if ((calls_before + c) % 10 == 0) block();
                    }
                    calls_before += calls;
                }
            }
            for (unsigned threads : {1u, 5u}) {
                for (uint64_t batch : {911ull, 20000ull, 1000000ull}) {
                    bgr_graph g{5u};
                    bgr_params prm = {BGR_MODE_EXHAUSTIVE, 2, 2, 0};
                    bgr_run_options opt;
                    memset(&opt, 0, sizeof(opt));
    opt.struct_size = sizeof(opt);
                    opt.n_gpus = batch == 911ull ? 8 : (threads == 1 ? 1 : 2);  // 16 / 1 / 4 stream workers over the stand-in devices
                    opt.threads = threads; opt.batch_reads = batch; opt.chunk_bytes = 30000; opt.fastq = fq; opt.echo_files = 1;
                    uint64_t tot[5]; double secs;
                    std::ostringstream cap;
                    std::streambuf* old = std::cout.rdbuf(cap.rdbuf());
                    const int rc = bgr_align_all(&g, &prm, &opt, (f1 + "," + f2).c_str(), (tmp + "/p").c_str(), (tmp + "/n").c_str(), tot, &secs);
                    std::cout.rdbuf(old);
                    if (rc != BGR_OK) { printf("FAIL progress run: %s\n", bgr_last_error()); return 1; }
                    std::string got;  // blank the time-dependent line
                    { std::istringstream ss(cap.str()); std::string line;
                      while (std::getline(ss, line)) got += (line.rfind("Reads/seconds : ", 0) == 0 ? std::string("Reads/seconds : X") : line) + "\n"; }
                    if (got != expect) {
                        printf("FAIL progress blocks %s threads=%u batch=%llu\n--- got\n%s--- expected\n%s", fq ? "fastq" : "fasta", threads, (unsigned long long)batch, got.c_str(), expect.c_str());
                        return 1;
                    }
                }
            }
        }
        printf("progress blocks ok\n");
    }
    return 0;
}
