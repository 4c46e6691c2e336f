// main.cpp -- `bgreat`: command-line twin of the reference driver (bgreat.cpp:54-130 + Aligner::alignAll,
// aligner.cpp:550-597) on top of the C-ABI of include/bgreat_gpu.h.  Same flags, same files, same stdout
// lines; the per-read work runs on the GPU(s).  Host code only -- no algorithm here.
//
//   bgreat -r reads.fa[,more.fa] -k 31 -g unitigs.fa -m 2 -t 8 [-e effort] [-f paths] [-a notAligned.fa] [-q] [-b] [-i]
//   extensions (opt-in, absent from the reference): --gpus N  (shard each batch over N devices, input order kept)
//                                                   --batch N (reads per device batch, default 4M)
//                                                   --write-exhaustive (-b normally writes nothing)
#include <getopt.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <string>
#include <thread>
#include <vector>

#include "../../include/bgreat_gpu.h"

static void die(const char* what) {
    fprintf(stderr, "bgreat: %s: %s\n", what, bgr_last_error());
    exit(2);
}

int main(int argc, char** argv) {
    std::string reads, unitigs("unitig.fa"), pathFile("paths"), notAlignedFile("notAligned.fa");
    int errors = 2, threads = 1, ka = 30, effort = 2, gpus = 1;  // bgreat.cpp:56-66 defaults (k is 30, not 31)
    long batch = 4 << 20;
    bool brute = false, incomplete = false, fastq = false, correction = false, dog = false, write_exh = false;
    static option longopts[] = {{"gpus", required_argument, nullptr, 1000}, {"batch", required_argument, nullptr, 1001},
                                {"write-exhaustive", no_argument, nullptr, 1002}, {nullptr, 0, nullptr, 0}};
    int c;
    while ((c = getopt_long(argc, argv, "r:k:g:m:t:e:f:o:a:biqpcG", longopts, nullptr)) != -1) {  // bgreat.cpp:67
        switch (c) {
            case 'r': reads = optarg; break;
            case 'k': ka = std::stoi(optarg); break;
            case 'g': unitigs = optarg; break;
            case 'm': errors = std::stoi(optarg); break;
            case 't': threads = std::stoi(optarg); break;
            case 'e': effort = std::stoi(optarg); break;
            case 'f': pathFile = optarg; break;
            case 'a': notAlignedFile = optarg; break;
            case 'b': brute = true; break;
            case 'i': incomplete = true; break;
            case 'q': fastq = true; break;
            case 'G': dog = true; break;
            case 'c': correction = true; break;
            case 1000: gpus = std::stoi(optarg); break;
            case 1001: batch = std::stol(optarg); break;
            case 1002: write_exh = true; break;
            default: break;  // -o and -p are accepted and ignored, as in the reference (no `case`)
        }
    }
    (void)threads;
    if (reads.empty()) {  // bgreat.cpp:117-127
        std::cout << "-r read_file" << std::endl << "-k k_value (30)" << std::endl << "-g unitig_file (unitig.dot)" << std::endl
                  << "-m n_missmatch (2)" << std::endl << "-t n_thread (1)" << std::endl << "-e effort put in mapping (2)" << std::endl
                  << "-f path_file (paths)" << std::endl << "-a not_aligned_file (notAligned.fa)" << std::endl
                  << "-p to align on paths instead of walks" << std::endl << "-q for fastq read file" << std::endl
                  << "-c to output corrected reads" << std::endl;
        return 0;
    }
    if (correction || dog) {
        fprintf(stderr, "bgreat: -c (correction) and -G (anchors mode) are outside the GPU mapping path and not implemented\n");
        return 2;
    }
    if (gpus < 1 || batch < 1) { fprintf(stderr, "bgreat: --gpus and --batch must be positive\n"); return 2; }

    FILE* pathF = fopen(pathFile.c_str(), "wb");              // aligner.h:85
    FILE* notMappedF = fopen(notAlignedFile.c_str(), "wb");  // aligner.h:86
    if (!pathF || !notMappedF) { fprintf(stderr, "bgreat: cannot open the output files\n"); return 2; }

    auto t0 = std::chrono::system_clock::now();
    bgr_graph* graph = nullptr;
    if (bgr_graph_build_from_fasta(unitigs.c_str(), (uint32_t)ka, 0.0, &graph) != BGR_OK) die("index");
    std::vector<bgr_aligner*> al((size_t)gpus, nullptr);
    for (int g = 0; g < gpus; ++g)
        if (bgr_aligner_create(graph, g, &al[(size_t)g]) != BGR_OK) die("device setup");
    auto t1 = std::chrono::system_clock::now();
    std::cout << "Indexing in seconds : " << std::chrono::duration_cast<std::chrono::seconds>(t1 - t0).count() << std::endl;  // aligner.cpp:546

    bgr_params prm = {brute ? (uint32_t)BGR_MODE_EXHAUSTIVE : (uint32_t)BGR_MODE_GREEDY, (uint32_t)errors, (uint32_t)effort, incomplete ? 1u : 0u};
    const bool writes = !brute || write_exh;
    auto start = std::chrono::system_clock::now();
    std::vector<int32_t> paths;
    std::vector<uint64_t> poffs;
    std::vector<uint8_t> status;
    size_t last = 0;
    for (size_t i = 0; i <= reads.size(); ++i) {  // aligner.cpp:552-586: comma-separated list
        if (i != reads.size() && reads[i] != ',') continue;
        std::string file = reads.substr(last, i - last);
        last = i + 1;
        std::cout << file << std::endl;
        bgr_readset* rs = nullptr;
        if (bgr_readset_load(file.c_str(), fastq ? 1 : 0, (uint32_t)ka, &rs) != BGR_OK) die("read file");
        const uint64_t n = bgr_readset_count(rs);
        const char *rd, *hd;
        const uint64_t *ro, *ho;
        bgr_readset_view(rs, &rd, &ro, &hd, &ho);
        for (uint64_t b0 = 0; b0 < n; b0 += (uint64_t)batch * (uint64_t)gpus) {
            const uint64_t bn = std::min<uint64_t>(n - b0, (uint64_t)batch * (uint64_t)gpus);
            // contiguous, input-ordered shard per device; concatenating the shards in device order is the -t 1 byte stream
            std::vector<uint64_t> s0((size_t)gpus + 1);
            for (int g = 0; g <= gpus; ++g) s0[(size_t)g] = b0 + bn * (uint64_t)g / (uint64_t)gpus;
            poffs.resize(bn + (size_t)gpus);
            status.resize(bn);
            const uint64_t cap = (ro[b0 + bn] - ro[b0]) + 8 * bn + 8;
            paths.resize(cap);
            std::vector<uint64_t> pbase((size_t)gpus + 1, 0), obase((size_t)gpus + 1, 0);
            for (int g = 0; g < gpus; ++g) {  // disjoint slices of the output buffers per device
                pbase[(size_t)g + 1] = pbase[(size_t)g] + (ro[s0[(size_t)g + 1]] - ro[s0[(size_t)g]]) + 8 * (s0[(size_t)g + 1] - s0[(size_t)g]) + 1;
                obase[(size_t)g + 1] = obase[(size_t)g] + (s0[(size_t)g + 1] - s0[(size_t)g]) + 1;
            }
            paths.resize(pbase[(size_t)gpus] + 8);
            std::vector<int> rcs((size_t)gpus, 0);
            std::vector<std::string> errs((size_t)gpus);
            std::vector<std::thread> ts;
            for (int g = 0; g < gpus; ++g) {
                ts.emplace_back([&, g]() {
                    const uint64_t a0 = s0[(size_t)g], an = s0[(size_t)g + 1] - a0;
                    rcs[(size_t)g] = bgr_align_batch(al[(size_t)g], &prm, rd, ro + a0, an, paths.data() + pbase[(size_t)g],
                                                     pbase[(size_t)g + 1] - pbase[(size_t)g], poffs.data() + obase[(size_t)g], status.data() + (a0 - b0));
                    if (rcs[(size_t)g] != BGR_OK) errs[(size_t)g] = bgr_last_error();
                });
            }
            for (auto& t : ts) t.join();
            for (int g = 0; g < gpus; ++g)
                if (rcs[(size_t)g] != BGR_OK) { fprintf(stderr, "bgreat: mapping on device %d: %s\n", g, errs[(size_t)g].c_str()); return 2; }
            if (writes) {
                for (int g = 0; g < gpus; ++g) {
                    const uint64_t a0 = s0[(size_t)g], an = s0[(size_t)g + 1] - a0;
                    if (bgr_write_records(pathF, notMappedF, an, hd, ho + a0, rd, ro + a0, paths.data() + pbase[(size_t)g], poffs.data() + obase[(size_t)g]) != BGR_OK)
                        die("output");
                }
            }
        }
        bgr_readset_destroy(rs);
    }
    uint64_t tot[5] = {0, 0, 0, 0, 0};
    for (int g = 0; g < gpus; ++g) {
        uint64_t c5[5];
        if (bgr_aligner_counters(al[(size_t)g], c5) != BGR_OK) die("counters");
        for (int j = 0; j < 5; ++j) tot[j] += c5[j];
    }
    const uint64_t rn = tot[0], no = tot[1], ali = tot[2], na = tot[3];
    std::cout << "The End" << std::endl;  // aligner.cpp:588-596
    std::cout << "Reads : " << rn << std::endl;
    std::cout << "No overlap : " << no << " Percent : " << (100 * float(no)) / rn << std::endl;
    std::cout << "Got overlap : " << ali + na << " Percent : " << (100 * float(ali + na)) / rn << std::endl;
    std::cout << "Overlap and aligned : " << ali << " Percent : " << (100 * float(ali)) / (ali + na) << std::endl;
    std::cout << "Overlap but not aligned : " << na << " Percent : " << (100 * float(na)) / (ali + na) << std::endl;
    auto end = std::chrono::system_clock::now();
    auto secs = std::chrono::duration_cast<std::chrono::seconds>(end - start).count();
    std::cout << "Reads/seconds : " << rn / (uint64_t)(secs + 1) << std::endl;
    std::cout << "Mapping in seconds : " << secs << std::endl;
    fclose(pathF);
    fclose(notMappedF);
    for (auto* a : al) bgr_aligner_destroy(a);
    bgr_graph_destroy(graph);
    return 0;
}
