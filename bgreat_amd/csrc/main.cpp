// main.cpp -- `bgreat`: command-line twin of the reference driver (bgreat.cpp:54-130 + Aligner::alignAll,
// aligner.cpp:550-597) on top of the C-ABI of include/bgreat_gpu.h.  Same flags, same files, same stdout
// lines; the per-read work runs on the GPU(s).  Host code only -- no algorithm here.
//
//   bgreat -r reads.fa[,more.fa] -k 31 -g unitigs.fa -m 2 -t 8 [-e effort] [-f paths] [-a notAligned.fa] [-q] [-b] [-i]
//   extensions (opt-in, absent from the reference): --gpus N  (shard each batch over N devices, input order kept)
//                                                   --batch N (reads per device batch, default 128k)
//                                                   --write-exhaustive (-b normally writes nothing)
//                                                   --chunk-bytes N (parser chunk size; tests use tiny chunks)
//                                                   --no-overlap FILE (reads without any anchor go there instead of notAligned.fa)
//                                                   --split-output (with --gpus N: one pipeline per device, device d writes <paths>.<d> /
//                                                                   <notAligned>.<d>; `cat` in device order = the reference's files)
//                                                   --host-route (parse and format on the host always; default: FASTA goes through the
//                                                                 device as text when the run writes the reference's two files)
//                                                   --set name=value (library option, bgr_set_option: INTEGRATION.md 5; e.g. --set timing=1)
#include <getopt.h>

#include <algorithm>
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
    std::string reads, unitigs("unitig.fa"), pathFile("paths"), notAlignedFile("notAligned.fa"), noOverlapFile;
    int errors = 2, threads = 1, ka = 30, effort = 2, gpus = 1;  // bgreat.cpp:56-66 defaults (k is 30, not 31)
    long batch = 0, chunk_bytes = 0;  // batch 0 = the pipeline's default per route
    bool brute = false, incomplete = false, fastq = false, correction = false, dog = false, write_exh = false, host_route = false, split_out = false;
    static option longopts[] = {{"gpus", required_argument, nullptr, 1000}, {"batch", required_argument, nullptr, 1001},
                                {"write-exhaustive", no_argument, nullptr, 1002}, {"chunk-bytes", required_argument, nullptr, 1003},
                                {"no-overlap", required_argument, nullptr, 1004}, {"host-route", no_argument, nullptr, 1005}, {"split-output", no_argument, nullptr, 1006},
                                {"set", required_argument, nullptr, 1007},
                                {nullptr, 0, nullptr, 0}};
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
            case 1003: chunk_bytes = std::stol(optarg); break;
            case 1004: noOverlapFile = optarg; break;
            case 1005: host_route = true; break;
            case 1006: split_out = true; break;
            case 1007: {  // the environment is not read anywhere: options come in here
                const std::string kv = optarg;
                const size_t eq = kv.find('=');
                if (eq == std::string::npos || bgr_set_option(kv.substr(0, eq).c_str(), std::stoll(kv.substr(eq + 1))) != BGR_OK) die("--set name=value");
                break;
            }
            default: break;  // -o and -p are accepted and ignored, as in the reference (no `case`)
        }
    }
    if (reads.empty()) {  // bgreat.cpp:117-127
        std::cout << "-r read_file" << std::endl << "-k k_value (30)" << std::endl << "-g unitig_file (unitig.dot)" << std::endl
                  << "-m n_missmatch (2)" << std::endl << "-t n_thread (1)" << std::endl << "-e effort put in mapping (2)" << std::endl
                  << "-f path_file (paths)" << std::endl << "-a not_aligned_file (notAligned.fa)" << std::endl
                  << "-p to align on paths instead of walks" << std::endl << "-q for fastq read file" << std::endl
                  << "-c to output corrected reads" << std::endl;
        return 0;
    }
    if (gpus < 1 || batch < 0) { fprintf(stderr, "bgreat: --gpus and --batch must be positive\n"); return 2; }

    auto t0 = std::chrono::system_clock::now();
    bgr_graph* graph = nullptr;
    bgr_set_build_threads((uint32_t)std::max(1, threads));
    if (bgr_graph_build_from_fasta_ex(unitigs.c_str(), (uint32_t)ka, 0.0, dog ? BGR_BUILD_ANCHORS : 0u, &graph) != BGR_OK) die("index");
    // one host -> device copy, then device to device over xGMI (RCCL broadcast, or peer copies): include/bgreat_gpu.h
    int64_t one_device = 0, timing = 0;  // (test hook: every lane on device 0)
    (void)bgr_get_option("test.lanes_on_one_device", &one_device);
    (void)bgr_get_option("timing", &timing);
    if (bgr_devices_init(graph, 0, one_device ? 1u : (uint32_t)gpus, BGR_FANOUT_AUTO) != BGR_OK) die("device setup");
    auto t1 = std::chrono::system_clock::now();
    std::cout << "Indexing in seconds : " << std::chrono::duration_cast<std::chrono::seconds>(t1 - t0).count() << std::endl;  // aligner.cpp:546

    // -b selects alignPartExhaustive, where -G has no effect (aligner.cpp:563-567, alignerGreedy.cpp:387)
    bgr_params prm = {brute ? (uint32_t)BGR_MODE_EXHAUSTIVE : (dog ? (uint32_t)BGR_MODE_ANCHORS : (uint32_t)BGR_MODE_GREEDY), (uint32_t)errors, (uint32_t)effort, incomplete ? 1u : 0u};
    bgr_run_options opt;
    memset(&opt, 0, sizeof(opt));
    opt.struct_size = sizeof(opt);
    opt.n_gpus = (uint32_t)gpus;
    opt.threads = (uint32_t)std::max(1, threads);   // -t: host threads of the pipeline (the reference: worker threads)
    opt.batch_reads = (uint64_t)batch;
    opt.chunk_bytes = (uint64_t)std::max(0L, chunk_bytes);
    opt.fastq = fastq ? 1 : 0;
    opt.write_exhaustive = write_exh ? 1 : 0;
    opt.echo_files = 1;
    opt.correction = correction ? 1 : 0;
    opt.no_overlap_file = noOverlapFile.empty() ? nullptr : noOverlapFile.c_str();
    opt.route = host_route ? 1u : 0u;
    opt.split_output = split_out ? 1u : 0u;
    auto start = std::chrono::system_clock::now();
    uint64_t tot[5] = {0, 0, 0, 0, 0};
    double map_secs = 0;
    const int arc = bgr_align_all(graph, &prm, &opt, reads.c_str(), pathFile.c_str(), notAlignedFile.c_str(), tot, &map_secs);
    if (arc == BGR_E_COMPACTION) {  // aligner.cpp:280-283: cout<<"bug compaction"<<endl; cout<<path<<" "<<unitig<<endl; exit(0);
        std::cout << bgr_last_error() << std::endl;
        bgr_host_cache_release();
        return 0;
    }
    if (arc != BGR_OK) die("mapping");
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
    if (timing) fprintf(stderr, "bgreat: mapping %.3f s, %.3f Mreads/s end to end\n", map_secs, map_secs > 0 ? rn / map_secs / 1e6 : 0.0);
    bgr_host_cache_release();  // the pipeline's page-locked staging sets (kept for a next run of the process): freed while the HIP runtime is up
    bgr_graph_destroy(graph);
    return 0;
}
