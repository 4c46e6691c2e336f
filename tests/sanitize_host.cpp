// Harness of tests/test_host_sanitizers.py: the host-side parser and index builder (incl. the anchors index, 1 and 3
// build threads) over the golden inputs, compiled with -fsanitize=address,undefined and again with -fsanitize=thread.
#include "fastx.h"
#include "graph_build.h"
#include "anchor_index.h"
#include <cstdio>
#include <fstream>
#include <sstream>
using namespace bgr;
int main(int argc, char** argv) {
    std::string gold = argv[1];
    const char* unitigs[] = {"toy_unitig.fa", "deg_unitig.fa", "deg_unitig_exc.fa", "syn_unitig.fa", "long_unitig.fa", "short_stop_unitig.fa"};
    int ks[] = {4, 5, 5, 31, 31, 5};
    for (int i = 0; i < 6; ++i) {
        for (unsigned T : {1u, 3u}) {
            set_build_threads(T);
            std::vector<char> seqs; std::vector<uint64_t> offs; std::string err; HostGraph g;
            if (!read_unitig_fasta(gold + "/" + unitigs[i], ks[i], seqs, offs, err)) { printf("read fail %s\n", err.c_str()); return 1; }
            if (!build_graph(ks[i], offs.size() - 1, seqs.data(), offs.data(), 0.0, BGR_BUILD_ANCHORS, g, err)) { printf("build fail %s\n", err.c_str()); return 1; }
            std::string e2;
            if (!validate_blob(g.blob.data(), g.bytes(), e2)) { printf("validate fail %s\n", e2.c_str()); return 1; }
            uint64_t hits = 0;
            for (uint64_t x = 0; x < 5000; ++x) hits += anchor_lookup(g.header(), g.base(), x * 0x9E3779B97F4A7C15ULL >> (64 - 2 * ks[i])) != ~0ULL;
            printf("%s k=%d T=%u unitigs=%llu keys=%llu anchors=%llu hits=%llu\n", unitigs[i], ks[i], T, (unsigned long long)g.header()->n_unitigs,
                   (unsigned long long)g.header()->n_keys, (unsigned long long)g.header()->anc_n, (unsigned long long)hits);
        }
    }
    const char* reads[] = {"toy_reads.fa", "deg_reads.fa", "edge_reads.fa", "edge_reads.fq", "edge_reads_nonl.fq", "syn_r150.fa", "long_r150.fq"};
    bool fq[] = {false, false, false, true, true, false, true};
    for (int i = 0; i < 7; ++i) {
        std::ifstream in(gold + "/" + reads[i], std::ios::binary); std::stringstream ss; ss << in.rdbuf(); std::string d = ss.str();
        ReadSet a; parse_reads(d.data(), d.size(), fq[i], 31, a);
        for (uint64_t chunk : {1ull, 13ull, 100ull, 5000ull}) {
            ReadSet b; parse_reads_parallel(d.data(), d.size(), fq[i], 31, 3, chunk, b);
            if (a.reads != b.reads || a.read_offs != b.read_offs || a.headers != b.headers) { printf("MISMATCH %s chunk %llu\n", reads[i], (unsigned long long)chunk); return 1; }
        }
        printf("%s: %llu records\n", reads[i], (unsigned long long)a.count());
    }
    return 0;
}
