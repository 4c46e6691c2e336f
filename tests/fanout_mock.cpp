// Harness of tests/test_fanout_mock.py: drives the bookkeeping of bgr_devices_init (bgreat_amd/csrc/fanout.h) with stand-in
// devices -- TEST INFRASTRUCTURE: "device memory" is host memory tagged with its device, the collective and the peer copies are
// memcpy with scripted failures -- for N in {1, 2, 3, 8}: every device of the range must end with the blob, devices that were
// resident keep their buffer, the collective's root is the holder, and after ANY failure nothing new is registered and every
// buffer allocated by the call has been released (the round-2 defect: uninitialised buffers stayed registered).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <set>

#include "fanout.h"

namespace {

const size_t kBytes = 4096;
struct Buf { int dev; unsigned char data[kBytes]; };
std::set<Buf*> live;
int fails = 0;
#define CHECK(c) do { if (!(c)) { printf("FAIL line %d: %s\n", __LINE__, #c); ++fails; } } while (0)

struct Script { int alloc_fail_at = -1; bool rccl_available = true; bool rccl_fails = false; int peer_fail_round = -1; };
struct Log { int allocs = 0, rounds = 0, broadcasts = 0; int root = -1; size_t max_copies_per_src = 0; };

bgr::FanoutOps make_ops(const Script& sc, Log& log) {
    bgr::FanoutOps ops;
    ops.alloc = [&sc, &log](int dev, void** out) {
        if (log.allocs++ == sc.alloc_fail_at) return false;
        Buf* b = new Buf();
        b->dev = dev;
        memset(b->data, 0xEE, kBytes);  // "uninitialised"
        live.insert(b);
        *out = b;
        return true;
    };
    ops.release = [](int dev, void* p) {
        Buf* b = static_cast<Buf*>(p);
        CHECK(live.count(b) == 1 && b->dev == dev);
        live.erase(b);
        delete b;
    };
    ops.broadcast = [&sc, &log](const std::vector<int>& devs, const std::vector<void*>& ptr, std::string& why) {
        ++log.broadcasts;
        if (!sc.rccl_available) { why = "librccl not loadable"; return false; }
        log.root = devs[0];
        for (size_t i = 1; i < devs.size(); ++i) {
            CHECK(static_cast<Buf*>(ptr[i])->dev == devs[i]);
            if (sc.rccl_fails && i == devs.size() - 1) { why = "RCCL broadcast failed"; return false; }  // (the others already written: partial)
            memcpy(static_cast<Buf*>(ptr[i])->data, static_cast<Buf*>(ptr[0])->data, kBytes);
        }
        return true;
    };
    ops.peer_round = [&sc, &log](const std::vector<bgr::FanoutCopy>& round, std::string& why) {
        if (log.rounds++ == sc.peer_fail_round) { why = "peer copy failed"; return false; }
        std::set<int> srcs, dsts;
        for (const bgr::FanoutCopy& c : round) {
            const Buf* s = static_cast<const Buf*>(c.src);
            Buf* d = static_cast<Buf*>(c.dst);
            CHECK(s->dev == c.src_dev && d->dev == c.dst_dev);
            CHECK(s->data[0] == 0x42 && s->data[kBytes - 1] == 0x42);   // a source must hold the blob already
            CHECK(srcs.insert(c.src_dev).second && dsts.insert(c.dst_dev).second);  // disjoint links within a round
            memcpy(d->data, s->data, kBytes);
        }
        return true;
    };
    return ops;
}

bool holds_blob(void* p) {
    const Buf* b = static_cast<const Buf*>(p);
    for (size_t i = 0; i < kBytes; ++i) if (b->data[i] != 0x42) return false;
    return true;
}

void run(const char* name, int first, uint32_t n, uint32_t how, const std::vector<int>& pre_resident, const Script& sc, bool expect_ok, int expect_method) {
    Log log;
    std::map<int, void*> resident;
    std::vector<Buf*> mine;
    auto add = [&](int dev) { Buf* b = new Buf(); b->dev = dev; memset(b->data, 0x42, kBytes); resident[dev] = b; mine.push_back(b); };
    add(first);
    for (int d : pre_resident) add(d);
    const std::map<int, void*> before = resident;
    const bgr::FanoutOps ops = make_ops(sc, log);
    const bgr::FanoutResult r = bgr::fanout_blob(first, n, how, resident, ops);
    const bool ok = r.error.empty();
    CHECK(ok == expect_ok);
    if (ok) {
        CHECK(r.method == expect_method);
        for (uint32_t i = 0; i < n; ++i) {
            CHECK(resident.count(first + (int)i) == 1);
            if (resident.count(first + (int)i)) CHECK(holds_blob(resident[first + (int)i]));
        }
        for (auto& kv : before) CHECK(resident[kv.first] == kv.second);  // resident devices keep their buffer
        if (r.method == bgr::kFanoutRccl && n > 1) CHECK(log.root == first);
        size_t fresh = 0;
        for (uint32_t i = 0; i < n; ++i) fresh += before.count(first + (int)i) ? 0 : 1;
        CHECK(live.size() == fresh);
        if (r.method == bgr::kFanoutPeer) { int need = 0; for (size_t h = 1; h < fresh + 1; h *= 2) ++need; CHECK(log.rounds == need); }
    } else {
        CHECK(resident == before);   // nothing new registered
        CHECK(live.empty());         // and nothing leaked
        CHECK(r.hip_error);
    }
    for (auto& kv : resident) if (!before.count(kv.first)) { live.erase(static_cast<Buf*>(kv.second)); delete static_cast<Buf*>(kv.second); }
    for (Buf* b : mine) delete b;
    CHECK(live.empty());
    printf("%-58s %s (method %d, %d allocs, %d broadcasts, %d peer rounds)\n", name, fails ? "FAIL" : "ok", r.method, log.allocs, log.broadcasts, log.rounds);
}

}  // namespace

int main() {
    Script plain, no_rccl, bad_rccl, bad_alloc, bad_peer;
    no_rccl.rccl_available = false;
    bad_rccl.rccl_fails = true;
    for (uint32_t n : {1u, 2u, 3u, 8u}) {
        char nm[96];
        snprintf(nm, sizeof nm, "N=%u auto, collective available", n);
        run(nm, 0, n, bgr::kFanoutAuto, {}, plain, true, n > 1 ? bgr::kFanoutRccl : 0);
        snprintf(nm, sizeof nm, "N=%u auto, no librccl -> peer doubling", n);
        run(nm, 0, n, bgr::kFanoutAuto, {}, no_rccl, true, n > 1 ? bgr::kFanoutPeer : 0);
        snprintf(nm, sizeof nm, "N=%u auto, collective fails half way -> peer", n);
        run(nm, 0, n, bgr::kFanoutAuto, {}, bad_rccl, true, n > 1 ? bgr::kFanoutPeer : 0);
        snprintf(nm, sizeof nm, "N=%u collective demanded", n);
        run(nm, 0, n, bgr::kFanoutRccl, {}, plain, true, bgr::kFanoutRccl);
        snprintf(nm, sizeof nm, "N=%u collective demanded, unavailable", n);
        run(nm, 0, n, bgr::kFanoutRccl, {}, no_rccl, false, 0);
        snprintf(nm, sizeof nm, "N=%u peer demanded", n);
        run(nm, 0, n, bgr::kFanoutPeer, {}, plain, true, n > 1 ? bgr::kFanoutPeer : 0);
    }
    // devices already resident take no part and keep their buffers; first device not 0
    run("N=8 from device 0, devices 2 and 5 resident", 0, 8, bgr::kFanoutAuto, {2, 5}, plain, true, bgr::kFanoutRccl);
    run("N=8 peer, devices 1,2,3 resident", 0, 8, bgr::kFanoutPeer, {1, 2, 3}, plain, true, bgr::kFanoutPeer);
    run("N=3 from device 4", 4, 3, bgr::kFanoutAuto, {}, no_rccl, true, bgr::kFanoutPeer);
    run("N=4, everything resident already", 0, 4, bgr::kFanoutAuto, {1, 2, 3}, plain, true, 0);
    // failures: allocation in the middle of the loop, a peer copy in the second round
    bad_alloc.alloc_fail_at = 3;
    run("N=8, fourth allocation fails", 0, 8, bgr::kFanoutAuto, {}, bad_alloc, false, 0);
    bad_peer.rccl_available = false;
    bad_peer.peer_fail_round = 1;
    run("N=8, second peer round fails", 0, 8, bgr::kFanoutAuto, {}, bad_peer, false, 0);
    bad_peer.peer_fail_round = 0;
    run("N=2, the only peer round fails", 0, 2, bgr::kFanoutPeer, {}, bad_peer, false, 0);
    printf("%s\n", fails ? "FAILED" : "ALL OK");
    return fails ? 1 : 0;
}
