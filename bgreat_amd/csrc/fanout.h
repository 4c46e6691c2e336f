// fanout.h -- the bookkeeping of bgr_devices_init (one-off distribution of the graph blob over the GPUs of one process),
// kept free of HIP so that a CPU test can drive it with stand-in devices (tests/fanout_mock.cpp): which devices need a copy,
// which method runs, what is registered as resident afterwards and what is freed when something fails.
// The device layer is a table of callbacks; capi.hip fills it with hipMalloc / RCCL / hipMemcpyPeerAsync.
//
// Replaces nothing in the reference (one process, no devices: the worker fan-out of aligner.cpp:577-586 is threads over one
// shared index); SURVEY.md 8e: the read-only graph is broadcast once, reads never move between GPUs.
#ifndef BGREAT_AMD_FANOUT_H
#define BGREAT_AMD_FANOUT_H

#include <stddef.h>
#include <stdint.h>

#include <functional>
#include <map>
#include <string>
#include <vector>

namespace bgr {

enum { kFanoutAuto = 0, kFanoutRccl = 1, kFanoutPeer = 2 };  // == BGR_FANOUT_* of include/bgreat_gpu.h

struct FanoutCopy { int src_dev; const void* src; int dst_dev; void* dst; };

struct FanoutOps {
    // a buffer of the blob's size on `dev`; false = out of memory / device error
    std::function<bool(int dev, void** out)> alloc;
    std::function<void(int dev, void* p)> release;
    // ONE collective broadcast from ptr[0] on devs[0] (the root) into ptr[i] on devs[i]; false (+ why) = not usable here,
    // in which case the destinations may hold anything
    std::function<bool(const std::vector<int>& devs, const std::vector<void*>& ptr, std::string& why)> broadcast;
    // a round of device-to-device copies over disjoint links, all complete on return; false = a copy failed
    std::function<bool(const std::vector<FanoutCopy>& round, std::string& why)> peer_round;
};

struct FanoutResult {
    int method = 0;           // kFanoutRccl / kFanoutPeer, 0 = nothing had to move
    bool hip_error = false;   // the failure came from the device layer (else: an argument error)
    std::string error;        // empty = success
};

// `resident`: device -> blob pointer, in and out.  The first device must hold the blob on entry.  Devices of the range that are
// resident already keep their copy and take no part.  New buffers are registered ONLY after the copies into them have succeeded;
// on any failure they are released and `resident` is left as it was, so that a per-device upload can still be tried afterwards.
inline FanoutResult fanout_blob(int first_device, uint32_t n_devices, uint32_t how, std::map<int, void*>& resident, const FanoutOps& ops) {
    FanoutResult res;
    auto holder = resident.find(first_device);
    if (holder == resident.end() || !holder->second) { res.error = "the first device does not hold the blob"; return res; }
    std::vector<int> devs(1, first_device);
    std::vector<void*> ptr(1, holder->second);
    auto drop_fresh = [&]() { for (size_t i = 1; i < devs.size(); ++i) ops.release(devs[i], ptr[i]); };
    for (uint32_t i = 1; i < n_devices; ++i) {
        const int d = first_device + (int)i;
        if (resident.count(d)) continue;  // already resident there
        void* p = nullptr;
        if (!ops.alloc(d, &p) || !p) {
            drop_fresh();
            res.hip_error = true;
            res.error = "allocation of the blob on device " + std::to_string(d) + " failed";
            return res;
        }
        devs.push_back(d);
        ptr.push_back(p);
    }
    auto commit = [&](int method) {
        for (size_t i = 1; i < devs.size(); ++i) resident[devs[i]] = ptr[i];
        res.method = method;
    };
    std::string why;
    if (devs.size() == 1) {
        // nothing to distribute.  Asked for the collective explicitly, the call still goes through a one-rank communicator and an
        // in-place broadcast, so that the run-time lookup of librccl and the call sequence can be checked on a single-GPU machine.
        if (how == kFanoutRccl) {
            if (!ops.broadcast(devs, ptr, why)) { res.hip_error = true; res.error = why; return res; }
            res.method = kFanoutRccl;
        }
        return res;
    }
    if (how != kFanoutPeer) {
        if (ops.broadcast(devs, ptr, why)) { commit(kFanoutRccl); return res; }
        if (how == kFanoutRccl) { drop_fresh(); res.hip_error = true; res.error = why; return res; }
    }
    // peer copies in a doubling schedule: holders 0 .. have-1 each feed device i + have (1 -> 2 -> 4 -> 8 holders, every round over
    // disjoint point-to-point links); whatever a failed broadcast left in the destinations is overwritten
    for (size_t have = 1; have < devs.size(); have *= 2) {
        std::vector<FanoutCopy> round;
        for (size_t i = 0; i < have && i + have < devs.size(); ++i) round.push_back({devs[i], ptr[i], devs[i + have], ptr[i + have]});
        if (!ops.peer_round(round, why)) { drop_fresh(); res.hip_error = true; res.error = why; return res; }
    }
    commit(kFanoutPeer);
    return res;
}

}  // namespace bgr

#endif
