// Process-wide cache of HIP runtime objects (device slabs, pinned host words, streams, events), per device.
// A batch of small problems creates and destroys one handle per problem; on MI355X / ROCm 7.2 the stream, event, pinned-allocation and
// hipFree calls of ONE handle cost 18-40 ms (HIPKKT_VERBOSE: "runtime objects", "destroy") against 20-50 ms for its whole solve, so a
// destroyed handle parks its objects here and the next one on the same device takes them.  Everything handed back is idle (the owner
// synchronises its streams first); recycled memory is NOT zero -- like fresh hipMalloc memory, nothing may rely on its contents
// The cache is never torn down: at process exit the HIP runtime may already be gone.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <map>
#include <mutex>
#include <utility>
#include <vector>

namespace hipkkt {

class RuntimePool {
public:
    static RuntimePool &get() {
        static RuntimePool *p = new RuntimePool();   // intentionally leaked
        return *p;
    }
    bool enabled() const { return on_; }

    // ---- device memory: a block of at least `bytes`; *cap receives its real size (hand it back with the same value)
    void *dev_alloc(int device, size_t bytes, size_t *cap) {
        if (on_) {
            std::lock_guard<std::mutex> lk(mu_);
            Dev &d = dev_[device];
            size_t best = d.slabs.size();
            for (size_t k = 0; k < d.slabs.size(); k++)
                if (d.slabs[k].second >= bytes && d.slabs[k].second <= 4 * bytes + ((size_t)8 << 20) &&
                    (best == d.slabs.size() || d.slabs[k].second < d.slabs[best].second))
                    best = k;
            if (best < d.slabs.size()) {
                void *p = d.slabs[best].first;
                *cap = d.slabs[best].second;
                d.slab_bytes -= *cap;
                d.slabs.erase(d.slabs.begin() + (long)best);
                return p;
            }
        }
        void *p = nullptr;
        if (hipMalloc(&p, bytes) != hipSuccess) {
            (void)hipGetLastError();                       // the failed attempt must not linger as the thread's "last error": callers
                                                           // that check hipGetLastError() after their launches would report it as theirs
            trim(device);                                  // give the cached blocks back and try once more
            if (hipMalloc(&p, bytes) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        }
        *cap = bytes;
        return p;
    }
    void dev_free(int device, void *p, size_t cap) {
        if (!p) return;
        if (on_) {
            std::lock_guard<std::mutex> lk(mu_);
            Dev &d = dev_[device];
            if (d.slabs.size() < 256 && d.slab_bytes + cap <= kMaxCachedBytes) {
                d.slabs.push_back({p, cap});
                d.slab_bytes += cap;
                return;
            }
        }
        (void)hipFree(p);
    }
    void trim(int device) {
        std::vector<std::pair<void *, size_t>> v;
        {
            std::lock_guard<std::mutex> lk(mu_);
            Dev &d = dev_[device];
            v.swap(d.slabs);
            d.slab_bytes = 0;
        }
        for (auto &s : v) (void)hipFree(s.first);
    }

    // ---- pinned host words: chunks of kPinned bytes
    static constexpr size_t kPinned = 256;
    void *pinned_alloc(int device) {
        if (on_) {
            std::lock_guard<std::mutex> lk(mu_);
            Dev &d = dev_[device];
            if (!d.pinned.empty()) { void *p = d.pinned.back(); d.pinned.pop_back(); return p; }
        }
        void *p = nullptr;
        if (hipHostMalloc(&p, kPinned, hipHostMallocDefault) != hipSuccess) return nullptr;
        return p;
    }
    void pinned_free(int device, void *p) {
        if (!p) return;
        if (on_) {
            std::lock_guard<std::mutex> lk(mu_);
            Dev &d = dev_[device];
            if (d.pinned.size() < 1024) { d.pinned.push_back(p); return; }
        }
        (void)hipHostFree(p);
    }

    // ---- streams (cls 0: highest priority, 1: lowest, 2: default) and events (default flags)
    hipStream_t stream_get(int device, int cls) {
        if (on_) {
            std::lock_guard<std::mutex> lk(mu_);
            std::vector<hipStream_t> &v = dev_[device].streams[cls];
            if (!v.empty()) { hipStream_t s = v.back(); v.pop_back(); return s; }
        }
        hipStream_t s = nullptr;
        if (cls == 2) {
            if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) return nullptr;
        } else {
            int lo = 0, hi = 0;
            if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) return nullptr;
            if (hipStreamCreateWithPriority(&s, hipStreamNonBlocking, cls == 0 ? hi : lo) != hipSuccess) return nullptr;
        }
        return s;
    }
    void stream_put(int device, int cls, hipStream_t s) {   // the caller has synchronised it
        if (!s) return;
        if (on_) {
            std::lock_guard<std::mutex> lk(mu_);
            std::vector<hipStream_t> &v = dev_[device].streams[cls];
            if (v.size() < 64) { v.push_back(s); return; }
        }
        (void)hipStreamDestroy(s);
    }
    hipEvent_t event_get(int device) {
        if (on_) {
            std::lock_guard<std::mutex> lk(mu_);
            std::vector<hipEvent_t> &v = dev_[device].events;
            if (!v.empty()) { hipEvent_t e = v.back(); v.pop_back(); return e; }
        }
        hipEvent_t e = nullptr;
        if (hipEventCreate(&e) != hipSuccess) return nullptr;
        return e;
    }
    void event_put(int device, hipEvent_t e) {
        if (!e) return;
        if (on_) {
            std::lock_guard<std::mutex> lk(mu_);
            std::vector<hipEvent_t> &v = dev_[device].events;
            if (v.size() < 512) { v.push_back(e); return; }
        }
        (void)hipEventDestroy(e);
    }

private:
    static constexpr size_t kMaxCachedBytes = (size_t)4 << 30;   // per device
    struct Dev {
        std::vector<std::pair<void *, size_t>> slabs;
        size_t slab_bytes = 0;
        std::vector<void *> pinned;
        std::vector<hipStream_t> streams[3];
        std::vector<hipEvent_t> events;
    };
    bool on_ = true;
    std::mutex mu_;
    std::map<int, Dev> dev_;
};

}  // namespace hipkkt
