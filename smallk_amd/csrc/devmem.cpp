// smallk_amd/csrc/devmem.cpp -- a caching allocator in front of hipMalloc / hipFree.
//
// HierNMF2, flat clustering and repeated Nmf() calls create and destroy a solver (~35 device buffers), a column subset
// (6 - 12) and sort workspaces per node; hipMalloc maps fresh pages and hipFree synchronises the device and unmaps them,
// ~0.1 - 0.3 ms each: on the C5-shaped run (15 node factorisations) that was ~0.1 s of a 0.9 s run.  Freed blocks of up to
// 512 MB are kept per device (at most 4 GB of them; SMK_DEVMEM_CACHE_MB) and handed out again to requests that fit (the smallest cached block
// that is large enough and not more than 4 x the request).  Everything larger -- the resident matrices of the big
// workloads -- goes straight to the runtime.
//   * dev_free() synchronises the block's device before the block can be reused: the same guarantee hipFree gives
//     (no kernel that still uses the memory can be in flight), at the cost of a sync on an idle device (~10 us);
//   * a failed hipMalloc empties the device's cache and tries once more;
//   * SMK_DEVMEM_CACHE=0 turns the cache off (every call goes to the runtime), SMK_POISON=1 still poisons every block a
//     caller receives (dev_alloc in solver.cpp), reused or fresh;
//   * smk_finalize / smk_thread_context_end / smk_device_trim return the cached blocks of their device (dev_trim).
#include "common.h"

#include <map>
#include <mutex>
#include <unordered_map>

namespace smk {

namespace {
constexpr size_t MAX_CACHED_BLOCK = (size_t)512 << 20;
// per device.  4 GB covers the per-node workspaces this cache exists for (a solver's ~35 buffers, a subset's 6 - 12, the
// sort workspaces: < 1 GB on the C5-shaped run); what is parked here is invisible to the other allocators of the process
// (torch, RCCL's channel buffers), so the cap is modest and adjustable (SMK_DEVMEM_CACHE_MB), and smk_device_trim() hands
// everything back before a caller creates communicators or large torch tensors
size_t max_cached_total()
{
    static const size_t cap = [] { const char* e = getenv("SMK_DEVMEM_CACHE_MB"); return e ? (size_t)atoll(e) << 20 : (size_t)4 << 30; }();
    return cap;
}
constexpr int MAX_DEVICES = 64;

struct Rec { int dev; size_t bytes; };
std::mutex g_mu;
std::unordered_map<void*, Rec> g_live;                       // every block handed out by dev_malloc
std::multimap<size_t, void*> g_free[MAX_DEVICES];            // cached blocks by size
size_t g_cached[MAX_DEVICES] = {};
unsigned long long g_hits = 0, g_misses = 0;

bool cache_on()
{
    static const bool on = [] { const char* e = getenv("SMK_DEVMEM_CACHE"); return !(e && atoi(e) == 0); }();
    return on;
}

size_t round_size(size_t b)
{
    if (b < 256) b = 256;
    if (b <= 65536) return (b + 255) / 256 * 256;
    size_t step = 4096;                                      // ~1/16 of the size, a power of two
    while (step * 32 < b) step <<= 1;
    return (b + step - 1) / step * step;
}

void trim_locked(int dev)
{
    for (auto& kv : g_free[dev]) (void)hipFree(kv.second);
    g_free[dev].clear();
    g_cached[dev] = 0;
}
}  // namespace

hipError_t dev_malloc(void** p, size_t bytes)
{
    if (!p) return hipErrorInvalidValue;
    *p = nullptr;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= MAX_DEVICES) return hipMalloc(p, bytes);
    size_t want = round_size(bytes);
    if (!cache_on() || want > MAX_CACHED_BLOCK) want = bytes ? bytes : 256;      // never cached: no rounding either (the resident matrices are tens of GB)
    if (cache_on() && want <= MAX_CACHED_BLOCK) {
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = g_free[dev].lower_bound(want);
        if (it != g_free[dev].end() && it->first <= 4 * want) {
            *p = it->second;
            g_live[*p] = Rec{dev, it->first};
            g_cached[dev] -= it->first;
            g_free[dev].erase(it);
            ++g_hits;
            return hipSuccess;
        }
    }
    e = hipMalloc(p, want);
    if (e != hipSuccess) {                                   // give the cache back and try once more
        (void)hipGetLastError();
        { std::lock_guard<std::mutex> lk(g_mu); (void)hipDeviceSynchronize(); trim_locked(dev); }
        e = hipMalloc(p, want);
        if (e != hipSuccess) return e;
    }
    std::lock_guard<std::mutex> lk(g_mu);
    g_live[*p] = Rec{dev, want};
    ++g_misses;
    return hipSuccess;
}

hipError_t dev_free(void* p)
{
    if (!p) return hipSuccess;
    Rec rec{-1, 0};
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = g_live.find(p);
        if (it != g_live.end()) { rec = it->second; g_live.erase(it); }
    }
    if (rec.dev < 0 || !cache_on() || rec.bytes > MAX_CACHED_BLOCK) return hipFree(p);      // not ours / too large: the runtime's
    // the block must be idle before anybody else gets it: what hipFree guarantees, on the block's own device
    int cur = 0;
    (void)hipGetDevice(&cur);
    if (cur != rec.dev) (void)hipSetDevice(rec.dev);
    (void)hipDeviceSynchronize();
    if (cur != rec.dev) (void)hipSetDevice(cur);
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_cached[rec.dev] + rec.bytes > max_cached_total()) return hipFree(p);
    g_free[rec.dev].emplace(rec.bytes, p);
    g_cached[rec.dev] += rec.bytes;
    return hipSuccess;
}

// return the cached blocks of the current device to the runtime
void dev_trim()
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEVICES) return;
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_free[dev].empty()) return;
    (void)hipDeviceSynchronize();
    trim_locked(dev);
}

void dev_cache_stats(unsigned long long* hits, unsigned long long* misses, size_t* cached_bytes)
{
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lk(g_mu);
    if (hits) *hits = g_hits;
    if (misses) *misses = g_misses;
    if (cached_bytes) *cached_bytes = (dev >= 0 && dev < MAX_DEVICES) ? g_cached[dev] : 0;
}

}  // namespace smk
