// Red-zone device allocator for the GPU test-suite (plugged into PyTorch through
// torch.cuda.memory.CUDAPluggableAllocator; tests/conftest.py enables it when DRM_GUARD_ALLOC=1).
//
// Every allocation becomes a region of its own:   [ red zone | body | red zone ]
//   * both red zones (GUARD_RZ_MB MiB each, default 4) are mapped memory filled with a canary word, so an access
//     that runs a little past either end of a tensor neither faults the GPU nor touches another tensor;
//   * the body is filled with 0xFF bytes (fp32 NaN, uint8 255, int32 / int64 -1), so a kernel that consumes memory
//     nobody wrote shows up as NaNs or as an index that lands in a red zone, not as a value that happens to work;
//   * when a tensor is freed (and on demand for every live one) a kernel compares the red zones with the canary:
//     a write past the end of any tensor, by any kernel of any library, is recorded with the size of the tensor it
//     belongs to, the side and the byte distance.
//
// GUARD_FLUSH=1 switches to the other form, which finds READS past the end as well: a tensor's region comes from the
// virtual-memory interface (hipMemAddressReserve / hipMemCreate / hipMemMap) with one more granule of address space
// reserved behind it that is never mapped, and the body ends on the region's last 16-byte boundary.  Any access of
// 16 bytes or more past the end of any tensor is then a memory fault of the kernel that made it, whatever the order
// of the tests (diagnostic runs only: the fault ends the process; DRM_SYNC_CHECK=2 names the library launch).  The
// leading red zone (GUARD_RZ_MB) stays mapped and canary-checked.  A memory fault is fatal for more than the process on
// shared machines (the node's GPU service may end the whole box): use this form only where that is acceptable, for one
// run with a known question, never in a loop.
// Test infrastructure only: nothing in the product path loads this file.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <sys/types.h>

#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <unordered_map>
#include <vector>

namespace {

constexpr uint32_t CANARY = 0xA5C3A5C3u;
constexpr int MAX_DEV = 16;

struct Report {
    unsigned long long count;        // corrupted canary words seen so far
    unsigned long long seq, size;    // allocation of the first one: ordinal and byte size
    unsigned long long side;         // 0: before the tensor, 1: after it
    unsigned long long distance;     // bytes between the tensor's edge and the corrupted word
    unsigned long long value;        // what the word held
};

struct Block {
    char* base; size_t total, size; unsigned long long seq; hipEvent_t ev;
    size_t lead;                     // bytes between base and the body (the leading red zone; flush form: varies with size)
    hipMemGenericAllocationHandle_t handle; bool flush;
};

std::mutex g_mu;
std::unordered_map<void*, Block> g_live;
std::unordered_map<size_t, std::vector<Block>> g_pool[MAX_DEV];
Report* g_report[MAX_DEV] = {};
unsigned long long g_seq = 0, g_mallocs = 0, g_reused = 0;
size_t g_pooled_bytes = 0;
constexpr size_t POOL_CAP_BYTES = (size_t)96 << 30;     // regions kept for reuse; beyond it a freed region goes back to the runtime

size_t rz_bytes() {
    static const size_t v = [] { const char* e = std::getenv("GUARD_RZ_MB"); const long mb = e ? atol(e) : 4; return (size_t)(mb < 1 ? 1 : mb) << 20; }();
    return v;
}
size_t up(size_t v, size_t a) { return (v + a - 1) / a * a; }
bool flush_mode() {
    static const bool v = [] { const char* e = std::getenv("GUARD_FLUSH"); return e != nullptr && e[0] != '0'; }();
    return v;
}
size_t g_gran[MAX_DEV] = {};
hipMemAllocationProp vm_prop(int device) {
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = device;
    return prop;
}
size_t granule(int device) {
    if (g_gran[device] == 0) {
        hipMemAllocationProp prop = vm_prop(device);
        size_t g = 0;
        if (hipMemGetAllocationGranularity(&g, &prop, hipMemAllocationGranularityMinimum) != hipSuccess || g == 0) g = (size_t)2 << 20;
        g_gran[device] = g;
    }
    return g_gran[device];
}
// a mapped region of `total` bytes followed by one reserved, unmapped granule
bool vm_region(int device, size_t total, Block* b) {
    const size_t g = granule(device);
    hipMemAllocationProp prop = vm_prop(device);
    void* va = nullptr;
    if (hipMemAddressReserve(&va, total + g, g, nullptr, 0) != hipSuccess) return false;
    if (hipMemCreate(&b->handle, total, &prop, 0) != hipSuccess) { (void)hipMemAddressFree(va, total + g); return false; }
    if (hipMemMap(va, total, 0, b->handle, 0) != hipSuccess) { (void)hipMemRelease(b->handle); (void)hipMemAddressFree(va, total + g); return false; }
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    if (hipMemSetAccess(va, total, &acc, 1) != hipSuccess) {
        (void)hipMemUnmap(va, total); (void)hipMemRelease(b->handle); (void)hipMemAddressFree(va, total + g); return false;
    }
    b->base = (char*)va; b->total = total; b->flush = true;
    return true;
}
void vm_release(int device, Block& b) {
    (void)hipMemUnmap(b.base, b.total);
    (void)hipMemRelease(b.handle);
    (void)hipMemAddressFree(b.base, b.total + granule(device));
}

__global__ __launch_bounds__(256) void k_fill(uint32_t* p, size_t words, uint32_t v) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < words; i += (size_t)gridDim.x * 256) p[i] = v;
}

// lo: `lo_words` canary words that end where the tensor begins; hi: `hi_words` that begin where it ends
__global__ __launch_bounds__(256) void k_check(const uint32_t* lo, size_t lo_words, const uint32_t* hi, size_t hi_words, Report* rep,
                                               unsigned long long seq, unsigned long long size) {
    const size_t total = lo_words + hi_words;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const bool after = i >= lo_words;
        const uint32_t v = after ? hi[i - lo_words] : lo[i];
        if (v != CANARY && atomicAdd(&rep->count, 1ull) == 0ull) {
            rep->seq = seq; rep->size = size; rep->side = after ? 1ull : 0ull;
            rep->distance = after ? (unsigned long long)(i - lo_words) * 4ull : (unsigned long long)(lo_words - i) * 4ull;
            rep->value = v;
        }
    }
}

Report* report_of(int dev) {
    if (g_report[dev] == nullptr) {
        if (hipMalloc((void**)&g_report[dev], sizeof(Report)) != hipSuccess) return nullptr;
        (void)hipMemset(g_report[dev], 0, sizeof(Report));
    }
    return g_report[dev];
}

void launch_check(const Block& b, Report* rep, hipStream_t s) {
    const size_t body4 = up(b.size, 4);
    const uint32_t* lo = reinterpret_cast<const uint32_t*>(b.base);
    const uint32_t* hi = reinterpret_cast<const uint32_t*>(b.base + b.lead + body4);
    const size_t hi_words = (b.total - b.lead - body4) / 4;
    k_check<<<dim3(256), dim3(256), 0, s>>>(lo, b.lead / 4, hi, hi_words, rep, b.seq, (unsigned long long)b.size);
}

}  // namespace

extern "C" void* guard_malloc(ssize_t size, int device, hipStream_t stream) {
    if (size <= 0 || device < 0 || device >= MAX_DEV) return nullptr;
    int prev = 0;
    (void)hipGetDevice(&prev);
    if (prev != device) (void)hipSetDevice(device);
    const size_t rz = rz_bytes();
    const bool flush = flush_mode();
    const size_t total = flush ? up(rz + up((size_t)size, 16), granule(device)) : rz + up((size_t)size, 256) + rz;
    Block b{};
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto& bucket = g_pool[device][total];
        if (!bucket.empty()) { b = bucket.back(); bucket.pop_back(); ++g_reused; g_pooled_bytes -= b.total; }
        b.seq = ++g_seq;
        ++g_mallocs;
    }
    if (b.base == nullptr) {
        const bool ok = flush ? vm_region(device, total, &b) : hipMalloc((void**)&b.base, total) == hipSuccess;
        if (!ok) { if (prev != device) (void)hipSetDevice(prev); return nullptr; }
        b.total = total;
        (void)hipEventCreateWithFlags(&b.ev, hipEventDisableTiming);
    } else {
        (void)hipStreamWaitEvent(stream, b.ev, 0);     // the check of its previous life has finished
    }
    b.size = (size_t)size;
    b.lead = flush ? total - up(b.size, 16) : rz;      // flush form: the body ends on the region's last 16-byte boundary
    const size_t body4 = up(b.size, 4);
    k_fill<<<dim3(64), dim3(256), 0, stream>>>(reinterpret_cast<uint32_t*>(b.base), b.lead / 4, CANARY);
    (void)hipMemsetAsync(b.base + b.lead, 0xFF, body4, stream);
    if (total > b.lead + body4)
        k_fill<<<dim3(64), dim3(256), 0, stream>>>(reinterpret_cast<uint32_t*>(b.base + b.lead + body4), (total - b.lead - body4) / 4, CANARY);
    void* body = b.base + b.lead;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        g_live[body] = b;
    }
    if (prev != device) (void)hipSetDevice(prev);
    return body;
}

extern "C" void guard_free(void* ptr, ssize_t /*size*/, int device, hipStream_t stream) {
    if (ptr == nullptr || device < 0 || device >= MAX_DEV) return;
    Block b{};
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = g_live.find(ptr);
        if (it == g_live.end()) return;
        b = it->second;
        g_live.erase(it);
    }
    int prev = 0;
    (void)hipGetDevice(&prev);
    if (prev != device) (void)hipSetDevice(device);
    if (Report* rep = report_of(device)) launch_check(b, rep, stream);
    (void)hipEventRecord(b.ev, stream);
    bool keep;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        keep = g_pooled_bytes + b.total <= POOL_CAP_BYTES;
        if (keep) { g_pool[device][b.total].push_back(b); g_pooled_bytes += b.total; }
    }
    if (!keep) {
        (void)hipStreamSynchronize(stream);
        (void)hipEventDestroy(b.ev);
        if (b.flush) vm_release(device, b); else (void)hipFree(b.base);
    }
    if (prev != device) (void)hipSetDevice(prev);
}

// Checks every live allocation as well, waits for the device, and copies the record out:
// out = {corrupted words, ordinal, tensor bytes, side, distance, value, allocations so far, of which reused regions}.
// reset != 0 clears the device-side record afterwards.  Returns 0, or -1 when the device cannot be reached.
extern "C" int guard_report(int device, unsigned long long* out, int reset) {
    if (device < 0 || device >= MAX_DEV || out == nullptr) return -1;
    int prev = 0;
    (void)hipGetDevice(&prev);
    if (prev != device) (void)hipSetDevice(device);
    int rc = -1;
    Report* rep = report_of(device);
    if (rep != nullptr && hipDeviceSynchronize() == hipSuccess) {
        std::vector<Block> live;
        {
            std::lock_guard<std::mutex> lk(g_mu);
            for (auto& kv : g_live) live.push_back(kv.second);
        }
        for (auto& b : live) launch_check(b, rep, nullptr);
        Report h{};
        if (hipDeviceSynchronize() == hipSuccess && hipMemcpy(&h, rep, sizeof(h), hipMemcpyDeviceToHost) == hipSuccess) {
            out[0] = h.count; out[1] = h.seq; out[2] = h.size; out[3] = h.side; out[4] = h.distance; out[5] = h.value;
            std::lock_guard<std::mutex> lk(g_mu);
            out[6] = g_mallocs; out[7] = g_reused;
            rc = 0;
            if (reset) {
                (void)hipMemset(rep, 0, sizeof(Report));
                if (h.count != 0) {      // heal the live regions so that one stray write is reported once
                    for (auto& b : live) {
                        const size_t body4 = up(b.size, 4);
                        k_fill<<<dim3(64), dim3(256), 0, nullptr>>>(reinterpret_cast<uint32_t*>(b.base), b.lead / 4, CANARY);
                        if (b.total > b.lead + body4)
                            k_fill<<<dim3(64), dim3(256), 0, nullptr>>>(reinterpret_cast<uint32_t*>(b.base + b.lead + body4), (b.total - b.lead - body4) / 4, CANARY);
                    }
                    (void)hipDeviceSynchronize();
                }
            }
        }
    }
    if (prev != device) (void)hipSetDevice(prev);
    return rc;
}
