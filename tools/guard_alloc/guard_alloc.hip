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
    const size_t rz = rz_bytes();
    const size_t body4 = up(b.size, 4);
    const uint32_t* lo = reinterpret_cast<const uint32_t*>(b.base);
    const uint32_t* hi = reinterpret_cast<const uint32_t*>(b.base + rz + body4);
    const size_t hi_words = (b.total - rz - body4) / 4;
    k_check<<<dim3(256), dim3(256), 0, s>>>(lo, rz / 4, hi, hi_words, rep, b.seq, (unsigned long long)b.size);
}

}  // namespace

extern "C" void* guard_malloc(ssize_t size, int device, hipStream_t stream) {
    if (size <= 0 || device < 0 || device >= MAX_DEV) return nullptr;
    int prev = 0;
    (void)hipGetDevice(&prev);
    if (prev != device) (void)hipSetDevice(device);
    const size_t rz = rz_bytes();
    const size_t total = rz + up((size_t)size, 256) + rz;
    Block b{};
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto& bucket = g_pool[device][total];
        if (!bucket.empty()) { b = bucket.back(); bucket.pop_back(); ++g_reused; g_pooled_bytes -= b.total; }
        b.seq = ++g_seq;
        ++g_mallocs;
    }
    if (b.base == nullptr) {
        if (hipMalloc((void**)&b.base, total) != hipSuccess) { if (prev != device) (void)hipSetDevice(prev); return nullptr; }
        b.total = total;
        (void)hipEventCreateWithFlags(&b.ev, hipEventDisableTiming);
    } else {
        (void)hipStreamWaitEvent(stream, b.ev, 0);     // the check of its previous life has finished
    }
    b.size = (size_t)size;
    const size_t body4 = up(b.size, 4);
    k_fill<<<dim3(64), dim3(256), 0, stream>>>(reinterpret_cast<uint32_t*>(b.base), rz / 4, CANARY);
    (void)hipMemsetAsync(b.base + rz, 0xFF, body4, stream);
    k_fill<<<dim3(64), dim3(256), 0, stream>>>(reinterpret_cast<uint32_t*>(b.base + rz + body4), (total - rz - body4) / 4, CANARY);
    void* body = b.base + rz;
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
        (void)hipFree(b.base);
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
                    const size_t rz = rz_bytes();
                    for (auto& b : live) {
                        const size_t body4 = up(b.size, 4);
                        k_fill<<<dim3(64), dim3(256), 0, nullptr>>>(reinterpret_cast<uint32_t*>(b.base), rz / 4, CANARY);
                        k_fill<<<dim3(64), dim3(256), 0, nullptr>>>(reinterpret_cast<uint32_t*>(b.base + rz + body4), (b.total - rz - body4) / 4, CANARY);
                    }
                    (void)hipDeviceSynchronize();
                }
            }
        }
    }
    if (prev != device) (void)hipSetDevice(prev);
    return rc;
}
