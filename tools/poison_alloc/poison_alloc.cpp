// Poisoning device allocator for torch.cuda.memory.CUDAPluggableAllocator (debug tool, GPU box; never part of the product).
//
// Finds reads of memory a kernel has no business reading: every byte this allocator owns that is not inside a live, written tensor
// holds 0xFF (fp32 / fp16 NaN):
//   * one arena (GRL_POISON_ARENA_GB, default 32) is taken from the driver at the first request and filled with 0xFF;
//   * a block is handed out poisoned and followed by a 4 KiB red zone that is never handed out: reading a tensor that was never
//     written, or up to 4 KiB past its end, yields NaN instead of whatever the caching allocator's neighbour held;
//   * free() poisons the block again (stream-ordered; outside stream capture after a device synchronisation), so a launch that
//     still reads a tensor after its last reference went away reads NaN as well;
//   * requests made while the stream is capturing a HIP graph come from, and return to, a pool of their own (what the caching
//     allocator's private pools do): replays cannot scribble over blocks that eager code received later.  No driver call other than
//     the (captured) memset happens during capture.
// Build: make -C tools/poison_alloc ; use: tools/poison_alloc/run_poisoned.py
#include <hip/hip_runtime.h>
#include <sys/types.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <mutex>
#include <vector>

namespace {
constexpr size_t ALIGN = 512, RED = 4096;
std::mutex mu;
char* arena = nullptr;
size_t arena_bytes = 0, bump = 0;
struct Block { size_t size; bool graph; };
std::map<void*, Block> live;                                   // handed-out blocks
std::map<size_t, std::vector<void*>> free_eager, free_graph;   // rounded size -> blocks
size_t n_malloc = 0, n_free = 0, n_graph = 0;

bool capturing(hipStream_t s) {
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &st) != hipSuccess) { (void)hipGetLastError(); return false; }
    return st == hipStreamCaptureStatusActive;
}
void die(const char* what) { fprintf(stderr, "[poison_alloc] %s\n", what); abort(); }
}  // namespace

extern "C" void* grl_poison_malloc(ssize_t size, int device, hipStream_t stream) {
    std::lock_guard<std::mutex> g(mu);
    if (!arena) {
        const char* gb = getenv("GRL_POISON_ARENA_GB");
        arena_bytes = (size_t)(gb ? atof(gb) : 32.0) * (size_t(1) << 30);
        if (hipMalloc((void**)&arena, arena_bytes) != hipSuccess) die("arena allocation failed");
        if (hipMemset(arena, 0xFF, arena_bytes) != hipSuccess || hipDeviceSynchronize() != hipSuccess) die("arena fill failed");
    }
    const size_t need = ((size_t)(size > 0 ? size : 1) + ALIGN - 1) / ALIGN * ALIGN;
    const bool cap = capturing(stream);
    auto& fl = cap ? free_graph : free_eager;
    void* p = nullptr;
    auto it = fl.find(need);
    if (it != fl.end() && !it->second.empty()) { p = it->second.back(); it->second.pop_back(); }
    else {
        if (bump + need + RED > arena_bytes) die("arena exhausted (raise GRL_POISON_ARENA_GB)");
        p = arena + bump;
        bump += need + RED;
    }
    live[p] = Block{need, cap};
    ++n_malloc; n_graph += cap;
    return p;
}

extern "C" void grl_poison_free(void* ptr, ssize_t size, int device, hipStream_t stream) {
    if (!ptr) return;
    std::lock_guard<std::mutex> g(mu);
    auto it = live.find(ptr);
    if (it == live.end()) die("free of a block this allocator does not own");
    const Block b = it->second;
    live.erase(it);
    const bool cap = capturing(stream);
    if (!cap) (void)hipDeviceSynchronize();                   // every launch that may still use the block has finished
    if (hipMemsetAsync(ptr, 0xFF, b.size, stream) != hipSuccess) die("poisoning a freed block failed");
    if (!cap) (void)hipStreamSynchronize(stream);
    (b.graph ? free_graph : free_eager)[b.size].push_back(ptr);
    ++n_free;
}

extern "C" void grl_poison_stats() {
    std::lock_guard<std::mutex> g(mu);
    fprintf(stderr, "[poison_alloc] %zu mallocs (%zu while capturing), %zu frees, %zu live, high water %.2f GiB of %.0f\n", n_malloc, n_graph,
            n_free, live.size(), bump / double(size_t(1) << 30), arena_bytes / double(size_t(1) << 30));
}
