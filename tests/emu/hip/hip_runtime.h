/*
 * tests/emu/hip/hip_runtime.h -- TEST-ONLY debugging harness, NOT a backend and NOT product code.
 *
 * A tiny fiber-based interpreter of the HIP execution model, good enough to run the kernels of
 * pfbwt-f_amd/csrc/ *.hip on the CPU of the build container (which has no GPU) at toy sizes, so that
 * indexing / synchronisation mistakes are found before spending GPU minutes.  It is injected with
 * `g++ -I tests/emu` when building tests/emu/build/libpfbwt_emu.so; the product library
 * (pfbwt-f_amd/lib/libpfbwt_hip.so, built by hipcc for gfx950) never sees this file, the python
 * package never loads the emu library, and no parity claim rests on it.
 *
 * Model: one workgroup at a time; each work-item is a ucontext fiber; __syncthreads() and the wave64
 * cross-lane builtins are rendezvous points.  A kernel that would deadlock on the GPU (divergent
 * barrier) aborts here with a message.
 */
#ifndef PFBWT_EMU_HIP_RUNTIME_H
#define PFBWT_EMU_HIP_RUNTIME_H
#include <sys/mman.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <ucontext.h>
#include <vector>
#include <functional>

#define __global__
#define __device__
#define __host__
#define __shared__ static
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __restrict__ __restrict
#define __constant__ static

struct dim3 { unsigned x, y, z; dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {} };
struct uint2 { unsigned x, y; };
struct uint4 { unsigned x, y, z, w; };
static inline uint4 make_uint4(unsigned a, unsigned b, unsigned c, unsigned d) { uint4 r = {a, b, c, d}; return r; }
static inline uint2 make_uint2(unsigned a, unsigned b) { uint2 r = {a, b}; return r; }
struct ulonglong2 { unsigned long long x, y; };
static inline ulonglong2 make_ulonglong2(unsigned long long a, unsigned long long b) { ulonglong2 r = {a, b}; return r; }

typedef int hipError_t;
typedef struct emu_stream_s *hipStream_t;
typedef struct emu_event_s { double t; } *hipEvent_t;
enum { hipSuccess = 0, hipErrorOutOfMemory = 2, hipErrorInvalidValue = 1 };
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
#define hipStreamNonBlocking 1
#define hipHostMallocDefault 0

namespace emu {
// Fiber switch.  x86-64: six callee-saved registers and the stack pointer (glibc's swapcontext also saves the signal
// mask with a system call per switch, and AddressSanitizer's interceptor of it clears shadow memory for the whole stack:
// a third of the CPU suite's time, five times that in the sanitizer build).  Elsewhere: ucontext.
#if defined(__x86_64__)
struct Ctx { void *sp; };
__attribute__((naked, noinline)) static void emu_switch(void ** /*save sp here: rdi*/, void * /*continue on this sp: rsi*/)
{
    __asm__ volatile("pushq %rbp\n\tpushq %rbx\n\tpushq %r12\n\tpushq %r13\n\tpushq %r14\n\tpushq %r15\n\t"
                     "movq %rsp, (%rdi)\n\tmovq %rsi, %rsp\n\t"
                     "popq %r15\n\tpopq %r14\n\tpopq %r13\n\tpopq %r12\n\tpopq %rbx\n\tpopq %rbp\n\tret");
}
inline void ctx_switch(Ctx *from, Ctx *to) { emu_switch(&from->sp, to->sp); }
inline void ctx_make(Ctx *c, char *stack, size_t size, void (*entry)(), Ctx *)
{
    void **sp = (void **)(((uintptr_t)stack + size) & ~(uintptr_t)15);
    *--sp = nullptr;                    // where entry would return to: it never does
    *--sp = (void *)entry;              // taken by the ret of the first switch: entry starts with rsp = 8 mod 16, as after a call
    for (int i = 0; i < 6; ++i) *--sp = nullptr;
    c->sp = sp;
}
#else
struct Ctx { ucontext_t u; };
inline void ctx_switch(Ctx *from, Ctx *to) { swapcontext(&from->u, &to->u); }
inline void ctx_make(Ctx *c, char *stack, size_t size, void (*entry)(), Ctx *back)
{
    getcontext(&c->u); c->u.uc_stack.ss_sp = stack; c->u.uc_stack.ss_size = size; c->u.uc_link = &back->u;
    makecontext(&c->u, entry, 0);
}
#endif
struct State {
    dim3 tid, bid, bdim, gdim;
    int cur = -1;               // running fiber
    int nthreads = 0;
    Ctx sched;
    std::vector<Ctx> ctx;
    std::vector<char *> stacks;
    std::vector<int> st;        // 0 ready, 1 at block barrier, 2 done, 3 at wave rendezvous
    std::vector<uint64_t> lane_val;
    std::vector<int> lane_pred;
    std::function<void()> body;
};
inline State &S() { static State s; return s; }

inline void yield_to_sched() { State &s = S(); int me = s.cur; ctx_switch(&s.ctx[me], &s.sched); }
inline void fiber_main() { State &s = S(); s.body(); s.st[s.cur] = 2; ctx_switch(&s.ctx[s.cur], &s.sched); abort(); /* a finished fiber is never resumed */ }

inline void set_tid(int t) { State &s = S(); s.cur = t; s.tid.x = t % s.bdim.x; s.tid.y = (t / s.bdim.x) % s.bdim.y; s.tid.z = t / (s.bdim.x * s.bdim.y); }

inline void run_block()
{
    State &s = S();
    const size_t STK = 256 * 1024;
    int T = s.nthreads;
    if ((int)s.ctx.size() < T) { s.ctx.resize(T); s.st.resize(T); s.lane_val.resize(T); s.lane_pred.resize(T); }
    while ((int)s.stacks.size() < T) s.stacks.push_back((char *)malloc(STK));
    for (int t = 0; t < T; ++t) {
        ctx_make(&s.ctx[t], s.stacks[t], STK, (void (*)())fiber_main, &s.sched);
        s.st[t] = 0;
    }
    for (;;) {
        bool progress = false; int done = 0;
        for (int t = 0; t < T; ++t) {
            if (s.st[t] == 0) { set_tid(t); ctx_switch(&s.sched, &s.ctx[t]); progress = true; }
        }
        // block barrier release
        int at_bar = 0; done = 0;
        for (int t = 0; t < T; ++t) { at_bar += s.st[t] == 1; done += s.st[t] == 2; }
        if (done == T) break;
        if (at_bar && at_bar + done == T) { for (int t = 0; t < T; ++t) if (s.st[t] == 1) s.st[t] = 0; progress = true; }
        // wave rendezvous release
        for (int w0 = 0; w0 < T; w0 += 64) {
            int w1 = w0 + 64 < T ? w0 + 64 : T, at = 0, dn = 0;
            for (int t = w0; t < w1; ++t) { at += s.st[t] == 3; dn += s.st[t] == 2; }
            if (at && at + dn == w1 - w0) { for (int t = w0; t < w1; ++t) if (s.st[t] == 3) s.st[t] = 4; progress = true; }
        }
        // state 4 = wave released: all lanes may now read peers' deposits; flip to ready
        for (int t = 0; t < T; ++t) if (s.st[t] == 4) s.st[t] = 0;
        if (!progress) {
            fprintf(stderr, "[emu] deadlock in block (%u,%u): states:", s.bid.x, s.bid.y);
            for (int t = 0; t < T && t < 64; ++t) fprintf(stderr, " %d", s.st[t]);
            fprintf(stderr, "\n"); abort();
        }
    }
}

inline void block_barrier() { State &s = S(); s.st[s.cur] = 1; yield_to_sched(); }
inline void wave_sync() { State &s = S(); s.st[s.cur] = 3; yield_to_sched(); }

template <typename F> inline void launch(dim3 grid, dim3 block, F f)
{
    State &s = S();
    s.gdim = grid; s.bdim = block; s.nthreads = (int)(block.x * block.y * block.z);
    s.body = f;
    for (unsigned bz = 0; bz < grid.z; ++bz) for (unsigned by = 0; by < grid.y; ++by) for (unsigned bx = 0; bx < grid.x; ++bx) {
        s.bid = dim3(bx, by, bz);
        run_block();
    }
}
inline int lane() { return S().cur & 63; }
inline int wave_base() { return S().cur & ~63; }
inline int wave_end() { State &s = S(); int e = wave_base() + 64; return e < s.nthreads ? e : s.nthreads; }
} // namespace emu

#define threadIdx (emu::S().tid)
#define blockIdx (emu::S().bid)
#define blockDim (emu::S().bdim)
#define gridDim (emu::S().gdim)
#define warpSize 64

static inline void __syncthreads() { emu::block_barrier(); }
static inline void __threadfence() {}
static inline void __threadfence_block() {}

// ---- wave64 cross-lane: every live lane of the wave must call (no divergence) -------------
static inline unsigned long long __ballot(int pred)
{
    emu::State &s = emu::S(); int me = s.cur;
    s.lane_pred[me] = pred ? 1 : 0; emu::wave_sync();
    unsigned long long m = 0; int b = emu::wave_base(), e = emu::wave_end();
    for (int t = b; t < e; ++t) if (s.st[t] != 2 && s.lane_pred[t]) m |= 1ULL << (t - b);
    emu::wave_sync();
    (void)me; return m;
}
// the hardware runs a wave's lanes in lock step; a kernel that RELIES on it (the lanes' LDS atomics of one instruction all
// execute before those of the next) marks the points with this builtin, which is where the emulator lines its fibers up
static inline void __builtin_amdgcn_wave_barrier() { emu::wave_sync(); }
static inline unsigned long long __builtin_amdgcn_ballot_w64(bool p) { return __ballot(p ? 1 : 0); }
static inline unsigned __builtin_amdgcn_mbcnt_lo(unsigned m, unsigned acc) { int l = emu::lane(); return acc + (unsigned)__builtin_popcount(l >= 32 ? m : (l ? (m & (0xFFFFFFFFu >> (32 - l))) : 0u)); }
static inline unsigned __builtin_amdgcn_mbcnt_hi(unsigned m, unsigned acc) { int l = emu::lane(); return acc + (unsigned)__builtin_popcount(l <= 32 ? 0u : (m & (0xFFFFFFFFu >> (64 - l)))); }
static inline int __any(int p) { return __ballot(p) != 0; }
static inline int __all(int p) { emu::State &s = emu::S(); unsigned long long live = 0; int b = emu::wave_base(), e = emu::wave_end(); unsigned long long m = __ballot(p); for (int t = b; t < e; ++t) if (s.st[t] != 2) live |= 1ULL << (t - b); return m == live; }
template <typename T> static inline T emu_shfl_from(T v, int src)
{
    emu::State &s = emu::S(); int me = s.cur;
    static_assert(sizeof(T) <= 8, "shfl payload");
    uint64_t raw = 0; memcpy(&raw, &v, sizeof(T)); s.lane_val[me] = raw; emu::wave_sync();
    int b = emu::wave_base(); int e = emu::wave_end();
    uint64_t got = (src >= 0 && b + src < e) ? s.lane_val[b + src] : raw;
    emu::wave_sync();
    T r; memcpy(&r, &got, sizeof(T)); return r;
}
template <typename T> static inline T __shfl(T v, int src, int width = 64) { int l = emu::lane(); int base = l & ~(width - 1); return emu_shfl_from(v, base + (src & (width - 1))); }
template <typename T> static inline T __shfl_up(T v, unsigned d, int width = 64) { int l = emu::lane(); int base = l & ~(width - 1); int src = l - (int)d; return emu_shfl_from(v, src < base ? l : src); }
template <typename T> static inline T __shfl_down(T v, unsigned d, int width = 64) { int l = emu::lane(); int base = l & ~(width - 1); int src = l + (int)d; return emu_shfl_from(v, src >= base + width ? l : src); }
template <typename T> static inline T __shfl_xor(T v, int m, int width = 64) { int l = emu::lane(); (void)width; return emu_shfl_from(v, l ^ m); }

static inline int __popc(unsigned x) { return __builtin_popcount(x); }
static inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
static inline int __clz(int x) { return x ? __builtin_clz((unsigned)x) : 32; }
static inline int __clzll(long long x) { return x ? __builtin_clzll((unsigned long long)x) : 64; }
static inline int __ffs(int x) { return __builtin_ffs(x); }
static inline int __ffsll(long long x) { return __builtin_ffsll(x); }
static inline unsigned long long wall_clock64() { return 0ULL; }
static inline unsigned long long __umul64hi(unsigned long long a, unsigned long long b) { return (unsigned long long)(((unsigned __int128)a * b) >> 64); }
static inline unsigned __umulhi(unsigned a, unsigned b) { return (unsigned)(((uint64_t)a * b) >> 32); }

// ---- atomics (fibers are cooperative: plain RMW is atomic) --------------------------------
template <typename T> static inline T atomicAdd(T *p, T v) { T o = *p; *p = o + v; return o; }
template <typename T> static inline T atomicMax(T *p, T v) { T o = *p; if (v > o) *p = v; return o; }
template <typename T> static inline T atomicMin(T *p, T v) { T o = *p; if (v < o) *p = v; return o; }
template <typename T> static inline T atomicOr(T *p, T v) { T o = *p; *p = o | v; return o; }
template <typename T> static inline T atomicExch(T *p, T v) { T o = *p; *p = v; return o; }
template <typename T> static inline T atomicCAS(T *p, T c, T v) { T o = *p; if (o == c) *p = v; return o; }

#define __HIP_MEMORY_SCOPE_AGENT 4
template <typename T> static inline T __hip_atomic_load(const T *p, int, int) { return *p; }
template <typename T, typename V> static inline void __hip_atomic_store(T *p, V v, int, int) { *p = (T)v; }
static inline void __builtin_amdgcn_s_sleep(int) {}
// v_perm_b32 with selectors 0..7: byte i of the result = byte (sel.byte[i] & 7) of the 64-bit value {hi, lo}
static inline unsigned __builtin_amdgcn_perm(unsigned hi, unsigned lo, unsigned sel)
{
    const unsigned long long tbl = ((unsigned long long)hi << 32) | lo; unsigned r = 0;
    for (int i = 0; i < 4; ++i) r |= (unsigned)((tbl >> (8 * ((sel >> (8 * i)) & 7))) & 0xff) << (8 * i);
    return r;
}

// ---- host runtime ---------------------------------------------------------------------------
static inline hipError_t hipMalloc(void **p, size_t n) { *p = malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
template <typename T> static inline hipError_t hipMalloc(T **p, size_t n) { return hipMalloc((void **)p, n); }
static inline hipError_t hipFree(void *p) { free(p); return hipSuccess; }
static inline hipError_t hipHostMalloc(void **p, size_t n, unsigned = 0) { *p = malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
template <typename T> static inline hipError_t hipHostMalloc(T **p, size_t n, unsigned f = 0) { return hipHostMalloc((void **)p, n, f); }
static inline hipError_t hipHostFree(void *p) { free(p); return hipSuccess; }
#define hipHostRegisterDefault 0
static inline hipError_t hipHostRegister(void *, size_t, unsigned) { return hipSuccess; }
static inline hipError_t hipHostUnregister(void *) { return hipSuccess; }
static inline hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t = 0) { memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipMemset(void *d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t = 0) { memset(d, v, n); return hipSuccess; }
static inline hipError_t hipMemcpy2DAsync(void *d, size_t dp, const void *s, size_t sp, size_t w, size_t h, hipMemcpyKind, hipStream_t = 0)
{ for (size_t r = 0; r < h; ++r) memmove((char *)d + r * dp, (const char *)s + r * sp, w); return hipSuccess; }
static inline hipError_t hipMemset2DAsync(void *d, size_t dp, int v, size_t w, size_t h, hipStream_t = 0)
{ for (size_t r = 0; r < h; ++r) memset((char *)d + r * dp, v, w); return hipSuccess; }
// virtual memory management (csrc/devmem.h): a reservation is an inaccessible anonymous mapping, mapping a handle makes its
// range readable and writable -- a touch outside what the engine committed faults here, as it would on the card
typedef struct emu_memhandle_s { size_t bytes; } *hipMemGenericAllocationHandle_t;
enum hipMemAllocationType { hipMemAllocationTypePinned = 1 };
enum hipMemLocationType { hipMemLocationTypeDevice = 1 };
enum hipMemAccessFlags { hipMemAccessFlagsProtReadWrite = 3 };
struct hipMemLocation { hipMemLocationType type; int id; };
struct hipMemAllocationProp { hipMemAllocationType type; int requestedHandleTypes; hipMemLocation location; void *win32HandleMetaData; unsigned char allocFlags[8]; };
struct hipMemAccessDesc { hipMemLocation location; hipMemAccessFlags flags; };
static inline hipError_t hipMemAddressReserve(void **p, size_t size, size_t align, void *, unsigned long long)
{
    if (!align) align = 4096;
    char *q = (char *)mmap(nullptr, size + align, PROT_NONE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (q == (char *)MAP_FAILED) return hipErrorOutOfMemory;
    char *a = (char *)(((uintptr_t)q + align - 1) / align * align);
    if (a > q) munmap(q, (size_t)(a - q));
    if (a + size < q + size + align) munmap(a + size, (size_t)(q + size + align - (a + size)));
    *p = a; return hipSuccess;
}
static inline hipError_t hipMemAddressFree(void *p, size_t size) { munmap(p, size); return hipSuccess; }
static inline hipError_t hipMemCreate(hipMemGenericAllocationHandle_t *h, size_t size, const hipMemAllocationProp *, unsigned long long) { *h = new emu_memhandle_s{size}; return hipSuccess; }
static inline hipError_t hipMemRelease(hipMemGenericAllocationHandle_t h) { delete h; return hipSuccess; }
// PFP_EMU_POISON=1: freshly mapped memory is filled with 0xA5 (the card hands out whatever the last owner left; anonymous pages are
// zero): a read of memory no kernel wrote then yields wild indices here too instead of benign zeros
static inline hipError_t hipMemMap(void *p, size_t size, size_t, hipMemGenericAllocationHandle_t, unsigned long long)
{
    if (mprotect(p, size, PROT_READ | PROT_WRITE) != 0) return hipErrorInvalidValue;
    static const bool poison = getenv("PFP_EMU_POISON") != nullptr;
    if (poison) memset(p, 0xA5, size);
    return hipSuccess;
}
static inline hipError_t hipMemUnmap(void *p, size_t size)
{ return mmap(p, size, PROT_NONE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE | MAP_FIXED, -1, 0) == p ? hipSuccess : hipErrorInvalidValue; }
static inline hipError_t hipMemSetAccess(void *, size_t, const hipMemAccessDesc *, size_t) { return hipSuccess; }
static inline hipError_t hipStreamCreate(hipStream_t *s) { *s = 0; return hipSuccess; }
static inline hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = 0; return hipSuccess; }
static inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
static inline hipError_t hipSetDevice(int) { return hipSuccess; }
static inline hipError_t hipGetDevice(int *d) { *d = 0; return hipSuccess; }
static inline hipError_t hipGetDeviceCount(int *d) { *d = 1; return hipSuccess; }
enum hipMemoryType { hipMemoryTypeUnregistered = 0, hipMemoryTypeHost = 1, hipMemoryTypeDevice = 2 };
struct hipPointerAttribute_t { int type; };
static inline hipError_t hipPointerGetAttributes(hipPointerAttribute_t *a, const void *) { a->type = hipMemoryTypeUnregistered; return hipErrorInvalidValue; }   // every host pointer is "pageable": the staging ring is what the tests exercise
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipPeekAtLastError() { return hipSuccess; }
static inline const char *hipGetErrorString(hipError_t) { return "emu"; }
static inline hipError_t hipMemGetInfo(size_t *f, size_t *t) { *f = *t = (size_t)8 << 30; return hipSuccess; }
static inline hipError_t hipEventCreate(hipEvent_t *e) { *e = new emu_event_s{0}; return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
#define hipEventDisableTiming 2
static inline hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { *e = new emu_event_s{0}; return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t = 0) { e->t = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b) { *ms = (float)(b->t - a->t); return hipSuccess; }

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
    emu::launch(dim3(grid), dim3(block), [=]() { kernel(__VA_ARGS__); })

#endif
