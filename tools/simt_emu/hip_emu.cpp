/* hip_emu.cpp — fiber scheduler of the SIMT emulator (TEST TOOLING ONLY, see hip_emu.h). */
#include "hip_emu.h"

#include <cstdio>
#include <dlfcn.h>
#include <cstdint>
#include <ucontext.h>
#include <mutex>
#include <vector>

namespace simt_emu {

enum State { RUN, WAIT_BLOCK, WAIT_WAVE, DONE };

/* Fiber switch: on x86-64 six callee-saved registers and the stack pointer, by hand — swapcontext() also saves and restores
 * the signal mask (two system calls per switch), and a wave-level exchange is 64+ switches: a third of the suite's time */
#if defined(__x86_64__)
#define SIMT_EMU_ASM_SWITCH 1
extern "C" void simt_emu_switch(void **save_sp, void *load_sp);
asm(".text\n.globl simt_emu_switch\n.type simt_emu_switch,@function\nsimt_emu_switch:\n"
    "  pushq %rbp\n  pushq %rbx\n  pushq %r12\n  pushq %r13\n  pushq %r14\n  pushq %r15\n"
    "  movq %rsp, (%rdi)\n  movq %rsi, %rsp\n"
    "  popq %r15\n  popq %r14\n  popq %r13\n  popq %r12\n  popq %rbx\n  popq %rbp\n  ret\n"
    ".size simt_emu_switch, .-simt_emu_switch\n");
#endif

struct Fiber {
#ifdef SIMT_EMU_ASM_SWITCH
    void *sp;
#else
    ucontext_t ctx;
#endif
    Lane lane;
    State st;
    int xchg_val;    /* value published for a shuffle / ballot */
    int xchg_arg;
    int xchg_res;
    char *stack;
    const void *where = nullptr;   /* return address of the pending barrier / wave operation (deadlock report) */
};

Lane *cur = nullptr;
dim3 g_blockIdx, g_blockDim, g_gridDim;

static std::vector<Fiber> fibers;
#ifdef SIMT_EMU_ASM_SWITCH
static void *sched_sp;
static inline void to_sched(Fiber *f) { simt_emu_switch(&f->sp, sched_sp); }
static inline void to_fiber(Fiber *f) { simt_emu_switch(&sched_sp, f->sp); }
#else
static ucontext_t sched_ctx;
static inline void to_sched(Fiber *f) { swapcontext(&f->ctx, &sched_ctx); }
static inline void to_fiber(Fiber *f) { swapcontext(&sched_ctx, &f->ctx); }
#endif
static Fiber *cur_fiber = nullptr;
static const std::function<void()> *cur_body = nullptr;
static const size_t STACK = 256 * 1024;

static void trampoline()
{
    (*cur_body)();
    cur_fiber->st = DONE;
    to_sched(cur_fiber);
    std::abort();                      /* a finished fiber is never resumed */
}

static void yield_as(State s)
{
    Fiber *f = cur_fiber;
    f->st = s;
    to_sched(f);
    /* resumed */
}

void barrier_block() { cur_fiber->where = __builtin_return_address(0); yield_as(WAIT_BLOCK); }

/* All live lanes of the wave publish, yield, and are resumed once every live lane of
 * the wave has published; the resolver (scheduler) fills xchg_res. */
static int wave_op(int v, int arg, int mode_width)
{
    Fiber *f = cur_fiber;
    f->xchg_val = v;
    f->xchg_arg = arg;
    f->xchg_res = mode_width;
    yield_as(WAIT_WAVE);
    return f->xchg_res;
}

int shfl_exchange(int v, int a, int width, int mode) { cur_fiber->where = __builtin_return_address(0); return wave_op(v, a, (mode << 8) | (width & 0xFF)); }
unsigned long long ballot(int pred)
{
    int lo = wave_op(pred != 0, 0, (4 << 8) | 64);
    int hi = wave_op(pred != 0, 0, (5 << 8) | 64);
    return ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo;
}

static void resolve_wave(size_t w0, size_t w1)
{
    /* snapshot published values first (results overwrite xchg_res which holds mode) */
    int vals[64], args[64], modes[64];
    bool live[64];
    for (size_t i = w0; i < w1; i++) {
        Fiber &f = fibers[i];
        live[i - w0] = f.st == WAIT_WAVE;
        vals[i - w0] = f.xchg_val;
        args[i - w0] = f.xchg_arg;
        modes[i - w0] = f.xchg_res;
    }
    for (size_t i = w0; i < w1; i++) {
        Fiber &f = fibers[i];
        if (f.st != WAIT_WAVE) continue;
        int lane = (int)(i - w0);
        int mode = modes[lane] >> 8, width = modes[lane] & 0xFF;
        if (width == 0) width = 64;
        if (mode >= 4) { /* ballot halves */
            unsigned m = 0;
            for (int k = 0; k < 32; k++) {
                int l = k + (mode == 5 ? 32 : 0);
                if ((size_t)l < w1 - w0 && live[l] && vals[l]) m |= 1u << k;
            }
            f.xchg_res = (int)m;
        } else {
            int base = lane & ~(width - 1), rel = lane & (width - 1), src;
            switch (mode) {
            case 0: src = base + (args[lane] & (width - 1)); break;
            case 1: src = lane ^ args[lane]; if ((src & ~(width - 1)) != base) src = lane; break;
            case 2: src = rel - args[lane] >= 0 ? lane - args[lane] : lane; break;
            default: src = rel + args[lane] < width ? lane + args[lane] : lane; break;
            }
            if (src < 0 || (size_t)src >= w1 - w0 || !live[src]) src = lane; /* inactive source: own value */
            f.xchg_res = vals[src];
        }
        f.st = RUN;
    }
}

static void run_block(const std::function<void()> &body, unsigned nthreads)
{
    cur_body = &body;
    if (fibers.size() < nthreads) {
        size_t old = fibers.size();
        fibers.resize(nthreads);
        for (size_t i = old; i < nthreads; i++) fibers[i].stack = (char *)std::malloc(STACK);
    }
    for (unsigned t = 0; t < nthreads; t++) {
        Fiber &f = fibers[t];
#ifdef SIMT_EMU_ASM_SWITCH
        /* the first switch pops six registers and returns into trampoline() with the stack as after a call */
        void **top = reinterpret_cast<void **>((reinterpret_cast<uintptr_t>(f.stack) + STACK) & ~(uintptr_t)15);
        void **sp = top - 8;
        for (int k = 0; k < 6; k++) sp[k] = nullptr;
        sp[6] = reinterpret_cast<void *>(&trampoline);
        sp[7] = nullptr;
        f.sp = sp;
#else
        getcontext(&f.ctx);
        f.ctx.uc_stack.ss_sp = f.stack;
        f.ctx.uc_stack.ss_size = STACK;
        f.ctx.uc_link = &sched_ctx;
        makecontext(&f.ctx, trampoline, 0);
#endif
        f.lane.tid = dim3(t % g_blockDim.x, (t / g_blockDim.x) % g_blockDim.y, t / (g_blockDim.x * g_blockDim.y));
        f.st = RUN;
    }
    for (;;) {
        bool progressed = false;
        for (unsigned t = 0; t < nthreads; t++) {
            Fiber &f = fibers[t];
            if (f.st != RUN) continue;
            cur_fiber = &f;
            cur = &f.lane;
            to_fiber(&f);
            progressed = true;
        }
        /* wave-level rendezvous */
        for (size_t w0 = 0; w0 < nthreads; w0 += 64) {
            size_t w1 = w0 + 64 < nthreads ? w0 + 64 : nthreads;
            bool any = false, all = true;
            for (size_t i = w0; i < w1; i++) {
                if (fibers[i].st == WAIT_WAVE) any = true;
                else if (fibers[i].st != DONE) all = false;
            }
            if (any && all) { resolve_wave(w0, w1); progressed = true; }
        }
        /* block-level barrier */
        bool anyb = false, allb = true, alldone = true;
        for (unsigned t = 0; t < nthreads; t++) {
            if (fibers[t].st == WAIT_BLOCK) anyb = true;
            else if (fibers[t].st != DONE) allb = false;
            if (fibers[t].st != DONE) alldone = false;
        }
        if (alldone) break;
        if (anyb && allb) {
            for (unsigned t = 0; t < nthreads; t++)
                if (fibers[t].st == WAIT_BLOCK) fibers[t].st = RUN;
            progressed = true;
        }
        if (!progressed) {
            std::fprintf(stderr, "simt_emu: deadlock in block (%u,%u,%u): divergent barrier/shuffle "
                                 "(on a GPU this kernel would hang or read undefined lanes)\n",
                         g_blockIdx.x, g_blockIdx.y, g_blockIdx.z);
            for (unsigned t = 0; t < nthreads; t++)
                std::fprintf(stderr, "%d", (int)fibers[t].st);
            std::fprintf(stderr, "\n");
            /* where the stuck lanes wait (addresses inside the library: addr2line -e tests/_emu/libmi355dsp_emu.so <offset>) */
            {
                const void *seen[8]; int ns = 0;
                for (unsigned t = 0; t < nthreads; t++) {
                    if (fibers[t].st == DONE) continue;
                    bool dup = false;
                    for (int k = 0; k < ns; k++) dup = dup || seen[k] == fibers[t].where;
                    if (dup || ns >= 8) continue;
                    seen[ns++] = fibers[t].where;
                    Dl_info di;
                    if (dladdr(fibers[t].where, &di) && di.dli_fbase)
                        std::fprintf(stderr, "  lane %u (state %d) waits at %s+0x%lx\n", t, (int)fibers[t].st, di.dli_fname, (unsigned long)((const char *)fibers[t].where - (const char *)di.dli_fbase));
                }
            }
            std::abort();
        }
    }
}

static long g_launches[16], g_alloc_bytes[16];
int device_count()
{
    const char *e = std::getenv("MI355_EMU_DEVICES");
    const int n = e ? std::atoi(e) : 1;
    return n < 1 ? 1 : (n > 16 ? 16 : n);
}
int &current_device() { static thread_local int d = 0; return d; }
void note_alloc(size_t n) { g_alloc_bytes[current_device() & 15] += (long)n; }

/* one kernel at a time: the fibers, the block index and the current lane are the emulator's own globals, and a host with several decoder
 * threads (the bridges' tests) launches from all of them — on the device those launches queue in the stream, here behind this lock */
static std::mutex g_launch_lock;

void launch(dim3 grid, dim3 block, const std::function<void()> &body)
{
    std::lock_guard<std::mutex> hold(g_launch_lock);
    g_launches[current_device() & 15]++;
    g_gridDim = grid;
    g_blockDim = block;
    unsigned nthreads = block.x * block.y * block.z;
    for (unsigned bz = 0; bz < grid.z; bz++)
        for (unsigned by = 0; by < grid.y; by++)
            for (unsigned bx = 0; bx < grid.x; bx++) {
                g_blockIdx = dim3(bx, by, bz);
                run_block(body, nthreads);
            }
}

}  // namespace simt_emu

extern "C" void simt_emu_device_stats(int device, long out[2])
{
    out[0] = simt_emu::g_launches[device & 15];
    out[1] = simt_emu::g_alloc_bytes[device & 15];
}
