// hipemu.cpp -- TEST INFRASTRUCTURE ONLY (see hipemu.h).
#include "hipemu.h"
#include <omp.h>

namespace hipemu {

thread_local State S;

static const size_t STACK_BYTES = 96 * 1024;

static void fiber_main() {
    (*S.body)();
    Fiber& f = S.fibers[S.cur];
    f.done = 1;
    f.yield_kind = 0;
    hipemu_switch(&f.sp, S.sched_sp);
    abort();  // a finished fibre is never resumed
}

static void init_fiber(Fiber& f) {
    f.done = 0;
    f.yield_kind = 0;
    // initial frame for hipemu_switch: 6 callee-saved registers, then the return address.
    uintptr_t top = (uintptr_t)(f.stack + STACK_BYTES);
    top &= ~(uintptr_t)15;
    void** sp = (void**)top;
    *--sp = nullptr;                 // top-8: fake return address of fiber_main; after the 'ret' below rsp
                                     // points here, and (top-8) % 16 == 8 exactly as at a normal call entry
    *--sp = (void*)&fiber_main;      // top-16: 'ret' target
    for (int i = 0; i < 6; ++i) *--sp = nullptr;   // r15 r14 r13 r12 rbx rbp
    f.sp = sp;
}

static void run_block(dim3 grid, dim3 block, size_t smem_bytes, const std::function<void()>& body,
                      unsigned bx, unsigned by, unsigned bz) {
    const unsigned nthreads = block.x * block.y * block.z;
    S.bid = dim3(bx, by, bz);
    S.bdim = block;
    S.gdim = grid;
    S.body = &body;
    if (S.fibers.size() < nthreads) {
        size_t old = S.fibers.size();
        S.fibers.resize(nthreads);
        for (size_t i = old; i < nthreads; ++i) S.fibers[i].stack = (char*)malloc(STACK_BYTES);
    }
    S.xchg.assign(nthreads, 0);
    S.xchg2.assign(nthreads, 0);
    S.xq.assign(4 * (size_t)nthreads, 0);
    S.xq2.assign(4 * (size_t)nthreads, 0);
    std::vector<char> smem(smem_bytes + 64);
    memset(smem.data(), 0xFF, smem.size());   // NaN pattern: uninitialised LDS reads show up
    S.smem = (char*)(((uintptr_t)smem.data() + 63) & ~(uintptr_t)63);
    for (unsigned t = 0; t < nthreads; ++t) init_fiber(S.fibers[t]);
    // Cooperative scheduler.  A fibre runs until it yields at a block barrier (kind 1) or at a wave-level
    // rendezvous (kind 2: shuffles, MFMA operand exchange).  Wave rendezvous release as soon as every live lane of
    // that wave has arrived, so waves of one block may execute different numbers of them between block barriers.
    enum { RUNNABLE = 0, AT_BARRIER = 1, AT_WAVE = 2 };
    std::vector<unsigned char> st(nthreads, RUNNABLE);
    unsigned alive = nthreads;
    while (alive) {
        bool progressed = false;
        for (unsigned t = 0; t < nthreads; ++t) {
            Fiber& f = S.fibers[t];
            if (f.done || st[t] != RUNNABLE) continue;
            S.cur = (int)t;
            S.tid = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
            hipemu_switch(&S.sched_sp, f.sp);
            progressed = true;
            if (f.done) { --alive; continue; }
            st[t] = f.yield_kind == 1 ? AT_BARRIER : AT_WAVE;
        }
        for (unsigned w0 = 0; w0 < nthreads; w0 += WAVE) {
            unsigned live = 0, waiting = 0;
            for (unsigned t = w0; t < w0 + WAVE && t < nthreads; ++t) {
                if (S.fibers[t].done) continue;
                ++live;
                if (st[t] == AT_WAVE) ++waiting;
            }
            if (live && waiting == live) {
                for (unsigned t = w0; t < w0 + WAVE && t < nthreads; ++t)
                    if (!S.fibers[t].done) st[t] = RUNNABLE;
                progressed = true;
            }
        }
        unsigned at_barrier = 0;
        for (unsigned t = 0; t < nthreads; ++t)
            if (!S.fibers[t].done && st[t] == AT_BARRIER) ++at_barrier;
        if (alive && at_barrier == alive) {
            if (alive != nthreads) {
                // legal on a GPU only if the finished threads never reach that barrier; our kernels never
                // do that, so flag it.
                fprintf(stderr, "hipemu: %u threads exited while %u wait at a barrier (block %u,%u,%u)\n",
                        nthreads - alive, alive, bx, by, bz);
                abort();
            }
            for (unsigned t = 0; t < nthreads; ++t) st[t] = RUNNABLE;
            progressed = true;
        }
        if (alive && !progressed) {
            fprintf(stderr, "hipemu: deadlock (divergent synchronisation) in block (%u,%u,%u)\n", bx, by, bz);
            abort();
        }
    }
}

void launch(dim3 grid, dim3 block, size_t smem_bytes, const std::function<void()>& body) {
    const long nblocks = (long)grid.x * grid.y * grid.z;
#pragma omp parallel for schedule(dynamic, 1)
    for (long b = 0; b < nblocks; ++b) {
        unsigned bx = (unsigned)(b % grid.x), by = (unsigned)((b / grid.x) % grid.y), bz = (unsigned)(b / ((long)grid.x * grid.y));
        run_block(grid, block, smem_bytes, body, bx, by, bz);
    }
}

}  // namespace hipemu

asm(R"(
.text
.globl hipemu_switch
.type hipemu_switch,@function
hipemu_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
.size hipemu_switch,.-hipemu_switch
)");
