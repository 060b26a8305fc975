// Regions in flight as fibers on a few host threads.
//
// The reference runs one process per region (scripts/rambl.py:190-194, Pool(cores)); the first versions of this
// library ran one host thread per region in flight.  A region spends nearly all of its life waiting for its current
// level on the GPU, so with hundreds of regions in flight that is hundreds of threads that wake up, work for ~15 us
// and sleep again -- a futex round trip per level on both sides, and far more runnable threads than the cgroup of a
// GPU box has cores.  Here a region is a fiber (its own stack, switched in user space); a fixed number of executor
// threads, sized from the CPU quota of the rank, run whichever fibers are ready.  A fiber that waits for its level parks;
// whoever sees the level's completion stamp makes it ready again.  No HIP in this file: tests/native/fiber_check.cpp
// drives it on the CPU.
#pragma once
#include <sys/mman.h>

#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <cstdlib>
#include <deque>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

// The switch itself: callee-saved registers, the SSE and x87 control words (the library computes in long double: the
// precision control travels with the fiber), the stack pointer.  No signal mask (swapcontext makes two system calls per
// switch for it; a region switches twice per level).  x86-64 System V.
extern "C" void sc_fiber_switch(void** save_sp, void* load_sp);
asm(R"(
    .text
    .weak sc_fiber_switch
    .type sc_fiber_switch,@function
sc_fiber_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    subq $8, %rsp
    stmxcsr (%rsp)
    fnstcw 4(%rsp)
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    ldmxcsr (%rsp)
    fldcw 4(%rsp)
    addq $8, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
    .size sc_fiber_switch,.-sc_fiber_switch
)");

namespace sc {

class FiberPool;

// A lock for sections of a few nanoseconds that are entered hundreds of thousands of times a second from a handful of
// threads (the scheduler's queues, the level server's inbox).  It never sleeps in the kernel: a mutex does when it is
// contended, and the ~50 us until the sleeper runs again are several levels' worth of host work.
class SpinLock {
public:
    void lock() {
        for (unsigned n = 0;; n++) {
            if (!busy_.load(std::memory_order_relaxed) && !busy_.exchange(true, std::memory_order_acquire)) return;
            if ((n & 0xFFFu) == 0xFFFu) std::this_thread::yield(); else __builtin_ia32_pause();       // (its holder has lost its CPU)
        }
    }
    void unlock() { busy_.store(false, std::memory_order_release); }
private:
    std::atomic<bool> busy_{false};
};

struct Fiber {
    void* sp = nullptr;                 // saved stack pointer while the fiber is not running
    void* stack = nullptr;
    size_t stack_bytes = 0;
    std::function<void()> body;
    // true from the moment an executor switches the fiber in until that executor is back on its own stack: whoever
    // makes a parked fiber ready may do so before the fiber has finished switching out, and the executor that picks
    // it up waits for this flag (a few nanoseconds) instead of running a stack that is still in use
    std::atomic<bool> on_cpu{false};
    std::atomic<bool> finished{false};
    FiberPool* pool = nullptr;
    void** back = nullptr;              // where the executor that runs the fiber keeps its own stack pointer
};

class FiberPool {
public:
    // n_threads executor threads, of which n_long take the fibers that were made ready with `later` first (the long,
    // CPU-bound stretches: a region's set-up) and the others never do: a continuation of a region whose level has come back
    // (a few tens of microseconds) does not wait behind a graph construction (tens of milliseconds).  With n_long = 0 every
    // thread takes both.  `on_thread_start` runs once on each (hipSetDevice).
    explicit FiberPool(int n_threads, std::function<void()> on_thread_start = nullptr, int n_long = 0)
        : on_start_(std::move(on_thread_start)) {
        if (n_threads < 1) n_threads = 1;
        if (n_long < 0 || n_long >= n_threads) n_long = 0;
        split_ = n_long > 0;
        if (const char* e = getenv("SC_EXEC_SPINNERS")) max_spinners_ = atoi(e) < 0 ? 0 : atoi(e);
        for (int i = 0; i < n_threads; i++) { const bool lng = i >= n_threads - n_long; threads_.emplace_back([this, lng] { run(lng); }); }
    }
    ~FiberPool() { shutdown(); }
    FiberPool(const FiberPool&) = delete;
    FiberPool& operator=(const FiberPool&) = delete;

    int threads() const { return (int)threads_.size(); }

    // A new fiber, not ready yet (make_ready starts it).  The pool owns it until shutdown.
    Fiber* create(std::function<void()> body, size_t stack_bytes = (size_t)1 << 20) {
        Fiber* f = new Fiber();
        const size_t page = 4096;
        stack_bytes = (stack_bytes + page - 1) & ~(page - 1);
        void* p = mmap(nullptr, stack_bytes + page, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE | MAP_STACK, -1, 0);
        if (p == MAP_FAILED) { delete f; return nullptr; }
        mprotect(p, page, PROT_NONE);                      // guard page below the stack
        f->stack = p; f->stack_bytes = stack_bytes + page;
        f->body = std::move(body);
        f->pool = this;
        // the frame sc_fiber_switch pops on the first switch into the fiber: control words, six registers, the address of
        // entry(); above it a null return address, so that entry() starts on the alignment a call would have left
        uintptr_t top = ((uintptr_t)p + page + stack_bytes) & ~(uintptr_t)15;
        uint64_t* w = (uint64_t*)(top - 72);
        uint32_t mxcsr; uint16_t fpucw;
        asm volatile("stmxcsr %0" : "=m"(mxcsr));
        asm volatile("fnstcw %0" : "=m"(fpucw));
        w[0] = (uint64_t)mxcsr | ((uint64_t)fpucw << 32);
        for (int i = 1; i <= 6; i++) w[i] = 0;
        w[7] = (uint64_t)(uintptr_t)&FiberPool::entry;
        w[8] = 0;
        f->sp = w;
        { std::lock_guard<std::mutex> lk(sleep_mu_); all_.push_back(f); }
        return f;
    }

    // The fiber may run (again).  Callable from any thread, also before the fiber has finished parking.  Every
    // make_ready must be matched by exactly one park() / the fiber's start: a fiber that is made ready twice for one
    // park would be resumed a second time from wherever it parks next.
    // `later`: behind every fiber that was made ready the ordinary way -- for long, CPU-bound stretches (the set-up of
    // a new region) that must not delay the short continuations of the regions whose levels are coming back.
    void make_ready(Fiber* f, bool later = false) {
        qlock();
        (later ? later_ : ready_).push_back(f);
        qunlock();
        // (the queue is filled before its counter rises: whoever sees the counter finds the fiber)
        if (later) {
            n_later_.fetch_add(1, std::memory_order_seq_cst);
            if (split_) { if (long_sleepers_.load(std::memory_order_seq_cst) > 0) wake(cv_long_); return; }
            if (sleepers_.load(std::memory_order_seq_cst) > 0 && spinners_.load(std::memory_order_acquire) == 0) wake(cv_);
            return;
        }
        const int nf = n_front_.fetch_add(1, std::memory_order_seq_cst) + 1;
        // a spinning executor sees the counter at once; a sleeping one is woken only when no spinner is there to take this
        // fiber (a futex call per level would cost the level server more than the level's own bookkeeping)
        if (sleepers_.load(std::memory_order_seq_cst) > 0 && nf > spinners_.load(std::memory_order_acquire)) wake(cv_);
    }

    // From inside a fiber: give the thread back until somebody calls make_ready(this fiber).
    static void park() {
        Fiber* f = current();
        sc_fiber_switch(&f->sp, *f->back);
    }
    // From inside a fiber: let the other ready fibers run first.
    static void yield() {
        Fiber* f = current();
        f->pool->make_ready(f, true);
        sc_fiber_switch(&f->sp, *f->back);
    }
    // fibers made ready the ordinary way and not taken yet (a fiber that polls for something may sleep instead of yielding
    // while this is zero)
    int ready_now() const { return n_front_.load(std::memory_order_acquire); }
    // fibers waiting in the `later` queue (regions in set-up that have yielded)
    int later_now() const { return n_later_.load(std::memory_order_acquire); }
    static Fiber*& current() { return tl_current(); }
    static bool in_fiber() { return tl_current() != nullptr; }

    // Stops the executor threads once every fiber has finished.  Fibers still parked at that point are abandoned
    // (their stacks are unmapped): the owner makes its fibers finish first.
    void shutdown() {
        {
            std::lock_guard<std::mutex> lk(sleep_mu_);
            if (stop_.load()) return;
            stop_.store(true);
        }
        cv_.notify_all();
        cv_long_.notify_all();
        for (auto& t : threads_) if (t.joinable()) t.join();
        for (Fiber* f : all_) { if (f->stack) munmap(f->stack, f->stack_bytes); delete f; }
        all_.clear();
    }

    // What the continuation threads do between two fibers and while they spin for one: the owner's way of finding out that
    // a parked fiber may run again without a thread of its own watching for it (the completion stamps of the levels in
    // flight).  It returns whether anything is still being waited for: a spinning thread then keeps spinning instead of
    // going to sleep (nobody else would wake the parked fibers).  Set before any fiber is made ready; called from several
    // threads at once.
    void set_poll(std::function<bool()> poll) { poll_ = std::move(poll); if (max_spinners_ < 1) max_spinners_ = 1; }

    // A fiber is about to park for something only the poll can see: somebody must be watching.  (Call after the thing to
    // watch has been published.)
    void ensure_poller() {
        if (spinners_.load(std::memory_order_seq_cst) == 0 && sleepers_.load(std::memory_order_seq_cst) > 0) wake(cv_);
    }

    // diagnostics / tests
    void set_diag(bool on) { diag_ = on; }
    // time stamp counter ticks the two kinds of executor have spent inside fibers, and how long the stretches were
    // (log2 buckets of ticks)
    uint64_t busy_ticks(bool lng) const { return (lng ? busy_long_ : busy_fast_).load(std::memory_order_relaxed); }
    long stretch_count(bool lng, int bucket) const { return hist_[lng ? 1 : 0][bucket].load(std::memory_order_relaxed); }
    long switches() const { return switches_.load(std::memory_order_relaxed); }
    int max_running() const { return max_running_.load(std::memory_order_relaxed); }

private:
    // a fiber that migrates between threads must not see a cached address of a thread_local: keep the access in a
    // function the compiler cannot fold across the context switch
    static __attribute__((noinline)) Fiber*& tl_current() {
        static thread_local Fiber* cur = nullptr;
        asm volatile("" ::: "memory");
        return cur;
    }
    static void entry() {
        Fiber* f = tl_current();
        f->body();
        f->finished.store(true, std::memory_order_release);
        // back to the executor for good (the executor of NOW: the fiber may have migrated since it started)
        f = tl_current();
        sc_fiber_switch(&f->sp, *f->back);
        __builtin_trap();
    }
    // The two queues are touched for a few nanoseconds at a time by the level server (a fiber per finished level) and by
    // every executor: a lock that never sleeps.  (They were under the mutex the executors also sleep on: eight executors
    // asking it for work at once put each other -- and the level server -- to sleep in the kernel, ~50 us per level at
    // 224 regions in flight.)
    void qlock() { qlk_.lock(); }
    void qunlock() { qlk_.unlock(); }
    // Wakes one sleeping executor.  Passing through the sleepers' mutex first: a sleeper that has announced itself and
    // checked the counters before they rose is inside wait() by the time this notifies.
    void wake(std::condition_variable& cv) {
        // (a thread that is on its way to sleep polls once more with the mutex held; what it finds it makes ready itself and
        // sees before it waits: it must not ask for the mutex again)
        if (!holds_sleep_mu()) { std::lock_guard<std::mutex> lk(sleep_mu_); }
        cv.notify_one();
    }
    static bool& holds_sleep_mu() { static thread_local bool h = false; return h; }
    // what a thread of this kind may take: the set-up threads take the long stretches first, the others never take them
    // when the pool is split
    bool has_work(bool lng) const {
        if (n_front_.load(std::memory_order_seq_cst) > 0) return true;
        return (lng || !split_) && n_later_.load(std::memory_order_seq_cst) > 0;
    }
    Fiber* take(bool lng) {
        if (!has_work(lng)) return nullptr;
        Fiber* f = nullptr;
        qlock();
        if (lng && !later_.empty()) { f = later_.front(); later_.pop_front(); n_later_.fetch_sub(1, std::memory_order_relaxed); }
        else if (!ready_.empty()) { f = ready_.front(); ready_.pop_front(); n_front_.fetch_sub(1, std::memory_order_relaxed); }
        else if (!lng && !split_ && !later_.empty()) { f = later_.front(); later_.pop_front(); n_later_.fetch_sub(1, std::memory_order_relaxed); }
        qunlock();
        return f;
    }
    void run(bool lng) {
        if (on_start_) on_start_();
        void* self_sp = nullptr;
        for (;;) {
            if (!lng && poll_) poll_();
            Fiber* f = take(lng);
            if (!f) {
                // Nothing ready (for this kind of thread).  A rank's CPU share is a quota of CPU TIME (a GPU box hands out 16
                // CPUs of its host as a cgroup bandwidth limit): an executor that spins spends the quota the regions'
                // bookkeeping needs.  So at most `max_spinners_` executors wait by spinning (they pick a level's continuation
                // up within a fraction of a microsecond); the others sleep and are woken when fibers queue up behind the spinners.
                if (!lng && spinners_.load(std::memory_order_acquire) < max_spinners_) {
                    spinners_.fetch_add(1, std::memory_order_acq_rel);
                    bool watch = false;
                    for (unsigned spins = 0; (spins < 6000 || watch) && !has_work(false) && !stop_.load(std::memory_order_relaxed); spins++) {
                        if (poll_ && (spins & 7u) == 7u) watch = poll_();
                        __builtin_ia32_pause();
                    }
                    spinners_.fetch_sub(1, std::memory_order_acq_rel);
                    if (has_work(false)) continue;
                }
                std::atomic<int>& sl = lng ? long_sleepers_ : sleepers_;
                std::unique_lock<std::mutex> lk(sleep_mu_);
                sl.fetch_add(1, std::memory_order_seq_cst);
                // (with a poll: the last thread that could watch does not go to sleep while something is waited for)
                if (stop_.load() && !has_work(lng)) { sl.fetch_sub(1, std::memory_order_seq_cst); return; }
                holds_sleep_mu() = true;
                const bool must_watch = !lng && poll_ && spinners_.load(std::memory_order_seq_cst) == 0 && poll_();
                holds_sleep_mu() = false;
                if (!has_work(lng) && !must_watch) {
                    (lng ? cv_long_ : cv_).wait_for(lk, std::chrono::milliseconds(lng ? 5 : 20));
                }
                sl.fetch_sub(1, std::memory_order_seq_cst);
                continue;
            }
            // more fibers wait than spinners stand by: bring a sleeping executor in
            if (sleepers_.load(std::memory_order_acquire) > 0 && n_front_.load(std::memory_order_acquire) > spinners_.load(std::memory_order_acquire)) wake(cv_);
            if (f->finished.load(std::memory_order_acquire)) continue;                              // (a stale entry: never resume a finished fiber)
            while (f->on_cpu.exchange(true, std::memory_order_acquire)) __builtin_ia32_pause();     // still switching out elsewhere
            const int r = running_.fetch_add(1, std::memory_order_relaxed) + 1;
            int m = max_running_.load(std::memory_order_relaxed);
            while (r > m && !max_running_.compare_exchange_weak(m, r, std::memory_order_relaxed)) {}
            f->back = &self_sp;
            tl_current() = f;
            const uint64_t c0 = diag_ ? __builtin_ia32_rdtsc() : 0;
            sc_fiber_switch(&self_sp, f->sp);
            if (diag_) { const uint64_t d = __builtin_ia32_rdtsc() - c0; (lng ? busy_long_ : busy_fast_).fetch_add(d, std::memory_order_relaxed); int b = 0; while ((d >> b) > 1 && b < 39) b++; hist_[lng][b].fetch_add(1, std::memory_order_relaxed); }
            tl_current() = nullptr;
            running_.fetch_sub(1, std::memory_order_relaxed);
            switches_.fetch_add(1, std::memory_order_relaxed);
            f->on_cpu.store(false, std::memory_order_release);
        }
    }

    std::function<void()> on_start_;
    std::function<bool()> poll_;
    std::vector<std::thread> threads_;
    std::mutex sleep_mu_;                        // the sleeping executors' (and all_'s); never held while a queue is touched
    std::condition_variable cv_, cv_long_;
    SpinLock qlk_;                               // the queues' lock
    std::deque<Fiber*> ready_, later_;
    std::vector<Fiber*> all_;
    std::atomic<int> n_front_{0}, n_later_{0}, sleepers_{0}, long_sleepers_{0}, spinners_{0}, running_{0}, max_running_{0};
    int max_spinners_ = 4;
    bool split_ = false;
    std::atomic<long> switches_{0};
    std::atomic<bool> stop_{false};
    bool diag_ = false;
    std::atomic<uint64_t> busy_fast_{0}, busy_long_{0};
    std::atomic<long> hist_[2][40] = {};
};

}  // namespace sc
