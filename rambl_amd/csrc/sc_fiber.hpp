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
#include <ucontext.h>

#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <cstdlib>
#include <deque>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace sc {

class FiberPool;

struct Fiber {
    ucontext_t ctx{};
    void* stack = nullptr;
    size_t stack_bytes = 0;
    std::function<void()> body;
    // true from the moment an executor switches the fiber in until that executor is back on its own stack: whoever
    // makes a parked fiber ready may do so before the fiber has finished switching out, and the executor that picks
    // it up waits for this flag (a few nanoseconds) instead of running a stack that is still in use
    std::atomic<bool> on_cpu{false};
    std::atomic<bool> finished{false};
    FiberPool* pool = nullptr;
    ucontext_t* back = nullptr;         // the executor context the fiber returns to when it parks
};

class FiberPool {
public:
    // n_threads executor threads; `on_thread_start` runs once on each (hipSetDevice)
    explicit FiberPool(int n_threads, std::function<void()> on_thread_start = nullptr)
        : on_start_(std::move(on_thread_start)) {
        if (n_threads < 1) n_threads = 1;
        for (int i = 0; i < n_threads; i++) threads_.emplace_back([this] { run(); });
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
        getcontext(&f->ctx);
        f->ctx.uc_stack.ss_sp = (char*)p + page;
        f->ctx.uc_stack.ss_size = stack_bytes;
        f->ctx.uc_link = nullptr;
        const uintptr_t v = (uintptr_t)f;
        makecontext(&f->ctx, (void (*)())&FiberPool::entry, 2, (unsigned)(v & 0xffffffffu), (unsigned)(v >> 32));
        { std::lock_guard<std::mutex> lk(mu_); all_.push_back(f); }
        return f;
    }

    // The fiber may run (again).  Callable from any thread, also before the fiber has finished parking.  Every
    // make_ready must be matched by exactly one park() / the fiber's start: a fiber that is made ready twice for one
    // park would be resumed a second time from wherever it parks next.
    // `later`: behind every fiber that was made ready the ordinary way -- for long, CPU-bound stretches (the set-up of
    // a new region) that must not delay the short continuations of the regions whose levels are coming back.
    void make_ready(Fiber* f, bool later = false) {
        {
            std::lock_guard<std::mutex> lk(mu_);
            (later ? later_ : ready_).push_back(f);
        }
        n_ready_.fetch_add(1, std::memory_order_release);
        if (sleepers_.load(std::memory_order_acquire) > 0) cv_.notify_one();
    }

    // From inside a fiber: give the thread back until somebody calls make_ready(this fiber).
    static void park() {
        Fiber* f = current();
        swapcontext(&f->ctx, f->back);
    }
    // From inside a fiber: let the other ready fibers run first.
    static void yield() {
        Fiber* f = current();
        f->pool->make_ready(f, true);
        swapcontext(&f->ctx, f->back);
    }
    static Fiber*& current() { return tl_current(); }
    static bool in_fiber() { return tl_current() != nullptr; }

    // Stops the executor threads once every fiber has finished.  Fibers still parked at that point are abandoned
    // (their stacks are unmapped): the owner makes its fibers finish first.
    void shutdown() {
        {
            std::lock_guard<std::mutex> lk(mu_);
            if (stop_) return;
            stop_ = true;
        }
        cv_.notify_all();
        for (auto& t : threads_) if (t.joinable()) t.join();
        for (Fiber* f : all_) { if (f->stack) munmap(f->stack, f->stack_bytes); delete f; }
        all_.clear();
    }

    // diagnostics / tests
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
    static void entry(unsigned lo, unsigned hi) {
        Fiber* f = (Fiber*)((uintptr_t)lo | ((uintptr_t)hi << 32));
        f->body();
        f->finished.store(true, std::memory_order_release);
        // back to the executor for good (re-read the executor context: the fiber may have migrated since it started)
        setcontext(f->back);
    }
    void run() {
        if (on_start_) on_start_();
        ucontext_t self{};
        unsigned spins = 0;
        for (;;) {
            Fiber* f = nullptr;
            if (n_ready_.load(std::memory_order_acquire) > 0) {
                std::lock_guard<std::mutex> lk(mu_);
                if (!ready_.empty()) { f = ready_.front(); ready_.pop_front(); n_ready_.fetch_sub(1, std::memory_order_relaxed); }
                else if (!later_.empty()) { f = later_.front(); later_.pop_front(); n_ready_.fetch_sub(1, std::memory_order_relaxed); }
            }
            if (!f) {
                // nothing ready: spin briefly (a level lasts ~0.5 ms, completions arrive all the time under load), then sleep
                if (++spins < 2000) { __builtin_ia32_pause(); continue; }
                std::unique_lock<std::mutex> lk(mu_);
                if (stop_ && ready_.empty() && later_.empty()) return;
                if (ready_.empty() && later_.empty()) {
                    sleepers_.fetch_add(1, std::memory_order_release);
                    cv_.wait_for(lk, std::chrono::milliseconds(50), [&] { return stop_ || !ready_.empty() || !later_.empty(); });
                    sleepers_.fetch_sub(1, std::memory_order_release);
                }
                spins = 0;
                continue;
            }
            spins = 0;
            if (f->finished.load(std::memory_order_acquire)) continue;                              // (a stale entry: never resume a finished fiber)
            while (f->on_cpu.exchange(true, std::memory_order_acquire)) __builtin_ia32_pause();     // still switching out elsewhere
            const int r = running_.fetch_add(1, std::memory_order_relaxed) + 1;
            int m = max_running_.load(std::memory_order_relaxed);
            while (r > m && !max_running_.compare_exchange_weak(m, r, std::memory_order_relaxed)) {}
            f->back = &self;
            tl_current() = f;
            swapcontext(&self, &f->ctx);
            tl_current() = nullptr;
            running_.fetch_sub(1, std::memory_order_relaxed);
            switches_.fetch_add(1, std::memory_order_relaxed);
            f->on_cpu.store(false, std::memory_order_release);
        }
    }

    std::function<void()> on_start_;
    std::vector<std::thread> threads_;
    std::mutex mu_;
    std::condition_variable cv_;
    std::deque<Fiber*> ready_, later_;
    std::vector<Fiber*> all_;
    std::atomic<int> n_ready_{0}, sleepers_{0}, running_{0}, max_running_{0};
    std::atomic<long> switches_{0};
    bool stop_ = false;
};

}  // namespace sc
