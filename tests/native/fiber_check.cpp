// CPU check of rambl_amd/csrc/sc_fiber.hpp: R regions-as-fibers on T executor threads, a "server" thread that
// completes their levels.  Usage: fiber_check R T LEVELS [TL [poll]].  Prints one line of counters; exit 0 when every fiber
// has walked all its levels, no more than T fibers ever ran at once, and the process never had more than T + 2 threads.
// With a fifth argument the server only STAMPS the levels (as the GPU does) and the executors find the stamps themselves
// (FiberPool::set_poll, the way resident contexts run): a fiber raises its flag, asks for a watcher and parks.
#include <dirent.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <stdexcept>
#include <thread>
#include <vector>

#include "../../rambl_amd/csrc/sc_fiber.hpp"

using namespace sc;

static int thread_count() {
    int n = 0;
    if (DIR* d = opendir("/proc/self/task")) {
        while (dirent* e = readdir(d)) if (e->d_name[0] != '.') n++;
        closedir(d);
    }
    return n;
}

struct Region {
    Fiber* f = nullptr;
    std::atomic<int> state{0};       // 1: level posted, 2: done
    std::atomic<int> stamp{0};       // poll mode: the number of the last level the "device" has finished (a sequence, never reset)
    std::atomic<int> want{0};        // poll mode: the number of the level the fiber waits for
    std::atomic<unsigned char> flag{0};   // poll mode: parked until the stamp is seen
    long levels_done = 0;
    double acc = 0;                   // something on the fiber's stack frame must survive the migrations
};

int main(int argc, char** argv) {
    const int R = argc > 1 ? atoi(argv[1]) : 512, T = argc > 2 ? atoi(argv[2]) : 4, L = argc > 3 ? atoi(argv[3]) : 300;
    const int TL = argc > 4 ? atoi(argv[4]) : 0;            // of the T threads: those that take the `later` fibers first
    const bool poll = argc > 5;
    std::vector<Region> regs((size_t)R);
    std::mutex mu;
    std::vector<Region*> posted;
    std::atomic<int> finished{0};
    std::atomic<bool> stop{false};
    int max_threads = 0;
    {
        FiberPool pool(T, nullptr, TL);
        if (poll)
            pool.set_poll([&] {
                bool waiting = false;
                for (Region& r : regs) {
                    if (r.flag.load(std::memory_order_acquire) != 1) continue;
                    if (r.stamp.load(std::memory_order_acquire) != r.want.load(std::memory_order_acquire)) { waiting = true; continue; }
                    unsigned char one = 1;
                    if (!r.flag.compare_exchange_strong(one, 0)) continue;
                    // (the flag may by now be the one of the region's NEXT level -- another thread saw this stamp first and the
                    // region ran on: with the flag in hand, look again)
                    if (r.stamp.load(std::memory_order_acquire) != r.want.load(std::memory_order_acquire)) { r.flag.store(1); waiting = true; continue; }
                    r.state.store(2, std::memory_order_release);
                    pool.make_ready(r.f);
                }
                return waiting;
            });
        // the level server: takes what was posted, "runs" it, makes the fiber ready again
        std::thread server([&] {
            std::vector<Region*> mine;
            while (!stop.load()) {
                { std::lock_guard<std::mutex> lk(mu); mine.swap(posted); }
                for (Region* r : mine) {
                    if (poll) { r->stamp.store(r->want.load(std::memory_order_acquire), std::memory_order_release); continue; }      // the executors will see it
                    r->state.store(2, std::memory_order_release); pool.make_ready(r->f);
                }
                mine.clear();
                const int tc = thread_count();
                if (tc > max_threads) max_threads = tc;
                std::this_thread::yield();
            }
        });
        for (int i = 0; i < R; i++) {
            Region* r = &regs[(size_t)i];
            r->f = pool.create([&, r, i] {
                double local[64];                                  // lives on the fiber's stack across every park
                for (int k = 0; k < 64; k++) local[k] = i + k;
                long double ext = 0;
                int caught = 0;
                for (int lv = 0; lv < L; lv++) {
                    r->state.store(1, std::memory_order_release);
                    r->want.store(lv + 1, std::memory_order_release);
                    { std::lock_guard<std::mutex> lk(mu); posted.push_back(r); }
                    if (poll) { r->flag.store(1, std::memory_order_seq_cst); pool.ensure_poller(); }
                    // the protocol of the library: whoever completes the level makes the fiber ready exactly once, so the
                    // fiber parks exactly once per level -- also when the level is already done by now
                    FiberPool::park();
                    if (r->state.load(std::memory_order_acquire) != 2) { r->levels_done = -1000000; break; }
                    if (poll && r->stamp.load(std::memory_order_acquire) != lv + 1) { r->levels_done = -2000000; break; }      // woken before its level was done
                    local[lv & 63] += 1.0;
                    r->levels_done++;
                    // long double bookkeeping and a C++ exception thrown and caught inside the fiber, wherever it runs by now
                    ext += 1.0L / 3.0L;
                    if ((lv & 127) == 5) {
                        try { throw std::runtime_error("x"); } catch (const std::runtime_error&) { caught++; }
                    }
                    if ((lv & 31) == 0) FiberPool::yield();
                }
                for (int k = 0; k < 64; k++) r->acc += local[k];
                long double want = 0;
                for (int lv = 0; lv < L; lv++) want += 1.0L / 3.0L;
                int want_caught = 0;
                for (int lv = 0; lv < L; lv++) want_caught += (lv & 127) == 5;
                if (ext != want || caught != want_caught) r->levels_done = -1;       // (80-bit sums: the x87 control word came along)
                finished.fetch_add(1);
            }, 256 * 1024);
            pool.make_ready(r->f);
        }
        const auto t0 = std::chrono::steady_clock::now();
        while (finished.load() < R && std::chrono::steady_clock::now() - t0 < std::chrono::seconds(120))
            std::this_thread::sleep_for(std::chrono::milliseconds(1));
        stop = true;
        server.join();
        bool ok = finished.load() == R && pool.max_running() <= T && max_threads <= T + 2;
        for (int i = 0; i < R && ok; i++) {
            double want = 0;
            for (int k = 0; k < 64; k++) want += i + k;
            want += L;
            ok = regs[(size_t)i].levels_done == L && regs[(size_t)i].acc == want;
        }
        printf("{\"regions\": %d, \"threads\": %d, \"levels\": %d, \"finished\": %d, \"max_running\": %d, \"max_os_threads\": %d, \"switches\": %ld, \"ok\": %s}\n",
               R, T, L, finished.load(), pool.max_running(), max_threads, pool.switches(), ok ? "true" : "false");
        return ok ? 0 : 1;
    }
}
