// CPU-tier self-test of iba_group's worker handshake (iba_workers.hpp) behind stub jobs — built with -fsanitize=thread and with
// -fsanitize=address,undefined by `make -C csrc san`, run by tests/test_sanitizers_cpu.py. What iba_group does per call is
// reproduced with the GPU work replaced by arithmetic: the calling thread writes a shared block in two halves (values, then —
// while the workers run — derivatives behind a release flag), every worker reads its half-copies, meets its peers at the barrier,
// "reduces", and the caller sums the per-worker results. Failure paths: a worker that fails before the barrier (nobody may enter
// the collective), a worker that raises the abort flag after it (the others must leave their wait).
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../iba_workers.hpp"

using namespace iba;

struct Block { double values[64]; double derivs[64]; };

static int failures = 0;
#define CHECK(c) do { if (!(c)) { std::fprintf(stderr, "workers_selftest: %s failed at line %d\n", #c, __LINE__); ++failures; } } while (0)

int main(int argc, char** argv) {
    const int n = argc > 1 ? std::atoi(argv[1]) : 4;
    const int calls = argc > 2 ? std::atoi(argv[2]) : 3000;
    WorkerPool pool;
    std::vector<int> inited((size_t)n, 0);
    pool.start(n, [&](int i) { inited[(size_t)i] = 1; });
    Block blk;
    std::atomic<int> derivs_ready{0};
    std::vector<double> part((size_t)n, 0.0);
    std::vector<int> entered((size_t)n, 0);
    for (int c = 0; c < calls; ++c) {
        const int fail_rank = (c % 7 == 3) ? c % n : -1;        // fails before the barrier
        const int abort_rank = (c % 11 == 5) ? (c / 11) % n : -1;   // raises the abort flag after it
        for (int k = 0; k < 64; ++k) blk.values[k] = c + k;       // the value half, before the workers are released
        derivs_ready.store(0, std::memory_order_release);
        std::fill(entered.begin(), entered.end(), 0);
        const int bad = pool.run_all([&](int i) -> iba_status {
            double v = 0;
            for (int k = 0; k < 64; ++k) v += blk.values[k];      // (only the value half: the caller is writing the other)
            iba_status mine = IBA_OK;
            if (i == fail_rank) mine = pool.fail(i, IBA_ERR_STATE, "injected");
            const bool all_ok = pool.meet(i, mine == IBA_OK);
            if (!all_ok) return mine != IBA_OK ? mine : pool.fail(i, IBA_ERR_STATE, "peer failed", true);
            entered[(size_t)i] = 1;                                // "the collective"
            if (i == abort_rank) { pool.raise_abort(); return pool.fail(i, IBA_ERR_HIP, "collective did not enqueue"); }
            while (derivs_ready.load(std::memory_order_acquire) == 0) cpu_relax();
            double d = 0;
            for (int k = 0; k < 64; ++k) d += blk.derivs[k];
            // "wait for the stream": leaves at once when a peer has raised the abort flag
            if (abort_rank >= 0) { const auto t0 = std::chrono::steady_clock::now(); while (!pool.aborted()) { cpu_relax(); if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(5)) return pool.fail(i, IBA_ERR_STATE, "abort flag never seen"); } return pool.fail(i, IBA_ERR_STATE, "abandoned", true); }
            part[(size_t)i] = v + d;
            return IBA_OK;
        }, [&]() {
            for (int k = 0; k < 64; ++k) blk.derivs[k] = 2.0 * (c + k);   // the derivative half, while the workers run
            derivs_ready.store(1, std::memory_order_release);
        });
        if (fail_rank >= 0) {
            CHECK(bad == fail_rank);   // the primary failure is the one reported, whatever its index
            CHECK(pool.status_of(fail_rank) == IBA_ERR_STATE && pool.error_of(fail_rank) == "injected");
            for (int i = 0; i < n; ++i) CHECK(entered[(size_t)i] == 0);   // nobody entered the collective
        } else if (abort_rank >= 0) {
            CHECK(bad == abort_rank);
            for (int i = 0; i < n; ++i) CHECK(pool.status_of(i) != IBA_OK);
            CHECK(pool.error_of(abort_rank) == "collective did not enqueue");
        } else {
            CHECK(bad == -1);
            double want = 0; for (int k = 0; k < 64; ++k) want += 3.0 * (c + k);
            for (int i = 0; i < n; ++i) CHECK(part[(size_t)i] == want);
        }
        if (c % 500 == 499) std::this_thread::sleep_for(std::chrono::milliseconds(2));   // let the workers fall asleep on the condition variable now and then
    }
    for (int i = 0; i < n; ++i) CHECK(inited[(size_t)i] == 1);
    pool.stop();
    pool.stop();   // idempotent
    std::printf("workers_selftest: %d workers, %d calls, %d failure(s)\n", n, calls, failures);
    return failures ? 1 : 0;
}
