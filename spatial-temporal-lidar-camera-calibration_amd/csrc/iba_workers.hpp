// The worker handshake of iba_group (one issuing thread per device), HOST ONLY — no HIP in here, so that the same code runs under
// -fsanitize=thread / address in the CPU tier (csrc/san/workers_selftest.cpp).
//
//   run_all(fn, meanwhile)  hands fn(i) to every worker, runs `meanwhile` on the calling thread, waits for all of them, reports
//                           the first failure. Workers spin for a short while after a job (an optimiser calls back within
//                           microseconds) and sleep on a condition variable otherwise.
//   meet(i, ok)             a barrier among the workers INSIDE a job: every worker calls it exactly once per job that uses it and
//                           learns whether ALL of them were ok. iba_group puts it between a device's launch chain and its
//                           collective: a worker that failed must not leave its peers alone in an all-reduce that can never
//                           complete (round 3: the caller hung forever).
//   raise_abort() / aborted()  a flag any worker may raise after the barrier (the collective itself failed to enqueue, a bounded
//                           wait ran out); the others see it in their wait loop and abandon the collective.
#pragma once
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/iba_mi355x.h"

namespace iba {

inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#else
    std::this_thread::yield();
#endif
}

class WorkerPool {
public:
    static constexpr std::chrono::microseconds kSpinFor{200};

    ~WorkerPool() { stop(); }
    int size() const { return n_; }

    // thread_init(i) runs once on worker i before its first job (iba_group: hipSetDevice)
    void start(int n, std::function<void(int)> thread_init) {
        n_ = n;
        wstatus_.assign((size_t)n, IBA_OK); werr_.assign((size_t)n, ""); secondary_.assign((size_t)n, 0);
        for (int i = 0; i < n; ++i) workers_.emplace_back([this, i, thread_init]() { if (thread_init) thread_init(i); loop(i); });
    }

    void stop() {
        if (workers_.empty()) return;
        { std::lock_guard<std::mutex> lk(mu_); quit_.store(true, std::memory_order_release); }
        cv_job_.notify_all();
        for (auto& t : workers_) if (t.joinable()) t.join();
        workers_.clear();
    }

    // a worker's failure: message kept per worker, status returned from the job. secondary: the worker only gave up because a PEER
    // failed (run_all reports a primary failure when there is one)
    iba_status fail(int i, iba_status s, const std::string& m, bool secondary = false) { werr_[(size_t)i] = m; secondary_[(size_t)i] = secondary ? 1 : 0; return s; }
    const std::string& error_of(int i) const { return werr_[(size_t)i]; }
    iba_status status_of(int i) const { return wstatus_[(size_t)i]; }

    // returns the index of the first worker that failed of its own (else of the first that gave up because of a peer), -1 if none did
    int run_all(std::function<iba_status(int)> fn, const std::function<void()>& meanwhile = nullptr) {
        job_ = std::move(fn);
        std::fill(secondary_.begin(), secondary_.end(), 0);
        arrived_.store(0, std::memory_order_relaxed); not_ok_.store(0, std::memory_order_relaxed); abort_.store(false, std::memory_order_relaxed);
        pending_.store(n_, std::memory_order_release);
        { std::lock_guard<std::mutex> lk(mu_); gen_.fetch_add(1, std::memory_order_acq_rel); }
        cv_job_.notify_all();
        if (meanwhile) meanwhile();
        const auto t0 = std::chrono::steady_clock::now();
        int polls = 0;
        while (pending_.load(std::memory_order_acquire) != 0) {
            cpu_relax();
            if ((++polls & 63) == 0 && std::chrono::steady_clock::now() - t0 > kSpinFor) {
                std::unique_lock<std::mutex> lk(mu_);
                cv_done_.wait(lk, [&]() { return pending_.load(std::memory_order_acquire) == 0; });
            }
        }
        for (int i = 0; i < n_; ++i) if (wstatus_[(size_t)i] != IBA_OK && !secondary_[(size_t)i]) return i;
        for (int i = 0; i < n_; ++i) if (wstatus_[(size_t)i] != IBA_OK) return i;
        return -1;
    }

    // barrier among the workers of the running job; true when every worker arrived with ok == true
    bool meet(int /*i*/, bool ok) {
        if (!ok) not_ok_.fetch_add(1, std::memory_order_acq_rel);
        arrived_.fetch_add(1, std::memory_order_acq_rel);
        int polls = 0;
        while (arrived_.load(std::memory_order_acquire) < n_) { cpu_relax(); if ((++polls & 1023) == 0) std::this_thread::yield(); }
        return not_ok_.load(std::memory_order_acquire) == 0;
    }
    void raise_abort() { abort_.store(true, std::memory_order_release); }
    bool aborted() const { return abort_.load(std::memory_order_acquire); }

private:
    void loop(int i) {
        uint64_t seen = 0;
        for (;;) {
            const auto t0 = std::chrono::steady_clock::now();
            int polls = 0;
            while (gen_.load(std::memory_order_acquire) == seen && !quit_.load(std::memory_order_acquire)) {
                cpu_relax();
                if ((++polls & 63) == 0 && std::chrono::steady_clock::now() - t0 > kSpinFor) {
                    std::unique_lock<std::mutex> lk(mu_);
                    cv_job_.wait(lk, [&]() { return gen_.load(std::memory_order_acquire) != seen || quit_.load(std::memory_order_acquire); });
                }
            }
            if (quit_.load(std::memory_order_acquire) && gen_.load(std::memory_order_acquire) == seen) return;
            seen = gen_.load(std::memory_order_acquire);
            wstatus_[(size_t)i] = job_(i);
            if (pending_.fetch_sub(1, std::memory_order_acq_rel) == 1) { std::lock_guard<std::mutex> lk(mu_); cv_done_.notify_all(); }
        }
    }

    int n_ = 0;
    std::vector<std::thread> workers_;
    std::vector<iba_status> wstatus_;
    std::vector<std::string> werr_;
    std::vector<char> secondary_;
    std::function<iba_status(int)> job_;
    std::mutex mu_;
    std::condition_variable cv_job_, cv_done_;
    std::atomic<uint64_t> gen_{0};
    std::atomic<int> pending_{0}, arrived_{0}, not_ok_{0};
    std::atomic<bool> quit_{false}, abort_{false};
};

}  // namespace iba
