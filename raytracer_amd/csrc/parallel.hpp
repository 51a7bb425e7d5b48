// parallel.hpp -- a small fork-join helper for the host half of the scene upload (BVH build, flatten, record loops).
// Header-only, no HIP: it also runs under the host sanitizers (tests/host_san).
//
// run(n_chunks, fn) calls fn(chunk) for every chunk in [0, n_chunks), chunks handed out dynamically to the pool's threads;
// the calling thread takes part.  Nothing escapes a worker thread: the first exception is kept and rethrown by run() on
// the calling thread (so std::bad_alloc still reaches the ABI guard), and chunks not yet started are skipped after it.
// What is computed must not depend on which thread runs which chunk: callers write per-chunk results and combine them
// in chunk order (the builder's determinism for any thread count rests on that).
#pragma once
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <exception>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace rt {

class ForkJoin {
public:
    explicit ForkJoin(int threads) {
        if (threads > 1) th_.reserve((size_t)threads - 1);
        try { for (int t = 1; t < threads; ++t) th_.emplace_back([this] { loop(); }); }
        catch (...) {}                                   // fewer helpers (std::system_error / bad_alloc): still correct
    }
    ~ForkJoin() {
        { std::lock_guard<std::mutex> g(mu_); quit_ = true; }
        cv_.notify_all();
        for (auto& t : th_) t.join();
    }
    ForkJoin(const ForkJoin&) = delete;
    ForkJoin& operator=(const ForkJoin&) = delete;
    int size() const { return (int)th_.size() + 1; }

    template <class F>
    void run(uint32_t n_chunks, F&& fn) {
        if (n_chunks == 0) return;
        if (th_.empty() || n_chunks == 1) { for (uint32_t i = 0; i < n_chunks; ++i) fn(i); return; }
        const std::function<void(uint32_t)> f = std::ref(fn);
        {
            std::lock_guard<std::mutex> g(mu_);
            job_ = &f; total_ = n_chunks; next_.store(0); failed_.store(false); fail_ = nullptr;
            active_ = (int)th_.size(); ++gen_;
        }
        cv_.notify_all();
        work();
        std::unique_lock<std::mutex> g(mu_);
        done_.wait(g, [&] { return active_ == 0; });
        job_ = nullptr;
        if (fail_) { std::exception_ptr e = fail_; fail_ = nullptr; g.unlock(); std::rethrow_exception(e); }
    }

    // fn(begin, end) over [0, n) in pieces of about `grain` elements
    template <class F>
    void range(uint64_t n, uint64_t grain, F&& fn) {
        if (n == 0) return;
        if (grain == 0) grain = 1;
        const uint64_t chunks = (n + grain - 1) / grain;
        const uint32_t nc = (uint32_t)(chunks > 0xffffffffull ? 0xffffffffull : chunks);
        const uint64_t per = (n + nc - 1) / nc;
        run(nc, [&](uint32_t c) { const uint64_t b = (uint64_t)c * per, e = b + per < n ? b + per : n; if (b < e) fn(b, e); });
    }

private:
    void work() {
        for (;;) {
            const uint32_t i = next_.fetch_add(1);
            if (i >= total_) return;
            if (failed_.load()) continue;
            try { (*job_)(i); }
            catch (...) {
                std::lock_guard<std::mutex> g(mu_);
                if (!fail_) fail_ = std::current_exception();
                failed_.store(true);
            }
        }
    }
    void loop() {
        uint64_t seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> g(mu_);
                cv_.wait(g, [&] { return quit_ || gen_ != seen; });
                if (quit_) return;
                seen = gen_;
            }
            work();
            { std::lock_guard<std::mutex> g(mu_); if (--active_ == 0) done_.notify_one(); }
        }
    }
    std::vector<std::thread> th_;
    std::mutex mu_;
    std::condition_variable cv_, done_;
    const std::function<void(uint32_t)>* job_ = nullptr;
    uint32_t total_ = 0;
    std::atomic<uint32_t> next_{0};
    std::atomic<bool> failed_{false};
    std::exception_ptr fail_;
    int active_ = 0;
    uint64_t gen_ = 0;
    bool quit_ = false;
};

inline int host_threads(int asked) {
    if (asked > 0) return asked;
    const unsigned hw = std::thread::hardware_concurrency();
    return (int)(hw == 0 ? 1u : (hw > 16u ? 16u : hw));
}

}  // namespace rt
