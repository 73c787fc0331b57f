// Host-side worker pool of a call slot: the O(#components) box geometry of a detector pass and the beam search fan out over it.
// Sized from what this PROCESS may use -- the scheduler affinity mask and the cgroup CPU quota -- not from the machine
// (std::thread::hardware_concurrency() reports every core of an 8-GPU host to each of its 8 ranks), optionally divided by the ranks that
// share the host (bbocr_config::host_threads), and created ONCE per slot: no thread is spawned per call.
#pragma once
#include <atomic>
#include <condition_variable>
#include <exception>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#include <sched.h>
#include <stdio.h>

// CPUs this process may run on: min(affinity mask, cgroup v2 cpu.max / v1 cfs quota), at least 1
inline int host_cpu_share() {
    int n = 0;
    cpu_set_t set;
    CPU_ZERO(&set);
    if (sched_getaffinity(0, sizeof(set), &set) == 0) n = CPU_COUNT(&set);
    if (n <= 0) n = (int)std::thread::hardware_concurrency();
    if (n <= 0) n = 1;
    auto quota = [&](const char* path, const char* path_period) -> double {
        double q = -1, per = -1;
        if (FILE* f = fopen(path, "r")) {
            char a[64] = {0}, b[64] = {0};
            const int got = fscanf(f, "%63s %63s", a, b);
            fclose(f);
            if (got >= 1 && a[0] != 'm' && a[0] != '-') q = atof(a);          // "max" / -1: unlimited
            if (got >= 2) per = atof(b);
        }
        if (path_period && q > 0)
            if (FILE* f = fopen(path_period, "r")) { if (fscanf(f, "%lf", &per) != 1) per = -1; fclose(f); }
        return (q > 0 && per > 0) ? q / per : -1.0;
    };
    double lim = quota("/sys/fs/cgroup/cpu.max", nullptr);
    if (lim <= 0) lim = quota("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us");
    if (lim > 0) n = std::min(n, std::max(1, (int)(lim + 0.5)));
    return n;
}

class HostPool {
public:
    explicit HostPool(int nthreads) {
        for (int i = 0; i < nthreads; ++i) workers_.emplace_back([this] { loop(); });
    }
    ~HostPool() {
        {
            std::lock_guard<std::mutex> lk(mu_);
            stop_ = true;
        }
        cv_.notify_all();
        for (auto& t : workers_) t.join();
    }
    HostPool(const HostPool&) = delete;
    HostPool& operator=(const HostPool&) = delete;
    int size() const { return (int)workers_.size() + 1; }       // the calling thread works too
    // fn(i) for i in [0, n); returns when all are done; the first exception is rethrown in the caller.  One job at a time (a slot has one caller).
    void parallel_for(int n, const std::function<void(int)>& fn) {
        if (n <= 0) return;
        if (workers_.empty() || n == 1) { for (int i = 0; i < n; ++i) fn(i); return; }
        {
            std::lock_guard<std::mutex> lk(mu_);
            fn_ = &fn; n_ = n; next_.store(0); active_ = (int)workers_.size(); err_ = nullptr; ++gen_;
        }
        cv_.notify_all();
        run();
        std::unique_lock<std::mutex> lk(mu_);
        done_cv_.wait(lk, [&] { return active_ == 0; });
        fn_ = nullptr;
        if (err_) std::rethrow_exception(err_);
    }

private:
    void run() {
        try {
            for (int i = next_.fetch_add(1); i < n_; i = next_.fetch_add(1)) (*fn_)(i);
        } catch (...) {
            std::lock_guard<std::mutex> lk(mu_);
            if (!err_) err_ = std::current_exception();
            next_.store(n_);                                     // nobody starts another item
        }
    }
    void loop() {
        unsigned long long seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return stop_ || gen_ != seen; });
                if (stop_) return;
                seen = gen_;
            }
            run();
            {
                std::lock_guard<std::mutex> lk(mu_);
                --active_;
            }
            done_cv_.notify_one();
        }
    }
    std::vector<std::thread> workers_;
    std::mutex mu_;
    std::condition_variable cv_, done_cv_;
    const std::function<void(int)>* fn_ = nullptr;
    int n_ = 0, active_ = 0;
    std::atomic<int> next_{0};
    unsigned long long gen_ = 0;
    bool stop_ = false;
    std::exception_ptr err_;
};
