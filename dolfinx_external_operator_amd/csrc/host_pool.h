// host_pool.h — worker threads for the host half of the DXO_MEM_HOST pipeline (plain C++17, no HIP types: also built on
// its own under ThreadSanitizer by tests/test_sanitizers.py).
//
// A job is fn(begin, end) over [0, n) in ranges of `grain`; ranges are handed out through one atomic counter (no lock on
// the hot path). Between jobs a worker first SPINS on the generation counter for up to DXO_POOL_SPIN_US and only then
// sleeps on the condition variable: the pipeline hands over a chunk every few hundred microseconds, and waking 31
// sleeping threads through the futex cost more than rebuilding the chunk (measured on the GPU box's 2 x EPYC 9575F,
// 10^7 points, 32 threads: 23-30 ms as 153 jobs of 2^16 points against 12 ms as one job — scripts/exp/archive/host_rebuild_bench.hip).
// Every worker checks in for every job, so when dxo_pool_parallel_for returns no thread is inside fn or can still read
// the job's fields.
#pragma once

#include <emmintrin.h>
#include <sched.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#ifndef DXO_POOL_SPIN_US
#define DXO_POOL_SPIN_US 1000
#endif

// CPUs this process may actually use: the affinity mask, and the cgroup CPU quota when there is one (a container with a
// quota of 64 CPUs on a 256-thread host reports 256 from hardware_concurrency(); 64 spinning workers plus the runtime's
// own threads then exhaust the quota and the whole process is throttled for the rest of every 100 ms period — measured:
// a 13 ms call took 100 ms).
inline int dxo_host_cpu_budget() {
    if (const char* e = std::getenv("DXO_HOST_CPU_BUDGET")) {   // explicit override (experiments, unusual schedulers)
        const int k = std::atoi(e);
        if (k > 0) return k;
    }
    int cpus = (int)std::thread::hardware_concurrency();
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof(set), &set) == 0) {
        const int k = CPU_COUNT(&set);
        if (k > 0 && (cpus <= 0 || k < cpus)) cpus = k;
    }
    long long quota = -1, period = -1;
    if (FILE* f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {   // cgroup v2: "<quota|max> <period>"
        char q[32] = {0};
        if (std::fscanf(f, "%31s %lld", q, &period) == 2 && q[0] != 'm') quota = std::atoll(q);
        std::fclose(f);
    } else if (FILE* g = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {   // cgroup v1
        if (std::fscanf(g, "%lld", &quota) != 1) quota = -1;
        std::fclose(g);
        if (FILE* h = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
            if (std::fscanf(h, "%lld", &period) != 1) period = -1;
            std::fclose(h);
        }
    }
    if (quota > 0 && period > 0) {
        const int k = (int)((quota + period - 1) / period);
        if (k > 0 && (cpus <= 0 || k < cpus)) cpus = k;
    }
    return cpus > 0 ? cpus : 1;
}

struct dxo_host_pool {
    std::vector<std::thread> threads;
    std::mutex m;
    std::condition_variable cv_work, cv_done;
    // the job: plain fields, published by the release increment of `generation`
    const std::function<void(int64_t, int64_t)>* fn = nullptr;
    int64_t n = 0, grain = 1;
    std::atomic<int64_t> next{0};
    std::atomic<uint64_t> generation{0};
    std::atomic<int> checked_in{0};   // workers that are through with the current generation
    std::atomic<bool> stop{false};

    void run_ranges() {
        for (;;) {
            const int64_t b = next.fetch_add(grain, std::memory_order_relaxed);
            if (b >= n) return;
            (*fn)(b, b + grain < n ? b + grain : n);
        }
    }

    void worker() {
        uint64_t seen = 0;
        for (;;) {
            bool got = false;
            const auto t0 = std::chrono::steady_clock::now();
            for (int spin = 0;; ++spin) {
                if (stop.load(std::memory_order_acquire)) return;
                if (generation.load(std::memory_order_acquire) != seen) {
                    got = true;
                    break;
                }
                _mm_pause();
                if ((spin & 63) == 63 &&
                    std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count() >= DXO_POOL_SPIN_US)
                    break;
            }
            if (!got) {
                std::unique_lock<std::mutex> lk(m);
                cv_work.wait(lk, [&] { return stop.load(std::memory_order_acquire) || generation.load(std::memory_order_acquire) != seen; });
                if (stop.load(std::memory_order_acquire)) return;
            }
            seen = generation.load(std::memory_order_acquire);
            run_ranges();
            if (checked_in.fetch_add(1, std::memory_order_acq_rel) + 1 == (int)threads.size()) {
                { std::lock_guard<std::mutex> lk(m); }   // the caller is either before its predicate check or inside wait()
                cv_done.notify_one();
            }
        }
    }
};

inline void dxo_host_pool_destroy(dxo_host_pool* pool) {
    if (!pool) return;
    {
        std::lock_guard<std::mutex> lk(pool->m);
        pool->stop.store(true, std::memory_order_release);
    }
    pool->cv_work.notify_all();
    for (auto& t : pool->threads) t.join();
    delete pool;
}

// Run fn(begin, end) over [0, n) in ranges of `grain` on `want` threads (the caller is one of them); returns when all
// ranges are done. `pool` is created on first use and re-created when `want` changes. One caller at a time per pool.
inline void dxo_pool_parallel_for(dxo_host_pool*& pool, int want, int64_t n, int64_t grain,
                                  const std::function<void(int64_t, int64_t)>& fn) {
    if (n <= 0) return;
    if (grain < 1) grain = 1;
    static const int budget = dxo_host_cpu_budget();
    const int room = budget > 3 ? budget - 2 : 1;   // the caller's pipeline keeps two more threads busy (enqueue + stage)
    if (want > room) want = room;
    if (want <= 1 || n <= grain) {
        fn(0, n);
        return;
    }
    if (pool && (int)pool->threads.size() != want - 1) {
        dxo_host_pool_destroy(pool);
        pool = nullptr;
    }
    if (!pool) {
        pool = new dxo_host_pool();
        pool->threads.reserve(want - 1);   // threads.size() is read by the workers: fixed before the first job
        for (int i = 0; i < want - 1; ++i) pool->threads.emplace_back();
        for (int i = 0; i < want - 1; ++i) pool->threads[i] = std::thread([p = pool] { p->worker(); });
    }
    dxo_host_pool* p = pool;
    {
        std::lock_guard<std::mutex> lk(p->m);
        p->fn = &fn;
        p->n = n;
        p->grain = grain;
        p->next.store(0, std::memory_order_relaxed);
        p->checked_in.store(0, std::memory_order_relaxed);
        p->generation.fetch_add(1, std::memory_order_release);
    }
    p->cv_work.notify_all();
    p->run_ranges();   // the calling thread works too
    const int workers = (int)p->threads.size();
    for (int spin = 0; spin < 4096 && p->checked_in.load(std::memory_order_acquire) != workers; ++spin) _mm_pause();
    if (p->checked_in.load(std::memory_order_acquire) != workers) {
        std::unique_lock<std::mutex> lk(p->m);
        p->cv_done.wait(lk, [&] { return p->checked_in.load(std::memory_order_acquire) == workers; });
    }
    p->fn = nullptr;
}
