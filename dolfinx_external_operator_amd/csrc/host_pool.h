// host_pool.h — worker threads for the host half of the DXO_MEM_HOST pipeline (plain C++17, no HIP types: also built on
// its own under ThreadSanitizer by tests/test_sanitizers.py).
#pragma once

#include <condition_variable>
#include <cstdint>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

struct dxo_host_pool {
    std::vector<std::thread> threads;
    std::mutex m;
    std::condition_variable cv_work, cv_done;
    const std::function<void(int64_t, int64_t)>* fn = nullptr;
    int64_t n = 0, grain = 1, next = 0;
    int active = 0;          // workers inside the current job
    uint64_t generation = 0;
    bool stop = false;

    void worker() {
        uint64_t seen = 0;
        std::unique_lock<std::mutex> lk(m);
        for (;;) {
            cv_work.wait(lk, [&] { return stop || generation != seen; });
            if (stop) return;
            seen = generation;
            ++active;
            while (next < n) {
                const int64_t b = next, e = b + grain < n ? b + grain : n;
                next = e;
                lk.unlock();
                (*fn)(b, e);
                lk.lock();
            }
            if (--active == 0) cv_done.notify_all();
        }
    }
};

inline void dxo_host_pool_destroy(dxo_host_pool* pool) {
    if (!pool) return;
    {
        std::lock_guard<std::mutex> lk(pool->m);
        pool->stop = true;
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
    const int hw = (int)std::thread::hardware_concurrency();
    if (hw > 0 && want > hw) want = hw;
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
        for (int i = 0; i < want - 1; ++i) pool->threads.emplace_back([p = pool] { p->worker(); });
    }
    dxo_host_pool* p = pool;
    std::unique_lock<std::mutex> lk(p->m);
    p->fn = &fn;
    p->n = n;
    p->grain = grain;
    p->next = 0;
    ++p->generation;
    p->cv_work.notify_all();
    while (p->next < p->n) {   // the calling thread works too
        const int64_t b = p->next, e = b + grain < n ? b + grain : n;
        p->next = e;
        lk.unlock();
        fn(b, e);
        lk.lock();
    }
    p->cv_done.wait(lk, [&] { return p->active == 0; });
    // a worker that wakes up late finds next == n and leaves at once; none is inside fn any more
    p->fn = nullptr;
}
