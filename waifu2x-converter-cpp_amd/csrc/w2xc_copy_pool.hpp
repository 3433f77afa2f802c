// w2xc_copy_pool.hpp -- the host-side gather/scatter workers of the tile farm.
//
// The reference stitches block interiors into the output plane with cv::Mat::copyTo on one thread
// (/root/reference/src/convertRoutine.cpp:143-161) and spends modelUtility's nJob threads on the
// convolutions (src/modelHandler.cpp:42-69).  Here the convolutions are on the GPU, so nJob is spent
// where the host still has work: copying the caller's pageable planes into / out of the pinned
// staging rings that feed the H2D / D2H streams (SURVEY 8b "map nJob to host staging threads").
//
// A tiny persistent pool: copy_rows() splits one strided row copy into `parts` contiguous row ranges,
// the caller takes part in its own job, idle workers help.  Several device threads may submit at once.
#pragma once
#include <sched.h>
#include <pthread.h>
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#if defined(__x86_64__)
#include <immintrin.h>
#endif

namespace w2xc_host {

// Large row copies between the caller's planes and the staging rings never get re-read by this core: streaming (non-temporal)
// stores skip the read-for-ownership of every destination line, a third of the memory traffic of a plain memcpy.
#if defined(__x86_64__)
__attribute__((target("avx2"))) static inline void copy_stream_avx2(char *d, const char *s, size_t n)
{
    size_t head = (32 - ((uintptr_t)d & 31)) & 31;
    if (head > n) head = n;
    memcpy(d, s, head);
    d += head; s += head; n -= head;
    for (; n >= 128; n -= 128, d += 128, s += 128) {
        const __m256i a = _mm256_loadu_si256((const __m256i *)s), b = _mm256_loadu_si256((const __m256i *)(s + 32));
        const __m256i c = _mm256_loadu_si256((const __m256i *)(s + 64)), e = _mm256_loadu_si256((const __m256i *)(s + 96));
        _mm256_stream_si256((__m256i *)d, a);
        _mm256_stream_si256((__m256i *)(d + 32), b);
        _mm256_stream_si256((__m256i *)(d + 64), c);
        _mm256_stream_si256((__m256i *)(d + 96), e);
    }
    _mm_sfence();
    memcpy(d, s, n);
}
#endif
static inline void copy_bytes(char *d, const char *s, size_t n)
{
#if defined(__x86_64__)
    static const bool avx2 = __builtin_cpu_supports("avx2");
    if (avx2 && n >= (64u << 10)) { copy_stream_avx2(d, s, n); return; }
#endif
    memcpy(d, s, n);
}

class CopyPool {
public:
    static CopyPool &get()
    {
        static CopyPool *p = [] {
            // after fork() the child has this object but none of its (detached) worker threads, and mu_ may be held by a thread that
            // does not exist there: the child copies inline (nothing to lock, nothing to wake)
            pthread_atfork(nullptr, nullptr, [] { forked().store(true, std::memory_order_relaxed); });
            return new CopyPool();   // leaked on purpose: workers may outlive static destruction
        }();
        return *p;
    }
    // create the workers up to `n` now (callers that are about to bind threads to a NUMA node call this first)
    void reserve(int n) { if (n > 0 && !forked().load(std::memory_order_relaxed)) ensure_threads(n); }
    static std::atomic<bool> &forked()
    {
        static std::atomic<bool> f{false};
        return f;
    }

    // dst/src rows are `row_bytes` long, `ds` / `ss` bytes apart.  nthreads <= 1 copies inline.
    void copy_rows(char *dst, size_t ds, const char *src, size_t ss, size_t row_bytes, int rows, int nthreads)
    {
        if (rows <= 0 || row_bytes == 0) return;
        const size_t total = row_bytes * (size_t)rows;
        int parts = nthreads;
        if ((size_t)parts > total / (256u << 10)) parts = (int)(total / (256u << 10));   // >= 256 KiB per part
        if (parts > rows) parts = rows;
        if (parts <= 1 || forked().load(std::memory_order_relaxed)) {
            run(dst, ds, src, ss, row_bytes, 0, rows);
            return;
        }
        ensure_threads(parts - 1);
        auto j = std::make_shared<Job>();
        j->dst = dst; j->ds = ds; j->src = src; j->ss = ss; j->rb = row_bytes; j->rows = rows; j->parts = parts;
        {
            std::lock_guard<std::mutex> lk(mu_);
            q_.push_back(j);
            pending_.fetch_add(1, std::memory_order_release);
        }
        cv_.notify_all();
        work_on(*j);
        for (int spin = 0; spin < 5000 && j->done.load(std::memory_order_acquire) != j->parts; spin++) {   // (~0.2 ms, then sleep)
#if defined(__x86_64__)
            __builtin_ia32_pause();
#endif
        }
        std::unique_lock<std::mutex> lk(mu_);
        done_cv_.wait(lk, [&] { return j->done.load() == j->parts; });
    }

private:
    CopyPool() { have_base_ = sched_getaffinity(0, sizeof base_, &base_) == 0; }
    cpu_set_t base_;
    bool have_base_ = false;
    struct Job {
        char *dst; size_t ds; const char *src; size_t ss; size_t rb; int rows, parts;
        std::atomic<int> next{0}, done{0};
    };

    static void run(char *dst, size_t ds, const char *src, size_t ss, size_t rb, int r0, int r1)
    {
        if (ds == rb && ss == rb) {
            copy_bytes(dst + (size_t)r0 * rb, src + (size_t)r0 * rb, (size_t)(r1 - r0) * rb);
            return;
        }
        for (int r = r0; r < r1; r++) copy_bytes(dst + (size_t)r * ds, src + (size_t)r * ss, rb);
    }

    void work_on(Job &j)
    {
        for (;;) {
            const int p = j.next.fetch_add(1);
            if (p >= j.parts) return;
            const int r0 = (int)((long long)j.rows * p / j.parts), r1 = (int)((long long)j.rows * (p + 1) / j.parts);
            run(j.dst, j.ds, j.src, j.ss, j.rb, r0, r1);
            if (j.done.fetch_add(1) + 1 == j.parts) {
                std::lock_guard<std::mutex> lk(mu_);   // pairs with the submitter's predicate check
                done_cv_.notify_all();
            }
        }
    }

    void ensure_threads(int n)
    {
        std::lock_guard<std::mutex> lk(mu_);
        if (n > 31) n = 31;
        while ((int)th_.size() < n) {
            th_.emplace_back([this] { worker(); });
            th_.back().detach();
        }
    }

    void worker()
    {
        // The pool is global and serves every device's pipeline: a worker must not keep the NUMA binding of whichever (bound) thread happened to
        // create it.  Its affinity is reset to the mask the process had when the pool was built.
        if (have_base_) sched_setaffinity(0, sizeof base_, &base_);
        for (;;) {
            std::shared_ptr<Job> j;
            // jobs arrive in bursts (one per staged chunk, ~100 us apart while a plane streams through the rings): poll for
            // a short while before sleeping, a condition-variable wake-up costs more than the copy of a 256 KiB part
            for (int spin = 0; spin < 5000 && pending_.load(std::memory_order_acquire) == 0; spin++) {   // (~0.2 ms: two chunk intervals)
#if defined(__x86_64__)
                __builtin_ia32_pause();
#endif
            }
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return !q_.empty(); });
                j = q_.front();
                if (j->next.load() >= j->parts) {   // fully handed out: retire it from the queue
                    q_.pop_front();
                    pending_.fetch_sub(1, std::memory_order_release);
                    continue;
                }
            }
            work_on(*j);
        }
    }

    std::atomic<int> pending_{0};   // jobs in the queue (lock-free peek for the workers' polling phase)
    std::mutex mu_;
    std::condition_variable cv_, done_cv_;
    std::deque<std::shared_ptr<Job>> q_;
    std::vector<std::thread> th_;
};

}  // namespace w2xc_host
