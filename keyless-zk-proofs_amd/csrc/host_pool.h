// host_pool.h -- the library's host worker threads (no HIP in here: tests/cpp/host_pool_check.cpp builds it with g++ and
// ThreadSanitizer)
#pragma once
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

// Host threads for the host-side loops that sit on a proof's critical path: packing the witness for the compact upload
// (prover.hip) and combining the per-window partial sums of an MSM (msm_api.hip).  ONE pool per PROCESS (round 4): a pool
// per context -- round 3 -- meant 12 threads per prover, i.e. 350 idle-or-fighting threads under a service that keeps four
// provers on each of eight GPUs; the reference has one TBB arena per process (multiexp.cpp:46).  Several callers (provers
// of different contexts) may run jobs at once: a job is a task counter on its caller's stack, the workers take tasks from
// the oldest job that still has some, and the caller works on its own job too, so a job never waits for a free worker.
struct k16_host_pool {
    struct Job {
        const std::function<void(unsigned)>* f = nullptr;
        unsigned                             n = 0;
        unsigned                             next = 0;      // guarded by mu
        std::atomic<unsigned>                done{0};
    };
    std::vector<std::thread> workers;
    std::mutex               mu;
    std::condition_variable  cv_go, cv_done;
    std::vector<Job*>        jobs; // jobs with unclaimed tasks, oldest first
    bool                     quit = false;
    explicit k16_host_pool(unsigned n_workers)
    {
        for (unsigned t = 0; t < n_workers; t++) workers.emplace_back([this] { loop(); });
    }
    unsigned width() const { return (unsigned)workers.size() + 1; }
    // next task of job j (mu held); removes the job from the list when it hands out the last one
    bool take(Job* j, unsigned* t)
    {
        if (j->next >= j->n) return false;
        *t = j->next++;
        if (j->next == j->n) {
            for (size_t i = 0; i < jobs.size(); i++)
                if (jobs[i] == j) {
                    jobs.erase(jobs.begin() + i);
                    break;
                }
        }
        return true;
    }
    void finish_one(Job* j)
    {
        const unsigned n = j->n; // the job may be gone as soon as the count below reaches n
        if (j->done.fetch_add(1) + 1 == n) {
            std::lock_guard<std::mutex> lk(mu);
            cv_done.notify_all();
        }
    }
    void loop()
    {
        for (;;) {
            Job*     j = nullptr;
            unsigned t = 0;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_go.wait(lk, [&] { return quit || !jobs.empty(); });
                if (quit) return;
                j = jobs.front();
                if (!take(j, &t)) continue;
            }
            (*j->f)(t);
            finish_one(j);
        }
    }
    void run(unsigned tasks, const std::function<void(unsigned)>& f) // f(task) for task in [0, tasks); returns when all are done
    {
        if (tasks == 0) return;
        Job j;
        j.f = &f;
        j.n = tasks;
        {
            std::lock_guard<std::mutex> lk(mu);
            jobs.push_back(&j);
        }
        cv_go.notify_all();
        for (;;) {
            unsigned t;
            {
                std::lock_guard<std::mutex> lk(mu);
                if (!take(&j, &t)) break;
            }
            f(t);
            j.done.fetch_add(1);
        }
        std::unique_lock<std::mutex> lk(mu);
        cv_done.wait(lk, [&] { return j.done.load() == tasks; });
    }
    ~k16_host_pool()
    {
        {
            std::lock_guard<std::mutex> lk(mu);
            quit = true;
        }
        cv_go.notify_all();
        for (auto& w : workers) w.join();
    }
};

