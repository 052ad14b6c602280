// tests/cpp/host_pool_check.cpp -- CPU check of the process-wide host pool (keyless-zk-proofs_amd/csrc/host_pool.h): several
// callers run jobs at the same time (provers of different contexts packing witnesses / combining windows), every task runs
// exactly once, run() returns only when its own job is complete, jobs of zero and one task work, and a pool without workers
// degrades to the caller's loop.  Built with -fsanitize=thread by tests/test_host_pool.py.  Prints "ok <tasks>".
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include "host_pool.h"

static int run_case(unsigned workers, unsigned callers, unsigned rounds)
{
    k16_host_pool                         pool(workers);
    std::atomic<unsigned long>            total{0};
    std::atomic<int>                      bad{0};
    std::vector<std::thread>              th;
    for (unsigned c = 0; c < callers; c++)
        th.emplace_back([&, c] {
            for (unsigned r = 0; r < rounds; r++) {
                const unsigned        tasks = (c * 7 + r * 13) % 40; // includes 0 and 1
                std::vector<unsigned> hit(tasks, 0);
                unsigned long         local = 0;
                std::mutex            mu;
                pool.run(tasks, [&](unsigned t) {
                    hit[t]++; // each task index belongs to exactly one thread: no lock
                    unsigned long x = 0;
                    for (unsigned k = 0; k < 200 + 50 * (t % 5); k++) x += k * (t + 1);
                    std::lock_guard<std::mutex> lk(mu);
                    local += x ? 1 : 1;
                });
                // run() has returned: every task of THIS job is done and visible
                for (unsigned t = 0; t < tasks; t++)
                    if (hit[t] != 1) bad++;
                if (local != tasks) bad++;
                total += tasks;
            }
        });
    for (auto& t : th) t.join();
    if (bad) {
        printf("FAIL workers=%u callers=%u: %d inconsistencies\n", workers, callers, bad.load());
        return -1;
    }
    return (int)total.load();
}

int main()
{
    long all = 0;
    const unsigned cases[][3] = {{0, 1, 50}, {0, 4, 50}, {1, 1, 200}, {3, 4, 300}, {7, 8, 300}, {11, 3, 300}};
    for (auto& c : cases) {
        int n = run_case(c[0], c[1], c[2]);
        if (n < 0) return 1;
        all += n;
    }
    printf("ok %ld\n", all);
    return 0;
}
