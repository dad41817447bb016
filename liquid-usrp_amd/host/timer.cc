// timer.cc -- stopwatch behind include/timer.h (reference: lib/timer.cc:40-84).
#include <chrono>
#include <cstdio>
#include "timer.h"

struct timer_s {
    std::chrono::steady_clock::time_point t0;
    bool started;
};

timer timer_create() { timer q = new timer_s; q->started = false; return q; }
void timer_destroy(timer _q) { delete _q; }
void timer_tic(timer _q) { _q->t0 = std::chrono::steady_clock::now(); _q->started = true; }
float timer_toc(timer _q)
{
    if (!_q->started) { fprintf(stderr, "warning: timer_toc(), timer was never started\n"); return 0.0f; }
    return std::chrono::duration<float>(std::chrono::steady_clock::now() - _q->t0).count();
}
