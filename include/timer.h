// timer.h -- wall-clock stopwatch with the interface of liquid-usrp's include/timer.h:32-44
// (used by src/multichannel_rx.cc:31,181-182,215,225).
#ifndef LIQUID_USRP_AMD_TIMER_H
#define LIQUID_USRP_AMD_TIMER_H

typedef struct timer_s *timer;

timer timer_create();
void timer_destroy(timer _q);
void timer_tic(timer _q);
float timer_toc(timer _q);      // seconds since the last tic

#endif
