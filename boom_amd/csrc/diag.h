// The diagnostic builds' cycle stamps, all in one place (never in the product: every macro here
// is empty unless its -D flag is given; the Makefile's tools/build/libboomamd_*stamps*.so targets make the builds, tools/
// ss_phase_profile.py and tools/phase_profile.py read them).
//
//   flag                      macro            what the 8 slots time                    clock
//   -DBA_STAMPS               STAMP(i)         phases of one SSVS sweep (ssvs_sweep_body.h)   core
//   -DBA_STAMPS -DBA_STAMPS2  SUBSTAMP(c, i)   the inside of a proposal batch: 0 uniform +
//                                              log, 1 classify, 2 V gather, 3 V solve, 4 A
//                                              gather, 5 A solve, 6 epilogue, 7 outside       core
//   -DBA_STAMPS -DBA_STAMPS3  TSTAMP(c, i)     the master's pieces of a forked sweep: 0
//                                              commit, 1 sweep-start copy, 2 fork, 3 swap
//                                              proposal, 4 sigma, 5 normals, 6 back
//                                              substitution, 7 everything else                core
//   -DBA_STAMPS -DBA_STAMPS4  HSTAMP(c, i)     helper wave 1: 0 shuffle uniforms, 1 matching
//                                              rounds, 2 links, 3 walks, 4 table walk, 5
//                                              waiting for commands, 6 its share of proposal
//                                              rounds, 7 other                                core
//   -DBA_PSTAMPS              PSTAMP(i)        the sweep kernel's prologue (absolute stamps)  core
//   -DBA_KSTAMPS              KSTAMP(i)        the local-level Kalman kernel's phases, and
//                             SSTAMP(i)        the structural kernels' (chain 0 prints)       core
//   -DBA_RSTAMPS              RSTAMP(i)        the bsts round kernel's phases per chain and
//                                              wave, summed over the launch's rounds          100 MHz
//
// The sites declare their own accumulators (`ph[8]` + `last`, under the same flag); the macros
// only add "now - last" to slot i and move `last` on.
#pragma once

#define BA_STAMP_ADD(ph, last, i, now, T) do { const long long t_ = (long long)(now); (ph)[i] += (T)(t_ - (last)); (last) = t_; } while (0)
#define BA_STAMP_OFF do { } while (0)
#define BA_CORE_CLOCK __builtin_readcyclecounter()

struct StampCtx { long long last; double ph[8]; };

#ifdef BA_STAMPS
#define STAMP_DECL long long st_last = (long long)BA_CORE_CLOCK; double st_ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define STAMP(i) BA_STAMP_ADD(st_ph, st_last, i, BA_CORE_CLOCK, double)
#else
#define STAMP_DECL BA_STAMP_OFF
#define STAMP(i) BA_STAMP_OFF
#endif
#if defined(BA_STAMPS) && defined(BA_STAMPS2)
#define SUBSTAMP(c, i) BA_STAMP_ADD((c).ph, (c).last, i, BA_CORE_CLOCK, double)
#else
#define SUBSTAMP(c, i) BA_STAMP_OFF
#endif
#if defined(BA_STAMPS) && defined(BA_STAMPS3)
#define TSTAMP(c, i) BA_STAMP_ADD((c).ph, (c).last, i, BA_CORE_CLOCK, double)
#else
#define TSTAMP(c, i) BA_STAMP_OFF
#endif
#if defined(BA_STAMPS) && defined(BA_STAMPS4)
#define HSTAMP(c, i) BA_STAMP_ADD((c).ph, (c).last, i, BA_CORE_CLOCK, double)
#else
#define HSTAMP(c, i) BA_STAMP_OFF
#endif
#ifdef BA_PSTAMPS
#define PSTAMP(i) pst[i] = (long long)BA_CORE_CLOCK
#else
#define PSTAMP(i) BA_STAMP_OFF
#endif
#ifdef BA_KSTAMPS
#define KSTAMP(i) BA_STAMP_ADD(kph, klast, i, BA_CORE_CLOCK, long long)
#define SSTAMP(i) BA_STAMP_ADD(kph, klast, i, BA_CORE_CLOCK, long long)
#else
#define KSTAMP(i) BA_STAMP_OFF
#define SSTAMP(i) BA_STAMP_OFF
#endif
#ifdef BA_RSTAMPS
#define RSTAMP(i) BA_STAMP_ADD(rph, rlast, i, wall_clock64(), long long)
#else
#define RSTAMP(i) BA_STAMP_OFF
#endif
