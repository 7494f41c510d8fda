// TEST INFRASTRUCTURE ONLY -- sequential "lane loop" stand-in for simple-mpc_amd/csrc/smpc_backend.h.
//
// The test build puts tests/emu first on the include path, so the unmodified kernel bodies and host
// logic of simple-mpc_amd/csrc are compiled by g++ with every SMPC_LANES phase executed as a loop
// over lanes (in ascending or descending order, see smpc::emu_reverse: running both orders and
// comparing results exposes missing phase barriers).  It exists so that the numerics of the HIP
// kernel bodies can be checked against the oracle in the CPU-only CI tier and under sanitizers;
// the shipped library is never built with it and has no CPU path.
#pragma once
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include "smpc_alloc_scope.h"

#define SMPC_CPU_EMU_BUILD 1
#ifndef SMPC_CROSSCHECK
#define SMPC_CROSSCHECK 1 // the alternative paths of the engines and their environment switches (smpc_model.h)
#endif
#define SMPC_HD inline
#define SMPC_DEV inline
#define SMPC_DEV_NOINLINE inline
// LDS is NOT zero on the GPU: poison it on every kernel-body entry (0xFF bytes = NaN doubles) so that a
// read-before-write shows up here instead of only on hardware.
#define SMPC_LDS(type, name, n)                                                                                        \
  static thread_local type name[n];                                                                                    \
  std::memset((void *)name, 0xFF, sizeof(type) * (n))
#define SMPC_LANES(NT)                                                                                                 \
  for (int _l = 0; _l < (NT); ++_l)                                                                                    \
  {                                                                                                                    \
    const int lane = ::smpc::emu_reverse ? ((NT)-1 - _l) : _l;                                                         \
    (void)lane;
#define SMPC_LANES_END }
#define SMPC_LANES_END_WAVE }
#define SMPC_LAMBDA_INLINE
#define SMPC_WAVE_SYNC() ((void)0) // (lockstep-only code has a per-lane form here)
#define SMPC_TEST_CHECK(cond, msg)                                                                                     \
  do                                                                                                                   \
  {                                                                                                                    \
    if (!(cond))                                                                                                       \
      throw std::runtime_error(msg);                                                                                   \
  } while (0)
#define SMPC_PL(type, name, NT) type name[NT]
#define SMPC_PLA(type, name, NT, n) type name[NT][n]
#define SMPC_PLV(name) name[lane]
#define SMPC_PL_REF(type, name, NT) type(&name)[NT]
#define SMPC_XLANE(name, src) name[src]
#define SMPC_XLANE_A(name, idx, src) name[src][idx]
#define SMPC_SCHED_FENCE() ((void)0)
#define SMPC_LOCKSTEP 0 // (lanes run one after the other)
// wave-collective matrix-core step (see the HIP backend for the lane maps); called between lane phases
#define SMPC_ACC(name, NT, n) double name[NT][n][4]
#define SMPC_ACCV(name, t, v) name[lane][t][v]
#define SMPC_MFMA(acc, t, av, ia, bv, ib) ::smpc::emu_mfma_f64_16x16x4(acc, t, av, ia, bv, ib)
#define SMPC_CLOCK() (0LL)
#define SMPC_UNIFORM_U32(x) ((unsigned)(x))
#define SMPC_UNIFORM_F64(x) (x)
#define SMPC_PIN(x) (x)
#define SMPC_TOUCH(gptr, lds_sink) ((void)(gptr), (void)(lds_sink))
#define SMPC_COPY16_TO_LDS(gptr, lds_dst) std::memcpy((char *)(lds_dst) + 16 * lane, (const void *)(gptr), 16)
#define SMPC_COPY_TO_LDS_WAIT() ((void)0)
#define SMPC_RSQRT(x) (1.0 / std::sqrt(x))
#define SMPC_RCP(x) (1.0 / (x))
#define SMPC_RCP1(x) (1.0 / (x))

namespace smpc
{
  // a double that may alias any other type: the model blocks (doubles and ints) are copied into LDS eight bytes at a time
  typedef double __attribute__((may_alias)) alias_double;
  // D[(l>>4)+4v][l&15] += sum_k A[.][k] B[k][.], k ascending with fused multiply-adds (bitwise what the hardware does)
  template <class Acc, class AV, class BV>
  inline void emu_mfma_f64_16x16x4(Acc & acc, int t, const AV & av, int ia, const BV & bv, int ib)
  {
    for (int l = 0; l < 64; l++)
      for (int v = 0; v < 4; v++)
      {
        const int row = (l >> 4) + 4 * v, col = l & 15;
        double sacc = acc[l][t][v];
        for (int k = 0; k < 4; k++)
          sacc = std::fma(av[row + 16 * k][ia], bv[col + 16 * k][ib], sacc);
        acc[l][t][v] = sacc;
      }
  }
  inline bool emu_reverse = std::getenv("SMPC_EMU_REVERSE") != nullptr && std::getenv("SMPC_EMU_REVERSE")[0] == '1';
  typedef int stream_t;
  inline void store2_nowait(double * dst, double v0, double v1)
  {
    dst[0] = v0;
    dst[1] = v1;
  }
  inline void * stream_native(stream_t) { return nullptr; } // (no streams in the sequential build)
  inline void * dev_alloc(size_t bytes)
  {
    void * p = std::calloc(bytes ? bytes : 8, 1);
    if (!p)
      throw std::runtime_error("emu: out of memory");
    AllocScope::note_alloc(p);
    return p;
  }
  inline void dev_clear_error() {}
  inline void dev_free(void * p)
  {
    if (p)
      AllocScope::note_free(p);
    std::free(p);
  }
  inline void h2d(void * dst, const void * src, size_t bytes, stream_t) { std::memcpy(dst, src, bytes); }
  inline void d2h(void * dst, const void * src, size_t bytes, stream_t) { std::memcpy(dst, src, bytes); }
  inline void d2h_2d(void * dst, size_t dpitch, const void * src, size_t spitch, size_t width, size_t height, stream_t)
  {
    for (size_t r = 0; r < height; r++)
      std::memcpy((char *)dst + r * dpitch, (const char *)src + r * spitch, width);
  }
  inline void d2d(void * dst, const void * src, size_t bytes, stream_t) { std::memmove(dst, src, bytes); }
  inline void d2peer(void * dst, int, const void * src, int, size_t bytes, stream_t) { std::memmove(dst, src, bytes); }
  inline void dev_zero(void * dst, size_t bytes, stream_t) { std::memset(dst, 0, bytes); }
  inline void stream_sync(stream_t) {}
  inline stream_t stream_create() { return 0; }
  inline void stream_destroy(stream_t) {}
  inline void set_device(int) {}
  inline int dev_cu_count(int) { return 2; } // (small on purpose: the persistent grids of the dense engines loop over many work items per block here)
  inline int device_count() { return 1; }
  struct event_t
  {
    int e;
  };
  inline event_t event_create() { return event_t{0}; }
  inline void event_destroy(event_t) {}
  inline void event_record(event_t, stream_t) {}
  inline void stream_wait_event(stream_t, event_t) {}
  inline float event_elapsed_ms(event_t, event_t) { return 0.f; }
  struct UploadRing // (sequential build: a copy)
  {
    void upload(void * dst, const void * src, size_t bytes, stream_t) { std::memcpy(dst, src, bytes); }
  };

  template <class Args, void (*Body)(const Args &, int), int NT, int MINW = 1, int TAG = 0>
  inline void launch(int grid, stream_t, const Args & a)
  {
#pragma omp parallel for schedule(dynamic)
    for (int b = 0; b < grid; b++)
      Body(a, b);
  }
} // namespace smpc
