// smpc_backend.h -- HIP/gfx950 backend glue for the kernel bodies in smpc_kernels*.h.
//
// Kernel bodies are written in a "lane phase" style:
//
//     SMPC_LANES(NT)            // code executed by every lane of the workgroup, `lane` = threadIdx.x
//       ...
//     SMPC_LANES_END            // workgroup barrier
//
// Code between phases is workgroup-uniform.  Data crossing a phase boundary lives in LDS.  This
// keeps every cross-lane dependency explicit (one barrier per phase end) and lets the test suite
// re-compile the very same bodies with a sequential lane loop (tests/emu/, found first on the
// include path there) to check them against the oracle without a GPU.  This header is the only
// backend the shipped library is built with; it has no CPU path.
#pragma once
#include <hip/hip_runtime.h>
#include <cstring>
#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include <string>
#include "smpc_alloc_scope.h"

#define SMPC_HD __host__ __device__ __forceinline__
#define SMPC_DEV __device__ __forceinline__
#define SMPC_DEV_NOINLINE __device__ __noinline__
#define SMPC_LDS(type, name, n) __shared__ type name[n]
// The lane index is re-materialised through an opaque asm in every phase: otherwise the compiler hoists the index
// arithmetic of ALL phases out of the stage loops (hundreds of loop-invariant values), spills them to scratch,
// and every scratch access then drains the outstanding global prefetches (shared vmcnt).
#define SMPC_LANES(NT)                                                                                                 \
  {                                                                                                                    \
    int _smpc_lane = (int)threadIdx.x;                                                                                 \
    asm volatile("" : "+v"(_smpc_lane));                                                                               \
    __builtin_assume(_smpc_lane >= 0 && _smpc_lane < (NT));                                                            \
    const int lane = _smpc_lane;                                                                                       \
    (void)lane;
#define SMPC_LANES_END                                                                                                 \
  }                                                                                                                    \
  __syncthreads();
// Phase end for kernels whose workgroup is exactly ONE wavefront: lanes run in lockstep and the LDS
// serves one wave's requests in order, so no s_barrier and no vmcnt/lgkmcnt drain is needed -- only a
// compiler-level ordering point.  Outstanding global loads (register prefetch) stay in flight.
#define SMPC_LANES_END_WAVE                                                                                            \
  }                                                                                                                    \
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");                                                               \
  __builtin_amdgcn_wave_barrier();                                                                                     \
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
// the same ordering point INSIDE a lane phase (one-wavefront workgroups, uniform control flow): LDS written by the lanes before it is visible to
// every lane after it.  Only for code that is written for lockstep lanes (SMPC_LOCKSTEP); the sequential test backend has the per-lane form.
#define SMPC_WAVE_SYNC()                                                                                               \
  do                                                                                                                   \
  {                                                                                                                    \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");                                                             \
    __builtin_amdgcn_wave_barrier();                                                                                   \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");                                                             \
  } while (0)
// a lambda that must be inlined into the kernel body (device code: a called function would save / restore the caller's registers)
#define SMPC_LAMBDA_INLINE __attribute__((always_inline))
// a consistency check of the kernel SOURCE that only the sequential test build evaluates (there it throws); nothing on the device
#define SMPC_TEST_CHECK(cond, msg) ((void)0)
// per-lane value that must survive a phase boundary (register on the GPU)
#define SMPC_PL(type, name, NT) type name
#define SMPC_PLA(type, name, NT, n) type name[n]
#define SMPC_PLV(name) name
// a per-lane persistent variable as a function parameter
#define SMPC_PL_REF(type, name, NT) type & name
// value of a per-lane persistent variable / array element in lane `src` (wave-uniform src): v_readlane pairs, the
// result lives in SGPRs and feeds VALU FMAs directly -- no LDS round trip, no barrier
#define SMPC_XLANE(name, src) ::smpc::readlane_f64(name, src)
#define SMPC_XLANE_A(name, idx, src) ::smpc::readlane_f64(name[idx], src)
// no instruction is scheduled across this point (keeps the machine scheduler from hoisting a whole unrolled loop's cross-lane reads
// to its top, where they overflow the scalar registers and are parked in vector lanes)
#define SMPC_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
// 1: the lanes of a wavefront run a lane phase in lockstep -- an exchange through LDS between the lanes is legal INSIDE a phase as long as the
// control flow around it is uniform (the sequential-lane test backend says 0 and takes the per-lane form of such code)
#define SMPC_LOCKSTEP 1
// FP64 matrix cores (v_mfma_f64_16x16x4_f64), one wave:  D(16x16) += A(16x4) B(4x16).
//   operands, one double per lane:   A[i][k] in lane i + 16 k ,  B[k][j] in lane j + 16 k
//   accumulator tile t, 4 doubles per lane:   D[(lane >> 4) + 4 v][lane & 15]  in  SMPC_ACCV(acc, t, v)
// (lane maps verified on hardware by tools/micro/mfma_f64_layout.hip).  av / bv are SMPC_PLA operand arrays.
#define SMPC_ACC(name, NT, n) ::smpc::d4 name[n]
#define SMPC_ACCV(name, t, v) name[t][v]
#define SMPC_MFMA(acc, t, av, ia, bv, ib) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[ia], bv[ib], acc[t], 0, 0, 0)
#define SMPC_CLOCK() ((long long)__builtin_readcyclecounter())
// a value that is the same in every lane, moved to a scalar register (branches on it are scalar branches)
#define SMPC_UNIFORM_U32(x) ((unsigned)__builtin_amdgcn_readfirstlane((int)(x)))
// the same for a double (two scalar registers): constants of a kernel that the compiler would otherwise keep -- and spill -- in vector registers
#define SMPC_UNIFORM_F64(x) ::smpc::readlane_first_f64(x)
// an int the compiler cannot see through: loads addressed with it stay after this point (the optimiser otherwise
// hoists loads of read-only buffers across whole phases and then spills what it loaded)
#define SMPC_PIN(x) ::smpc::pin_int(x)
// fire-and-forget touch of one cache line per lane: a 4-byte load that lands in an LDS sink nobody reads -- no destination
// register, hence nothing to wait for and nothing to spill; the line is in L2 when the real load comes
#define SMPC_TOUCH(gptr, lds_sink) ::smpc::touch_line((gptr), (lds_sink))
// asynchronous copy global -> LDS without registers (global_load_lds_dwordx4, gfx950): lane l moves the 16 bytes at its `gptr` (16-byte aligned)
// to lds_dst + 16 l.  All LDS reads issued before it have completed (the destination may be a block the wave has just read); the data is
// there after SMPC_COPY_TO_LDS_WAIT() (it drains the wave's outstanding global loads, the copy among them).
#define SMPC_COPY16_TO_LDS(gptr, lds_dst) ::smpc::global_to_lds_b128((gptr), (lds_dst))
#define SMPC_COPY_TO_LDS_WAIT() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
// 1/sqrt(x): hardware estimate (v_rsq_f64) + two Newton steps (full FP64 accuracy, no division)
#define SMPC_RSQRT(x) ::smpc::rsqrt_nr(x)
// 1/x: hardware estimate (v_rcp_f64) + two Newton steps (a full IEEE division is ~3x the dependent latency)
#define SMPC_RCP(x) ::smpc::rcp_nr(x)
// 1/x with ONE Newton step: 2e-15 relative error (tools/micro/rcp_accuracy.hip), two dependent FMAs less per pivot
#define SMPC_RCP1(x) ::smpc::rcp_nr1(x)

namespace smpc
{
  // 16-byte store the compiler does not track: a known store makes a non-inlined function wait for the write to complete before it returns
  // (1 .. 2 us); for data nothing in the kernel reads back.  dst: 16-byte aligned.
  // Valid under three conditions, all of them true here and none of them visible to the compiler: (1) memory operations retire through
  // ONE in-order counter (vmcnt of the gfx9 family: the s_endpgm drain covers this store), (2) NO instruction of the same kernel reads
  // the stored bytes back (the compiler would not wait for the store), (3) `s_nop 1` covers the wait state a > 64-bit store needs before
  // its data registers are overwritten.  A target with split load / store counters must not compile this.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "store2_nowait: inline global_store_dwordx4 relies on the gfx950 (gfx9 family) single in-order vmcnt"
#endif
  __device__ __forceinline__ void store2_nowait(double * dst, double v0, double v1)
  {
    typedef double d2_t __attribute__((ext_vector_type(2)));
    const d2_t vv = {v0, v1};
    // (s_nop: a store of more than 64 bits needs a wait state before a VALU write of its data registers -- the hazard recogniser does not
    //  look into inline assembly)
    asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" : : "v"(dst), "v"(vv) : "memory");
  }
  // a double that may alias any other type: the model blocks (doubles and ints) are copied into LDS eight bytes at a time
  typedef double __attribute__((may_alias)) alias_double;
  __device__ __forceinline__ int pin_int(int x)
  {
    asm volatile("" : "+v"(x));
    return x;
  }
  __device__ __forceinline__ void touch_line(const void * gptr, void * lds_sink)
  {
    // LDS destination of lane l: M0 + 4 l (the sink is 256 bytes)
    const unsigned lds_off = (unsigned)(__UINTPTR_TYPE__)(__attribute__((address_space(3))) char *)lds_sink;
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" : : "v"(gptr), "s"(lds_off) : "memory", "m0");
  }
  __device__ __forceinline__ void global_to_lds_b128(const void * gptr, void * lds_dst)
  {
    const unsigned lds_off = (unsigned)(__UINTPTR_TYPE__)(__attribute__((address_space(3))) char *)lds_dst;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(gptr), "s"(lds_off) : "memory", "m0");
  }
  typedef double d4 __attribute__((ext_vector_type(4)));
  __device__ __forceinline__ double readlane_f64(double v, int src)
  {
    union
    {
      double d;
      int i[2];
    } u;
    u.d = v;
    u.i[0] = __builtin_amdgcn_readlane(u.i[0], src);
    u.i[1] = __builtin_amdgcn_readlane(u.i[1], src);
    return u.d;
  }
  __device__ __forceinline__ double readlane_first_f64(double v)
  {
    union
    {
      double d;
      int i[2];
    } u;
    u.d = v;
    u.i[0] = __builtin_amdgcn_readfirstlane(u.i[0]);
    u.i[1] = __builtin_amdgcn_readfirstlane(u.i[1]);
    return u.d;
  }
  __device__ __forceinline__ double rcp_nr(double x)
  {
    double r = __builtin_amdgcn_rcp(x);
    r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
    r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
    return r;
  }
  __device__ __forceinline__ double rcp_nr1(double x)
  {
    double r = __builtin_amdgcn_rcp(x);
    r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
    return r;
  }
  __device__ __forceinline__ double rsqrt_nr(double x)
  {
    double y = __builtin_amdgcn_rsq(x);
    y = y * (1.5 - 0.5 * x * y * y);
    y = y * (1.5 - 0.5 * x * y * y);
    return y;
  }

  inline void hip_check(hipError_t e, const char * what, const char * file, int line)
  {
    if (e != hipSuccess)
    {
      char buf[512];
      std::snprintf(buf, sizeof(buf), "HIP error %d (%s) at %s:%d: %s", (int)e, hipGetErrorString(e), file, line, what);
      throw std::runtime_error(buf);
    }
  }
#define SMPC_HIP(x) ::smpc::hip_check((x), #x, __FILE__, __LINE__)

  typedef hipStream_t stream_t;
  inline void * stream_native(stream_t s) { return (void *)s; } // the hipStream_t itself, for callers that enqueue their own work on it

  inline void * dev_alloc(size_t bytes)
  {
    void * p = nullptr;
    SMPC_HIP(hipMalloc(&p, bytes ? bytes : 8));
    // the engine's streams are non-blocking (no implicit ordering with the null stream the memset
    // runs on): make the zero-fill complete before any kernel can touch the buffer
    hipError_t e = hipMemset(p, 0, bytes ? bytes : 8);
    if (e == hipSuccess)
      e = hipDeviceSynchronize();
    if (e != hipSuccess)
    {
      (void)hipFree(p); // (a caller that handles the failure -- the hand-over fall-backs -- must not leak the block)
      hip_check(e, "zero-fill of a fresh allocation", __FILE__, __LINE__);
    }
    AllocScope::note_alloc(p); // (released again if the constructor that asked for it throws: smpc_alloc_scope.h)
    return p;
  }
  inline void dev_clear_error() { (void)hipGetLastError(); } // after a failed allocation that the caller handles
  inline void dev_free(void * p)
  {
    if (p)
    {
      AllocScope::note_free(p);
      (void)hipFree(p);
    }
  }
  inline void h2d(void * dst, const void * src, size_t bytes, stream_t s)
  {
    SMPC_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, s));
  }
  inline void d2h(void * dst, const void * src, size_t bytes, stream_t s)
  {
    SMPC_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, s));
  }
  // `height` rows of `width` bytes, source rows `spitch` bytes apart on the device, destination rows `dpitch` bytes apart on the host
  inline void d2h_2d(void * dst, size_t dpitch, const void * src, size_t spitch, size_t width, size_t height, stream_t s)
  {
    SMPC_HIP(hipMemcpy2DAsync(dst, dpitch, src, spitch, width, height, hipMemcpyDeviceToHost, s));
  }
  inline void d2d(void * dst, const void * src, size_t bytes, stream_t s)
  {
    SMPC_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, s));
  }
  // device -> another device of the node (xGMI; the runtime stages through host memory when peer access is not available)
  inline void d2peer(void * dst, int dst_device, const void * src, int src_device, size_t bytes, stream_t s)
  {
    SMPC_HIP(hipMemcpyPeerAsync(dst, dst_device, src, src_device, bytes, s));
  }
  inline void dev_zero(void * dst, size_t bytes, stream_t s) { SMPC_HIP(hipMemsetAsync(dst, 0, bytes, s)); }
  inline void stream_sync(stream_t s) { SMPC_HIP(hipStreamSynchronize(s)); }
  inline stream_t stream_create()
  {
    stream_t s;
    SMPC_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    return s;
  }
  inline void stream_destroy(stream_t s) { (void)hipStreamDestroy(s); }
  inline void set_device(int id) { SMPC_HIP(hipSetDevice(id)); }
  // compute units of a device (grids of kernels whose blocks own a slice of device scratch for their lifetime are sized by it)
  inline int dev_cu_count(int id)
  {
    int n = 0;
    SMPC_HIP(hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, id));
    return n > 0 ? n : 256;
  }
  inline int device_count()
  {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess)
      return 0;
    return n;
  }

  struct event_t
  {
    hipEvent_t e;
  };
  inline event_t event_create()
  {
    event_t ev;
    SMPC_HIP(hipEventCreate(&ev.e));
    return ev;
  }
  inline void event_destroy(event_t ev) { (void)hipEventDestroy(ev.e); }
  inline void event_record(event_t ev, stream_t s) { SMPC_HIP(hipEventRecord(ev.e, s)); }
  inline void stream_wait_event(stream_t s, event_t ev) { SMPC_HIP(hipStreamWaitEvent(s, ev.e, 0)); }
  // Per-step uploads of small host tables (the shared stage descriptors): a ring of pinned staging buffers, each guarded by an event, so that
  // the host may run several control steps ahead of the device (no host-side wait inside a closed loop that lives on the handle's stream)
  // without a later step's table overwriting one a queued copy has not read yet.
  struct UploadRing
  {
    static constexpr int N = 4;
    void * host[N] = {};
    event_t ev[N] = {};
    bool used[N] = {};
    size_t cap = 0;
    int next = 0;
    void upload(void * dst, const void * src, size_t bytes, stream_t s)
    {
      if (bytes > cap)
      {
        release();
        for (int i = 0; i < N; i++)
        {
          SMPC_HIP(hipHostMalloc(&host[i], bytes, hipHostMallocDefault));
          ev[i] = event_create();
        }
        cap = bytes;
      }
      const int i = next;
      next = (next + 1) % N;
      if (used[i])
        SMPC_HIP(hipEventSynchronize(ev[i].e)); // the copy that read this slot N uploads ago has completed
      std::memcpy(host[i], src, bytes);
      SMPC_HIP(hipMemcpyAsync(dst, host[i], bytes, hipMemcpyHostToDevice, s));
      event_record(ev[i], s);
      used[i] = true;
    }
    void release()
    {
      for (int i = 0; i < N; i++)
        if (host[i])
        {
          if (used[i])
            (void)hipEventSynchronize(ev[i].e);
          (void)hipHostFree(host[i]);
          event_destroy(ev[i]);
          host[i] = nullptr;
          used[i] = false;
        }
      cap = 0;
    }
    ~UploadRing() { release(); }
    UploadRing() = default;
    UploadRing(const UploadRing &) = delete;
    UploadRing & operator=(const UploadRing &) = delete;
  };
  inline float event_elapsed_ms(event_t a, event_t b)
  {
    float ms = 0;
    SMPC_HIP(hipEventSynchronize(b.e));
    SMPC_HIP(hipEventElapsedTime(&ms, a.e, b.e));
    return ms;
  }

  // MINW = minimum waves per SIMD the register allocator must leave room for (caps VGPRs at 512 / MINW)
  // TAG gives auxiliary launches (cold start on one instance, list-mode launches of the backtracking path) their own
  // kernel symbol, so that a profiler's per-kernel average of the TAG = 0 symbol is the full-batch launch only
  template <class Args, void (*Body)(const Args &, int), int NT, int MINW, int TAG>
  __global__ __launch_bounds__(NT, MINW) void kernel_entry(const Args a)
  {
    Body(a, (int)blockIdx.x);
  }

  template <class Args, void (*Body)(const Args &, int), int NT, int MINW = 1, int TAG = 0>
  inline void launch(int grid, stream_t s, const Args & a)
  {
    if (grid <= 0)
      return;
    static const bool dbg_occ = std::getenv("SMPC_DEBUG_OCCUPANCY") != nullptr;
    if (dbg_occ)
    {
      static bool once = false;
      if (!once)
      {
        once = true;
        int nb = 0;
        SMPC_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kernel_entry<Args, Body, NT, MINW, TAG>, NT, 0));
        hipFuncAttributes fa;
        SMPC_HIP(hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(kernel_entry<Args, Body, NT, MINW, TAG>)));
        std::fprintf(stderr, "[smpc] %s: %d blocks/CU (NT %d, LDS %zu B, %d regs)\n", __PRETTY_FUNCTION__, nb, NT, (size_t)fa.sharedSizeBytes, fa.numRegs);
      }
    }
    hipLaunchKernelGGL((kernel_entry<Args, Body, NT, MINW, TAG>), dim3((unsigned)grid), dim3(NT), 0, s, a);
    SMPC_HIP(hipGetLastError());
  }
} // namespace smpc
