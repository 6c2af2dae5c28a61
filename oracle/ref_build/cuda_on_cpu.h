// TEST INFRASTRUCTURE (see oracle/__init__.py) -- never part of the product.
//
// The reference's tracking kernels (/root/reference/src/lib/{droid_kernels,correlation_kernels,altcorr_kernel}.cu) are
// CUDA sources; this container has neither nvcc nor a GPU.  Their kernel BODIES, though, are plain C++ over
// torch::PackedTensorAccessor32, __shared__ arrays, __syncthreads() and atomicAdd -- no warp intrinsics, no textures,
// no streams.  This header gives those few names a CPU meaning, so that oracle/build_ref.py can compile the reference's
// OWN files (read where they lie, launch syntax rewritten, nothing copied into the repository) into oracle/_ref/ and
// the CPU test-suite can put oracle/droid_oracle.py -- the restatement every GPU parity test is judged against --
// next to the code it restates:
//
//   * a kernel launch `k<<<grid, block>>>(args)` becomes gs_cpu::launch(grid, block, [&] { k(args); }): the blocks of
//     the grid run one after the other, the threads of a block are cooperative FIBERS of which one runs at a time,
//     highest index first, handing over at barriers (gs_cpu::BlockSched): __syncthreads() is a real barrier, shared
//     memory (`__shared__` -> a function-local static) is really shared, the stretch between two barriers is atomic
//     per thread, and the result is deterministic;
//   * a thread that returns from the kernel drops out of the block's barriers (as on the GPU);
//   * atomicAdd is a plain read-modify-write (no two threads of a block run at once);
//   * the warp-synchronous tail of the reference's blockReduce (32 lanes adding in lockstep without a barrier) is the one
//     idiom that needs more than a name: build_ref.py rewrites those six statements to read / GS_WARP0_SYNC / write /
//     GS_WARP0_SYNC, which is what lockstep execution does;
//   * torch::kCUDA means the CPU here (the host code moves small index tensors "to the GPU").
#pragma once
#include <torch/extension.h>
#include <ATen/ATen.h>
#include <ATen/NativeFunctions.h>
#include <ATen/Parallel.h>

#include <algorithm>
#include <ucontext.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <cmath>
#include <tuple>
#include <vector>

// (ATen defines RestrictPtrTraits for device compilers only)
namespace at {
template <typename T>
struct RestrictPtrTraits {
  typedef T* __restrict__ PtrType;
};
}  // namespace at
namespace torch {
using at::RestrictPtrTraits;
}

// AT_DISPATCH_*(tensor.type(), ...): this PyTorch no longer converts the deprecated type object itself
namespace detail {
inline at::ScalarType scalar_type(const at::DeprecatedTypeProperties& t) { return t.scalarType(); }
}  // namespace detail

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __shared__ static
#define kCUDA kCPU

struct gs_uint3 { unsigned x, y, z; };
struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
inline thread_local gs_uint3 threadIdx{0, 0, 0}, blockIdx{0, 0, 0};
inline thread_local dim3 blockDim(1, 1, 1), gridDim(1, 1, 1);

namespace gs_cpu {
// ONE thread of a block runs at a time, and always the runnable thread with the HIGHEST index: a cooperative schedule.
// The stretch between two barriers is therefore atomic per thread and thread 0 runs LAST in every such stretch -- the
// two things the reference's kernels silently rely on the GPU for:
//   * frame_distance_kernel has ALL 256 threads compute the relative pose into the SAME shared arrays (write R(t_i),
//     then t_ij = t_j - t_ij in place, droid_kernels.cu:578): harmless when the warps of a block run nearly in lockstep,
//     but freely interleaved CPU threads turn it into t_j - (t_j - R t_i);
//   * the same kernel lets thread 0 swap the shared frame indices after ITS pixel loop (:639-643) while, formally, other
//     threads may still be reading them: on the GPU everybody is done within a few cycles of each other, here thread 0
//     must not run its epilogue before the others have run their loops.
// Deterministic: the same inputs give the same bits.
struct BlockSched {
  enum : unsigned char { RUNNABLE = 0, AT_BARRIER = 1, AT_WARP0 = 2, DONE = 3 };
  // The threads of a block are FIBERS (ucontext) of the launching OS thread: a hand-over costs a register swap, not a
  // futex round trip -- the projective-transform kernel alone hands over ~150 000 times per block (90 block reductions
  // x (5 block barriers x 256 threads + 12 warp-level syncs x 32 lanes)).
  static constexpr size_t STACK = 256 * 1024;
  std::vector<ucontext_t> ctx;
  std::vector<char> stacks;
  std::vector<unsigned char> st;
  ucontext_t main_ctx;
  unsigned nt;
  int cur = -1;
  dim3 block, grid;
  gs_uint3 bidx{0, 0, 0};
  const std::function<void()>* body = nullptr;

  BlockSched(dim3 grid_, dim3 block_)
      : ctx(block_.x * block_.y * block_.z), stacks((size_t)block_.x * block_.y * block_.z * STACK),
        st(block_.x * block_.y * block_.z, RUNNABLE), nt(block_.x * block_.y * block_.z), block(block_), grid(grid_) {}

  int pick() {                               // the next fiber to run, or -1 when every thread of the block is done
    for (;;) {
      for (int t = (int)nt - 1; t >= 0; --t)
        if (st[(size_t)t] == RUNNABLE) return t;
      bool live = false, all_at_barrier = true;
      for (unsigned t = 0; t < nt; ++t)
        if (st[t] != DONE) {
          live = true;
          if (st[t] != AT_BARRIER) all_at_barrier = false;
        }
      if (!live) return -1;
      if (all_at_barrier) {                  // every live thread has arrived: __syncthreads() releases
        for (unsigned t = 0; t < nt; ++t)
          if (st[t] == AT_BARRIER) st[t] = RUNNABLE;
        continue;
      }
      bool any = false, warp_ready = true;   // the live lanes of warp 0 have all arrived at the warp-level sync
      for (unsigned t = 0; t < nt && t < 32; ++t)
        if (st[t] != DONE) {
          any = true;
          if (st[t] != AT_WARP0) warp_ready = false;
        }
      if (!(any && warp_ready)) {
        fprintf(stderr, "cuda_on_cpu: the threads of a block wait at different barriers (deadlock)\n");
        abort();
      }
      for (unsigned t = 0; t < nt && t < 32; ++t)
        if (st[t] == AT_WARP0) st[t] = RUNNABLE;
    }
  }
  void install(int t) {                      // the identity the running fiber sees
    cur = t;
    const unsigned u = (unsigned)t;
    threadIdx = gs_uint3{u % block.x, (u / block.x) % block.y, u / (block.x * block.y)};
    blockIdx = bidx;
    blockDim = block;
    gridDim = grid;
  }
  void wait(unsigned char kind) {
    const int me = cur;
    st[(size_t)me] = kind;
    const int nxt = pick();
    if (nxt == me) return;
    install(nxt);
    swapcontext(&ctx[(size_t)me], &ctx[(size_t)nxt]);
  }
  static void entry(unsigned lo, unsigned hi);
  void run_block(gs_uint3 b, const std::function<void()>& f) {
    bidx = b;
    body = &f;
    for (unsigned t = 0; t < nt; ++t) {
      st[t] = RUNNABLE;
      getcontext(&ctx[t]);
      ctx[t].uc_stack.ss_sp = stacks.data() + (size_t)t * STACK;
      ctx[t].uc_stack.ss_size = STACK;
      ctx[t].uc_link = &main_ctx;
      const uintptr_t self = (uintptr_t)this;
      makecontext(&ctx[t], (void (*)())entry, 2, (unsigned)(self & 0xffffffffu), (unsigned)(self >> 32));
    }
    install((int)nt - 1);
    swapcontext(&main_ctx, &ctx[nt - 1]);
  }
};
inline thread_local BlockSched* cur_sched = nullptr;
inline void BlockSched::entry(unsigned lo, unsigned hi) {
  BlockSched* s = (BlockSched*)(((uintptr_t)hi << 32) | (uintptr_t)lo);
  (*s->body)();
  s->st[(size_t)s->cur] = DONE;              // a finished thread no longer counts at later barriers
  const int nxt = s->pick();
  if (nxt < 0) {
    setcontext(&s->main_ctx);
  } else {
    s->install(nxt);
    setcontext(&s->ctx[(size_t)nxt]);
  }
}

template <class F>
void launch(dim3 grid, dim3 block, F&& body) {
  if (block.x * block.y * block.z == 0) return;
  BlockSched sched(grid, block);
  BlockSched* outer = cur_sched;
  cur_sched = &sched;
  const std::function<void()> fn(body);
  for (unsigned bz = 0; bz < grid.z; ++bz)
    for (unsigned by = 0; by < grid.y; ++by)
      for (unsigned bx = 0; bx < grid.x; ++bx) sched.run_block(gs_uint3{bx, by, bz}, fn);
  cur_sched = outer;
}
}  // namespace gs_cpu

inline void __syncthreads() { gs_cpu::cur_sched->wait(gs_cpu::BlockSched::AT_BARRIER); }
#define GS_WARP0_SYNC() gs_cpu::cur_sched->wait(gs_cpu::BlockSched::AT_WARP0)
#define GS_LOCKSTEP() __syncthreads()      /* hand-over point of a one-warp block (build_ref.py) */

template <class T, class U>
inline T atomicAdd(T* addr, U val) {                  // (only one thread of the block runs at a time)
  const T old = *addr;
  *addr = old + (T)val;
  return old;
}

using std::max;
using std::min;
