// 3x3 / pad 1 / stride 1 convolution, NHWC fp16 -> NHWC fp16 (fp32 accumulation): the update operator's large
// convolutions (reference src/modules/gru.py:10-12, src/droid_net.py:76,83-92,40) as an implicit GEMM on MFMA with a
// TWO-GROUP PING-PONG schedule -- the production kernel of round 2.
//
// Why a second kernel: rocprofv3 on conv3x3_kernel (profiles/r02_pmc_conv3x3.md) shows the matrix pipes busy 46 % and
// the LDS 40 % of the kernel's cycles, waves parked at barriers / waitcnt 35 % of theirs: its four waves read all their
// fragments together, then compute together (the LDS pipe idles while the MFMAs run and vice versa), and the second
// workgroup of the CU only overlaps with it by chance.  Here the overlap is built in:
//
//  * 512 threads = 8 waves = two groups of four; waves w and w + 4 share a SIMD.  Every wave alternates a READ phase
//    (12 ds_read_b128: its A/B fragments of one tap) and a MATH phase (16 v_mfma_f32_32x32x16_f16), separated by
//    workgroup barriers; group 1 runs one phase behind group 0 (one extra barrier up front), so on every SIMD one wave
//    is on the matrix pipe while its partner is on the LDS pipe.  One instruction stream, no role branches.
//  * the workgroup owns 512 pixels (row-stacked tiling: the n images are one image of n*H rows, tiles of 512 / TW rows x
//    TW columns run across image boundaries; vertical taps that would cross an image boundary read a zero slot) x 128
//    output channels; K = 9 taps x C in chunks of 32 channels; the (TH+2) x (TW+2) input patch of a chunk is staged ONCE
//    and serves all 9 taps.
//  * ALL staging is LDS-DMA (global_load_lds_dwordx4): no staging VGPRs, no ds_write pass.  The pre-packed weight image
//    of a tap (8 KB, exactly one 16-byte piece per thread) goes into a ring of 4 buffers, issued 3 taps ahead; the next
//    chunk's patch goes into the second patch buffer, one 512-slot round per tap.  Out-of-image pixels load from a
//    16-byte zero page, so there is no masking code.  Loads stay in flight across the barriers (raw
//    s_barrier, counted s_waitcnt vmcnt(N), never 0 in the loop); a buffer is read only in the phase AFTER the
//    barrier that follows the wait which retires it (both groups' pieces).
//  * patch layout [pixel][5 x 16-byte slots]: four channel groups + one pad slot, i.e. an 80-BYTE pixel stride.  The four
//    DMA lanes of a pixel still read its 64 contiguous bytes (the fifth lane reads the zero page), and 16 consecutive
//    pixels start 20 banks apart = 16 distinct multiples of 4 banks modulo 64: with the lane-permuted fragment mapping (a
//    ds_read_b128 service group = 16 consecutive pixels of one patch row, conv3x3_common.h) every B-fragment read is
//    bank-conflict-free WITHOUT an address swizzle, so a tap's shift and the second k-step are immediate offsets of the
//    ds_read: the READ phase is 9 VALU + 12 LDS instructions (the XOR-swizzled 64-byte layout of rounds 2-3 needed 46
//    VALU per tap for the same reads; bench keyframe 12.66 -> 12.48 ms).  Taps that would cross an image boundary read
//    a zero region behind the DMA rounds (written once per workgroup) through a per-lane base selected up front.  Weight
//    planes [8-channel group][128 channels][8] are conflict-free as they are.
//  * LDS: 2 x 51.3 KB patch (6 DMA rounds + zero region) + 4 x 8 KB weights = 134.5 KB (TW = 16; 148 KB at TW = 8); one
//    workgroup per CU, 2 waves per SIMD.
//  * BN = 64 instantiation (flow_encoder[2], 128 -> 64 channels): the same schedule with ONE 32-channel fragment per wave
//    (a wave owns 128 pixels x 32 channels: 8 MFMAs per phase against 10 fragment reads, so it is bound by the READ
//    phase, not the matrix pipe); a tap's weight image is 4 KB = one piece per thread of waves 0-3, waves 4-7 issue a
//    dummy 16-byte DMA from the zero page so that every wave's vmcnt arithmetic stays the same.
#include "common.h"
#include "conv3x3_common.h"

namespace {

constexpr int PP_KG = 4;            // 8-channel groups per 32-channel chunk
constexpr int PP_PSTR = 5;          // 16-byte slots per patch pixel (80-byte pixel stride, see the header)
// zero slots behind a patch buffer's DMA rounds: the largest tap offset (2 rows + 2 pixels) + the second k-step's + 2
constexpr int pp_zpad(int tw) { return (((2 * (tw + 2) + 2) * PP_PSTR + 3 + 15) / 16) * 16; }
constexpr int pp_patch_slots(int tw) {
  return (((512 / tw + 2) * (tw + 2) * PP_PSTR + 511) / 512) * 512 + pp_zpad(tw);
}

__device__ __attribute__((aligned(16))) const uint32_t pp_zero_page[4] = {0u, 0u, 0u, 0u};

__device__ __forceinline__ void pp_glds16(const void* src, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

#define PP_BAR()                              \
  do {                                        \
    asm volatile("" ::: "memory");            \
    __builtin_amdgcn_s_barrier();             \
    asm volatile("" ::: "memory");            \
    __builtin_amdgcn_sched_barrier(0);        \
  } while (0)

#define PP_LGKM0()                                        \
  do {                                                    \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    \
    __builtin_amdgcn_sched_barrier(0);                    \
  } while (0)

// Fused epilogues (EPI != 0): arithmetic applied to the convolution's fp16-rounded pre-activations while they pass
// through the epilogue's LDS tile -- the formulas and rounding points of gru_gates.hip (gs_gru_gate_zr / gs_gru_gate_q /
// gs_bias_act), so the results equal convolution + gate kernel bit for bit, but the pre-activations never travel to HBM
// and back.
//   EPI 1 (ConvGRU z|r, 256 outputs): block 0 -> z = sigm(pre + inp_pre[:, 0:128] + b + glo) -> out0;
//                                     block 1 -> r likewise from channels 128:256, out1 = r * net (net = x[:, 0:128]).
//   EPI 2 (ConvGRU q, 128 outputs):   q = tanh(pre + inp_pre[:, 256:384] + b + glo), out0 = (1 - z) net + z q with
//                                     z = aux0, net = aux1; the convolution's input is [x (first `split` channels,
//                                     stride xs) | xb (stride xsb)], i.e. [r * net | corr, flow features] without a cat.
//   EPI 3 (any width):                y = relu(pre + b); y / ys may address a channel slice of a wider tensor.
struct PpEpi {
  const float* bias;           // [256] (EPI 1) / [128] (EPI 2) / [n_out] (EPI 3)
  const float* glo;            // [n, 256] / [n, 128] global-context terms
  const _Float16* inp_pre;     // [n*h*w, 384] hoisted context-feature convolutions, or nullptr
  const _Float16* aux0;        // EPI 2: z [n*h*w, 128]
  const _Float16* aux1;        // EPI 2: net [n*h*w, 128]
  _Float16* out0;              // EPI 1: z, EPI 2: new net   [n*h*w, 128]
  _Float16* out1;              // EPI 1: r * net             [n*h*w, 128]
  const _Float16* xb;          // EPI 2: second input source (channels >= split)
  int xsb, split;
  // Read by the PROBE instantiation only (built with -DGS_BUILD_PROBES for tools/conv3x3_pp_probe.py; the production
  // instantiations contain no trace of them): per-workgroup s_memtime stamps, and A/B bits -- 1 = no s_setprio around
  // the MFMAs, 2 = read phase without masks, 4 = no LDS-DMA in the main loop (stale operands), 8 = no fragment reads in
  // the main loop (stale fragments): what the schedule costs without that traffic (results wrong on purpose).
  long long* dbg;
  int variant;
};

template <bool PROBE>
__device__ __forceinline__ void pp_stamp(long long* dbg, int slot) {
  if constexpr (PROBE) {
    if (dbg && threadIdx.x == 0) dbg[(size_t)blockIdx.x * 4 + slot] = (long long)__builtin_amdgcn_s_memtime();
  }
}

__device__ __forceinline__ float pp_sigm(float v) { return gs_sigmoid(v); }

template <int TW, int EPI, int BN = 128, bool PROBE = false>
__global__ __launch_bounds__(512, 1) void conv3x3_pp_kernel(const _Float16* __restrict__ x, int xs, int C,
                                                            const half8* __restrict__ wpack, _Float16* __restrict__ y,
                                                            int ys, int H, int W, int rows, int tiles_x, int NB,
                                                            int xcd, PpEpi ep) {
  static_assert(BN == 128 || (BN == 64 && (EPI == 0 || EPI == 3)), "64-channel workgroups: plain / bias + ReLU only");
  constexpr int PP_BN = BN;                               // output channels per workgroup
  constexpr int PP_WTAP = PP_KG * PP_BN;                  // 16-byte slots of one tap's weight image (512 or 256)
  constexpr int NJ = BN / 64;                             // 32-channel fragments per wave
  constexpr int PP_TS = NJ == 2 ? 72 : 40;                // epilogue tile row stride in halves (16-byte aligned rows)
  constexpr int TH_ = 512 / TW, PW_ = TW + 2, NPX = (TH_ + 2) * PW_;
  constexpr int PSTR = PP_PSTR;                           // 16-byte slots per patch pixel: 4 channel groups + 1 pad
  constexpr int ZPAD = pp_zpad(TW);                       // zero slots behind the DMA rounds (masked taps read there)
  constexpr int NROUND = (NPX * PSTR + 511) / 512;        // DMA rounds (512 slots each) per patch
  constexpr int PSLOTS = NROUND * 512 + ZPAD;             // slots of a patch buffer
  constexpr int ZSLOT = NROUND * 512;                     // first slot of the zero region
  static_assert(PSLOTS == pp_patch_slots(TW), "launch_pp sizes the LDS with the same formula");
  static_assert(NROUND <= 9, "one DMA round per tap");
  extern __shared__ half8 smem[];                         // patch [2][PSLOTS] | weights [4][PP_WTAP] (| 256 dummy slots)
  half8* const pbuf = smem;
  half8* const wbuf = smem + 2 * PSLOTS;
  half8* const wjunk = wbuf + 4 * PP_WTAP;                // BN = 64 only: target of waves 4-7's dummy weight DMA
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp2 = wv >> 2;                               // 0: leading group, 1: one phase behind
  const int wm = (wv & 1) + 2 * grp2;                     // pixel quarter of the 512-pixel tile
  const int wn = (wv >> 1) & 1;                           // channel half (64 of 128, or 32 of 64)
  const int r = lane & 31, kgl = lane >> 5;
  int tix, nb;
  decode_block(blockIdx.x, gridDim.x / NB, NB, xcd, tix, nb);
  const int tx0 = (tix % tiles_x) * TW;
  const int g0 = (tix / tiles_x) * TH_;                   // first stacked row (img * H + y) of the tile
  const int nchunk = C / 32;
  const int T = nchunk * 9;                               // taps in all
  const half8* wsrc = wpack + (size_t)nb * T * PP_WTAP;

  // this lane's pixel in the 4 pixel fragments of its wave, and whether it sits on the first / last row of its image
  int pb[4];
  bool top[4], bot[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int ty, tx;
    tile_pixel<TW, true>(wm, i, r, ty, tx);
    pb[i] = ty * PW_ + tx;
    const int yy = (g0 + ty) % H;
    top[i] = yy == 0;
    bot[i] = yy == H - 1;
  }
  // DMA sources of this thread's patch slots: 4 * (pixel index) + channel group, or < 0 for the zero page
  int poff[NROUND];
#pragma unroll
  for (int q = 0; q < NROUND; ++q) {
    const int s = q * 512 + tid, p = s / PSTR, kg = s - PSTR * p;
    int off = -1;
    if (p < NPX && kg < 4) {                              // the fifth slot of a pixel and the tail of the last round: zeros
      const int pr = p / PW_, pc = p - pr * PW_;
      const int gv = g0 + pr - 1, gx = tx0 + pc - 1;
      if (gv >= 0 && gv < rows && gx >= 0 && gx < W) off = (gv * W + gx) * 4 + kg;
    }
    poff[q] = off;
  }

  // slot (pixel * 5 + channel group of this half-wave) of this lane's pixel in each fragment: the tap shift and the
  // second k-step are immediate offsets of the ds_read.  Lanes whose row above (below) belongs to another image read the
  // zero region instead for the taps of the upper (lower) kernel row.
  int vb[4], vt[4], vo[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    vb[i] = pb[i] * PSTR + kgl;
    vt[i] = top[i] ? ZSLOT : vb[i];
    vo[i] = bot[i] ? ZSLOT : vb[i];
  }
  {                                                       // the zero regions are written once; no DMA ever targets them
    const half8 z8 = {0, 0, 0, 0, 0, 0, 0, 0};
    if (tid < ZPAD) pbuf[ZSLOT + tid] = z8;
    else if (tid < 2 * ZPAD) pbuf[PSLOTS + ZSLOT + tid - ZPAD] = z8;
    static_assert(2 * ZPAD <= 512, "one store per thread");
  }

  float16v acc[NJ][4];
#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[j][i][e] = 0.0f;
  half8 a[2][NJ], b[2][4];

  // ---- the three building blocks --------------------------------------------------------------------------------
  auto issue_patch = [&](int src_chunk, int buf, int q, int off) {        // one 512-slot round of a patch
    const void* src = (const void*)pp_zero_page;
    if (off >= 0) {
      const size_t pix = (size_t)(off >> 2);
      const int c0 = src_chunk * 32 + (off & 3) * 8;
      if constexpr (EPI == 1 || EPI == 2)                  // two input tensors: [x (first `split` channels) | xb]
        src = c0 < ep.split ? (const void*)(x + pix * xs + c0) : (const void*)(ep.xb + pix * ep.xsb + (c0 - ep.split));
      else
        src = (const void*)(x + pix * xs + c0);
    }
    pp_glds16(src, pbuf + buf * PSLOTS + q * 512 + wv * 64);
  };
  auto issue_w = [&](int src_tap, int buf) {                              // one tap's weight image
    if constexpr (BN == 128) {
      pp_glds16(wsrc + (size_t)src_tap * PP_WTAP + tid, wbuf + buf * PP_WTAP + wv * 64);
    } else {                                                              // 256 pieces: waves 0-3; 4-7 keep vmcnt in step
      if (wv < 4) pp_glds16(wsrc + (size_t)src_tap * PP_WTAP + tid, wbuf + buf * PP_WTAP + wv * 64);
      else pp_glds16((const void*)pp_zero_page, wjunk + (wv - 4) * 64);
    }
  };
  auto read_frags = [&](int pbuf_ix, int wbuf_ix, int tap) {              // READ phase body: 4 A + 8 B fragments
    const half8* wb = wbuf + wbuf_ix * PP_WTAP;
    const half8* pp = pbuf + pbuf_ix * PSLOTS;
    const int dy = tap / 3, toff = dy * PW_ + (tap % 3);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int kg = 2 * s + kgl;
#pragma unroll
      for (int j = 0; j < NJ; ++j) a[s][j] = wb[kg * PP_BN + wn * (PP_BN / 2) + 32 * j + r];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int base = dy == 0 ? vt[i] : dy == 2 ? vo[i] : vb[i];     // the row above / below belongs to another image: zeros
      if constexpr (PROBE) {
        if (ep.variant & 2) base = vb[i];
      }
      asm volatile("" : "+v"(base));                      // opaque: one address register per read pair, not 24 hoisted ones
      const half8* q = pp + base;
      b[0][i] = q[toff * PSTR];                           // channel group kgl (first k-step) ...
      b[1][i] = q[toff * PSTR + 2];                       // ... and 2 + kgl (second): immediate offsets
    }
  };
  // MATH phase body: 16 MFMAs; `between` (the LDS-DMA issue of this tap) is placed after the first four, so that its
  // address arithmetic issues in the shadow of running MFMAs instead of delaying the first one after the barrier
  bool prio = true;
  if constexpr (PROBE) prio = !(ep.variant & 1);
  auto math = [&](auto&& between) {
    if (prio) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j)
        acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0][j], b[0][i], acc[j][i], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    between();
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 2; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j)
        acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0][j], b[0][i], acc[j][i], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j)
        acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1][j], b[1][i], acc[j][i], 0, 0, 0);
    if (prio) __builtin_amdgcn_s_setprio(0);
  };

  // ---- prologue: patch of chunk 0, weights of taps 0..2 ------------------------------------------------------------
  pp_stamp<PROBE>(ep.dbg, 0);
#pragma unroll
  for (int q = 0; q < NROUND; ++q) issue_patch(0, 0, q, poff[q]);
  issue_w(0, 0);
  issue_w(1, 1);
  issue_w(2, 2);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // lgkmcnt: the zero regions' stores (V >= 1)
  PP_BAR();
  pp_stamp<PROBE>(ep.dbg, 1);
  if (grp2 == 1) PP_BAR();                                // group 1 runs one phase behind from here on
  read_frags(0, 0, 0);
  PP_LGKM0();
  PP_BAR();

  for (int ck = 0; ck < nchunk; ++ck) {
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int tg = ck * 9 + tap;
      // ---- MATH phase of tap tg; first put the loads for 3 taps ahead (and a round of the next patch) in flight.
      // Past the end the same number of loads is issued from clamped sources into buffers nobody reads any more, so
      // that the counted waits below stay compile-time constants.
      __builtin_amdgcn_sched_barrier(0);
      math([&]() {
        if constexpr (PROBE) {
          if (ep.variant & 4) return;
        }
        const int tw = tg + 3 < T ? tg + 3 : T - 1;
        issue_w(tw, (tg + 3) & 3);
        if (tap < NROUND) issue_patch(ck + 1 < nchunk ? ck + 1 : nchunk - 1, (ck + 1) & 1, tap, poff[tap]);
      });
      __builtin_amdgcn_sched_barrier(0);
      // everything issued before this tap has landed (this wave's pieces); the barrier publishes it.  Tap tg + 2's
      // weights are first read two phases from now, the next patch at the earliest two taps from now.
      if (tap < NROUND) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
      PP_BAR();
      // ---- READ phase: fragments of tap tg + 1
      bool reads = true;
      if constexpr (PROBE) reads = !(ep.variant & 8);
      if (reads) {
        if (tap < 8) read_frags(ck & 1, (tg + 1) & 3, tap + 1);
        else if (ck + 1 < nchunk) read_frags((ck + 1) & 1, (tg + 1) & 3, 0);
      }
      PP_LGKM0();
      PP_BAR();
    }
  }
  pp_stamp<PROBE>(ep.dbg, 2);
  if (grp2 == 0) PP_BAR();                                // pairs with group 1's last barrier
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the clamped tail loads: they target LDS the epilogue reuses
  PP_BAR();

  // ---- epilogue: [32 pixels][64 channels] at a time through a wave-private LDS tile (aliases patch buffer 0)
  _Float16* tile = reinterpret_cast<_Float16*>(smem) + wv * 32 * PP_TS;
  _Float16* yb = y + nb * PP_BN + wn * (PP_BN / 2);
  constexpr int PIECES = NJ * 4;                          // 16-byte pieces per pixel row of this wave's channels (8 / 4)
  constexpr int PXIT = 64 / PIECES;                       // pixels covered per store round (8 / 16)
  constexpr int NIT = 32 / PXIT;                          // store rounds per 32-pixel fragment (4 / 2)
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    // (1) where this lane's four output pieces go, and -- for the gate epilogues -- their operands, requested NOW so
    //     that the loads are in flight while the accumulators are transposed through LDS
    size_t pix4[NIT];
    bool ok4[NIT];
    half8 pi4[NIT], u4[NIT], v4[NIT];
    const int piece = lane & (PIECES - 1);
    const int c8 = wn * (PP_BN / 2) + piece * 8;            // first of this lane's 8 channels inside the channel block
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int pxr = it * PXIT + lane / PIECES;
      int ty, tx;
      tile_pixel<TW, true>(wm, i, pxr, ty, tx);
      const int gv = g0 + ty, gx = tx0 + tx;
      ok4[it] = gv < rows && gx < W;
      pix4[it] = (size_t)gv * W + gx;
      if constexpr (EPI == 1 || EPI == 2) {
        const half8 z8 = {0, 0, 0, 0, 0, 0, 0, 0};
        pi4[it] = z8; u4[it] = z8; v4[it] = z8;
        if (ok4[it]) {
          if constexpr (EPI == 1) {
            if (ep.inp_pre) pi4[it] = *reinterpret_cast<const half8*>(ep.inp_pre + pix4[it] * 384 + nb * 128 + c8);
            if (nb == 1) u4[it] = *reinterpret_cast<const half8*>(x + pix4[it] * xs + c8);          // net
          } else {
            if (ep.inp_pre) pi4[it] = *reinterpret_cast<const half8*>(ep.inp_pre + pix4[it] * 384 + 256 + c8);
            u4[it] = *reinterpret_cast<const half8*>(ep.aux0 + pix4[it] * 128 + c8);               // z
            v4[it] = *reinterpret_cast<const half8*>(ep.aux1 + pix4[it] * 128 + c8);               // net
          }
        }
      }
    }
    // (2) accumulators -> [32 pixels][64 (32) channels] fp16 tile
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) {                       // C layout: row (channel) = 8 g + 4 (lane >> 5) + e, col = pixel
        half4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (_Float16)acc[j][i][4 * g + e];
        *reinterpret_cast<half4*>(tile + r * PP_TS + j * 32 + 8 * g + 4 * kgl) = o;
      }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // (3) rows of 64 (32) channels leave as 128 (64) contiguous bytes per pixel
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int pxr = it * PXIT + lane / PIECES;
      if (ok4[it]) {
        const half8 v = *reinterpret_cast<const half8*>(tile + pxr * PP_TS + piece * 8);
        const size_t pix = pix4[it];
        if constexpr (EPI == 0) {
          *reinterpret_cast<half8*>(yb + pix * ys + piece * 8) = v;
        } else if constexpr (EPI == 3) {
          const float* bb = ep.bias + nb * PP_BN + c8;
          half8 o;
#pragma unroll
          for (int k = 0; k < 8; ++k) o[k] = (_Float16)fmaxf((float)v[k] + bb[k], 0.0f);
          *reinterpret_cast<half8*>(yb + pix * ys + piece * 8) = o;
        } else if constexpr (EPI == 1) {
          const int img = (int)(pix / ((size_t)H * W));
          const half8 pi = pi4[it];
          const float* bb = ep.bias + nb * 128 + c8;
          const float* gg = ep.glo + (size_t)img * 256 + nb * 128 + c8;
          half8 o;
          if (nb == 0) {
#pragma unroll
            for (int k = 0; k < 8; ++k) o[k] = (_Float16)pp_sigm((float)v[k] + (float)pi[k] + bb[k] + gg[k]);
            *reinterpret_cast<half8*>(ep.out0 + pix * 128 + c8) = o;
          } else {
            const half8 net = u4[it];
#pragma unroll
            for (int k = 0; k < 8; ++k)
              o[k] = (_Float16)(pp_sigm((float)v[k] + (float)pi[k] + bb[k] + gg[k]) * (float)net[k]);
            *reinterpret_cast<half8*>(ep.out1 + pix * 128 + c8) = o;
          }
        } else {
          const int img = (int)(pix / ((size_t)H * W));
          const half8 pi = pi4[it], zz = u4[it], nn = v4[it];
          const float* bb = ep.bias + c8;
          const float* gg = ep.glo + (size_t)img * 128 + c8;
          half8 o;
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            const float a_ = (float)v[k] + (float)pi[k] + bb[k] + gg[k];
            const float q = gs_tanh(a_);
            const float zf = (float)zz[k];
            o[k] = (_Float16)((1.0f - zf) * (float)nn[k] + zf * q);
          }
          *reinterpret_cast<half8*>(ep.out0 + pix * 128 + c8) = o;
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  pp_stamp<PROBE>(ep.dbg, 3);
}

template <int TW, int EPI, int BN = 128, bool PROBE = false>
int launch_pp(const void* x, int x_stride, int c_in, const void* wpack, void* y, int y_stride, int n_out, int n, int h,
              int w, int xcd, hipStream_t st, PpEpi ep = PpEpi()) {
  constexpr int PP_BN = BN, PP_WTAP = PP_KG * BN;
  constexpr size_t lds = (size_t)(2 * pp_patch_slots(TW) + 4 * PP_WTAP + (BN == 64 ? 256 : 0)) * sizeof(half8);
  static_assert(lds <= 160 * 1024, "conv3x3_pp: more LDS than a CU has");
  static GsLdsLimit limit;
  if (int rc = limit.raise((const void*)conv3x3_pp_kernel<TW, EPI, BN, PROBE>, lds, "conv3x3_pp")) return rc;
  const long long rows = (long long)n * h;
  GS_REQUIRE(rows * w < (1ll << 29), "conv3x3_pp: too many pixels");
  const int tiles_x = gs_cdiv(w, TW), tiles_y = gs_cdiv((int)rows, 512 / TW);
  const int NB = n_out / PP_BN;
  const long long blocks = (long long)tiles_x * tiles_y * NB;
  GS_REQUIRE(blocks < (1ll << 31), "conv3x3_pp: too many workgroups");
  conv3x3_pp_kernel<TW, EPI, BN, PROBE><<<dim3((unsigned)blocks), 512, lds, st>>>(
      (const _Float16*)x, x_stride, c_in, (const half8*)wpack, (_Float16*)y, y_stride, h, w, (int)rows, tiles_x, NB, xcd,
      ep);
  GS_CHECK_LAUNCH("conv3x3_pp");
  return GS_OK;
}

template <int EPI, int BN = 128>
int dispatch_pp(const void* x, int x_stride, int c_in, const void* wpack, int tw, void* y, int y_stride, int n_out, int n,
                int h, int w, int xcd, hipStream_t st, PpEpi ep = PpEpi()) {
  if (tw == 8) return launch_pp<8, EPI, BN>(x, x_stride, c_in, wpack, y, y_stride, n_out, n, h, w, xcd, st, ep);
  return launch_pp<16, EPI, BN>(x, x_stride, c_in, wpack, y, y_stride, n_out, n, h, w, xcd, st, ep);
}

// output channels per workgroup for a layer: 128, or 64 when n_out is only a multiple of 64 (the weight image follows)
int pp_block_channels(int n_out) { return n_out % 128 == 0 ? 128 : 64; }

int pp_tile_width(int w) { return gs_cdiv(w, 8) * 8 < gs_cdiv(w, 16) * 16 ? 8 : 16; }   // least column padding

}  // namespace

extern "C" int gs_conv3x3_pp(const void* x, int x_stride, int c_in, const void* wpack, int tw, void* y, int y_stride,
                             int n_out, int n, int h, int w, int xcd_order, gs_stream_t stream) {
  GS_REQUIRE(x && wpack && y, "conv3x3_pp: null pointer");
  GS_REQUIRE(tw == 8 || tw == 16, "conv3x3_pp: tile width must be 8 or 16");
  GS_REQUIRE(c_in > 0 && c_in % 32 == 0, "conv3x3_pp: c_in must be a multiple of 32 (weights packed with kc = 32)");
  GS_REQUIRE(n_out > 0 && n_out % 64 == 0, "conv3x3_pp: n_out must be a multiple of 64");
  GS_REQUIRE(x_stride >= c_in && x_stride % 8 == 0, "conv3x3_pp: x_stride must be >= c_in and a multiple of 8");
  GS_REQUIRE(y_stride >= n_out && y_stride % 8 == 0, "conv3x3_pp: y_stride must be >= n_out and a multiple of 8");
  GS_REQUIRE(n >= 0 && h > 0 && w > 0, "conv3x3_pp: bad shape");
  if (n == 0) return GS_OK;
  if (pp_block_channels(n_out) == 64)
    return dispatch_pp<0, 64>(x, x_stride, c_in, wpack, tw, y, y_stride, n_out, n, h, w, xcd_order ? 1 : 0,
                              (hipStream_t)stream);
  return dispatch_pp<0>(x, x_stride, c_in, wpack, tw, y, y_stride, n_out, n, h, w, xcd_order ? 1 : 0,
                        (hipStream_t)stream);
}

// ---- ConvGRU with the gate arithmetic fused into the convolutions' epilogues (see PpEpi) -----------------------------
extern "C" int gs_conv3x3_gru_zr(const void* hx, int hx_stride, int c_in, const void* wpack, const float* bias_zr,
                                 const float* glo_zr, const void* inp_pre, void* z_out, void* rnet_out, int n, int h,
                                 int w, gs_stream_t stream) {
  GS_REQUIRE(hx && wpack && bias_zr && glo_zr && z_out && rnet_out, "conv3x3_gru_zr: null pointer");
  GS_REQUIRE(c_in >= 128 && c_in % 32 == 0, "conv3x3_gru_zr: c_in must be >= 128 and a multiple of 32");
  GS_REQUIRE(hx_stride >= c_in && hx_stride % 8 == 0, "conv3x3_gru_zr: bad hx_stride");
  GS_REQUIRE(n >= 0 && h > 0 && w > 0, "conv3x3_gru_zr: bad shape");
  if (n == 0) return GS_OK;
  PpEpi ep = PpEpi();
  ep.bias = bias_zr;
  ep.glo = glo_zr;
  ep.inp_pre = (const _Float16*)inp_pre;
  ep.out0 = (_Float16*)z_out;
  ep.out1 = (_Float16*)rnet_out;
  ep.xb = (const _Float16*)hx;                              // one tensor: every channel comes from `hx`
  ep.xsb = hx_stride;
  ep.split = 1 << 30;
  return dispatch_pp<1>(hx, hx_stride, c_in, wpack, pp_tile_width(w), nullptr, 0, 256, n, h, w, 1, (hipStream_t)stream, ep);
}

// The same with the GRU input given as two tensors, [net (128 channels, dense) | x_rest (c_rest channels, pixels
// x_rest_stride apart)], as gs_conv3x3_gru_q takes it: the caller no longer copies net into the first 128 channels of the
// input buffer before every step.
extern "C" int gs_conv3x3_gru_zr2(const void* net, const void* x_rest, int x_rest_stride, int c_rest, const void* wpack,
                                  const float* bias_zr, const float* glo_zr, const void* inp_pre, void* z_out,
                                  void* rnet_out, int n, int h, int w, gs_stream_t stream) {
  GS_REQUIRE(net && x_rest && wpack && bias_zr && glo_zr && z_out && rnet_out, "conv3x3_gru_zr2: null pointer");
  GS_REQUIRE(c_rest > 0 && c_rest % 32 == 0, "conv3x3_gru_zr2: c_rest must be a multiple of 32");
  GS_REQUIRE(x_rest_stride >= c_rest && x_rest_stride % 8 == 0, "conv3x3_gru_zr2: bad x_rest_stride");
  GS_REQUIRE(n >= 0 && h > 0 && w > 0, "conv3x3_gru_zr2: bad shape");
  if (n == 0) return GS_OK;
  PpEpi ep = PpEpi();
  ep.bias = bias_zr;
  ep.glo = glo_zr;
  ep.inp_pre = (const _Float16*)inp_pre;
  ep.out0 = (_Float16*)z_out;
  ep.out1 = (_Float16*)rnet_out;
  ep.xb = (const _Float16*)x_rest;
  ep.xsb = x_rest_stride;
  ep.split = 128;
  return dispatch_pp<1>(net, 128, 128 + c_rest, wpack, pp_tile_width(w), nullptr, 0, 256, n, h, w, 1,
                        (hipStream_t)stream, ep);
}

extern "C" int gs_conv3x3_gru_q(const void* rnet, const void* x_rest, int x_rest_stride, int c_rest, const void* wpack,
                                const float* bias_q, const float* glo_q, const void* inp_pre, const void* z,
                                const void* net, void* net_out, int n, int h, int w, gs_stream_t stream) {
  GS_REQUIRE(rnet && x_rest && wpack && bias_q && glo_q && z && net && net_out, "conv3x3_gru_q: null pointer");
  GS_REQUIRE(c_rest > 0 && c_rest % 32 == 0, "conv3x3_gru_q: c_rest must be a multiple of 32");
  GS_REQUIRE(x_rest_stride >= c_rest && x_rest_stride % 8 == 0, "conv3x3_gru_q: bad x_rest_stride");
  GS_REQUIRE(n >= 0 && h > 0 && w > 0, "conv3x3_gru_q: bad shape");
  if (n == 0) return GS_OK;
  PpEpi ep = PpEpi();
  ep.bias = bias_q;
  ep.glo = glo_q;
  ep.inp_pre = (const _Float16*)inp_pre;
  ep.aux0 = (const _Float16*)z;
  ep.aux1 = (const _Float16*)net;
  ep.out0 = (_Float16*)net_out;
  ep.xb = (const _Float16*)x_rest;
  ep.xsb = x_rest_stride;
  ep.split = 128;
  return dispatch_pp<2>(rnet, 128, 128 + c_rest, wpack, pp_tile_width(w), nullptr, 0, 128, n, h, w, 1,
                        (hipStream_t)stream, ep);
}

// 3x3 convolution + bias + ReLU in one kernel (EPI 3): corr_encoder[2] writing straight into its slice of the GRU input,
// agg.conv2.  Same arithmetic as gs_conv3x3_pp followed by gs_bias_act(relu).
extern "C" int gs_conv3x3_bias_relu(const void* x, int x_stride, int c_in, const void* wpack, const float* bias, void* y,
                                    int y_stride, int n_out, int n, int h, int w, gs_stream_t stream) {
  GS_REQUIRE(x && wpack && bias && y, "conv3x3_bias_relu: null pointer");
  GS_REQUIRE(c_in > 0 && c_in % 32 == 0, "conv3x3_bias_relu: c_in must be a multiple of 32");
  GS_REQUIRE(n_out > 0 && n_out % 64 == 0, "conv3x3_bias_relu: n_out must be a multiple of 64");
  GS_REQUIRE(x_stride >= c_in && x_stride % 8 == 0, "conv3x3_bias_relu: bad x_stride");
  GS_REQUIRE(y_stride >= n_out && y_stride % 8 == 0, "conv3x3_bias_relu: bad y_stride");
  GS_REQUIRE(n >= 0 && h > 0 && w > 0, "conv3x3_bias_relu: bad shape");
  if (n == 0) return GS_OK;
  PpEpi ep = PpEpi();
  ep.bias = bias;
  if (pp_block_channels(n_out) == 64)
    return dispatch_pp<3, 64>(x, x_stride, c_in, wpack, pp_tile_width(w), y, y_stride, n_out, n, h, w, 1,
                              (hipStream_t)stream, ep);
  return dispatch_pp<3>(x, x_stride, c_in, wpack, pp_tile_width(w), y, y_stride, n_out, n, h, w, 1, (hipStream_t)stream,
                        ep);
}

extern "C" size_t gs_conv3x3_wpack_elems(int c_in, int n_out) { return (size_t)9 * c_in * n_out; }

#ifdef GS_BUILD_PROBES
// Tools only (tools/conv3x3_pp_probe.py; `make PROBES=1`, never part of the shipped library): gs_conv3x3_pp at tw = 16
// with per-workgroup s_memtime stamps (dbg: int64 [workgroups][4] = start, prologue done, main loop done, end; wave 0 of
// each workgroup) and A/B variant bits (see PpEpi).
extern "C" int gs_conv3x3_pp_probe(const void* x, int x_stride, int c_in, const void* wpack, int tw, void* y, int y_stride,
                                   int n_out, int n, int h, int w, int variant, long long* dbg, gs_stream_t stream) {
  GS_REQUIRE(x && wpack && y, "conv3x3_pp_probe: null pointer");
  GS_REQUIRE(tw == 16 && c_in > 0 && c_in % 32 == 0 && n_out > 0 && n_out % 128 == 0 && n > 0,
             "conv3x3_pp_probe: bad arguments (tw must be 16)");
  PpEpi ep = PpEpi();
  ep.dbg = dbg;
  ep.variant = variant;
  return launch_pp<16, 0, 128, true>(x, x_stride, c_in, wpack, y, y_stride, n_out, n, h, w, 1, (hipStream_t)stream, ep);
}
#endif
