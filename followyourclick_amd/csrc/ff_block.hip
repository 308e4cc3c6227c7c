// One GEGLU feed-forward block of a transformer in ONE kernel (reference diffusers/models/attention.py:772-775 FeedForward,
// 819-821 GEGLU; called from animatediff/models/attention.py:489-564 BasicTransformerBlock and motion_module.py:270-283
// TemporalTransformerBlock), merged with the output projection that follows it as engine/weights.py::_ff does:
//
//   h   = GEGLU( LN(tok) W1^T + b1 )                       [rows][4C]
//   out = residual + b_out + [tok | h] [Wp | Wp W2]^T      [rows][C]
//
// The unfused schedule (fyc_row_stats, fyc_gemm GEGLU, fyc_gemm dual-K) writes and re-reads the 4C-wide hidden tensor (335 MB
// each way at the 64x64 level) and runs a K = C GEMM whose epilogue is as long as its MFMA work.  Here the hidden activation never
// leaves the registers and there is no per-tile epilogue in the main loop:
//
//   * a workgroup = 4 wave64 (one per SIMD, the whole 512-register file each) owns 128 token rows, a wave 32 of them for ALL
//     columns.  Its 32 x C tokens live in registers as MFMA operands for the whole kernel (80 VGPRs): they are the operand of
//     the projection phase and of all 40 FF1 chunks and the source of the LayerNorm statistics - no statistics pass, no LDS
//     traffic for the activation side.  The 32 x C output accumulators (160) sit in the accumulator half of the file;
//   * the matrix instruction is v_mfma_f32_32x32x16_bf16 (32 cycles), used as D^T = W x^T: a wave that is alone on its SIMD pays
//     for every other instruction BETWEEN two matrix instructions (tools/exp/mfma_fill_bench.hip: ~8 cycles per instruction
//     beside 16-cycle MFMAs, ~3 beside 32-cycle ones; the first version of this kernel on 16x16x32 ran 3.7 other instructions per
//     MFMA at 37-42 % matrix-pipe occupancy, profiles/r03_rr_kernels_pmc.txt).  A lane holds ONE token (lane % 32) and 8
//     consecutive k (lane / 32) of it per operand register, and of a 32-feature result block the features 8 b + 4 (lane / 32) + e;
//   * the weights arrive as one pre-packed stream (engine/weights.py::pack_ff_block): half-stages of 32 KiB, every 1-KiB piece
//     already the A operand of one MFMA in lane order (lane l holds W[32 j + l % 32][16 s + 8 (l / 32) .. +8]), so a half is a run
//     of global -> LDS DMAs (global_load_lds, 16 B / lane, lane-linear image = conflict-free ds_read_b128 at base + lane * 16)
//     into a 4-deep ring: half h + 2 is requested while half h computes;
//   * per chunk of 32 hidden units a wave runs 40 MFMAs of FF1 (value block, gate block; K = C from registers), the GEGLU gate on
//     its 32 x 32 results - which ARE the B operand of FF2 as they stand (the k-slots of the W2' fragments are packed in the
//     order the gate outputs sit in the lanes: no shuffle, no LDS round trip) - and 20 MFMAs of FF2 for the PREVIOUS chunk, so
//     that the gate's VALU work has independent matrix work beside it;
//   * epilogue: residual tile by DMA into the (now idle) ring (row pitch padded against bank conflicts), + bias + accumulators
//     in f32, one rounding to bf16 in LDS, the tile leaves in 640-B rows (16 B / lane); per-(row tile, channel) {sum, sum sq} of
//     the stored values for the GroupNorm that consumes the block (fyc_gemm's chan_parts layout with tile_rows = 128, one slot).
//
// Built for the level where it pays (C = 320, hidden 1280, bf16, rows % 128 == 0); other shapes keep the unfused schedule.
// Compiled WITHOUT -amdgpu-mfma-vgpr-form and with -fno-slp-vectorize (see _build.py).
#include <mutex>
#include <type_traits>

#include "fyc_common.h"

namespace {

constexpr int C_ = 320, HID = 1280, ROWS = 128, NW = 4, NT = 64 * NW;
constexpr int KS = C_ / 16;                    // 20 MFMA k-steps (of 16) over C
constexpr int NB = C_ / 32;                    // 10 feature blocks (of 32) of the output
constexpr int CHUNKS = HID / 32;               // 40 hidden chunks of 32 units
constexpr int PIECE = 1024;                    // one MFMA A operand for all 64 lanes: a 32 x 16 weight block
// The weight stream is cut into HALF-STAGES of HP = 32 pieces (32 KiB) that go through a 4-deep LDS ring: the pieces of half
// h + 2 are requested while half h computes, i.e. a whole stage-time before they are needed.
//   half t < 10           k-steps 2 t, 2 t + 1 of the projection: pieces 10 s' + j (s' < 2, j < 10) = Wp rows 32 j .. +32, columns 16 (2 t + s') .. +16
//   half 10 + 2 c  ("A")  pieces 2 s + v (s < 14; v = 0 value block, 1 gate block): the 32 value / 32 gate rows of W1 of chunk c,
//                         columns 16 s .. +16; piece 28: f32 bias[32 value | 32 gate] of chunk c - 1
//   half 11 + 2 c  ("B")  pieces 2 (s - 14) + v (s = 14 .. 19): the rest of W1 of chunk c; pieces 12 + 10 sg + j (sg < 2, j < 10): W2' of chunk c - 1
//   halves 90, 91         piece 28 / pieces 12 + ... of the same for chunk 39
constexpr int HP = 32, HALF_BYTES = HP * PIECE, NSLOT = 4, GPW = HP / 4 / NW;     // GPW: groups of four pieces per wave and half
constexpr int SA = 14;                         // FF1 k-steps in half A
constexpr int P_BIAS = 2 * SA, P_W2 = 2 * (KS - SA);
constexpr int NPROJ = KS / 2;                  // 10 projection halves of two k-steps
constexpr int NHALF = NPROJ + 2 * CHUNKS + 2;  // 92
// the residual / output tile of the epilogue (overlays the ring).  Row pitch 656 B: consecutive rows are 164 dwords = 36 (mod 64
// banks) apart, so the 32 rows x 2 lane halves x 8 B of one in-place add spread over all banks 2-way (the natural 640 B puts
// them into 2 banks; that defect cost csrc/temporal_block_rr.hip a fifth of its tile time: profiles/r03_temporal_block_rr_phases.txt)
constexpr int TP = C_ * 2 + 16;
constexpr int TILE_BYTES = ROWS * TP;          // 83968 = 82 pieces
constexpr int NSL = NT / 40;                   // row slices of the copy-out pass
constexpr int SCR_BYTES = NSL * 40 * 16 * 4;   // statistics partials of the epilogue: [row slice][column group][8 sums | 8 sums of squares]
constexpr int LDS_BYTES = NSLOT * HALF_BYTES;  // 131072
static_assert(TILE_BYTES % PIECE == 0 && TILE_BYTES <= ((NHALF - 2) % NSLOT) * HALF_BYTES + P_BIAS * PIECE && (NHALF - 1) % NSLOT == 3,
              "the residual tile lands while the last two halves (slots 2 and 3) still hold the bias and W2' pieces of chunk 39");
static_assert(TILE_BYTES + SCR_BYTES <= LDS_BYTES, "epilogue tile + statistics scratch overlay the ring");
static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct FFP {
  const bf16_t* x; const bf16_t* res; bf16_t* out;
  const char* ws;            // packed weight stream, NHALF * HALF_BYTES
  const float* b_out;        // [C]
  float* parts;              // [rows / 128][C][2] or null
  int ntiles;                // rows / 128
  float eps;
};

// Four consecutive pieces (4 KiB of the stream -> 4 KiB of the ring) by one wave, issued from inline asm: ONE M0 write, the
// instruction's immediate offset moves the global and the LDS address together.  NOT the builtin: while a builtin LDS-DMA is
// outstanding hipcc turns every counted lgkmcnt wait of the fragment reads into lgkmcnt(0) (it models the DMA as a FLAT access
// that may touch LDS), so each k-step paid the full LDS round trip.  The compiler does not count these loads: the barriers of
// this kernel carry their own s_waitcnt vmcnt.  M0 is saved and restored inside the statement (as in panel_linear.hip and
// temporal_block_rr.hip): it is compiler-reserved, and an "m0" clobber is only a warning, so a statement must leave it as it found it.
__device__ __forceinline__ void dma16x4(const char* gbase, unsigned voff, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
               "global_load_lds_dwordx4 %1, %2\n\t"
               "global_load_lds_dwordx4 %1, %2 offset:1024\n\t"
               "global_load_lds_dwordx4 %1, %2 offset:2048\n\t"
               "global_load_lds_dwordx4 %1, %2 offset:3072\n\t"
               "s_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(gbase), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void dma16v(const void* gsrc, unsigned lds_dst) {                        // one piece, per-lane source address
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void dma_landed_barrier() {       // all of this wave's DMA pieces have landed, then the workgroup meets
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
}
__device__ __forceinline__ void half_landed_barrier() {      // all but this wave's newest half (4 GPW loads) have landed, then the workgroup meets
  asm volatile("s_waitcnt vmcnt(%0)" :: "n"(4 * GPW) : "memory");
  __syncthreads();
}

// sum over the two lane halves of a wave (every lane gets it): gfx950's v_permlane32_swap, plain VALU.  NOT __shfl_xor: that is
// ds_bpermute_b32, an LDS-queue instruction, and hipcc (which cannot see the asm DMAs above) waits for it with a COUNTED lgkmcnt
// between the fragment reads of the projection stages it sinks this code into - with LDS-DMA writes in flight the result was
// consumed early in ~12 % of the row blocks (row statistics off by ~1e-3: tools/ff_stress.py found the kernel's output changing
// from launch to launch; profiles/r03_ff_block_race.txt).  -DFF_BPERMUTE rebuilds that form.
__device__ __forceinline__ float halves_sum(float v) {
#ifdef FF_BPERMUTE
  return v + __shfl_xor(v, 32);
#else
  return swap32_sum(v);          // (fyc_common.h: the two results of the swap must stay opaque to hipcc)
#endif
}

__device__ __forceinline__ f32x16 mfma(bf16x8 a, bf16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x16 mfma(f16x8 a, f16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
template <typename Frag> __device__ __forceinline__ Frag frag(const char* sl, int piece) { return *reinterpret_cast<const Frag*>(sl + piece * PIECE); }

#ifdef FF_TIMING                                                // phase timestamps of wave 0 of every workgroup (tools/ff_probe.py prints them)
__device__ unsigned long long g_ff_time[1024 * 16];
#define FF_MARK(k) do { if (tid == 0) g_ff_time[(blockIdx.x & 1023) * 16 + (k)] = __builtin_readcyclecounter(); } while (0)
#else
#define FF_MARK(k) do {} while (0)
#endif

// T: the 16-bit element type of the tokens, the residual, the output and the weight stream (bf16_t or f16_t; FFP's pointers are
// typed bf16_t for both: same arithmetic)
template <typename T>
__global__ void __launch_bounds__(NT) ff_block_kernel(const FFP p) {
  typedef typename Pair16<T>::Vec8 Frag;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  // One row tile per workgroup.  (A persistent loop over tiles was tried - with and without requesting the next tile's tokens
  // during the last stage: inside an outer loop the register allocator spilled 85-95 registers of this ~500-register kernel.)
  {
  const int tile = blockIdx.x;
  const long long row0 = (long long)tile * ROWS;
  const int lane = tid & 63;
  const int kh = lane >> 5, tok = lane & 31;                  // lane = (k half, token of the wave's 32)
  const unsigned lane16 = (unsigned)lane * 16u;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;     // LDS byte address of the dynamic region
  // n-th group of four pieces of this wave of half h (n < GPW): group wave + NW n; every wave issues exactly 4 GPW loads per
  // half, so "everything but the newest half has landed" is the constant s_waitcnt vmcnt(4 GPW)
  auto dma_group = [&](int h, int n) {
#ifdef FF_NO_DMA
    if (h >= 2) return;                                       // ablation build (tools/ff_probe.py): no weight stream, wrong results
#endif
    const int grp = wave + NW * n;
    dma16x4(p.ws + (long long)h * HALF_BYTES + grp * (4 * PIECE), lane16, lds0 + (h & (NSLOT - 1)) * HALF_BYTES + grp * (4 * PIECE));
  };

  // ---- the wave's 32 token rows as MFMA B operands: lane (token, k half) holds x[token][16 s + 8 kh .. +8] ----------------------
  FF_MARK(0);
  Frag xa[KS];
  {
    const bf16_t* xr = p.x + (row0 + wave * 32 + tok) * C_ + kh * 8;
#pragma unroll
    for (int s = 0; s < KS; ++s) xa[s] = *reinterpret_cast<const Frag*>(xr + s * 16);
  }
#pragma unroll
  for (int n = 0; n < GPW; ++n) dma_group(0, n);
#pragma unroll
  for (int n = 0; n < GPW; ++n) dma_group(1, n);

  // LayerNorm statistics of the lane's row (two-pass, in registers; the two lanes of a row meet through the lane-half swap)
  float mu, rs;
  {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < KS; ++k) {
      const u32x4 t = __builtin_bit_cast(u32x4, xa[k]);
#pragma unroll
      for (int e = 0; e < 4; ++e) s += Pair16<T>::lo(t[e]) + Pair16<T>::hi(t[e]);
    }
    mu = halves_sum(s) * (1.0f / C_);
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < KS; ++k) {
      const u32x4 t = __builtin_bit_cast(u32x4, xa[k]);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float a = Pair16<T>::lo(t[e]) - mu, b = Pair16<T>::hi(t[e]) - mu;
        q = __builtin_fmaf(a, a, q);
        q = __builtin_fmaf(b, b, q);
      }
    }
    rs = rsqrtf(halves_sum(q) * (1.0f / C_) + p.eps);
  }

  f32x16 oacc[NB];                                            // out^T block j: features 32 j + 8 b + 4 kh + e (register 4 b + e) of token `tok`
#pragma unroll
  for (int j = 0; j < NB; ++j)
#pragma unroll
    for (int e = 0; e < 16; ++e) oacc[j][e] = 0.f;

  FF_MARK(1);
  // ---- projection: out = tok Wp^T, half t = k-steps 2 t, 2 t + 1 of all 10 feature blocks -------------------------------------
#pragma unroll
  for (int t = 0; t < NPROJ; ++t) {
    half_landed_barrier();                                    // half t landed; slot (t + 2) & 3 is free (half t - 2 is done): refill it
    const char* sl = smem + (t & (NSLOT - 1)) * HALF_BYTES + lane16;
#pragma unroll
    for (int u = 0; u < 2 * NB; ++u) {
      oacc[u % NB] = mfma(frag<Frag>(sl, u), xa[2 * t + u / NB], oacc[u % NB]);
      if (u % 8 == 0 && u / 8 < GPW) dma_group(t + 2, u / 8);
    }
  }

  FF_MARK(2);
  // The tokens were needed raw for the projection; from here on they are only the FF1 operand: normalise them in place,
  // (x - mean) rstd rounded to bf16 (what the reference's autocast feeds its Linear), so that no LayerNorm term is left in the
  // per-chunk gate (gamma is folded into W1, beta into b1: engine/weights.py::fold_layernorm)
#pragma unroll
  for (int k = 0; k < KS; ++k) {
    u32x4 t = __builtin_bit_cast(u32x4, xa[k]);
#pragma unroll
    for (int e = 0; e < 4; ++e)
      t[e] = Pair16<T>::pack((Pair16<T>::lo(t[e]) - mu) * rs, (Pair16<T>::hi(t[e]) - mu) * rs);
    xa[k] = __builtin_bit_cast(Frag, t);
  }

  // ---- hidden chunks -----------------------------------------------------------------------------------------------------------
  // Chunk c:  half A = FF1(c) k-steps 0..13 -> hv, hg (28 MFMAs)  ||  elements 0..13 of the GEGLU gate of chunk c - 1 (VALU) -> hb;
  //           half B = FF1(c) k-steps 14..19 (12)  ||  elements 14, 15;  then  FF2(c - 1) with hb (20 MFMAs).
  // The gate of a chunk runs one chunk after its FF1 so that its VALU instructions have independent matrix work beside them.
  // The gate is cut into its 16 elements per lane (register e of the value / gate results = hidden unit 8 (e / 4) + 4 kh + e % 4):
  // ~18 VALU instructions each, ONE element behind the two MFMAs of a k-step and laid out between them by sched_group_barrier.
  // A lone wave overlaps VALU with the matrix pipe only where the two alternate instruction by instruction: with a whole
  // 75-instruction unit behind a k-step the phase times were exactly (MFMA cycles + 4 x other instructions), i.e. no overlap.
  // Pairs of elements are packed to bf16: FF2 k-step sg = e / 8, k-slots 8 kh + e % 8.
  f32x4 bv4, bg4;
  float glo = 0.f;
  auto gate_elem = [&](int e, const float* bias, const f32x16& hv, const f32x16& hg, u32x4 (&hbw)[2]) {
    if (e % 4 == 0) { bv4 = *reinterpret_cast<const f32x4*>(bias + 2 * e); bg4 = *reinterpret_cast<const f32x4*>(bias + 32 + 2 * e); }
    const float o = geglu_one(hv[e] + bv4[e % 4], hg[e] + bg4[e % 4]);
    if (e & 1) {
      unsigned pk = Pair16<T>::pack(glo, o);
      asm volatile("" : "+v"(pk));                              // "used" here: LLVM's sinking passes would otherwise move the gate down to FF2, its only consumer
      hbw[e >> 3][(e & 7) >> 1] = pk;
    } else {
      glo = o;
    }
  };
  // FF1 k-steps [S0, S1) from the pieces 2 (s - S0) + v of the half at sl; the next k-step's two fragments in flight;
  // `after(s)` runs behind the MFMAs of k-step s (DMA groups, gate units)
  auto ff1_steps = [&](const char* sl, auto s0_, auto s1_, f32x16& hv, f32x16& hg, auto after) {
    constexpr int S0 = decltype(s0_)::value, S1 = decltype(s1_)::value;
    Frag wv = frag<Frag>(sl, 0), wg = frag<Frag>(sl, 1);
#pragma unroll
    for (int s = S0; s < S1; ++s) {
      Frag nv = wv, ng = wg;
      if (s + 1 < S1) { nv = frag<Frag>(sl, 2 * (s + 1 - S0)); ng = frag<Frag>(sl, 2 * (s + 1 - S0) + 1); }
      hv = mfma(wv, xa[s], hv);
      hg = mfma(wg, xa[s], hg);
      after(s);
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);      // the two fragment reads of the next k-step, then MFMA / VALU alternating
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, 10, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, 12, 0);
      __builtin_amdgcn_sched_barrier(0);
      wv = nv; wg = ng;
    }
  };
  auto ff2 = [&](const char* sl, const u32x4 (&hbw)[2]) {
    const Frag hb[2] = {__builtin_bit_cast(Frag, hbw[0]), __builtin_bit_cast(Frag, hbw[1])};
    Frag w[2] = {frag<Frag>(sl, P_W2), frag<Frag>(sl, P_W2 + 1)};
#pragma unroll
    for (int u = 0; u < 2 * NB; u += 2) {                     // piece P_W2 + u: k-step u / NB, feature block u % NB; the next two in flight
      Frag n[2] = {w[0], w[1]};
      if (u + 2 < 2 * NB) { n[0] = frag<Frag>(sl, P_W2 + u + 2); n[1] = frag<Frag>(sl, P_W2 + u + 3); }
      oacc[u % NB] = mfma(w[0], hb[u / NB], oacc[u % NB]);
      oacc[(u + 1) % NB] = mfma(w[1], hb[(u + 1) / NB], oacc[(u + 1) % NB]);
      if ((u & 2) || u + 2 == 2 * NB) __builtin_amdgcn_sched_barrier(0);
      w[0] = n[0]; w[1] = n[1];
    }
  };
  using I0 = std::integral_constant<int, 0>;
  using IA = std::integral_constant<int, SA>;
  using IK = std::integral_constant<int, KS>;
  // one chunk = halves ha (A) and ha + 1 (B): FF1 -> hv, hg; with_gate: gate + FF2 of the previous chunk from pv, pg
  auto chunk = [&](int ha, f32x16& hv, f32x16& hg, const f32x16& pv, const f32x16& pg, auto with_gate) {
    constexpr bool WITH_GATE = decltype(with_gate)::value;
    u32x4 hbw[2];
    if (ha == NPROJ + 40) FF_MARK(4);
    half_landed_barrier();                                    // half A landed; slot (ha + 2) & 3 free
    if (ha == NPROJ + 40) FF_MARK(5);
    {
      const char* base = smem + (ha & (NSLOT - 1)) * HALF_BYTES;
      const float* bias = reinterpret_cast<const float*>(base + P_BIAS * PIECE) + 4 * kh;
#pragma unroll
      for (int e = 0; e < 16; ++e) { hv[e] = 0.f; hg[e] = 0.f; }
      ff1_steps(base + lane16, I0{}, IA{}, hv, hg, [&](int s) {
        if (s % 4 == 0 && s / 4 < GPW) dma_group(ha + 2, s / 4);                   // the refill goes out in the first k-steps
        if constexpr (WITH_GATE) gate_elem(s, bias, pv, pg, hbw);                  // gate elements 0 .. 13 of the previous chunk
      });
    }
    if (ha == NPROJ + 40) FF_MARK(6);
    half_landed_barrier();                                    // half B landed
    if (ha == NPROJ + 40) FF_MARK(7);
    {
      const char* base = smem + ((ha + 1) & (NSLOT - 1)) * HALF_BYTES;
      const float* bias = reinterpret_cast<const float*>(smem + (ha & (NSLOT - 1)) * HALF_BYTES + P_BIAS * PIECE) + 4 * kh;   // (half A's slot is refilled only two halves later)
      ff1_steps(base + lane16, IA{}, IK{}, hv, hg, [&](int s) {
        if ((s - SA) % 2 == 0 && (s - SA) / 2 < GPW) dma_group(ha + 3, (s - SA) / 2);
        if constexpr (WITH_GATE) {
          if (s < 16) gate_elem(s, bias, pv, pg, hbw);                             // ... and 14, 15
        }
      });
      if (ha == NPROJ + 40) FF_MARK(8);
      if constexpr (WITH_GATE) ff2(base + lane16, hbw);
      if (ha == NPROJ + 40) FF_MARK(9);
    }
  };

  f32x16 v0, g0, v1, g1;
  FF_MARK(3);
  chunk(NPROJ, v0, g0, v1, g1, std::false_type{});
  for (int c = 1; c + 1 < CHUNKS; c += 2) {                  // chunks (1, 2), (3, 4), ..., (37, 38): pre-activation buffers alternate
    chunk(NPROJ + 2 * c, v1, g1, v0, g0, std::true_type{});
    chunk(NPROJ + 2 * c + 2, v0, g0, v1, g1, std::true_type{});
  }
  FF_MARK(10);
  chunk(NHALF - 4, v1, g1, v0, g0, std::true_type{});         // chunk 39: its refills are halves 90, 91 (bias / W2' of chunk 39 only)
  {
    u32x4 hbw[2];
    half_landed_barrier();                                    // half 90 landed (half 91 may still be in flight)
    {
      const char* base = smem + ((NHALF - 2) & (NSLOT - 1)) * HALF_BYTES;
      const float* bias = reinterpret_cast<const float*>(base + P_BIAS * PIECE) + 4 * kh;
#pragma unroll
      for (int e = 0; e < 16; ++e) gate_elem(e, bias, v1, g1, hbw);
    }
    dma_landed_barrier();                                     // half 91 landed; every wave is done with slots 0 .. 2 below the bias piece
    if (p.res != nullptr) {                                   // residual tile over the idle part of the ring: the LDS image is linear, every
      const char* src = reinterpret_cast<const char*>(p.res + row0 * C_);    // lane fetches the 16 B that belong at its place
#pragma unroll 1
      for (int q = wave; q < TILE_BYTES / PIECE; q += NW) {
        const int off = q * PIECE + (int)lane16, row = off / TP;
        const int col = min(off - row * TP, C_ * 2 - 16);       // (the 16 pad bytes of a row re-fetch its last chunk)
        dma16v(src + row * (C_ * 2) + col, lds0 + q * PIECE);
      }
    }
    ff2(smem + ((NHALF - 1) & (NSLOT - 1)) * HALF_BYTES + lane16, hbw);
  }
  FF_MARK(11);

  // ---- epilogue -------------------------------------------------------------------------------------------------------------------
  dma_landed_barrier();                                            // residual tile landed; every wave is done with the ring
  FF_MARK(12);
  {
    char* rowp = smem + (wave * 32 + tok) * TP + kh * 8;
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        T* a = reinterpret_cast<T*>(rowp + (32 * j + 8 * b) * 2);
        const f32x4 bo = *reinterpret_cast<const f32x4*>(p.b_out + 32 * j + 8 * b + 4 * kh);
        float rr[4] = {0.f, 0.f, 0.f, 0.f}, v[4];
        if (p.res != nullptr) ElemIO<T>::ld4(a, rr);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = oacc[j][4 * b + r] + bo[r] + rr[r];
        ElemIO<T>::st4(a, v);
      }
  }
  dma_landed_barrier();
  // pass over the finished tile: thread (column group cg of 8 channels, row slice rsl) copies rows rsl, rsl + 6, ... to HBM,
  // 16 B per lane and 640 contiguous bytes per row, and sums its 8 columns for the statistics of the values as stored
  {
    const int cg = tid % 40, rsl = tid / 40;
    float cs8[8], cq8[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) cs8[e] = cq8[e] = 0.f;
    if (rsl < NSL) {
      char* dst = reinterpret_cast<char*>(p.out + row0 * C_) + cg * 16;
      const char* src = smem + cg * 16;
#pragma unroll 2
      for (int row = rsl; row < ROWS; row += NSL) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(src + row * TP);
        *reinterpret_cast<u32x4*>(dst + row * (C_ * 2)) = v;
        if (p.parts != nullptr) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float x0 = Pair16<T>::lo(v[e]), x1 = Pair16<T>::hi(v[e]);
            cs8[2 * e] += x0; cq8[2 * e] = __builtin_fmaf(x0, x0, cq8[2 * e]);
            cs8[2 * e + 1] += x1; cq8[2 * e + 1] = __builtin_fmaf(x1, x1, cq8[2 * e + 1]);
          }
        }
      }
    }
    if (p.parts != nullptr) {                                 // wave-uniform: every thread takes the barriers
      float* scr = reinterpret_cast<float*>(smem + TILE_BYTES);
      if (rsl < NSL) {
        float* d = scr + (rsl * 40 + cg) * 16;
        *reinterpret_cast<f32x4*>(d) = (f32x4){cs8[0], cs8[1], cs8[2], cs8[3]};
        *reinterpret_cast<f32x4*>(d + 4) = (f32x4){cs8[4], cs8[5], cs8[6], cs8[7]};
        *reinterpret_cast<f32x4*>(d + 8) = (f32x4){cq8[0], cq8[1], cq8[2], cq8[3]};
        *reinterpret_cast<f32x4*>(d + 12) = (f32x4){cq8[4], cq8[5], cq8[6], cq8[7]};
      }
      __syncthreads();
      if (tid < 160) {                                        // one thread per column pair: add the row slices
        const int c2 = tid >> 2, e2 = (tid & 3) * 2;
        float s0 = 0.f, q0 = 0.f, s1 = 0.f, q1 = 0.f;
#pragma unroll
        for (int sl6 = 0; sl6 < NSL; ++sl6) {
          const float* d = scr + (sl6 * 40 + c2) * 16;
          s0 += d[e2]; s1 += d[e2 + 1]; q0 += d[8 + e2]; q1 += d[8 + e2 + 1];
        }
        *reinterpret_cast<f32x4*>(p.parts + ((long long)tile * C_ + 2 * tid) * 2) = (f32x4){s0, q0, s1, q1};
      }
    }
  }
  FF_MARK(13);
  }
}

}  // namespace

// LDS per CU and CU count of this process's device (one GPU per process), queried once; 0 when no device answers
static void device_limits(int64_t& lds_cap, int64_t& n_cu) {
  static std::mutex mu;
  static int64_t cap = -1, cus = 0;
  std::lock_guard<std::mutex> lk(mu);
  if (cap < 0) {
    int64_t caps[8];
    if (fyc_device_caps(caps) == 0) { cap = caps[1]; cus = caps[0]; } else { cap = 0; cus = 0; }
  }
  lds_cap = cap;
  n_cu = cus;
}

#ifdef FF_TIMING
extern "C" int fyc_ff_timing(unsigned long long* host_out, int n) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_ff_time), sizeof(unsigned long long) * n);
}
#endif

extern "C" int64_t fyc_ff_block_wstream_bytes(void) { return (int64_t)NHALF * HALF_BYTES; }

extern "C" int fyc_ff_block_supported(const fyc_ff_block_args* a) {
  if (a == nullptr || (a->dtype != FYC_BF16 && a->dtype != FYC_F16) || a->C != C_ || a->hidden != HID || a->rows <= 0 || a->rows % ROWS != 0) return 0;
  if (a->chan_parts != nullptr && (a->cs_rows <= 0 || a->cs_rows % ROWS != 0 || a->rows % a->cs_rows != 0)) return 0;
  int64_t lds_cap = 0, n_cu = 0;
  device_limits(lds_cap, n_cu);
  if (lds_cap > 0 && lds_cap < LDS_BYTES) return 0;   // a device / partition mode with less LDS: the caller keeps the unfused schedule
  return 1;
}

extern "C" int fyc_ff_block(const fyc_ff_block_args* a, void* stream) {
  FYC_REQUIRE(a && a->x && a->out && a->wstream && a->b_out, "fyc_ff_block: null pointer");
  FYC_REQUIRE(fyc_ff_block_supported(a), "fyc_ff_block: built for bf16 / f16, C=320, hidden=1280, rows %% 128 == 0, cs_rows %% 128 == 0 and >= %d B of LDS (got C=%d hidden=%d rows=%d cs_rows=%d)",
              LDS_BYTES, a->C, a->hidden, a->rows, a->cs_rows);
  FYC_REQUIRE(a->x != a->out, "fyc_ff_block: out must not alias x");
  FYC_REQUIRE(((uintptr_t)a->x % 16) == 0 && ((uintptr_t)a->out % 16) == 0 && ((uintptr_t)a->wstream % 16) == 0 && ((uintptr_t)a->b_out % 16) == 0 &&
              ((uintptr_t)a->residual % 16) == 0 && ((uintptr_t)a->chan_parts % 16) == 0, "fyc_ff_block: operands must be 16-byte aligned");
  FFP p;
  p.x = (const bf16_t*)a->x; p.res = (const bf16_t*)a->residual; p.out = (bf16_t*)a->out; p.ws = (const char*)a->wstream;
  p.b_out = a->b_out; p.parts = a->chan_parts; p.eps = a->eps; p.ntiles = a->rows / ROWS;
  {  // dynamic LDS above 64 KB needs the function attribute once per device; one process may drive several GPUs from several threads
    constexpr int kMaxDev = 64;
    static std::mutex mu;
    static bool attr_done[kMaxDev] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lk(mu);
    if (dev < 0 || dev >= kMaxDev || !attr_done[dev]) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(ff_block_kernel<bf16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
      if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(ff_block_kernel<f16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
      if (e != hipSuccess) FYC_FAIL(-3, "fyc_ff_block: %d bytes of dynamic LDS refused: %s", LDS_BYTES, hipGetErrorString(e));
      if (dev >= 0 && dev < kMaxDev) attr_done[dev] = true;
    }
  }
  if (a->dtype == FYC_F16) hipLaunchKernelGGL(ff_block_kernel<f16_t>, dim3((unsigned)p.ntiles), dim3(NT), LDS_BYTES, (hipStream_t)stream, p);
  else hipLaunchKernelGGL(ff_block_kernel<bf16_t>, dim3((unsigned)p.ntiles), dim3(NT), LDS_BYTES, (hipStream_t)stream, p);
  FYC_CHECK_LAUNCH("fyc_ff_block");
  return 0;
}
