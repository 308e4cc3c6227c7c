// One GEGLU feed-forward block of a transformer in ONE kernel (reference diffusers/models/attention.py:772-775 FeedForward,
// 819-821 GEGLU; called from animatediff/models/attention.py:489-564 BasicTransformerBlock and motion_module.py:270-283
// TemporalTransformerBlock), merged with the output projection that follows it as engine/weights.py::_ff does:
//
//   h   = GEGLU( LN(tok) W1^T + b1 )                       [rows][4C]       (LayerNorm folded: rstd (tok W1'^T - mean colsum) + b1')
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
//   * the weights arrive as one pre-packed stream (engine/weights.py::pack_ff_block): 46 stages of <= 61 KiB, every 1-KiB piece
//     already in MFMA fragment order (lane l holds W[16 j + (l & 15)][32 s + 8 (l >> 4) .. +8]), so a stage is ONE contiguous
//     global -> LDS DMA (global_load_lds, 16 B / lane, lane-linear image = conflict-free ds_read_b128 at base + lane * 16) into a
//     2-deep ring: stages 0-4 = the tok Wp^T projection (2 k-steps each), stage 5 + c = {W1 rows of hidden chunk c (32 units =
//     2 x (16 value + 16 gate) rows), their colsum / bias, the W2' columns of chunk c - 1};
//   * per chunk a wave runs 80 MFMAs of FF1 (K = C from registers), the GEGLU gate on its 32 x 32 results - which ARE the MFMA
//     operand of FF2 as they stand (the k-slots of the W2' fragments are packed in the order the gate outputs sit in the lanes:
//     no shuffle, no LDS round trip) - and 40 MFMAs of FF2 for the PREVIOUS chunk, so that the gate's VALU work has independent
//     matrix work beside it.  One barrier per stage;
//   * epilogue: residual tile by DMA into the (now idle) ring (row pitch padded against bank conflicts), + bias + accumulators
//     in f32, one rounding to bf16 in LDS, the tile leaves in 640-B rows (16 B / lane); per-(row tile, channel) {sum, sum sq} of the stored values for the
//     GroupNorm that consumes the block (fyc_gemm's chan_parts layout with tile_rows = 128, one slot).
//
// Built for the level where it pays (C = 320, hidden 1280, bf16, rows % 128 == 0); other shapes keep the unfused schedule.
// Compiled WITHOUT -amdgpu-mfma-vgpr-form (see _build.py): the accumulators must live in AGPRs for the 512-register budget.
#include <mutex>
#include <type_traits>

#include "fyc_common.h"

namespace {

constexpr int C_ = 320, HID = 1280, ROWS = 128;
constexpr int KS = C_ / 32;                    // 10 MFMA k-steps over C
constexpr int NB = C_ / 16;                    // 20 column blocks of the output
constexpr int CHUNKS = HID / 32;               // 40 hidden chunks of 32 units
constexpr int PIECE = 1024;                    // one MFMA fragment for all 64 lanes
// The weight stream is cut into HALF-STAGES of HP = 32 pieces (32 KiB) that go through a 4-deep LDS ring: the pieces of half
// h + 2 are requested while half h computes, i.e. a whole stage-time before they are needed.  (The first version had 46 stages
// of 61 pieces in a 2-deep ring: the refill of a slot could only start when its previous stage was done, one stage before its
// use, and a 61-KiB refill alone takes 1.4 us of the 2.1 us a stage took - profiles/r03_ff_block_ablation.txt, DMA-only run.)
//   half t < 10           k-step t of the projection: pieces j < 20 = Wp rows 16 j .. +16, columns 32 t .. +32
//   half 10 + 2 c  ("A")  pieces 4 s + q (s < 7): W1 rows 64 c + 16 q .. +16, columns 32 s .. +32; piece 28: f32 bias[64] of chunk c - 1
//   half 11 + 2 c  ("B")  pieces 4 (s - 7) + q (s = 7, 8, 9): the rest of W1 of chunk c; pieces 12 + j (j < 20): W2' of chunk c - 1
//   halves 90, 91         piece 28 / pieces 12 + j of the same for chunk 39
constexpr int HP = 32, HALF_BYTES = HP * PIECE, NSLOT = 4;
constexpr int SA = 7;                          // FF1 k-steps in half A
constexpr int P_BIAS = 4 * SA, P_W2 = 4 * (KS - SA);
constexpr int NHALF = KS + 2 * CHUNKS + 2;     // 92
// the residual / output tile of the epilogue (overlays the ring).  Row pitch 672 B: consecutive rows are 168 dwords = 40 (mod 64
// banks) apart, so the 16 rows x 4 quads x 8 B of one in-place add spread over all banks; at the natural 640 B (32 mod 64) they
// met in two banks, 8-way (the same defect cost csrc/temporal_block_rr.hip a fifth of its tile time: profiles/r03_temporal_block_rr_phases.txt)
constexpr int TP = C_ * 2 + 32;
constexpr int TILE_BYTES = ROWS * TP;          // 86016 = 84 pieces
constexpr int SCR_BYTES = 12 * 40 * 16 * 4;      // statistics partials of the epilogue: [row slice (6 or 12)][column group][8 sums | 8 sums of squares]
constexpr int LDS_BYTES = NSLOT * HALF_BYTES;  // 131072
static_assert(TILE_BYTES % PIECE == 0 && TILE_BYTES <= ((NHALF - 2) % NSLOT) * HALF_BYTES + P_BIAS * PIECE && (NHALF - 1) % NSLOT == 3,
              "the residual tile lands while the last two halves (slots 2 and 3) still hold the bias and W2' pieces of chunk 39");
static_assert(TILE_BYTES + SCR_BYTES <= LDS_BYTES, "epilogue tile + statistics scratch overlay the ring");
static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");

struct FFP {
  const bf16_t* x; const bf16_t* res; bf16_t* out;
  const char* ws;            // packed weight stream, NHALF * HALF_BYTES
  const float* b_out;        // [C]
  float* parts;              // [rows / 128][C][2] or null
  int ntiles;                // rows / 128
  float eps;
};

// 1 KiB global -> LDS by DMA, issued from inline asm: wave-uniform 64-bit base + one 32-bit lane offset, destination = wave-uniform
// LDS byte address (+ 16 B per lane, implicit).  NOT the builtin: while a builtin LDS-DMA is outstanding hipcc turns every
// counted lgkmcnt wait of the fragment reads into lgkmcnt(0) (it treats the DMA as a possible out-of-order LDS event), so each
// k-step paid the full LDS round trip for fragments that were requested a k-step ahead (measured: 35 % of the wave's cycles in
// s_waitcnt with and without the DMA).  The compiler does not count these loads: every stage barrier has its own vmcnt(0).
__device__ __forceinline__ void dma16(const char* gbase, unsigned voff, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(gbase), "s"(lds_dst) : "memory");
}
// Four consecutive pieces (4 KiB of the stream -> 4 KiB of the ring) by one wave: ONE M0 write, the instruction's immediate
// offset moves the global and the LDS address together.  A wave alone on its SIMD issues one instruction per ~4 cycles, so
// beside a 16-cycle MFMA only three other instructions are free; piece by piece the DMA cost 8 issue slots per KiB (M0 save /
// set / restore, wait state, 64-bit address add, the load) = a quarter of a stage's issue budget (PMC: 49 % of the wave's
// cycles issuing at 3.7 non-MFMA instructions per MFMA, profiles/r03_rr_kernels_pmc.txt); this form costs 1.75.
// (M0 is not preserved: nothing else in these kernels uses it - hipcc treats it as reserved.)
__device__ __forceinline__ void dma16x4(const char* gbase, unsigned voff, unsigned lds_dst) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\t"
               "global_load_lds_dwordx4 %0, %1\n\t"
               "global_load_lds_dwordx4 %0, %1 offset:1024\n\t"
               "global_load_lds_dwordx4 %0, %1 offset:2048\n\t"
               "global_load_lds_dwordx4 %0, %1 offset:3072"
               :: "v"(voff), "s"(gbase), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void dma16v(const void* gsrc, unsigned lds_dst) {                        // per-lane source address
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void dma_landed_barrier() {       // this wave's DMA pieces have landed, then the workgroup meets
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
}

// sum over the four 16-lane rows of a wave (every lane gets it): gfx950's v_permlane16_swap / v_permlane32_swap, plain VALU.
// NOT __shfl_xor: that is ds_bpermute_b32, an LDS-queue instruction, and hipcc (which cannot see the asm DMAs above) waits for it
// with a COUNTED lgkmcnt between the fragment reads of the projection stages it sinks this code into - with LDS-DMA writes in
// flight the result was consumed early in ~12 % of the 16-row blocks (row statistics off by ~1e-3: tools/ff_stress.py found the
// kernel's output changing from launch to launch; profiles/r03_ff_block_race.txt)
__device__ __forceinline__ float rows_sum(float v) {
  auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
  auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

__device__ __forceinline__ f32x4 mfma(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }

__device__ __forceinline__ bf16x8 frag(const char* sl, int piece) { return *reinterpret_cast<const bf16x8*>(sl + piece * PIECE); }

// RI: 16-row blocks per wave.  2 = four waves, one per SIMD with the whole 512-register file (32 rows per wave, every weight
// fragment feeds two MFMAs);  1 = eight waves, TWO per SIMD at 256 registers (16 rows per wave, a fragment feeds one MFMA: twice the
// LDS reads, but a second wave to issue while the first one waits).  fyc_set_tuning key 8 = 2 selects RI = 1.
#ifdef FF_TIMING                                                // phase timestamps of wave 0 of every workgroup (tools/ff_probe.py prints them)
__device__ unsigned long long g_ff_time[1024 * 16];
#define FF_MARK(k) do { if (tid == 0) g_ff_time[(blockIdx.x & 1023) * 16 + (k)] = __builtin_readcyclecounter(); } while (0)
#else
#define FF_MARK(k) do {} while (0)
#endif

template <int RI>
__global__ void __launch_bounds__(64 * 8 / RI) ff_block_kernel(const FFP p) {
  constexpr int NW = 8 / RI, NT = 64 * NW, RW = 16 * RI;     // waves, threads, rows per wave
  constexpr int NSL = NT / 40;                                 // row slices of the copy-out pass (6 or 12)
  constexpr int GPW = HP / 4 / NW;                             // groups of four pieces per wave and half (2 or 1)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  // One row tile per workgroup.  (A persistent loop over tiles was tried - with and without requesting the next tile's tokens
  // during the last stage: inside an outer loop the register allocator spilled 85-95 registers of this 496-register kernel.)
  {
  const int tile = blockIdx.x;
  const long long row0 = (long long)tile * ROWS;
  const int lane = tid & 63;
  const int g = lane >> 4, r16 = lane & 15;
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
  auto half_landed_barrier = [&]() {                          // the oldest outstanding half of this wave has landed, then the workgroup meets
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(4 * GPW) : "memory");
    __syncthreads();
  };

  // ---- the wave's 32 token rows as MFMA operands: lane (row r16, quad g) holds x[row][32 s + 8 g .. +8] ---------------------
  bf16x8 xa[RI][KS];
  auto load_x = [&](int tile) {
    const bf16_t* xr = p.x + ((long long)tile * ROWS + wave * RW + r16) * C_ + g * 8;
#pragma unroll
    for (int i = 0; i < RI; ++i)
#pragma unroll
      for (int s = 0; s < KS; ++s) xa[i][s] = *reinterpret_cast<const bf16x8*>(xr + i * 16 * C_ + s * 32);
  };
  FF_MARK(0);
  load_x(tile);
#pragma unroll
  for (int n = 0; n < GPW; ++n) dma_group(0, n);
#pragma unroll
  for (int n = 0; n < GPW; ++n) dma_group(1, n);

  // LayerNorm statistics of the lane's two rows (two-pass, in registers; the four quads of a row meet through xor 16 / 32)
  float mu[RI], rs[RI];
#pragma unroll
  for (int i = 0; i < RI; ++i) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < KS; ++k) {
      const u32x4 t = __builtin_bit_cast(u32x4, xa[i][k]);
#pragma unroll
      for (int e = 0; e < 4; ++e) s += __uint_as_float(t[e] << 16) + __uint_as_float(t[e] & 0xffff0000u);
    }
#ifdef FF_BPERMUTE
    s += __shfl_xor(s, 16);
    s += __shfl_xor(s, 32);
#else
    s = rows_sum(s);
#endif
    const float m = s * (1.0f / C_);
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < KS; ++k) {
      const u32x4 t = __builtin_bit_cast(u32x4, xa[i][k]);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float a = __uint_as_float(t[e] << 16) - m, b = __uint_as_float(t[e] & 0xffff0000u) - m;
        q = __builtin_fmaf(a, a, q);
        q = __builtin_fmaf(b, b, q);
      }
    }
#ifdef FF_BPERMUTE
    q += __shfl_xor(q, 16);
    q += __shfl_xor(q, 32);
#else
    q = rows_sum(q);
#endif
    mu[i] = m;
    rs[i] = rsqrtf(q * (1.0f / C_) + p.eps);
  }

  f32x4 oacc[RI][NB];                                         // out rows 16 i + r16, columns 16 j + 4 g .. +4
#pragma unroll
  for (int i = 0; i < RI; ++i)
#pragma unroll
    for (int j = 0; j < NB; ++j) oacc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  FF_MARK(1);
  // ---- projection: out = tok Wp^T, half t = k-step t of all 20 column blocks ----------------------------------------------------
#pragma unroll
  for (int t = 0; t < KS; ++t) {
    half_landed_barrier();                                    // half t landed; slot (t + 2) & 3 is free (half t - 2 is done): refill it
    const char* sl = smem + (t & (NSLOT - 1)) * HALF_BYTES + lane16;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const bf16x8 wf = frag(sl, j);
#pragma unroll
      for (int i = 0; i < RI; ++i) oacc[i][j] = mfma(wf, xa[i][t], oacc[i][j]);
      if (j % 8 == 0 && j / 8 < GPW) dma_group(t + 2, j / 8);
    }
  }

  FF_MARK(2);
  // The tokens were needed raw for the projection; from here on they are only the FF1 operand: normalise them in place,
  // (x - mean) rstd rounded to bf16 (what the reference's autocast feeds its Linear), so that no LayerNorm term is left in the
  // per-chunk gate (gamma is folded into W1, beta into b1: engine/weights.py::fold_layernorm)
#pragma unroll
  for (int i = 0; i < RI; ++i)
#pragma unroll
    for (int k = 0; k < KS; ++k) {
      u32x4 t = __builtin_bit_cast(u32x4, xa[i][k]);
#pragma unroll
      for (int e = 0; e < 4; ++e)
        t[e] = pack_bf16x2((__uint_as_float(t[e] << 16) - mu[i]) * rs[i], (__uint_as_float(t[e] & 0xffff0000u) - mu[i]) * rs[i]);
      xa[i][k] = __builtin_bit_cast(bf16x8, t);
    }

  // ---- hidden chunks -----------------------------------------------------------------------------------------------------------
  // Chunk c:  half A = FF1(c) k-steps 0..6 -> hw (28 RI MFMAs)  ||  GEGLU gate of chunk c - 1 from hr (VALU) -> hb;
  //           half B = FF1(c) k-steps 7..9 (12 RI)  then  FF2(c - 1) with hb (20 RI MFMAs).
  // The gate of a chunk runs one chunk after its FF1 so that its ~250 VALU instructions have independent matrix work beside
  // them: a wave overlaps VALU with the matrix pipe only where the two alternate in program order, so the gate is cut into
  // units (row block, half) and one unit follows the MFMAs of a k-step.
  // gate unit u = (row block i, half h) of the pre-activations hr -> two packed bf16 pairs of the FF2 operand (two independent
  // polynomial chains: their dependent v_pk_fma steps fill each other's wait states): k-slots 8 g + e = hidden unit 4 g + e of
  // half 0 (e < 4), of half 1 (e >= 4)
  auto gate_unit = [&](int u, const f32x4 (&bi)[4], const f32x4 (&hr)[RI][4], u32x4 (&hbw)[RI]) {
    const int i = u >> 1, h = u & 1;
#ifdef FF_GATE_PACKED
    const f32x4 v = hr[i][2 * h] + bi[2 * h], gt = hr[i][2 * h + 1] + bi[2 * h + 1];
    const f32x2 lo = geglu_pair((f32x2){v[0], v[1]}, (f32x2){gt[0], gt[1]}), hi = geglu_pair((f32x2){v[2], v[3]}, (f32x2){gt[2], gt[3]});
    unsigned p0 = pack_bf16x2(lo.x, lo.y), p1 = pack_bf16x2(hi.x, hi.y);
#else
    // four independent SCALAR chains: beside MFMAs a packed f32 VALU instruction costs a lone wave ~22 cycles more than the two
    // scalar ones it replaces (MI355X_MICROARCH.md; half A of a chunk took 2 350 cycles for 900 cycles of MFMA with the packed gate)
    float o4[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) o4[e] = geglu_one(hr[i][2 * h][e] + bi[2 * h][e], hr[i][2 * h + 1][e] + bi[2 * h + 1][e]);
    unsigned p0 = pack_bf16x2(o4[0], o4[1]), p1 = pack_bf16x2(o4[2], o4[3]);
#endif
    asm volatile("" : "+v"(p0), "+v"(p1));                    // the unit's result is "used" here: LLVM's sinking passes would otherwise move the whole
    hbw[i][2 * h] = p0;                                       // unit down to FF2, its only real consumer, behind all the MFMAs it is meant to sit beside
    hbw[i][2 * h + 1] = p1;
  };
  auto load_consts = [&](const char* base, f32x4 (&bi)[4]) {  // bias (beta folded in) of the PREVIOUS chunk's 64 W1 rows
    const float* cst = reinterpret_cast<const float*>(base + P_BIAS * PIECE);
#pragma unroll
    for (int q = 0; q < 4; ++q) bi[q] = *reinterpret_cast<const f32x4*>(cst + q * 16 + g * 4);
  };
  // FF1 k-steps [S0, S1) of the chunk whose pieces start at `first` in the half at sl; the next k-step's fragments in flight;
  // `after(s)` runs behind the MFMAs of k-step s (DMA groups of half hnext, gate units)
  auto ff1_steps = [&](const char* sl, auto s0_, auto s1_, f32x4 (&hw)[RI][4], auto after) {
    constexpr int S0 = decltype(s0_)::value, S1 = decltype(s1_)::value;
    bf16x8 w[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) w[q] = frag(sl, q);
#pragma unroll
    for (int s = S0; s < S1; ++s) {
      bf16x8 n[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) n[q] = (s + 1 < S1) ? frag(sl, (s + 1 - S0) * 4 + q) : w[q];
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int i = 0; i < RI; ++i) hw[i][q] = mfma(w[q], xa[i][s], hw[i][q]);
      after(s);
      if (((s - S0) & 1) || s + 1 == S1) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < 4; ++q) w[q] = n[q];
    }
  };
  auto ff2 = [&](const char* sl, const u32x4 (&hbw)[RI]) {
    bf16x8 hb[RI];
#pragma unroll
    for (int i = 0; i < RI; ++i) hb[i] = __builtin_bit_cast(bf16x8, hbw[i]);
    bf16x8 w[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) w[q] = frag(sl, P_W2 + q);
#pragma unroll
    for (int jb = 0; jb < NB / 4; ++jb) {                     // four column blocks per step, the next four fragments in flight
      bf16x8 n[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) n[q] = (jb + 1 < NB / 4) ? frag(sl, P_W2 + (jb + 1) * 4 + q) : w[q];
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int i = 0; i < RI; ++i) oacc[i][jb * 4 + q] = mfma(w[q], hb[i], oacc[i][jb * 4 + q]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < 4; ++q) w[q] = n[q];
    }
  };
  using I0 = std::integral_constant<int, 0>;
  using IA = std::integral_constant<int, SA>;
  using IK = std::integral_constant<int, KS>;
  // one chunk = halves ha (A) and ha + 1 (B): FF1 -> hw; with_gate: gate + FF2 of the previous chunk from hr
  auto chunk = [&](int ha, f32x4 (&hw)[RI][4], const f32x4 (&hr)[RI][4], auto with_gate) {
    constexpr bool WITH_GATE = decltype(with_gate)::value;
    u32x4 hbw[RI];
    if (ha == KS + 40) FF_MARK(4);
    half_landed_barrier();                                    // half A landed; slot (ha + 2) & 3 free
    if (ha == KS + 40) FF_MARK(5);
    {
      const char* base = smem + (ha & (NSLOT - 1)) * HALF_BYTES;
      f32x4 bi[4];
      if constexpr (WITH_GATE) load_consts(base, bi);
#pragma unroll
      for (int i = 0; i < RI; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) hw[i][q] = (f32x4){0.f, 0.f, 0.f, 0.f};
      ff1_steps(base + lane16, I0{}, IA{}, hw, [&](int s) {
        if (!(s & 1) && (s >> 1) < GPW) dma_group(ha + 2, s >> 1);                 // the refill goes out in the first k-steps
        if constexpr (WITH_GATE) {                               // 2 RI gate units: behind k-steps 1, 3, 5, 6
          const int u = s == SA - 1 ? 3 : (s & 1) ? (s >> 1) : -1;
          if (u >= 0 && u < 2 * RI) gate_unit(u, bi, hr, hbw);
        }
      });
    }
    if (ha == KS + 40) FF_MARK(6);
    half_landed_barrier();                                    // half B landed
    if (ha == KS + 40) FF_MARK(7);
    {
      const char* base = smem + ((ha + 1) & (NSLOT - 1)) * HALF_BYTES;
      ff1_steps(base + lane16, IA{}, IK{}, hw, [&](int s) {
        if ((s - SA) < GPW) dma_group(ha + 3, s - SA);
      });
      if (ha == KS + 40) FF_MARK(8);
      if constexpr (WITH_GATE) ff2(base + lane16, hbw);
      if (ha == KS + 40) FF_MARK(9);
    }
  };

  f32x4 h0[RI][4], h1[RI][4];
  FF_MARK(3);
  chunk(KS, h0, h1, std::false_type{});
  for (int c = 1; c + 1 < CHUNKS; c += 2) {                  // chunks (1, 2), (3, 4), ..., (37, 38): pre-activation buffers alternate
    chunk(KS + 2 * c, h1, h0, std::true_type{});
    chunk(KS + 2 * c + 2, h0, h1, std::true_type{});
  }
  FF_MARK(10);
  chunk(NHALF - 4, h1, h0, std::true_type{});                 // chunk 39: its refills are halves 90, 91 (bias / W2' of chunk 39 only)
  {
    u32x4 hbw[RI];
    half_landed_barrier();                                    // half 90 landed (half 91 may still be in flight)
    {
      const char* base = smem + ((NHALF - 2) & (NSLOT - 1)) * HALF_BYTES;
      f32x4 bi[4];
      load_consts(base, bi);
#pragma unroll
      for (int u = 0; u < 2 * RI; ++u) gate_unit(u, bi, h1, hbw);
    }
    dma_landed_barrier();                                     // half 91 landed; every wave is done with slots 0 .. 2 below the bias piece
    if (p.res != nullptr) {                                   // residual tile over the idle part of the ring: the LDS image is linear, every
      const char* src = reinterpret_cast<const char*>(p.res + row0 * C_);    // lane fetches the 16 B that belong at its place
#pragma unroll 1
      for (int q = wave; q < TILE_BYTES / PIECE; q += NW) {
        const int off = q * PIECE + (int)lane16, row = off / TP;
        const int col = min(off - row * TP, C_ * 2 - 16);       // (the 32 pad bytes of a row re-fetch its last chunk)
        dma16v(src + row * (C_ * 2) + col, lds0 + q * PIECE);
      }
    }
    ff2(smem + ((NHALF - 1) & (NSLOT - 1)) * HALF_BYTES + lane16, hbw);
  }
  FF_MARK(11);

  // ---- epilogue -------------------------------------------------------------------------------------------------------------------
  dma_landed_barrier();                                            // residual tile landed; every wave is done with the ring
  FF_MARK(12);
  const float* bias_out = p.b_out;
#pragma unroll
  for (int i = 0; i < RI; ++i)
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      bf16_t* a = reinterpret_cast<bf16_t*>(smem + (wave * RW + i * 16 + r16) * TP) + j * 16 + g * 4;
      const f32x4 bo = *reinterpret_cast<const f32x4*>(bias_out + j * 16 + g * 4);
      float rr[4] = {0.f, 0.f, 0.f, 0.f}, v[4];
      if (p.res != nullptr) ElemIO<bf16_t>::ld4(a, rr);
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = oacc[i][j][r] + bo[r] + rr[r];
      ElemIO<bf16_t>::st4(a, v);
    }
  dma_landed_barrier();
  // flat pass over the finished tile: thread (column group cg of 8 channels, row slice rsl) copies rows rsl, rsl + 6, ... to HBM,
  // 16 B per lane and 3840 contiguous bytes per step, and sums its 8 columns for the statistics of the values as stored
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
            const float x0 = __uint_as_float(v[e] << 16), x1 = __uint_as_float(v[e] & 0xffff0000u);
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
      if (tid < 160) {                                        // one thread per column pair: add the six row slices
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
  if (a == nullptr || a->dtype != FYC_BF16 || a->C != C_ || a->hidden != HID || a->rows <= 0 || a->rows % ROWS != 0) return 0;
  if (a->chan_parts != nullptr && (a->cs_rows <= 0 || a->cs_rows % ROWS != 0 || a->rows % a->cs_rows != 0)) return 0;
  int64_t lds_cap = 0, n_cu = 0;
  device_limits(lds_cap, n_cu);
  if (lds_cap > 0 && lds_cap < LDS_BYTES) return 0;   // a device / partition mode with less LDS: the caller keeps the unfused schedule
  return 1;
}

extern "C" int fyc_ff_block(const fyc_ff_block_args* a, void* stream) {
  FYC_REQUIRE(a && a->x && a->out && a->wstream && a->b_out, "fyc_ff_block: null pointer");
  FYC_REQUIRE(fyc_ff_block_supported(a), "fyc_ff_block: built for bf16, C=320, hidden=1280, rows %% 128 == 0, cs_rows %% 128 == 0 and >= %d B of LDS (got C=%d hidden=%d rows=%d cs_rows=%d)",
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
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(ff_block_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
      if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(ff_block_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
      if (e != hipSuccess) FYC_FAIL(-3, "fyc_ff_block: %d bytes of dynamic LDS refused: %s", LDS_BYTES, hipGetErrorString(e));
      if (dev >= 0 && dev < kMaxDev) attr_done[dev] = true;
    }
  }
  const unsigned grid = (unsigned)p.ntiles;
  if (g_fyc_tuning[8] == 2) hipLaunchKernelGGL(ff_block_kernel<1>, dim3(grid), dim3(512), LDS_BYTES, (hipStream_t)stream, p);
  else hipLaunchKernelGGL(ff_block_kernel<2>, dim3(grid), dim3(256), LDS_BYTES, (hipStream_t)stream, p);
  FYC_CHECK_LAUNCH("fyc_ff_block");
  return 0;
}
