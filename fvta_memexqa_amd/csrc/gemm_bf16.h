// bf16 tile engine on v_mfma_f32_32x32x16_bf16 (bf16 operands, fp32 accumulate).
//
// Structure (MI355X guide, "glds + counted vmcnt + raw barrier"):
//  * block tile 256 x 128, BK = 32, four waves stacked in M (wave tile 64 x 128 =
//    2 x 4 MFMA tiles, 6 LDS fragments per 8 MFMAs), 256 threads;
//  * operands go global -> LDS directly (buffer_load_dwordx4 ... lds, 1 KiB per
//    wave-instruction, no VGPR staging, no ds_write), three LDS stages of 24 KiB
//    (72 KiB per workgroup -> two workgroups per CU), two tiles in flight behind a
//    counted s_waitcnt vmcnt and ONE raw s_barrier per k-tile;
//  * LDS images are unpadded (the DMA writes lane-linear) and XOR-swizzled in
//    16-byte chunks; the swizzle is applied to the per-lane SOURCE address and to
//    the fragment read address (never to the DMA destination);
//  * out-of-range rows/columns come back as zeros from the buffer descriptor's
//    range check (voffset sentinel), so there is no control flow around loads.
// Two image kinds:
//   "rows"    As[BM][32], Bs[BN][32]   (k contiguous)  -> ds_read_b128 fragments
//   "k-major" As[32][BM], Bs[32][BN]   (m/n contiguous, as the weight-gradient
//             operands lie in memory) -> ds_read_b64_tr_b16 fragments
// C/D layout equals the fp32 engine's (dtype independent on gfx950).
#pragma once
#include "fvta_common.h"

namespace fvta {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned short bf16_t;
typedef __attribute__((address_space(3))) void* lds_void_ptr;

union Pack8 {
  bf16x8 s;
  bf16x8_t b;
  f32x4 f;
};

constexpr unsigned GLDS_OOB = 0x80000000u;  // voffset sentinel: beyond any descriptor -> the load returns 0

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
}
// one wave-instruction: 64 lanes x 16 B from (rsrc + voff + soff) to LDS [dst, dst + 1 KiB) in lane order
__device__ __forceinline__ void glds16(__amdgpu_buffer_rsrc_t rsrc, bf16_t* dst, unsigned voff, unsigned soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void_ptr)dst, 16, voff, soff, 0, 0);
}
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// WN = wave columns: 1 -> 256 x 128 block tile, 4 waves, 72 KiB (two workgroups per CU);
//                    2 -> 256 x 256 block tile, 8 waves (4 x 2), 96 KiB (one workgroup per CU).  The wave tile is
// 64 x 128 either way (TM = 2 MFMA row tiles: 6 LDS fragments per 8 MFMAs); the wider block tile re-reads the A operand
// half as often.  WM < 4 wave rows: block tiles of 64 / 128 rows for calls with few sequences (the photo cell's backward
// step: a step is then a chain of K/32 k-tiles whose length is the DMA wave-instructions per k-tile).
// BK_ = k extent of a ring stage: 32 (64-byte rows; 1 KiB DMA pieces of 16 rows) or 64 (128-byte rows: a DMA piece is 8
// rows x one whole 128-byte line -- the CU's address unit takes a wave-instruction per ~31 cycles whatever it touches, but a
// piece of 16 half lines costs it more; row images only).
// TN_ = MFMA column tiles per wave (4: the 64 x 128 wave tile; 1: a 64 x 32 one -- WN = 4 waves of them side by side make the
// 64 x 128 block tile of the split engine's photo-cell backward step: the same rows and columns as ONE 64 x 128 wave, but four
// waves share the stage's DMA pieces and its MFMAs, so the step's chain of k-tiles is a quarter as long).
template <int WN, int TM_ = 2, int WM_ = 8 / TM_, int ST_ = 3, int BK_ = 32, int TN_ = 4>
struct TileCfgT {
  static_assert(TM_ == 2, "wave tile: two MFMA row tiles");
  static_assert(BK_ == 32 || BK_ == 64, "stage depth");
  static constexpr int TM = TM_, TN = TN_, WAVES_M = WM_, NWAVES = WAVES_M * WN;
  // ST_ = ring depth.  A k-loop's tile rate is (tiles in flight) / (load latency) -- a lone k-loop does not fill the
  // CU's load path -- so the weight-gradient GEMM (one 256 x 256 workgroup per CU, 32 KB per stage) runs a deeper ring;
  // the backward step measured no faster with four stages, and the 256 x 128 tile needs two workgroups per CU to fit.
  static constexpr int BM = 32 * TM_ * WM_, BN = 32 * TN_ * WN, BK = BK_, STAGES = ST_, NT = 64 * NWAVES;
  static constexpr int A_ELEMS = BM * BK, B_ELEMS = BN * BK;  // per stage
  static constexpr int STAGE_ELEMS = A_ELEMS + B_ELEMS;
  static constexpr int LDS_BYTES = STAGES * STAGE_ELEMS * 2;        // 73,728 (WN = 1)
  static constexpr int A_GLDS = A_ELEMS * 2 / 1024 / NWAVES;        // wave-instructions per wave per tile
  static constexpr int B_GLDS = B_ELEMS * 2 / 1024 / NWAVES;
};
typedef TileCfgT<1> TileCfg;

// ---- accumulators + fragment reads -------------------------------------------------------------
// chunk (16 bytes) c of row r of a row image is stored at chunk c ^ row_swz<BK>(r)
template <int BK>
__device__ __forceinline__ constexpr int row_swz(int r) { return BK == 32 ? ((r >> 2) & 3) : ((r >> 1) & 7); }

// X2 (the split engine, precision bf16x3: "two stored terms, three products"): every operand value is stored as TWO bf16
// terms, interleaved in groups of 32 -- element k of a row lies at il32(k) = (k / 32) * 64 + k % 32, its low term 32 elements
// further -- so that a 64-element stage row is one contiguous 128-byte line holding [hi of 32 k | lo of the same 32 k].  A
// k-step reads a_hi, a_lo, b_hi, b_lo ONCE and issues hi hi + hi lo + lo hi from those registers: 4 fragment sets per 3 MFMA
// groups (the three-term form of round 2-5 -- [hi | hi | lo] x [hi | lo | hi] rows through the plain loop -- staged and read 6).
template <int WN, int TM_ = 2, int WM_ = 8 / TM_, int ST_ = 3, int BK_ = 32, bool X2_ = false, int TN_ = 4>
struct MmaBT {
  typedef TileCfgT<WN, TM_, WM_, ST_, BK_, TN_> Cfg;
  static constexpr int BK = BK_;
  static constexpr bool X2 = X2_;
  static_assert(!X2_ || BK_ == 64, "split engine: 64-element stage rows (32 k of hi | lo)");
  static constexpr int TM = TM_, TN = TN_, WCOLS = 32 * TN_, WAVES_M = Cfg::WAVES_M, WAVES_N = 1, BM = Cfg::BM, BN = Cfg::BN;  // WAVES_N: per wave column
  static constexpr int WROWS = 32 * TM;  // rows of a wave tile
  f32x16 acc[TM][TN];
  int wave_all, wave, wn, lane, l31, hf;  // wave = row of the WAVES_M x WN wave grid (the M position), wn = its column

  static __device__ __forceinline__ void mfma(f32x16& c, bf16x8_t a, bf16x8_t b) {
#ifdef FVTA_GEMM_NO_MMA  // timing experiments: the operand stream without the matrix pipe
    asm volatile("" ::"v"(a), "v"(b));
#else
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
#endif
  }

  __device__ __forceinline__ void init(int tid) {
    wave_all = tid >> 6;
    wave = wave_all % WAVES_M;
    wn = wave_all / WAVES_M;
    lane = tid & 63;
    l31 = lane & 31;
    hf = lane >> 5;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  }

  // row images, BK-element rows, chunk c of row r stored at chunk c ^ row_swz<BK>(r)
  __device__ __forceinline__ void compute_rows(const bf16_t* __restrict__ As, const bf16_t* __restrict__ Bs) {
    if constexpr (X2) {  // chunks 0..3 of a stage row: hi of 32 k, chunks 4..7: lo
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        Pack8 ah[TM], al[TM], bh[TN], bl[TN];
        const int c = 2 * ks + hf;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const int r = wave * WROWS + i * 32 + l31;
          ah[i].f = *reinterpret_cast<const f32x4*>(As + r * 64 + ((c ^ row_swz<64>(r)) << 3));
          al[i].f = *reinterpret_cast<const f32x4*>(As + r * 64 + (((c + 4) ^ row_swz<64>(r)) << 3));
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int r = wn * WCOLS + j * 32 + l31;
          bh[j].f = *reinterpret_cast<const f32x4*>(Bs + r * 64 + ((c ^ row_swz<64>(r)) << 3));
          bl[j].f = *reinterpret_cast<const f32x4*>(Bs + r * 64 + (((c + 4) ^ row_swz<64>(r)) << 3));
        }
        // (term-major: the TM x TN independent accumulators between two MFMAs on the same one)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) mfma(acc[i][j], al[i].b, bh[j].b);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) mfma(acc[i][j], ah[i].b, bl[j].b);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) mfma(acc[i][j], ah[i].b, bh[j].b);
      }
      return;
    }
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      Pack8 a[TM], b[TN];
      const int c = 2 * ks + hf;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int r = wave * WROWS + i * 32 + l31;
        a[i].f = *reinterpret_cast<const f32x4*>(As + r * BK + ((c ^ row_swz<BK>(r)) << 3));
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int r = wn * WCOLS + j * 32 + l31;
        b[j].f = *reinterpret_cast<const f32x4*>(Bs + r * BK + ((c ^ row_swz<BK>(r)) << 3));
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          mfma(acc[i][j], a[i].b, b[j].b);
    }
  }

  // the same in two halves, so that a main loop can put other work between a k-step's fragment reads and its MFMAs
  struct Frags {
    bf16x8_t a[TM], b[TN];
  };
  __device__ __forceinline__ void load_rows(const bf16_t* __restrict__ As, const bf16_t* __restrict__ Bs, int ks, Frags& f) const {
    static_assert(BK == 32, "the software-pipelined loop is built for 32-deep stages");
    const int c = 2 * ks + hf;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int r = wave * WROWS + i * 32 + l31;
      Pack8 p;
      p.f = *reinterpret_cast<const f32x4*>(As + r * 32 + ((c ^ ((r >> 2) & 3)) << 3));
      f.a[i] = p.b;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int r = wn * WCOLS + j * 32 + l31;
      Pack8 p;
      p.f = *reinterpret_cast<const f32x4*>(Bs + r * 32 + ((c ^ ((r >> 2) & 3)) << 3));
      f.b[j] = p.b;
    }
  }
  __device__ __forceinline__ void mma_frags(const Frags& f) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) mfma(acc[i][j], f.a[i], f.b[j]);
  }

  // k-major image [32][LD] (LD = 256 or 128 elements), chunk c of row k stored at c ^ ((k & 3) << 2).
  // Transposing read: group g = lane>>4 reads the 4(k) x 16(m) blocks at k0 = kbase + 8(g>>1) (+4),
  // m0 = mbase + 16(g&1); lane 4q+p of the group supplies row k0+q, columns m0+4p..+3 and receives
  // column m0 + (lane&15).
  template <int LD>
  __device__ __forceinline__ bf16x8_t tr_frag(const bf16_t* img, int kbase, int mbase) const {
    const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
    const int k0 = kbase + 8 * (g >> 1) + q;  // (k0 & 3) == q, also for k0 + 4
    const int col = mbase + 16 * (g & 1) + 4 * p;
    const int off = (((col >> 3) ^ (q << 2)) << 3) + (col & 7);
    typedef s16x4 __attribute__((address_space(3))) * lds_ptr;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(img + k0 * LD + off));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(img + (k0 + 4) * LD + off));
    Pack8 r;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      r.s[e] = lo[e];
      r.s[e + 4] = hi[e];
    }
    return r.b;
  }
  __device__ __forceinline__ void compute_kmajor(const bf16_t* __restrict__ As, const bf16_t* __restrict__ Bs) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8_t a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) a[i] = tr_frag<BM>(As, ks * 16, wave * WROWS + i * 32);
#pragma unroll
      for (int j = 0; j < TN; ++j) b[j] = tr_frag<BN>(Bs, ks * 16, wn * WCOLS + j * 32);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) mfma(acc[i][j], a[i], b[j]);
    }
  }

  __device__ __forceinline__ int row_of(int i, int r) const { return wave * WROWS + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hf; }
  __device__ __forceinline__ int col_of(int j) const { return wn * WCOLS + j * 32 + l31; }
};
typedef MmaBT<1> MmaB;

// ---- the split engine's k-major tile (the weight gradient) ---------------------------------------
// 256 x 256 output tile, 8 waves (4 x 2), wave tile 64 x 128 like MmaBT<2>.  Both operands are k-major in memory with their
// columns in the il32 two-term layout, so a 256-column logical tile is 512 PHYSICAL columns: a stage is 16 k-rows (one
// k-step of 16 sequence rows) x 512 columns x 2 operands = 32 KB, four stages.  Per stage a wave reads a_hi, a_lo (2 row
// tiles) and b_hi, b_lo (4 column tiles) -- 12 fragments, 24 transposing reads -- for 24 MFMAs; the three-term form read 18
// fragments per 24 MFMAs and staged 1.5x the bytes.
struct DwX2Cfg {
  static constexpr int TM = 2, WAVES_M = 4, NWAVES = 8, BM = 256, BN = 256, KR = 16, PCOLS = 512, STAGES = 4, NT = 512;
  static constexpr int A_ELEMS = KR * PCOLS, B_ELEMS = KR * PCOLS, STAGE_ELEMS = A_ELEMS + B_ELEMS;
  static constexpr int LDS_BYTES = STAGES * STAGE_ELEMS * 2;  // 131,072
  static constexpr int A_GLDS = A_ELEMS * 2 / 1024 / NWAVES, B_GLDS = B_ELEMS * 2 / 1024 / NWAVES;  // 2 + 2
};
struct MmaX2K {
  typedef DwX2Cfg Cfg;
  static constexpr int TM = 2, TN = 4, WROWS = 64, BM = 256, BN = 256;
  f32x16 acc[TM][TN];
  int wave_all, wave, wn, lane, l31, hf;
  __device__ __forceinline__ void init(int tid) {
    wave_all = tid >> 6;
    wave = wave_all % 4;
    wn = wave_all / 4;
    lane = tid & 63;
    l31 = lane & 31;
    hf = lane >> 5;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  }
  // image [16 k][512 physical columns], chunk c of k-row k stored at c ^ ((k & 3) << 2) (KMajorSrc<512, .>); the fragment of
  // the 32 physical columns from pbase, k-rows 0..15 -- read exactly as MmaBT::tr_frag reads a 16-k block
  __device__ __forceinline__ bf16x8_t tr_frag(const bf16_t* img, int pbase) const {
    const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
    const int k0 = 8 * (g >> 1) + q;
    const int col = pbase + 16 * (g & 1) + 4 * p;
    const int off = (((col >> 3) ^ (q << 2)) << 3) + (col & 7);
    typedef s16x4 __attribute__((address_space(3))) * lds_ptr;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(img + k0 * Cfg::PCOLS + off));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(img + (k0 + 4) * Cfg::PCOLS + off));
    Pack8 r;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      r.s[e] = lo[e];
      r.s[e + 4] = hi[e];
    }
    return r.b;
  }
  __device__ __forceinline__ void compute_kmajor(const bf16_t* __restrict__ As, const bf16_t* __restrict__ Bs) {
    bf16x8_t ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) {  // logical row tile (wave, i) = physical columns 64 (2 wave + i) .. : [hi 32 | lo 32]
      ah[i] = tr_frag(As, 64 * (2 * wave + i));
      al[i] = tr_frag(As, 64 * (2 * wave + i) + 32);
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      bh[j] = tr_frag(Bs, 64 * (4 * wn + j));
      bl[j] = tr_frag(Bs, 64 * (4 * wn + j) + 32);
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
  }
  __device__ __forceinline__ int row_of(int i, int r) const { return wave * WROWS + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hf; }
  __device__ __forceinline__ int col_of(int j) const { return wn * 128 + j * 32 + l31; }
};

// ---- per-lane DMA source offsets ---------------------------------------------------------------
// Row image of ROWS rows: wave-instruction n (0 .. ROWS/16-1) fills LDS chunks [64n, 64n+64): unit
// U = 64n + lane -> row U>>2, physical chunk U&3, i.e. logical chunk (U&3) ^ ((row>>2)&3).
// voff[j] (instruction n = wave*PER + j) = row * ld_bytes + 16 * logical chunk, or GLDS_OOB for rows
// >= nrows.  The k position is added through the scalar offset at issue time.
template <int PER, int BK = 32>
struct RowSrc {
  unsigned voff[PER];
  // RPW < 64: a block tile whose 64-row wave tiles hold only RPW rows each (image row r = global row row0 + (r / 64) RPW +
  // r % 64, rows r % 64 >= RPW read as zeros): the same 256-row MFMA tile over fewer rows, so that a grid of them covers
  // more CUs (lstm_bwd_fused_bf16)
  template <int RPW = 64>
  __device__ __forceinline__ void setup(int wave, int lane, int row0, int nrows, unsigned ld_bytes) {
    constexpr int CPR = BK / 8;  // 16-byte chunks per image row
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      const int U = (wave * PER + j) * 64 + lane;
      const int row = U / CPR, c = (U % CPR) ^ row_swz<BK>(row);
      const int grow = RPW == 64 ? row0 + row : row0 + (row >> 6) * RPW + (row & 63);
      voff[j] = (grow < nrows && (RPW == 64 || (row & 63) < RPW)) ? (unsigned)grow * ld_bytes + 16u * c : GLDS_OOB;
    }
  }
  __device__ __forceinline__ void issue(__amdgpu_buffer_rsrc_t rsrc, bf16_t* stage, int wave, unsigned soff) const {
#pragma unroll
    for (int j = 0; j < PER; ++j) glds16(rsrc, stage + (wave * PER + j) * 512, voff[j], soff);
  }
};

// k-major image [32][COLS]: unit U = 64n + lane -> k-row U / (COLS/8), physical chunk U % (COLS/8),
// logical chunk pc ^ ((k&3)<<2).  voff = k * ld_bytes + (col0*2 + 16c) or GLDS_OOB for columns >= ncols.
// The k position (k0 * ld_bytes) is the scalar offset; rows beyond the operand's end fall off the
// descriptor (its size is exactly nrows * ld_bytes).
template <int COLS, int PER>
struct KMajorSrc {
  unsigned voff[PER];
  __device__ __forceinline__ void setup(int wave, int lane, int col0, int ncols, unsigned ld_bytes) {
    constexpr int CPR = COLS / 8;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      const int U = (wave * PER + j) * 64 + lane;
      const int k = U / CPR, c = (U % CPR) ^ ((k & 3) << 2);
      voff[j] = (col0 + 8 * c < ncols) ? (unsigned)k * ld_bytes + 2u * (unsigned)col0 + 16u * c : GLDS_OOB;
    }
  }
  __device__ __forceinline__ void issue(__amdgpu_buffer_rsrc_t rsrc, bf16_t* stage, int wave, unsigned soff) const {
#pragma unroll
    for (int j = 0; j < PER; ++j) glds16(rsrc, stage + (wave * PER + j) * 512, voff[j], soff);
  }
};

// ---- the pipeline --------------------------------------------------------------------------------
// issue(tile, a_stage, b_stage) starts the DMA of one k-tile; ntiles k-tiles are consumed.
// Every wave issues exactly A_GLDS + B_GLDS = 6 wave-instructions per tile, so "all but the newest
// tile have landed" is s_waitcnt vmcnt(6).  The barrier after the wait both publishes tile t to all
// waves and retires every wave's reads of stage (t-1)%3, which the next issue overwrites.
//
// Software-pipelined variant (row images only; the forward step's loop).  A k-tile is two k-steps of fragments; the loop
// keeps ONE k-step of fragments in flight at all times, across the barrier:
//     read F1 <- (tile t, step 1) | MFMA F0 | wait tile t+1, barrier | DMA tile t+3 -> stage t | read F0 <- (t+1, step 0) | MFMA F1
// so (a) the LDS latency of a k-step's fragment reads hides behind the other k-step's eight MFMAs also at the tile
// boundary, where the plain loop stalls on freshly issued reads after every barrier, and (b) a stage is released in the
// MIDDLE of its tile (both k-steps are in registers by then), which puts THREE tiles in flight on the same three stages.
// (Measured: forward -2.7 %; the backward / dx loops no faster, the k-major loop spills with it.)
template <class Mma, class Issue>
__device__ __forceinline__ void glds_mainloop_sp(Mma& mma, Issue&& issue, int ntiles, bf16_t* smem) {
  typedef typename Mma::Cfg TileCfg;
  constexpr int G = TileCfg::A_GLDS + TileCfg::B_GLDS;  // DMA wave-instructions per wave and tile
  auto a_stage = [&](int t) { return smem + (t % TileCfg::STAGES) * TileCfg::STAGE_ELEMS; };
  auto load = [&](int t, int ks, typename Mma::Frags& f) {
    const bf16_t* As = a_stage(t);
    mma.load_rows(As, As + TileCfg::A_ELEMS, ks, f);
  };
  if (ntiles <= 0) return;
  issue(0, a_stage(0), a_stage(0) + TileCfg::A_ELEMS);
  if (ntiles > 1) issue(1, a_stage(1), a_stage(1) + TileCfg::A_ELEMS);
  if (ntiles > 2) issue(2, a_stage(2), a_stage(2) + TileCfg::A_ELEMS);
  if (ntiles > 2)
    wait_vmcnt<2 * G>();
  else if (ntiles > 1)
    wait_vmcnt<G>();
  else
    wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  typename Mma::Frags f0, f1;
  load(0, 0, f0);
  for (int t = 0; t < ntiles; ++t) {
    load(t, 1, f1);
    __builtin_amdgcn_sched_barrier(0);
    mma.mma_frags(f0);
    __builtin_amdgcn_sched_barrier(0);
    if (t + 1 < ntiles) {
      // tiles 0 .. t+2 are issued; tile t+1 has landed once at most tile t+2's loads are outstanding
      if (t + 2 < ntiles)
        wait_vmcnt<G>();
      else
        wait_vmcnt<0>();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's reads of stage t are complete
      __builtin_amdgcn_s_barrier();                          // tile t+1 visible to all; stage t free
      asm volatile("" ::: "memory");
      if (t + 3 < ntiles) issue(t + 3, a_stage(t + 3), a_stage(t + 3) + TileCfg::A_ELEMS);
      load(t + 1, 0, f0);
      __builtin_amdgcn_sched_barrier(0);
    }
    mma.mma_frags(f1);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// What the stamps show (lstm_dx_bf16, 256 x 256 tile, 64-deep stages, per k-tile and wave, shader-clock counts): behind the
// barrier the eight waves' 64 DMA pieces queue at the CU's address unit, which accepts one 1-KB piece per ~35 cycles -- waves
// 0-3 are through after ~680, their SIMD partners 4-7 after ~2150 -- and a wave issues no MFMA while it stands there; then
// ~1300 of fragment reads + MFMAs.  A k-tile costs the address unit's time PLUS the matrix time (~4300), not their maximum.
// Spreading the refill over the MFMA groups (one piece per four MFMAs, SIMD partners in opposite phase; two 64-deep or four
// 32-deep stages) only moves the queueing: every piece then blocks its wave ~300 cycles, 4650-4950 per k-tile.
#ifdef FVTA_LOOP_STAMP  // -DFVTA_LOOP_STAMP (tools/r04_loop_stamps.py): where a k-tile's cycles go, per wave of one workgroup
static __device__ unsigned long long g_loop_stamp[16][4];  // [wave][wait, barrier, issue, compute] summed over the k-tiles
#define FVTA_LS(i) do { if (ls_on) { const unsigned long long now_ = __builtin_readcyclecounter(); ls_sum[i] += now_ - ls_t; ls_t = now_; } } while (0)
#else
#define FVTA_LS(i) do { } while (0)
#endif

template <bool KMAJOR, class Mma, class Issue>
__device__ __forceinline__ void glds_mainloop(Mma& mma, Issue&& issue, int ntiles, bf16_t* smem) {
  typedef typename Mma::Cfg TileCfg;
  constexpr int S = TileCfg::STAGES, G = TileCfg::A_GLDS + TileCfg::B_GLDS;
  static_assert(S >= 2 && S <= 5, "ring depth");
  auto a_stage = [&](int t) { return smem + (t % S) * TileCfg::STAGE_ELEMS; };
  if (ntiles <= 0) return;
#ifdef FVTA_LOOP_STAMP
  const bool ls_on = blockIdx.x == 8 && blockIdx.y == 0 && blockIdx.z == 3;
  unsigned long long ls_sum[4] = {0, 0, 0, 0}, ls_t = __builtin_readcyclecounter();
#endif
#pragma unroll
  for (int p = 0; p < S - 1; ++p)
    if (p < ntiles) issue(p, a_stage(p), a_stage(p) + TileCfg::A_ELEMS);
  FVTA_LS(2);
  for (int t = 0; t < ntiles; ++t) {
    // tiles up to t + S - 2 are issued; tile t has landed once only the newer ones are outstanding
    const int ahead = min(S - 2, ntiles - 1 - t);
    if (ahead >= 3)
      wait_vmcnt<3 * G>();
    else if (ahead == 2)
      wait_vmcnt<2 * G>();
    else if (ahead == 1)
      wait_vmcnt<G>();
    else
      wait_vmcnt<0>();
    FVTA_LS(0);
    __builtin_amdgcn_s_barrier();  // tile t visible to all waves; every wave's reads of stage (t - 1) % S are retired
    asm volatile("" ::: "memory");
    FVTA_LS(1);
    if (t + S - 1 < ntiles) issue(t + S - 1, a_stage(t + S - 1), a_stage(t + S - 1) + TileCfg::A_ELEMS);
    FVTA_LS(2);
    const bf16_t* As = a_stage(t);
    if constexpr (KMAJOR)
      mma.compute_kmajor(As, As + TileCfg::A_ELEMS);
    else
      mma.compute_rows(As, As + TileCfg::A_ELEMS);
#ifdef FVTA_LOOP_STAMP
    asm volatile("s_nop 0" ::: "memory");
#endif
    FVTA_LS(3);
  }
#ifdef FVTA_LOOP_STAMP
  if (ls_on && mma.lane == 0)
    for (int i = 0; i < 4; ++i) g_loop_stamp[mma.wave_all][i] = ls_sum[i];
#endif
}

}  // namespace fvta
