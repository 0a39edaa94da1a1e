// gemm_bf16_api.h - parameter block and host entry point of the bf16-resident GEMM (kernels: gemm_bf16.h, compiled in
// gemm_bf16_unit.hip).  engine.hip includes only this file, so the kernels can be rebuilt without it.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mcrn {

struct RowMap {           // off(r) = (r / inner) * hi + (r % inner) * lo ; inner <= 0: r * lo
    int inner;
    long long hi, lo;
};
static inline RowMap rm_plain(long long lo) { return RowMap{0, 0, lo}; }
static inline RowMap rm_two(int inner, long long hi, long long lo) { return RowMap{inner, hi, lo}; }
__host__ __device__ __forceinline__ long long rm_off(const RowMap& m, int r) {
    if (m.inner <= 0) return (long long)r * m.lo;
    const int q = r / m.inner;
    return (long long)q * m.hi + (long long)(r - q * m.inner) * m.lo;
}

// Operand contract (gemm_bf16.h): K-contiguous operands are readable and ZERO from seg_len to the next multiple of 64
// in every segment; [k][n] operands are readable and finite for k up to the next multiple of 64 of seg_len.
struct Bf16GemmP {
    const uint16_t* A;        // bf16 bits
    const uint16_t* B;
    RowMap am;                // A row m -> element offset of (m, k = 0) inside a segment
    RowMap bm;                // NT: B row n -> element offset ; NN: unused
    long long ldb;            // NN: elements between consecutive k rows of B
    int nseg, seg_len, tps;   // K segments, valid k per segment, k-tiles per segment = ceil(seg_len / BK) (set by the launcher)
    long long a_seg, b_seg;   // element offset between segments
    int nterm;                // 0 / 1: plain bf16 product.  3: hi/lo operand pairs - every K tile is FOUR images (A_hi, A_lo, B_hi, B_lo), fetched once,
    long long a_lo, b_lo;     //        multiplied as A_hi B_hi + A_hi B_lo + A_lo B_hi (the lo copies sit a_lo / b_lo elements behind the hi ones):
                              //        the library's bf16x3 arithmetic on bf16-resident operands.  K tiles are 16 or 32 deep there (plain: 32 or 64).
    int M, N;                 // valid rows of A ; valid columns (NT: rows of B; NN: multiple of 8)
    float* C;                 // fp32 result (nullable when only the bf16 copy is wanted)
    const float* Cin;         // nullable
    int cin_first_only;       // split-K: only split 0 adds Cin (the other splits write plain partial results)
    int cin_pre;              // 1: the workgroups that add Cin start their K loop FROM it (accumulators preloaded; alpha = beta = 1)
    RowMap cm;                // C row m -> element offset, columns contiguous unless cn_inner > 0
    int cn_inner;             // > 0: column j of C sits at (j / cn_inner) * cn_hi + j % cn_inner (cn_inner % 32 == 0): the
    int cn_hi;                //      h-channel block of every (node, sample) row of a plane set [n][b][Cp]; applies to C, Cin
    float alpha, beta;
    int nsplit, tiles_per_split;   // split-K over k-tiles ; C / Cin of split z at + z * slab
    long long slab;
    long long slab2;          // > 0: split z >= 1 lands at C + slab + (z - 1) * slab2 (split 0 at C): the partial sums of a product
                              //      whose first slab is the real output and whose further slabs are a separate run of planes
    uint16_t* Cb;             // optional bf16 copy of the result (row map cbm)
    RowMap cbm;
    int xcd;                  // 1: XCD-aware tile order
    int wide_cb;              // set by the launcher: bf16-only output staged through LDS into 16-byte stores
    // host side only (bench.py's roofline leg): when set, the launch attaches these two events to the dispatch itself
    // (hipExtLaunchKernelGGL), so that their elapsed time is the kernel's own begin -> end - what rocprofv3 reports - instead
    // of the span between two separately recorded event packets (~3 us longer per launch)
    void *ev0, *ev1;
    // roofline leg only, else null: workgroup 0 stores {shader clock counter, 100 MHz wall clock} at both ends of its K loop (4 x u64).
    // d(clock64) / d(wall_clock64) x 100 MHz = the clock the chip HOLDS under this kernel: MI355X clocks to its power budget, and a dense-MFMA
    // loop on non-trivial data runs well below the 2.4 GHz the 2.5 PF peak is quoted at (profiles/r6/experiments.md section 2).
    unsigned long long* clk;
};

// fills the derived fields (tps, split ranges) and launches tile configuration cfg (kCfgBf16) on stream st
hipError_t launch_gemm_bf16(Bf16GemmP p, bool btr, int cfg, int nsplit, int role, hipStream_t st);
static const int NCFG_BF16 = 16;   // {BM, BN, workgroups per CU}: see launch_cfg_bf16
static const int kCfgBf16[NCFG_BF16][3] = {{128, 128, 2}, {256, 128, 1}, {256, 256, 1}, {256, 256, 1}, {256, 256, 1},
                                           {320, 256, 1}, {192, 256, 1}, {256, 128, 1}, {192, 256, 1}, {256, 128, 1},
                                           {128, 256, 1}, {256, 128, 1}, {192, 256, 1},
                                           {256, 128, 1}, {128, 256, 1}, {256, 192, 1}};
// Slot 12 is RETIRED (the last of round 2's stream-K tiles; slots 10 and 11 were reused in round 6 for two three-stage ping-pong tiles - tile
// tables are tied to a build by mcrn_build_id now, so an old table cannot name them by mistake).  The tuner skips it, a table naming it is
// refused, forcing it (MCRN_BF16_CFG) fails the launch.
static const int CFG_BF16_SK0 = 12, CFG_BF16_SK1 = 13;
static inline bool bf16_cfg_is_sk(int c) { return c >= CFG_BF16_SK0 && c < CFG_BF16_SK1; }
// every live slot has a plain and a hi/lo (nterm == 3) form
static inline bool bf16_cfg_ok(int c, bool x3) { (void)x3; return c >= 0 && c < NCFG_BF16 && !bf16_cfg_is_sk(c); }
// ... and the slots the tuner times (duplicates skipped)
static inline bool bf16_cfg_tuned(int c, bool x3) { return bf16_cfg_ok(c, x3); }

}  // namespace mcrn
