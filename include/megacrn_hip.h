/*
 * megacrn_hip.h - C ABI of libmegacrn_hip.so (gfx950 / MI355X).
 *
 * The reference (deepkashiwa20/MegaCRN) has no native layer: its hot path is the
 * ATen ops invoked by model/MegaCRN.py:7-194.  This library replaces exactly
 * those ops with hand-written HIP kernels; each entry point below names the
 * reference lines it stands in for.  The reference-side binding is the ctypes
 * stub in INTEGRATION.md (and megacrn_amd/_lib.py).
 *
 * Conventions
 *  - All tensor pointers are DEVICE pointers to contiguous fp32 row-major
 *    arrays in the reference layout: x (B,T,N,C), state (B,N,H), support (N,N),
 *    AGCN weights (2*cheb_k*C, O) with row index k_global*C + c
 *    (model/MegaCRN.py:24-27).  int arguments named `teacher` are HOST arrays.
 *  - `stream` is a hipStream_t (passed as void*).  Kernels are enqueued on it;
 *    no entry point synchronises, allocates or frees device memory: all
 *    scratch and saved activations live in the caller-provided workspace `ws`
 *    of at least mcrn_*_workspace_bytes() bytes (256-byte aligned).
 *  - A forward call leaves its saved activations in `ws`; the matching
 *    backward call must receive the same `ws` untouched.
 *  - Return value: 0 on success, otherwise a hipError_t (>0) or
 *    MCRN_EINVAL (-1) for unsupported arguments; mcrn_last_error() returns a
 *    static message.
 *  - Process model: ONE host thread and ONE device per process (one process per GPU).  The library keeps a
 *    process-wide helper stream, tile cache and profiler; every compute entry point refuses (MCRN_EINVAL) a call
 *    that arrives while another host thread is inside the library, or on a different device than the first call.
 *  - Supported: cheb_k in 2 .. 8 (model/MegaCRN.py:21-22 recurses for any cheb_k >= 2; 2 and 3 have the fused fast
 *    paths, larger orders run the recursion step by step on the tiled GEMM; MCRN_BF16: 2 or 3); any B,N,H,
 *    input/output/ycov dims >= 1; num_layers == 1 in the fused model entry points (multi-layer models are
 *    composed from mcrn_cell_* by the host module).
 */
#ifndef MEGACRN_HIP_H
#define MEGACRN_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif
/* The library is built with -fvisibility=hidden: the declarations of this header are its whole dynamic interface. */
#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility push(default)
#endif

#define MCRN_EINVAL (-1)

/* compute precision of the MFMA contractions */
#define MCRN_F32 0     /* exact fp32 MFMA (v_mfma_f32_32x32x2_f32), 157 TF peak */
#define MCRN_BF16X3 1  /* fp32 operands split into bf16 hi+lo, 3x v_mfma_f32_32x32x16_bf16 with fp32
                          accumulate: ~1e-5 relative error, 5.3x the MFMA rate of MCRN_F32 (default) */
#define MCRN_BF16 2    /* fused model entry points only: the K-hop propagation, its transpose and the adjacency
                          gradient run on bf16-RESIDENT operands (one bf16 MFMA per product, fp32 accumulate; the
                          Chebyshev terms as the reference's own matrices [S, 2SS-I], model/MegaCRN.py:20-22).  With
                          rnn_units in {32, 64, 128} (H + mem_dim for the decoder likewise) the per-step propagation
                          covers the state channels only (the input channels of every step are propagated once per
                          stack) and writes bf16-resident planes that a streaming weight-pool kernel consumes; every
                          other contraction as MCRN_BF16X3.  Stated tolerance 1e-2 (tests: measured 5e-3); the large-graph mode. */

typedef struct mcrn_dims {
    int B;          /* batch */
    int N;          /* num_nodes */
    int T_in;       /* encoder sequence length */
    int T_out;      /* horizon */
    int input_dim;
    int output_dim;
    int ycov_dim;
    int H;          /* rnn_units */
    int mem_num;    /* M */
    int mem_dim;    /* D */
    int cheb_k;     /* 2 .. 8 (MCRN_BF16: 2 or 3) */
    int precision;  /* MCRN_F32, MCRN_BF16X3 or MCRN_BF16 */
} mcrn_dims_t;

/* parameter pointers, named after the reference state_dict keys (num_layers=1) */
typedef struct mcrn_params {
    const float *Memory, *Wq, *We1, *We2;                /* memory.*            MegaCRN.py:149-157 */
    const float *enc_gate_w, *enc_gate_b;                /* encoder.dcrnn_cells.0.gate.{weights,bias} */
    const float *enc_update_w, *enc_update_b;
    const float *dec_gate_w, *dec_gate_b;                /* decoder.dcrnn_cells.0.* */
    const float *dec_update_w, *dec_update_b;
    const float *proj_w, *proj_b;                        /* proj.0.{weight,bias}  MegaCRN.py:144 */
} mcrn_params_t;

typedef struct mcrn_grads {
    float *Memory, *Wq, *We1, *We2;
    float *enc_gate_w, *enc_gate_b, *enc_update_w, *enc_update_b;
    float *dec_gate_w, *dec_gate_b, *dec_update_w, *dec_update_b;
    float *proj_w, *proj_b;
} mcrn_grads_t;

const char* mcrn_last_error(void);
int mcrn_version(void);
/* 16 hex digits: sha256 over every source file the library was built from (csrc/Makefile).  Measured artefacts - PMC traffic tables,
 * GEMM tile tables - record it; a consumer ignores an artefact that was measured on another build. */
const char* mcrn_build_id(void);
/* Debug / test: which kernel FAMILY each planned launch site of the library chose since the last reset, as "family=count\n" lines
 * (NUL-terminated).  Returns the bytes needed incl. the NUL; writes nothing when cap is smaller.  reset != 0 clears the counters.
 * The plan picks kernels by shape; the GPU suite uses this to assert that a case runs the kernel it was written for. */
long long mcrn_launch_histogram(char* buf, long long cap, int reset);

/* arithmetic used by the stand-alone op entry points (the model entry points take it from
 * mcrn_dims_t.precision).  Returns 0 or MCRN_EINVAL. */
int mcrn_set_precision(int precision);
int mcrn_get_precision(void);
/* adjacency-gradient GEMMs run on an internal low-priority helper stream, forked from and joined back
 * to `stream` with events inside each call (default on); 0 keeps every kernel on `stream`. */
int mcrn_set_side_stream(int enable);

/* ---- whole model: MegaCRN.forward, model/MegaCRN.py:168-194, and its autograd backward ---- */
size_t mcrn_model_workspace_bytes(const mcrn_dims_t* d);

/* x (B,T_in,N,input_dim), ycov (B,T_out,N,ycov_dim), labels (B,T_out,N,output_dim) or NULL.
 * teacher[t] != 0  <=>  `go = labels[:, t]` after step t (the curriculum branch, :188-191);
 * NULL means all zero.  Outputs: output (B,T_out,N,output_dim), h_att/query/pos/neg (B,N,mem_dim). */
int mcrn_model_forward(const mcrn_dims_t* d, const mcrn_params_t* p,
                       const float* x, const float* ycov, const float* labels, const int* teacher,
                       void* ws, size_t ws_bytes,
                       float* output, float* h_att, float* query, float* pos, float* neg,
                       void* stream);

/* Gradients of every parameter given d(output), and optionally d(h_att), d(query), d(pos), d(neg)
 * (NULL = zero).  `grads` tensors are overwritten (not accumulated). */
int mcrn_model_backward(const mcrn_dims_t* d, const mcrn_params_t* p, const int* teacher,
                        const float* d_output, const float* d_hatt, const float* d_query,
                        const float* d_pos, const float* d_neg,
                        void* ws, size_t ws_bytes, const mcrn_grads_t* grads, void* stream);

/* ---- adaptive adjacency: model/MegaCRN.py:169-172 ---- */
size_t mcrn_supports_workspace_bytes(int N, int M, int D);
/* g1 = softmax(relu(E1 E2^T)), g2 = softmax(relu(E2 E1^T)); g1,g2 are (N,N) contiguous */
int mcrn_supports_forward(int N, int M, int D, const float* We1, const float* We2, const float* Mem,
                          void* ws, size_t ws_bytes, float* g1, float* g2, void* stream);
int mcrn_supports_backward(int N, int M, int D, const float* We1, const float* We2, const float* Mem,
                           const float* dg1, const float* dg2, void* ws, size_t ws_bytes,
                           float* dWe1, float* dWe2, float* dMem, void* stream);

/* ---- AGCN.forward: model/MegaCRN.py:16-28 ---- */
size_t mcrn_agcn_workspace_bytes(int B, int N, int C, int O, int cheb_k);
/* x (B,N,C), s1,s2 (N,N), W (2*cheb_k*C, O), b (O) -> y (B,N,O) */
int mcrn_agcn_forward(int B, int N, int C, int O, int cheb_k, const float* x, const float* s1,
                      const float* s2, const float* W, const float* b, void* ws, size_t ws_bytes,
                      float* y, void* stream);
int mcrn_agcn_backward(int B, int N, int C, int O, int cheb_k, const float* dy, const float* s1,
                       const float* s2, const float* W, void* ws, size_t ws_bytes, float* dx,
                       float* ds1, float* ds2, float* dW, float* db, void* stream);

/* ---- AGCRNCell.forward: model/MegaCRN.py:38-48 ---- */
size_t mcrn_cell_workspace_bytes(int B, int N, int din, int H, int cheb_k);
/* x (B,N,din), h (B,N,H) -> hn (B,N,H).  gate_w (2*cheb_k*(din+H), 2H), update_w (.., H) */
int mcrn_cell_forward(int B, int N, int din, int H, int cheb_k, const float* x, const float* h,
                      const float* s1, const float* s2, const float* gate_w, const float* gate_b,
                      const float* update_w, const float* update_b, void* ws, size_t ws_bytes,
                      float* hn, void* stream);
int mcrn_cell_backward(int B, int N, int din, int H, int cheb_k, const float* dhn, const float* s1,
                       const float* s2, const float* gate_w, const float* update_w, void* ws,
                       size_t ws_bytes, float* dx, float* dh, float* ds1, float* ds2,
                       float* dgate_w, float* dgate_b, float* dupdate_w, float* dupdate_b,
                       void* stream);

/* ---- MegaCRN.query_memory: model/MegaCRN.py:159-166 ---- */
size_t mcrn_memory_workspace_bytes(int B, int N, int H, int M, int D);
/* h (B,N,H) -> value, query, pos, neg (B,N,D); ind (B,N,2) int32 top-2 indices */
int mcrn_memory_forward(int B, int N, int H, int M, int D, const float* h, const float* Mem,
                        const float* Wq, void* ws, size_t ws_bytes, float* value, float* query,
                        float* pos, float* neg, int* ind, void* stream);
int mcrn_memory_backward(int B, int N, int H, int M, int D, const float* h, const float* Mem,
                         const float* Wq, const float* dvalue, const float* dquery,
                         const float* dpos, const float* dneg, void* ws, size_t ws_bytes,
                         float* dh, float* dMem, float* dWq, void* stream);

/* ---- trainer tail: clip_grad_norm_ + Adam on one flat fp32 buffer
 *      (model/traintest_MegaCRN.py:104,129-130) ---- */
/* p,g,m,v: flat arrays of n floats.  scratch: >= 1024 floats.  step is the 1-based Adam step.
 * grad_scale multiplies g first (1/world_size after an all-reduce(sum)).  total_norm_out (device,
 * 1 float, may be NULL) receives the pre-clip norm of grad_scale*g. */
int mcrn_flat_clip_adam(float* p, float* g, float* m, float* v, long long n, float lr, float beta1,
                        float beta2, float eps, int step, float max_norm, float grad_scale,
                        float* scratch, float* total_norm_out, void* stream);

/* ---- trainer loss, forward + backward (model/traintest_MegaCRN.py:118-125, model/utils.py:126-133) ----
 * loss = masked_mae(output*std+mean, labels*std+mean) + lamb*TripletMarginLoss(margin)(query,pos,neg)
 *        + lamb1*MSELoss(query,pos), pos/neg treated as constants (the trainer detaches them).
 * losses (device, 4 floats) = {total, masked_mae, triplet, mse}; d_output, d_query = gradients of `total`.
 * scratch: >= 4104 floats. */
int mcrn_loss_fwd_bwd(int B, int T, int N, int output_dim, int D, const float* output, const float* labels,
                      const float* query, const float* pos, const float* neg, float mean, float std, float lamb,
                      float lamb1, float margin, float* scratch, float* losses, float* d_output, float* d_query,
                      void* stream);

/* ---- evaluation metrics of the trainer, one launch per batch, accumulated on device
 *      (model/traintest_MegaCRN.py:63-93 with model/utils.py:126-160) ----
 * Adds this batch's values to acc (device, >= 17 floats, zero before the first batch of an evaluation):
 *   acc[0] += masked_mae + lamb*triplet + lamb1*mse (the evaluation loss, :66-72);
 *   acc[1+3s+{0,1,2}] += masked {MAE, MAPE, MSE} of slice s: s = 0 the whole batch (:75-77), s = 1..nh the
 *   single-step slices y[:, horizons[s-1]-1] (:79-88; horizons are 1-based step numbers, nh <= 3);
 *   acc[13] += 1 (batches); acc[14..16] = this batch's {masked_mae, triplet, mse}.
 * The epoch figures are acc[k]/acc[13] (RMSE = sqrt of the mean MSE, :89-93).  No host synchronisation.
 * scratch: >= 64 + 18*1024 floats, zero-filled once by the caller before the first call (word 0 is an arrival
 * counter that the kernel re-arms itself). */
int mcrn_eval_metrics(int B, int T, int N, int output_dim, int D, const float* output, const float* labels,
                      const float* query, const float* pos, const float* neg, float mean, float std, float lamb,
                      float lamb1, float margin, const int* horizons, int nh, float* scratch, float* acc,
                      void* stream);

/* ---- test hook: C = alpha*op(A)*op(B) + beta*C on the library's MFMA GEMM ---- */
/* A is (M,K) row-major if !transA else (K,M); B is (K,N) if !transB else (N,K); C (M,N). */
int mcrn_gemm_f32(int M, int N, int K, int transA, int transB, const float* A, const float* B,
                  float* C, float alpha, float beta, int nsplit, float* slabs, void* stream);

/* kernel launches issued since the last mcrn_model_forward began (bench bookkeeping) */
int mcrn_last_launch_count(void);

/* ---- live kernel timing for bench.py's roofline leg ----
 * Between mcrn_prof_begin(role) and mcrn_prof_end(), every launch of the GEMM role `role` is
 * bracketed by hipEventRecord on the stream it is launched on.  Roles (= last template argument
 * of mcrn::gemm_f32_kernel in rocprofv3 output): 1 propagation S x Z (model/MegaCRN.py:25),
 * 2 weight pool + GRU epilogue (:27,:43-47), 3 d-grad, 4 S^T propagation (backward),
 * 5 adjacency gradient, 6 weight gradient, 0 everything else; 7 (timing only, bf16 mode): the
 * once-per-stack products that propagate the input channels of every step (they run the role-1
 * kernels on a narrow operand and are timed apart from the per-step propagation).
 * mcrn_prof_end synchronises on the recorded events and returns the summed kernel time, the number
 * of launches, their algorithmic flops (true channel counts) and executed flops. */
int mcrn_prof_begin(int role);
/* One-off tile autotuning for a model shape: runs a scratch forward+backward in which every distinct
 * GEMM signature is timed once per tile configuration (HIP events) and the winner cached for all
 * later calls in this process.  The only entry point that synchronises and allocates (temporaries,
 * freed before return); call it outside the training loop.  `ws` is clobbered. */
int mcrn_model_autotune(const mcrn_dims_t* d, void* ws, size_t ws_bytes, void* stream);
int mcrn_autotune_entries(void);
int mcrn_autotune_clear(void);
/* The tile table as a flat int32 record list ({kind, nkey, key words..., cfg} per entry).  export returns the number of
 * words the table needs and fills `buf` when `cap` is large enough (call with NULL/0 first); import validates the whole list, then MERGES it into the table
 * (entries of other shapes stay; bench.py imports committed tables from profiles/tiles/ that way).
 * Data-parallel ranks tune independently and timing noise may choose different tiles (different fp32 summation
 * orders): rank 0 exports, every other rank imports, so all replicas run identical kernels (megacrn_amd/trainer.py). */
long long mcrn_autotune_export(int* buf, long long cap);
int mcrn_autotune_import(const int* buf, long long n);
/* tuning hook: force GEMM tile configuration 0..6 for every launch (-1 = automatic) */
int mcrn_set_gemm_cfg(int cfg);
/* ablation bits for the GEMM main loop (results are WRONG when non-zero; tools/ablate.py only) */
int mcrn_set_debug(int bits);
int mcrn_prof_end(double* total_ms, long long* launches, double* alg_flops, double* exec_flops);
/* Shader clock (MHz) the chip held inside the K loops of the bf16-resident products (gemm_bf16.h) profiled by the last
 * mcrn_prof_begin / mcrn_prof_end pair: workgroup 0 of each such launch stamps the shader cycle counter and the 100 MHz wall clock at both
 * ends of its K loop; *shader_mhz = sum of cycles / sum of wall time, *launches = stamped launches (0 and 0.0 when the role has no such
 * product - the small-graph kernels).  MI355X clocks to its power budget: dense-MFMA loops run well below the 2.4 GHz at which the
 * 2.5 PF bf16 peak is quoted, and bench.py reports the fraction of peak at the clock that was actually held beside the fraction of 2.5 PF. */
int mcrn_prof_clock_mhz(double* shader_mhz, long long* launches);

#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* MEGACRN_HIP_H */
