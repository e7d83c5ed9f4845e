/*
 * sot_hip.h -- C ABI of libsot_hip.so, the MI355X (gfx950) implementation of the
 * 1-D spectral optimal-transport loss.
 *
 * This is the drop-in boundary for the reference's hot path.  The reference is pure
 * Python on PyTorch (no FFI of its own), so each entry point replaces a *composition of
 * ATen ops* in the reference; the Python binding a maintainer adds is shown in
 * INTEGRATION.md (ctypes; the shipped host-side mirror is
 * 1d-spectral-optimal-transport_amd/losses.py).
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer unless stated;
 *   - the caller owns and allocates everything (outputs and workspace); functions only
 *     enqueue work on `stream` (a hipStream_t passed as void*; NULL = default stream):
 *     no allocation, no host synchronisation, graph-capturable;
 *   - return value: SOT_OK (0) or a negative sot_status; nothing aborts, nothing is
 *     enqueued when an error is returned;
 *   - rows are independent: a caller shards a batch by passing a sub-range of rows.
 */
#ifndef SOT_HIP_H
#define SOT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SOT_ABI_VERSION 13   /* bumped on every change of a signature or of a buffer contract below; the binding checks it
                              * (11, round 6: sot_workspace_bytes covers the per-row pre-sort; the MSS workspace is 16-byte aligned) */

typedef enum sot_status {
    SOT_OK = 0,
    SOT_ERR_INVALID_P = -1,        /* p < 1          -> AssertionError in the reference (losses.py:271) */
    SOT_ERR_BAD_SHAPE = -2,        /* B < 0, n < 1, m < 1, bad strides                                 */
    SOT_ERR_UNSUPPORTED_SIZE = -3, /* a row's working set exceeds one CU's 160 KiB LDS (n + m > ~19000) */
    SOT_ERR_NULL_POINTER = -4,
    SOT_ERR_WORKSPACE = -5,        /* workspace smaller than sot_workspace_bytes()                     */
    SOT_ERR_LAUNCH = -6            /* hipGetLastError() != hipSuccess after the launch                 */
} sot_status;

/* flag bits: the keyword arguments of Wasserstein1D (losses.py:90-121, 129-196) */
#define SOT_FLAG_SQUARE 1u         /* square_dist: weights are squared first       (losses.py:172-174) */
#define SOT_FLAG_DONT_NORMALIZE 2u /* y is divided by the mass of x, not its own   (losses.py:180-184) */
#define SOT_FLAG_LIMIT_Q 4u        /* levels Q_k > 1 contribute nothing            (losses.py:306-307) */
#define SOT_FLAG_REQUIRE_SORT 8u   /* sort supports, permute weights               (losses.py:286-290) */
#define SOT_FLAG_PRENORMALIZED 16u /* weights are used as given: the functional wasserstein_1d(u_values,
                                      v_values, u_weights, v_weights) of losses.py:223, which does not
                                      normalise (flags SQUARE / DONT_NORMALIZE are then ignored)       */
#define SOT_FLAG_NO_SPECIALIZE 32u /* diagnostic: always run the generic kernels, never the variants with the row
                                      length at compile time (257/512/513/1025/2048/8192 bins).  Results are
                                      bit-identical for 512/2048/8192 and agree to the last bits otherwise;
                                      the tests compare the two                                              */

#define SOT_FLAG_SAME_GRID 64u     /* the caller guarantees that xpos and ypos hold the SAME values (n == m, equal element by
                                      element after sorting; shared positions only): what every reference call site passes
                                      (y_pos = x_pos.clone(), trainer.py:191; the fixed_x buffer).  For p == 1 without
                                      SOT_FLAG_LIMIT_Q the forward then evaluates the area between the two CDFs on that grid,
                                      sum_i |U_i - V_i| (pos_{i+1} - pos_i) -- the same W_1, no merge (<= 3e-7 from the merge
                                      kernel's value).  A wrong guarantee gives wrong results                                */
#define SOT_FLAG_NO_AREA 128u      /* diagnostic: ignore SOT_FLAG_SAME_GRID (always the merge kernel)                         */
#define SOT_FLAG_TIE_FREE_GRADIENT 256u /* OPT-IN: the caller accepts, at exactly tied float32 CDF levels, the derivative of the loss
                                      as a function of the CDF values (what float64 autograd of the reference returns) instead of the
                                      tie-order artefact of the reference's float32 autograd (losses.py:295-298: ranks are constants
                                      to autograd, so a run of equal levels hands its whole gradient to one member).  With it, p == 1
                                      with SOT_FLAG_SAME_GRID and no cutoff gets a MERGE-FREE training form (gradient w.r.t. y alone):
                                      8192 x 2048 rows 72.7 -> 47.5 us, 16384 x 1025 rows 76 -> 69 us.  Gradient entries of rows without tied
                                      levels are unchanged; the loss value is the merge-free forward kernel's (<= 3e-7 from the merge walk's) */

/* One batch of spectrum pairs.  Mirrors the arguments of Wasserstein1D.forward
 * (losses.py:129) after its [batch,time,N] -> [B,N] reshape (losses.py:157-170). */
typedef struct sot_problem {
    const float *x;           /* [B, n] weights of the first measure (magnitude spectrum)             */
    const float *y;           /* [B, m] weights of the second measure                                 */
    const float *xpos;        /* support positions of x: [n] (row stride 0) or [B, n]                 */
    const float *ypos;        /* support positions of y: [m] (row stride 0) or [B, m]                 */
    int64_t B;                /* rows (spectrum pairs) in this call                                   */
    int32_t n, m;             /* support sizes (row lengths)                                          */
    int64_t x_row_stride;     /* elements between consecutive rows of x (>= n)                        */
    int64_t y_row_stride;     /* elements between consecutive rows of y (>= m)                        */
    int64_t xpos_row_stride;  /* 0 = one position row shared by all rows (the stride-0 expand of      */
    int64_t ypos_row_stride;  /*     losses.py:167-170), otherwise >= n / >= m                        */
    float p;                  /* order of the distance, p >= 1; result is W_p^p (no root)             */
    uint32_t flags;           /* SOT_FLAG_*                                                           */
    /* Optional position plan from sot_prepare_positions() (shared positions only).  When
     * perm_is_identity != NULL, xpos/ypos must be the SORTED positions that call produced and
     * SOT_FLAG_REQUIRE_SORT costs nothing extra (no internal sort, no workspace). NULL otherwise. */
    const int32_t *xperm;            /* [n] sort permutation of the original xpos             */
    const int32_t *yperm;            /* [m]                                                    */
    const int32_t *perm_is_identity; /* [2] device flags: 1 = positions were already sorted    */
    /* Round 5, per-row positions with SOT_FLAG_REQUIRE_SORT (the `torch.sort(u_values, 1)` of losses.py:286-288 on every row): the sort runs
     * ONCE per training step.  row_perm_out (or NULL): a call that sorts stores each row's two sort permutations here, [B, n + m] uint16 (the n
     * original columns of x's sorted supports, then the m of y's; n, m <= 16384).  row_perm_in (or NULL): permutations an earlier call produced
     * for the SAME position tensors -- sot_w1d_backward / sot_w1d_position_grad (and the forward itself) then gather the sorted supports
     * through them instead of sorting again.  Ignored for shared positions.
     * Round 6: rows of 2 ... 2048 positions are sorted AHEAD of the row kernel by a kernel of their own (one wavefront per row, a packed-word
     * network in registers: csrc/sot_wave_sort.hpp) whenever there is a place for the permutations -- row_perm_out, else the call's workspace
     * when it holds sot_workspace_bytes() bytes, else the row kernel sorts in LDS as before.  Either way an image that a call has written holds
     * the stable sort permutations of every row. */
    uint16_t *row_perm_out;
    const uint16_t *row_perm_in;
} sot_problem;

int sot_abi_version(void);
const char *sot_status_string(int status);

/* Bytes of device workspace the calls below need for this problem (host-side arithmetic only).  Shared positions with
 * SOT_FLAG_REQUIRE_SORT and no plan: required.  Per-row positions with SOT_FLAG_REQUIRE_SORT and neither row_perm_in nor row_perm_out
 * (round 6): room for the pre-sort's [B, n + m] uint16 permutations -- OPTIONAL: a call with a smaller (or no) workspace sorts inside the
 * row kernel.  0 otherwise. */
size_t sot_workspace_bytes(const sot_problem *prob);

/*
 * Position plan for row-invariant supports (the 1-D x_pos/y_pos of trainer.py:192-197 and the
 * fixed_x buffer of losses.py:124-127): checks sortedness and, if needed, sorts (position, index)
 * pairs once (torch.sort of losses.py:287-288 on a row that is the same for every batch row).
 * Outputs: sorted positions, int32 permutations, and two device flags.  A training loop that keeps
 * its position grid calls this once and passes the plan in sot_problem.
 */
int sot_prepare_positions(const float *xpos, const float *ypos, int32_t n, int32_t m,
                          float *xpos_sorted /* [n] */, float *ypos_sorted /* [m] */,
                          int32_t *xperm /* [n] */, int32_t *yperm /* [m] */,
                          int32_t *perm_is_identity /* [2] */, void *stream);
/* The same plan for xpos / max(xpos) and ypos / max(ypos) (round 6, ABI 12): the reference's trainer rebuilds its grid on every step --
 * `x_pos = x_pos / x_pos.max(); y_pos = x_pos.clone()` (trainer.py:196-197) -- and this call takes the bin frequencies as they are, so
 * that such a step pays one launch instead of a reduction, a division, a copy and the plan.  The sorted positions written are the
 * floats torch computes: torch.max's NaN rule, IEEE division. */
int sot_prepare_unit_positions(const float *xpos, const float *ypos, int32_t n, int32_t m,
                               float *xpos_sorted /* [n] */, float *ypos_sorted /* [m] */,
                               int32_t *xperm /* [n] */, int32_t *yperm /* [m] */,
                               int32_t *perm_is_identity /* [2] */, void *stream);

/*
 * Forward: row_loss[r] = W_p^p(x_r, y_r) for r in [0, B).
 * Replaces Wasserstein1D.forward steps 3-5 + wasserstein_1d + quantile_function
 * (losses.py:172-196, 214-220, 271-313): square, row mass (ATen summation order),
 * safe_divide (utils.py:135-142), position sort + weight gather, fp64-accumulated CDFs,
 * merge of the two CDFs (= sort(cat(U,V)) + 2x searchsorted + 2x take_along_dim), level
 * widths, cutoff mask, |q_x - q_y|^p, weighted row sum.
 */
int sot_w1d_forward(const sot_problem *prob, float *row_loss /* [B] */,
                    void *workspace, size_t workspace_bytes, void *stream);

/*
 * Batch reduction of losses.py:211 (`torch.mean(loss)`, dims=None) with a fixed-order fp64
 * accumulation: *mean_out = (float)(sum_r row_loss[r] / denom) and, if sum_out != NULL,
 * *sum_out = sum_r row_loss[r] (the per-shard partial that is all-reduced across GPUs).
 * hinge_threshold: rows are replaced by relu(row - threshold) first when apply_hinge != 0
 * (losses.py:203-205).
 */
int sot_w1d_reduce_mean(const float *row_loss, int64_t B, double denom, int apply_hinge, float hinge_threshold,
                        float *mean_out, double *sum_out, void *stream);

/*
 * Backward of the forward above w.r.t. the weights (closed form of the autograd graph of
 * losses.py:172-313; positions receive no gradient, as in every reference call site).
 * dL/d(row_loss[r]) = grad_scale * grad_row[r * grad_row_stride] (grad_row == NULL: 1 for every row).  grad_x and/or grad_y may be NULL (trainer.py only needs
 * grad_y: x is the target spectrum).  Row strides of the gradients equal n and m.
 * Tie convention: gradients of a run of equal quantile levels go to the run's last member in
 * stable-sort order (U before V, lower index first).
 */
int sot_w1d_backward(const sot_problem *prob,
                     const float *grad_row /* [B] if grad_row_stride == 1; ONE scalar if grad_row_stride == 0 */,
                     int64_t grad_row_stride, float grad_scale /* multiplies grad_row, e.g. 1/B of the batch mean */,
                     float *grad_x /* [B,n] or NULL */, float *grad_y /* [B,m] or NULL */,
                     void *workspace, size_t workspace_bytes, void *stream);

/*
 * Training form: row losses, their batch mean AND the gradient of `grad_scale * sum_r row_loss[r]` w.r.t. y in one call
 * (grad_scale = 1/denom: the gradient of the mean that Wasserstein1D.forward returns; trainer.py:220-228 backpropagates
 * exactly that, x being the target spectrum).  For the row lengths with a compile-time backward kernel this is ONE pass
 * over the rows (the backward kernel's merge walk also accumulates the loss, bit-identical to sot_w1d_forward) followed by
 * the reduction kernel; otherwise forward, backward and reduction are enqueued back to back.  No hinge (the hinge changes
 * which rows receive a gradient).  A caller whose upstream gradient turns out not to be 1 rescales with sot_scale_inplace.
 * completion_counters: see sot_w1d_loss.
 */
int sot_w1d_loss_and_grad(const sot_problem *prob, float *row_loss /* [B] */, double denom, float *mean_out, double *sum_out,
                          float grad_scale, float *grad_y /* [B,m] */, uint32_t *completion_counters /* or NULL */,
                          void *workspace, size_t workspace_bytes, void *stream);

/*
 * Kernel-attached timing for benchmarks: arm slot `slot` (0 .. 63) for the calling host thread; the NEXT launch of a kernel with
 * the row length at compile time (sot_w1d_forward / _loss / _backward / _loss_and_grad on such rows) is then issued with a
 * start / stop event pair attached to the dispatch itself (hipExtLaunchKernelGGL), so sot_profile_elapsed_ms() -- which waits
 * for that launch -- returns the kernel's own duration, the figure rocprofv3's kernel trace reports, free of the stream time
 * an event pair recorded around the call adds.  Not graph-capturable; no effect on results.
 */
int sot_profile_next_launch(int slot);
int sot_profile_elapsed_ms(int slot, float *ms);

/* data[i] *= *scalar for i < count (device scalar); returns without touching `data` when the scalar is exactly 1. */
int sot_scale_inplace(float *data, int64_t count, const float *scalar, void *stream);

/*
 * Forward + batch reduction in ONE call: the whole of Wasserstein1D.forward with dims=None (losses.py:129-211) behind a
 * single FFI crossing.
 *   completion_counters == NULL: sot_w1d_forward and sot_w1d_reduce_mean are enqueued back to back (two kernels).
 *   completion_counters != NULL: ONE kernel -- the workgroup of the forward kernel that finishes last reduces the row
 *     losses, with the summation order of sot_w1d_reduce_mean (bit-identical result, independent of timing).
 *     `completion_counters` points to SOT_COMPLETION_COUNTER_WORDS device words that the caller zero-fills ONCE; every
 *     launch leaves them zero again, so the same buffer serves all later calls -- as long as at most one launch that uses
 *     it is in flight at a time (calls on one stream are; give each concurrently used stream its own buffer).
 *     Measured on MI355X (8192 x 2048 rows): 55.1 us as one kernel, 46.9 us as two -- every workgroup pays a counter round
 *     trip at its end and the last one an agent-scope acquire plus a re-read of the row losses, which together cost more
 *     than the ~2.6 us kernel boundary + mean kernel they replace; the shipped Python binding therefore passes NULL unless
 *     asked.
 */
#define SOT_COMPLETION_COUNTER_WORDS 16
int sot_w1d_loss(const sot_problem *prob, float *row_loss /* [B] */, double denom, int apply_hinge,
                 float hinge_threshold, float *mean_out, double *sum_out, uint32_t *completion_counters /* or NULL */,
                 void *workspace, size_t workspace_bytes, void *stream);

/*
 * Forward for RAGGED supports in CSR form (BASELINE config 4: spectra after a per-row amplitude cutoff, where
 * the reference would be fed zero-masked dense rows; zero-weight points are inert, so both forms agree).
 * Row r owns entries [offsets[r], offsets[r+1]) of the concatenated weights/positions arrays; every row needs
 * 1 <= length <= max_n (resp. max_m), otherwise its loss is NaN.  x_nnz / y_nnz are offsets[B].  Positions are
 * per row; with SOT_FLAG_REQUIRE_SORT a row is sorted in LDS only if it is not already sorted.  The LDS footprint
 * is that of (max_n, max_m).  Semantics per row are those of sot_w1d_forward.
 */
int sot_w1d_forward_csr(const float *x_weights, const float *x_positions, const int64_t *x_offsets /* [B+1] */, int64_t x_nnz,
                        const float *y_weights, const float *y_positions, const int64_t *y_offsets /* [B+1] */, int64_t y_nnz,
                        int64_t B, int32_t max_n, int32_t max_m, float p, uint32_t flags, float *row_loss /* [B] */,
                        void *stream);

/*
 * return_quantiles=True (losses.py:198-201, 299-300): the five tensors the reference returns,
 * uq/vq/Q: [B, n+m], U: [B, n], V: [B, m]; any output pointer may be NULL.
 */
int sot_w1d_quantiles(const sot_problem *prob, float *uq, float *vq, float *Q, float *U, float *V,
                      void *workspace, size_t workspace_bytes, void *stream);

/*
 * Gradient of sum_r grad_scale * grad_row[r] * row_loss[r] w.r.t. the SUPPORT POSITIONS (losses.py:287-298 and 214-220: the positions
 * enter through torch.sort and take_along_dim, both differentiable; no reference call site asks for it, autograd supplies it):
 *     d row_loss / d xs[i] = sum over the merged levels k whose searchsorted rank in U (clamped to n-1) is i of
 *                            delta_k * p |xs[i_k] - ys[j_k]|^(p-1) sign(xs[i_k] - ys[j_k]);    ys[j]: minus the same.
 * One kernel, deterministic (no atomics).  Outputs are PER ROW, [B, n] / [B, m] dense, in the caller's original column order (the
 * sort of losses.py:287-288 is undone); either may be NULL.  For a position row shared by all batch rows (xpos_row_stride == 0)
 * the caller sums the rows -- sot_column_sum below -- which is what autograd's `expand` backward does (losses.py:167-170).
 * grad_row / grad_row_stride / grad_scale as in sot_w1d_backward.
 */
int sot_w1d_position_grad(const sot_problem *prob, const float *grad_row, int64_t grad_row_stride, float grad_scale,
                          float *grad_xpos /* [B,n] or NULL */, float *grad_ypos /* [B,m] or NULL */,
                          void *workspace, size_t workspace_bytes, void *stream);

/* out[c] = sum_r rows[r * row_stride + c], c < n, in a fixed order with fp64 accumulation. */
int sot_column_sum(const float *rows, int64_t B, int32_t n, int64_t row_stride, float *out /* [n] */, void *stream);

/*
 * Segmented (per-row) stable ascending sort with index payload: what torch.sort(keys, 1)
 * returns at losses.py:287-288 (indices are int64 like torch's; bit-identical to torch on
 * distinct keys; ties keep the lower index first).  row_stride in elements; outputs are
 * dense [B, n].  Either output may be NULL.  NaN keys: no defined order (torch puts them last; here a positive NaN orders above the
 * internal +inf pads, a negative one first), but every returned index is a valid column (< n) -- as are the gather / store indices of
 * the per-row-position kernels, which use the same sort.
 */
int sot_segmented_sort(const float *keys, int64_t B, int32_t n, int64_t row_stride,
                       float *sorted_keys, int64_t *indices, void *stream);

/* ---- The producer in front of the loss: the magnitude STFT of the reference's features.TorchSTFT
 * (features.py:85-113 -> compute_mag / stft, features.py:191-237; end padding utils.pad_for_stft, utils.py:252-275):
 * frames = ceil(samples / hop) frames of n_fft samples starting every `hop` samples (zeros past the end of the clip),
 * multiplied by `window` [n_fft], one-sided DFT, |.| / sqrt(n_fft) (torch.stft(center=False, normalized=True)).
 * audio [batch, samples] (row stride in elements), mag / grad_mag [batch, frames, n_fft/2 + 1] contiguous (frames-major:
 * each row is a spectrum for sot_w1d_*), grad_audio [batch, samples] contiguous.  n_fft: a power of two in [64, 4096].
 * Enqueue-only, no allocation: the backward takes a caller-owned scratch buffer of sot_stft_backward_workspace_bytes(). */
int64_t sot_stft_frames(int64_t samples, int hop);
int sot_stft_mag_forward(const float *audio, int64_t batch, int64_t samples, int64_t audio_row_stride,
                         const float *window, int n_fft, int hop, float *mag, void *stream);
/* the same transform of TWO signals of equal shape (target and estimate of a training step) in one launch:
 * mag[0 : batch_each] belongs to audio_a, mag[batch_each : 2 batch_each] to audio_b */
int sot_stft_mag_forward_pair(const float *audio_a, int64_t row_stride_a, const float *audio_b, int64_t row_stride_b,
                              int64_t batch_each, int64_t samples, const float *window, int n_fft, int hop,
                              float *mag /* [2 * batch_each, frames, n_fft/2+1] */, void *stream);
/* gradient of a scalar L w.r.t. the audio given dL/d(mag): closed form of abs o stft's autograd (bins with |X| = 0 pass
 * no gradient, as torch's sgn(0) = 0); deterministic (groups of frames overlap-added in a fixed order, no atomics).
 * grad_scale: optional DEVICE scalar that multiplies grad_mag (the upstream gradient of a loss whose dL/d(mag) was
 * computed ahead of the backward pass, sot_w1d_loss_and_grad), or NULL. */
/* Round 3: the forward can also hand out the COMPLEX spectrum X (what torch.abs's autograd node saves of torch.stft's output), `spec`:
 * [batch, frames, n_fft/2 + 1] interleaved (re, im) float pairs, 8-byte aligned, or NULL; in the pair form for the SECOND signal only
 * (`spec_b`: [batch_each, frames, n_fft/2 + 1, 2]; the estimate, the one a training step differentiates).  sot_stft_mag_backward_spec
 * then takes that spectrum instead of the audio (audio may be NULL) and skips the forward transform it would otherwise recompute per
 * frame: the same gradient bit for bit, n_fft 2048 on 4096 frames 46 -> ~30 us. */
int sot_stft_mag_forward_spec(const float *audio, int64_t batch, int64_t samples, int64_t audio_row_stride,
                              const float *window, int n_fft, int hop, float *mag, float *spec /* or NULL */, void *stream);
int sot_stft_mag_forward_pair_spec(const float *audio_a, int64_t row_stride_a, const float *audio_b, int64_t row_stride_b,
                                   int64_t batch_each, int64_t samples, const float *window, int n_fft, int hop,
                                   float *mag, float *spec_b /* or NULL */, void *stream);
int sot_stft_mag_backward_spec(const float *audio /* may be NULL when spec is given */, const float *spec /* or NULL */, int64_t batch,
                               int64_t samples, int64_t audio_row_stride, const float *window, int n_fft, int hop,
                               const float *grad_mag, const float *grad_scale, float *grad_audio, int accumulate,
                               void *workspace, size_t workspace_bytes, void *stream);
size_t sot_stft_backward_workspace_bytes(int64_t batch, int64_t samples, int n_fft, int hop);
int sot_stft_mag_backward(const float *audio, int64_t batch, int64_t samples, int64_t audio_row_stride,
                          const float *window, int n_fft, int hop, const float *grad_mag, const float *grad_scale,
                          float *grad_audio, int accumulate /* grad_audio += result instead of = */,
                          void *workspace, size_t workspace_bytes, void *stream);

/* ---- Additive oscillator bank in front of the STFT in the training step: ddsp.oscillator_bank
 * (ddsp.py:208-263 with use_angular_cumsum=False, sum_sinusoids=True) incl. remove_above_nyquist (ddsp.py:25-49):
 *   audio[b,t] = sum_k a'[b,t,k] sin(phase[b,t,k]),  a' = (f >= sample_rate/2) ? 0 : a,
 *   phase = cumsum_t((f * 2pi) / sample_rate)  (fp64 accumulation rounded to fp32 per sample, as ATen's CPU cumsum).
 * freq / amp / grad_freq / grad_amp: [batch, samples, sinusoids] contiguous; audio / grad_audio: [batch, samples].
 * samples <= 2^20, sinusoids <= 512 (else SOT_ERR_UNSUPPORTED_SIZE).  Deterministic.
 * `workspace`: device scratch of at least sot_oscillator_bank_workspace_bytes() bytes (per-segment phase carries, fp64;
 * the forward uses the first half only); 0 is returned for sizes the kernels do not take. */
size_t sot_oscillator_bank_workspace_bytes(int64_t batch, int64_t samples, int sinusoids);
int sot_oscillator_bank_forward(const float *freq, const float *amp, int64_t batch, int64_t samples, int sinusoids,
                                float sample_rate, float *audio, void *workspace, size_t workspace_bytes, void *stream);
/* gradients w.r.t. the envelopes (either may be NULL) given dL/d(audio) */
int sot_oscillator_bank_backward(const float *freq, const float *amp, int64_t batch, int64_t samples, int sinusoids,
                                 float sample_rate, const float *grad_audio, float *grad_freq, float *grad_amp,
                                 void *workspace, size_t workspace_bytes,
                                 int workspace_from_forward /* 1: `workspace` is the buffer sot_oscillator_bank_forward was
                                    given for the SAME envelopes and has not been written since: its segment start phases
                                    are reused instead of recomputed */,
                                 void *stream);

/* ---- Envelope upsampling of the synthesiser in front of the oscillator bank (synths.Sinusoidal.get_controls /
 * get_signal, synths.py:62-113 with amp_scale_fn = freq_scale_fn = None): frame-rate controls [batch, frames, sinusoids] ->
 * sample-rate envelopes [batch, samples, sinusoids] for sot_oscillator_bank_*.  harmonic != 0: freq_frames is [batch, frames, 1]
 * and partial k gets f0 * (k + 1) (ddsp.py:6-22).  Amplitudes of partials whose frame-rate frequency is >= sample_rate / 2 are
 * zeroed (ddsp.py:25-49), then upsampled with half-overlapping Hann windows (`window` = torch.hann_window(2 * samples / frames),
 * [2 hop] floats; ddsp.py:121-205, add_endpoint); frequencies are interpolated linearly (F.interpolate, align_corners=False, in
 * ATen's operation order): bit-identical to the reference's CPU result.  samples must be a multiple of frames, frames < samples.
 * The backward takes the gradients w.r.t. the two envelopes and returns those w.r.t. the controls (either output may be NULL;
 * grad_freq_frames is [batch, frames, 1] when harmonic).  Deterministic. */
int sot_synth_envelopes_forward(const float *amp_frames, const float *freq_frames, const float *window, int64_t batch, int frames,
                                int sinusoids, int harmonic, int64_t samples, float sample_rate, float *amp_env, float *freq_env,
                                void *stream);
int sot_synth_envelopes_backward(const float *amp_frames, const float *freq_frames, const float *window, int64_t batch, int frames,
                                 int sinusoids, int harmonic, int64_t samples, float sample_rate, const float *grad_amp_env,
                                 const float *grad_freq_env, float *grad_amp_frames, float *grad_freq_frames, void *stream);

/* ---- The synthesiser in one piece (synths.Sinusoidal.forward, synths.py:62-128): frame-rate controls -> audio [batch, samples]
 * without the sample-rate envelopes ever existing in memory: the oscillator-bank kernels evaluate them from the controls (the
 * same float32 operations as sot_synth_envelopes_forward, so the audio equals sot_oscillator_bank_forward on those envelopes bit
 * for bit).  The backward returns the gradients w.r.t. the controls (either may be NULL; grad_freq_frames [batch, frames, 1]
 * when harmonic) without sample-rate gradient arrays either: the tile kernel reduces its samples to per-frame partial sums (the
 * frequency gradient through cumulative tap weights, see csrc/sot_osc.hip), a last kernel adds the segments up.
 * workspace: sot_synth_workspace_bytes(batch, frames, samples, sinusoids, backward) bytes (0 for invalid sizes);
 * workspace_from_forward != 0: its first bytes still hold what sot_synth_forward left there for the SAME controls.  Deterministic. */
size_t sot_synth_workspace_bytes(int64_t batch, int frames, int64_t samples, int sinusoids, int backward);
int sot_synth_forward(const float *amp_frames, const float *freq_frames, const float *window, int64_t batch, int frames, int sinusoids,
                      int harmonic, int64_t samples, float sample_rate, float *audio, void *workspace, size_t workspace_bytes,
                      void *stream);
int sot_synth_backward(const float *amp_frames, const float *freq_frames, const float *window, int64_t batch, int frames, int sinusoids,
                       int harmonic, int64_t samples, float sample_rate, const float *grad_audio, float *grad_amp_frames,
                       float *grad_freq_frames, const void *tap_tables, void *workspace, size_t workspace_bytes,
                       int workspace_from_forward, void *stream);
/* The backward's weight tables depend on (window, frames, samples) only: a caller that keeps them across steps fills a buffer of
 * sot_synth_tap_table_bytes(frames, samples) bytes once with sot_synth_tap_tables and passes it as tap_tables; NULL: the backward
 * builds them in its workspace on every call (6 us for 16 frames x 4096 samples). */
size_t sot_synth_tap_table_bytes(int frames, int64_t samples);
int sot_synth_tap_tables(const float *window, int frames, int64_t samples, void *tables, void *stream);

/* ---- Spectral distance of the reference's MSSLoss (losses.py:365-425 with mean_difference
 * losses.py:7-36 and safe_log utils.py:145-151) over `count` magnitudes target[i], value[i]:
 *   out[0] = mag_weight * mean(D(t - v)) + logmag_weight * mean(D(slog t - slog v)),  D = |.| (l2 == 0) or (.)^2,
 *   slog(x) = log(x <= eps ? eps : x).  Deterministic (fixed-order fp64 partial sums in the caller-owned workspace). */
size_t sot_spec_distance_workspace_bytes(void);
int sot_spec_distance_forward(const float *target, const float *value, int64_t count, float mag_weight,
                              float logmag_weight, float eps, int l2, float *out,
                              int accumulate /* out[0] += the distance (MSSLoss sums its scales) instead of = */,
                              void *workspace, size_t workspace_bytes, void *stream);
/* gradients w.r.t. target and/or value (either may be NULL) given d(loss)/d(out) as a one-element device tensor times
 * grad_scale; |.|' (0) = 0 and no gradient through slog below eps, as torch */
int sot_spec_distance_backward(const float *target, const float *value, int64_t count, float mag_weight,
                               float logmag_weight, float eps, int l2, const float *upstream, float grad_scale,
                               float *grad_target, float *grad_value, void *stream);
/* The same distance PER ROW: out[r] for r < rows over the `count_per_row` consecutive magnitudes of row r -- MSSLoss called with
 * `dims` = the two spectrogram axes (one value per clip; losses.py:406-425 hands `dims` to mean_difference, losses.py:7-36).
 * One workgroup per row, fixed-order fp64 sums; upstream: [rows]. */
int sot_spec_distance_rows_forward(const float *target, const float *value, int64_t rows, int64_t count_per_row,
                                   float mag_weight, float logmag_weight, float eps, int l2, float *out /* [rows] */,
                                   int accumulate, void *stream);
int sot_spec_distance_rows_backward(const float *target, const float *value, int64_t rows, int64_t count_per_row,
                                    float mag_weight, float logmag_weight, float eps, int l2,
                                    const float *upstream /* [rows] */, float grad_scale,
                                    float *grad_target, float *grad_value, void *stream);

/* ---- Round 5: the WHOLE multi-scale spectrogram loss and its gradient in two launches (the reference's `MSSLoss.forward`,
 * losses.py:406-425, + autograd; per scale: compute_mag features.py:191-237 -- hann or any caller-given window, hop = n_fft / 4,
 * end padding utils.py:252-275, normalized -- and mean_difference losses.py:7-36 / safe_log utils.py:145-151):
 *   loss = sum_s [mag_weight * mean D(T_s - V_s) + logmag_weight * mean D(slog T_s - slog V_s)]      (per_clip == 0: one float;
 *   per_clip != 0: the means run over each clip's own spectrogram, `dims` = its two axes: loss[batch]),
 *   grad_value[b, t] = d loss (or d loss[b]) / d value[b, t]   (NULL: forward only; the caller applies its upstream gradient).
 * post_scale (ABI 13): the finished float32 loss value(s) and gradient entries are multiplied by it -- the `loss_fn(x, y) * weight` of
 *   MixOfLosses (losses.py:360) and its backward as ONE more float32 product each, i.e. the same floats as the two torch kernels; 1 = off.
 * target / value: [batch, samples] float32 with the given row strides (in floats); fft_sizes[n_scales]: powers of two in [64, 2048],
 * n_scales <= 8; windows[s]: n_fft taps of scale s, device pointers on 8-byte boundaries (host array of device pointers).
 * One workgroup per (scale, clip, 4096-sample chunk): 2048 / n_fft frames per wavefront through a register-resident FFT, magnitudes,
 * distance, gradient w.r.t. the spectrum, inverse transform and overlap-add without a spectrogram in memory; a second kernel adds the
 * scales per sample in a fixed order.  Deterministic, enqueue-only, graph-capturable.
 * workspace: sot_mss_workspace_bytes(batch, samples, fft_sizes, n_scales) bytes, 16-byte aligned (0 is returned for sizes it does not take). */
size_t sot_mss_workspace_bytes(int64_t batch, int64_t samples, const int *fft_sizes, int n_scales);
int sot_mss_loss_and_grad(const float *target, int64_t target_row_stride, const float *value, int64_t value_row_stride,
                          int64_t batch, int64_t samples, const int *fft_sizes, const float *const *windows, int n_scales,
                          float mag_weight, float logmag_weight, float eps, int l2, int per_clip, float post_scale,
                          float *loss /* [1] or [batch] */, float *grad_value /* [batch, samples] contiguous, or NULL */,
                          void *workspace, size_t workspace_bytes, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* SOT_HIP_H */
