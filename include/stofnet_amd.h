/*
 * stofnet_amd.h -- C ABI of the MI355X-native StofNet inference hot path.
 *
 * One shared library (libstofnet_amd.so, gfx950 code objects embedded) replaces
 * the ATen calls that the reference's hot path makes (the reference has no native
 * code of its own).  Every entry point cites the reference
 * interface it stands in for (paths relative to hahnec/stofnet).
 *
 * Conventions
 *   - plain pointers and sizes only; device pointers are raw HIP device
 *     addresses, `stream` is a hipStream_t passed as void*.
 *   - ownership: the caller owns every buffer including workspaces; the library
 *     allocates nothing per call and keeps no data between calls (its only state is a
 *     per-device "LDS limit already raised" bit per kernel, set with atomics), so one host
 *     thread per GPU -- or several GPUs from one process -- may call concurrently.
 *   - all device work is enqueued asynchronously on `stream`; no hidden syncs.
 *   - every function returns a stof_status (0 = ok); nothing throws or aborts
 *     across the ABI.  stof_status_string() gives a static message.
 *   - activations are fp32, contiguous, NCL as in the reference.
 */
#ifndef STOFNET_AMD_H
#define STOFNET_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define STOF_ABI_VERSION 3

typedef enum stof_status {
    STOF_OK = 0,
    STOF_ERR_BAD_ARG = 1,         /* null pointer, negative size, ...                          */
    STOF_ERR_ODD_SGB_REMAINDER = 2,/* L - 80*floor(L/80) is odd: the reference raises RuntimeError
                                      at models/stofnet.py:115                       */
    STOF_ERR_UNSUPPORTED = 3,     /* shape/mode outside what the kernels implement              */
    STOF_ERR_WORKSPACE = 4,       /* workspace or packed-weight buffer too small                */
    STOF_ERR_HIP = 5,             /* a HIP runtime call or launch failed                        */
    STOF_ERR_CHANNELS = 6,        /* channel count not divisible by r (view error in the reference,
                                      utils/sample_shuffle.py:24)                                */
    STOF_ERR_POOL_EMPTY = 7       /* SemiGlobalBlock on a row shorter than one pooling window (L < 80): the reference's
                                      MaxPool1d raises RuntimeError at models/stofnet.py:103       */
} stof_status;

const char* stof_status_string(int status);
int stof_abi_version(void);

/* ------------------------------------------------------------------------- *
 * Network description (models/stofnet.py:11 ctor arguments that the shipped
 * checkpoints use: num_features=64, num_blocks=13, kernel_sizes=[9,7,3],
 * in_channels=1).
 * ------------------------------------------------------------------------- */
typedef struct stof_net_desc {
    int32_t upsample_factor;     /* r: conv_last has r output channels (1..64)                  */
    int32_t semi_global_scale;   /* 80 = SemiGlobalBlock present, 1 = ablation without it       */
    int32_t precision;           /* STOF_PREC_*                                                 */
    int32_t seg_policy;           /* body sweep at small batches: 0 = automatic (waveforms are cut into 2^k segments
                                   * with +-38 rows of context while fewer than one per CU exists); k+1 forces 2^k   */
} stof_net_desc;

#define STOF_PREC_FP32 0          /* exact fp32 MFMA (v_mfma_f32_32x32x2_f32), parity baseline     */
#define STOF_PREC_F16X3 1         /* split-fp16 hi/lo operands, 3 MFMA passes, fp32 accumulate     */

/* Parameter order for stof_pack_weights(): the reference's state_dict tensors
 * (models/stofnet.py:23-31,88,94), each a host pointer to contiguous fp32:
 *   [0] conv1.weight (64,1,9)      [1] conv1.bias (64)
 *   [2+2i] conv{2+i}.weight (64,64,7), [3+2i] conv{2+i}.bias (64)   i = 0..10
 *   [24] conv_last.weight (r,64,3) [25] conv_last.bias (r)
 *   [26] semi_global_block.contract_conv.weight (512,64,5) [27] .bias (512)
 *   [28] semi_global_block.expand_conv.weight (64,512,5)   [29] .bias (64)
 * Entries 26..29 are ignored (may be NULL) when semi_global_scale == 1.      */
#define STOF_NUM_PARAMS 30

/* Bytes of the packed (kernel-layout) weight blob for `desc`. */
size_t stof_packed_weights_bytes(const stof_net_desc* desc);

/* One-time repack of the state_dict into the kernels' streaming layout.  Pure
 * host function (no GPU needed): writes `stof_packed_weights_bytes()` bytes to
 * `packed_host`; the caller uploads the blob to device memory once.
 * Replaces: nn.Module.load_state_dict + .to(device) (main.py:169,176-177).   */
int stof_pack_weights(const stof_net_desc* desc, const float* const* params,
                      void* packed_host, size_t packed_bytes);

/* Device workspace needed by stof_forward for a batch of N rows of length L. */
size_t stof_forward_workspace_bytes(const stof_net_desc* desc, int64_t N, int64_t L);

/* StofNet.forward (models/stofnet.py:42-67): x[N,1,L] -> y[N,1,L*r], both fp32
 * device buffers.  `packed_dev` is the uploaded blob from stof_pack_weights.
 * Returns STOF_ERR_ODD_SGB_REMAINDER for the lengths on which the reference's
 * SemiGlobalBlock raises (Q1).                                               */
int stof_forward(const stof_net_desc* desc, const void* packed_dev,
                 const float* x, float* y, int64_t N, int64_t L,
                 void* workspace, size_t workspace_bytes, void* stream);

/* Same as stof_forward, plus a range guard: bit 0 of *status_dev (a device int32 the caller has
 * zeroed) is set when the network produced a non-finite output.  In STOF_PREC_F16X3 that is what an
 * activation beyond the fp16 range (|a| > 65504; trained nets on max-abs-normalised inputs peak
 * near 50) turns into; the caller can then re-run in STOF_PREC_FP32.  No host sync.             */
int stof_forward_checked(const stof_net_desc* desc, const void* packed_dev,
                         const float* x, float* y, int64_t N, int64_t L,
                         void* workspace, size_t workspace_bytes, void* stream, int32_t* status_dev);

/* The default mode of the Python module (precision='auto'): STOF_PREC_F16X3 with the range guard of
 * stof_forward_checked, followed by an exact STOF_PREC_FP32 re-run of the same call whose kernels are gated ON THE
 * DEVICE by the guard word (they return at once while it is 0).  The result is the f16x3 map unless an activation left
 * the fp16 range, in which case it is the fp32 map -- no host sync either way.  `desc->precision` is ignored; the two
 * packed blobs come from stof_pack_weights with the respective precision.  *status_dev is zeroed by the call and reads
 * 1 afterwards iff the fp32 re-run happened.  `events` may be NULL or as in stof_forward_events (f16x3 pass only).   */
int stof_forward_auto(const stof_net_desc* desc, const void* packed_f16x3_dev, const void* packed_fp32_dev,
                      const float* x, float* y, int64_t N, int64_t L, void* workspace, size_t workspace_bytes,
                      void* stream, int32_t* status_dev, void* const* events);

/* StofNet.forward with get_maxima_positions in arg-max mode (utils/mask2samples.py:26-34, threshold = None) fused into
 * conv_last's epilogue (main.py:314 -> 320 for th = Null): per row the positions of the maximum of the NMS output, ties
 * and the constant-negative-row rule exactly as stof_pick_maxima reports them, WITHOUT the [N, 1, L*r] map going
 * through HBM -- `y` may be NULL (picker-only consumers: 4 bytes per onset instead of 4*L*r per waveform) or receive
 * the map as stof_forward would write it.  counts / idx / idx_cap as in stof_pick_maxima.  STOF_PREC_F16X3 and
 * upsample_factor <= 16 only (else STOF_ERR_UNSUPPORTED: use stof_forward + stof_pick_maxima); status_dev as in
 * stof_forward_checked (may be NULL).                                                                              */
size_t stof_forward_onsets_workspace_bytes(const stof_net_desc* desc, int64_t N, int64_t L);
int stof_forward_onsets(const stof_net_desc* desc, const void* packed_dev, const float* x, float* y, int64_t N,
                        int64_t L, int32_t window_size, int32_t* counts, int32_t* idx, int64_t idx_cap,
                        void* workspace, size_t workspace_bytes, void* stream, int32_t* status_dev);

/* Same as stof_forward, with instrumentation for bench.py: `events` is an array of
 * STOF_FORWARD_EVENTS hipEvent_t recorded on `stream` before the first kernel and after each
 * kernel of the first sub-batch (SemiGlobalBlock contract+pool, expand, body sweep), so the
 * caller can read per-kernel durations with stof_event_elapsed_ms after a sync.  `status_dev` may be NULL or
 * the range-guard word of stof_forward_checked.
 * Wraps nothing in the reference (its only timing is time.process_time(), main.py:313-315). */
#define STOF_FORWARD_EVENTS 4
int stof_forward_events(const stof_net_desc* desc, const void* packed_dev,
                        const float* x, float* y, int64_t N, int64_t L,
                        void* workspace, size_t workspace_bytes, void* stream, void* const* events,
                        int32_t* status_dev);
int stof_events_create(int32_t count, void** events_out);
int stof_events_destroy(int32_t count, void* const* events);
int stof_event_elapsed_ms(void* start, void* stop, float* ms_out);

/* SampleShuffle1D.forward (utils/sample_shuffle.py:10-28):
 * in[N, C_in, W] -> out[N, C_in/r, W*r], out[n,c,w*r+k] = in[n,k*C+c,w].     */
int stof_sample_shuffle(const float* in, float* out, int64_t N, int64_t C_in,
                        int64_t W, int32_t r, void* stream);
/* The same permutation for elements of elem_bytes = 1, 2, 4, 8 or 16 bytes, bit for bit: the reference's
 * view / permute / contiguous (utils/sample_shuffle.py:24-27) is dtype-agnostic (int64 ramps, float64, float16, complex). */
int stof_sample_shuffle_bytes(const void* in, void* out, int64_t N, int64_t C_in,
                              int64_t W, int32_t r, int32_t elem_bytes, void* stream);

/* get_maxima_positions (utils/mask2samples.py:26-34) per row of scores[N,1,M]:
 * NMS window `window_size` (made odd, :7), then threshold mode (has_threshold
 * != 0, the reference's `if threshold:`) or per-row arg-max mode.
 *   counts[N]       : detections per row (int32)
 *   idx[N, idx_cap] : time indices (int32, ascending) of the first idx_cap
 *                     detections of each row; unwritten tail is left untouched
 * The caller sizes idx_cap; a row with counts > idx_cap is truncated in `idx`
 * only (counts is exact), so the caller can re-run with a larger cap.        */
int stof_pick_maxima(const float* scores, int64_t N, int64_t M, int32_t window_size,
                     int32_t has_threshold, float threshold,
                     int32_t* counts, int32_t* idx, int64_t idx_cap, void* stream);

/* Second half of mask2coords (utils/mask2samples.py:92-112): scatter the first
 * kmax indices of each row into coords[N, kmax] (fp32, zero padded) and divide
 * by upsample_factor.                                                        */
int stof_indices_to_coords(const int32_t* counts, const int32_t* idx, int64_t idx_cap,
                           int64_t N, int64_t kmax, float upsample_factor,
                           float* coords, void* stream);

/* mask2coords with echo_max < kmax (utils/mask2samples.py:105-107,117-136): coords[N, echo_max] = the echo_max entries
 * of every row with the largest score amplitude (the reference's zero padding takes part with amplitude scores[row, 0]),
 * in ascending coordinate order, divided by upsample_factor.  counts / idx as written by stof_pick_maxima.            */
int stof_reduce_echoes(const float* scores, int64_t N, int64_t M, const int32_t* counts, const int32_t* idx,
                       int64_t idx_cap, int64_t kmax, int64_t echo_max, float upsample_factor, float* coords,
                       void* stream);

/* hilbert_transform (utils/hilbert.py:5-21) along the last dim of x[N, n]:
 * writes any of env[N,n] = |v|, re[N,n], im[N,n] that is non-NULL.           */
size_t stof_hilbert_workspace_bytes(int64_t N, int64_t n);
int stof_hilbert(const float* x, int64_t N, int64_t n, float* env, float* re, float* im,
                 void* workspace, size_t workspace_bytes, void* stream);
/* The same call for an output that nobody reads again soon (a result handed back to the framework): the envelope is
 * written with non-temporal stores and stays out of L2 / Infinity Cache ([4096,2000]: 23.9 -> 21.6 us).  A consumer that
 * follows immediately (GradPeak's row kernels) is better served by stof_hilbert, whose output it finds in the cache.  */
int stof_hilbert_streamed(const float* x, int64_t N, int64_t n, float* env, float* re, float* im,
                 void* workspace, size_t workspace_bytes, void* stream);

/* float64 rows: utils/hilbert.py:11 (torch.fft.fft) follows the input's dtype, so a float64 frame gives a complex128
 * analytic signal.  Same outputs in double; any n (mixed-radix plan over n's prime factors, O(n log n) for smooth n).
 * workspace: stof_hilbert_f64_workspace_bytes(N, n) bytes of device memory.                                       */
size_t stof_hilbert_f64_workspace_bytes(int64_t N, int64_t n);
int stof_hilbert_f64(const double* x, int64_t N, int64_t n, double* env, double* re, double* im,
                     void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------- *
 * GradPeak (models/gradpeak.py).  Common arguments:
 *   grad_step, taps[2*radius+1], radius : gradient spacing g and the Gaussian taps for sigma = (2g-1)/6, prepared
 *                                         by the host exactly as models/gradpeak.py:71-76 does (fp64 -> fp32)
 *   ival_min, ival_max                  : hysteresis gate ival_min < peak - onset < ival_max (:20,:49)
 *   echoes[N, cap, 3], counts[N]        : (onset, peak, env[peak]) of the first `cap` echoes of each row, exact counts
 *   echo_max, reduced[N, echo_max, 3]   : echo_max > 0 also writes toa_detect's reduction (:107-114: echo_max largest
 *                                         amplitudes, then ascending time, the reference's zero padding taking part);
 *                                         the caller uses it iff flags[1] > echo_max; echo_max > 0 with cap > 4096
 *                                         returns STOF_ERR_UNSUPPORTED (the reduction ranks <= 4096 entries per row)
 *   flags[2] (zeroed by the call)       : flags[0] = 1 if some row has edges but no surviving candidate (Q9: the
 *                                         reference then returns an empty tensor for the batch), flags[1] = max(counts)
 * Nothing synchronises with the host; one read of `flags` after the call tells the caller what to slice.
 * ------------------------------------------------------------------------- */

/* Pre-pass of the default threshold (Q7, models/gradpeak.py:18): adds the sum and the sum of squares of the smoothed
 * gradient of env[N, L] to stats[0], stats[1] (double, zeroed by the caller; the caller puts N*L -- or the
 * all-reduced values of a sharded batch -- into stats[0..2] = (sum, sum of squares, count)).                        */
int stof_gradpeak_moments(const float* env, int64_t N, int64_t L, int32_t grad_step, const float* taps,
                          int32_t radius, double* stats, void* stream);
/* The same pre-pass that also keeps what it computed: blurred[N, stof_gradpeak_blurred_stride(L, radius)] receives the
 * smoothed gradient of every row (in the kernels' streaming order, zero outside the row), so that the detection after
 * the threshold is known does not redo gradient and blur (stof_grad_peak_detect_blurred).                            */
int64_t stof_gradpeak_blurred_stride(int64_t L, int32_t radius);
int stof_gradpeak_moments_store(const float* env, int64_t N, int64_t L, int32_t grad_step, const float* taps,
                                int32_t radius, double* stats, float* blurred, void* stream);
/* thres_pos = std**16 * 1.2e13 (:18) from stats[3] = (sum, sum of squares, count) into threshold_out[0], on the device. */
int stof_gradpeak_threshold(const double* stats, float* threshold_out, void* stream);

/* grad_peak_detect (models/gradpeak.py:8-68) on an envelope env[N, L]: gradient -> Gaussian blur -> threshold
 * crossings -> hysteresis pairing in one launch (one wavefront per row).  The positive threshold is `threshold`, or
 * *threshold_dev when that device pointer is non-NULL (the default threshold from stof_gradpeak_threshold).        */
int stof_grad_peak_detect(const float* env, int64_t N, int64_t L, int32_t grad_step, const float* taps,
                          int32_t radius, float threshold, const float* threshold_dev, int32_t ival_min,
                          int32_t ival_max, int64_t echo_max, float* echoes, int64_t cap, float* reduced,
                          int32_t* counts, int32_t* flags, void* stream);

/* grad_peak_detect from the smoothed gradient kept by stof_gradpeak_moments_store: threshold crossings and pairing only
 * (env supplies the amplitudes of the kept peaks).  Same outputs as stof_grad_peak_detect.                            */
int stof_grad_peak_detect_blurred(const float* env, const float* blurred, int64_t N, int64_t L, int32_t radius,
                                  float threshold, const float* threshold_dev, int32_t ival_min, int32_t ival_max,
                                  int64_t echo_max, float* echoes, int64_t cap, float* reduced, int32_t* counts,
                                  int32_t* flags, void* stream);

/* toa_detect (models/gradpeak.py:99-116) with an explicit threshold in ONE launch on waveforms frame[N, L]:
 * Hilbert envelope (utils/hilbert.py:5-21) -> grad_peak_detect -> echo_max reduction; the envelope stays in LDS
 * (env_out, optional [N, L], receives a copy).  Available when stof_toa_detect_fused_ok(L, radius) != 0 (even L with
 * prime factors 2, 3, 5, a pair of rows within the LDS budget); otherwise STOF_ERR_UNSUPPORTED and the caller chains
 * stof_hilbert + stof_grad_peak_detect.                                                                              */
int stof_toa_detect_fused_ok(int64_t L, int32_t radius);
int stof_toa_detect(const float* frame, int64_t N, int64_t L, int32_t grad_step, const float* taps, int32_t radius,
                    float threshold, int32_t ival_min, int32_t ival_max, int64_t echo_max, float* echoes,
                    int64_t cap, float* reduced, int32_t* counts, int32_t* flags, float* env_out, void* stream);

/* Pre-pass of the default threshold (Q7, models/gradpeak.py:18) on WAVEFORMS frame[N, L], for row lengths the fused
 * kernel takes (stof_toa_detect_fused_ok): Hilbert envelope -> gradient -> blur in LDS as in stof_toa_detect; the
 * envelope goes to env_out[N, L] (required: the detection launch after the threshold, stof_grad_peak_detect, reads it)
 * and the sum / sum of squares of the smoothed gradient are added to stats[0], stats[1] as stof_gradpeak_moments does.
 * partials: device scratch of STOF_MOMENT_SLOTS * 16 doubles (zeroed here; the work-groups' sums are spread over that
 * many cache lines and folded into stats at the end of the call, on the stream).                                      */
#define STOF_MOMENT_SLOTS 64
int stof_toa_moments(const float* frame, int64_t N, int64_t L, int32_t grad_step, const float* taps, int32_t radius,
                     float* env_out, double* partials, double* stats, void* stream);

/* ------------------------------------------------------------------------- *
 * Neighbours of the hot path (SURVEY.md section 8f "next rows").
 * ------------------------------------------------------------------------- */

/* ChirpDataset.iq2rf (datasets/chirp_dataset.py:80-91) + NormalizeVol
 * (utils/transforms.py:13): iq[N, len, 2] (re, im) fp32 -> rf[N, int(len*rescale_factor)] fp32:
 * linear resampling on endpoint-inclusive grids, Re{y * exp(2 pi i fc t)}, then (normalize != 0)
 * division by the per-waveform max |.|.                                         */
int stof_iq2rf(const float* iq, float* rf, int64_t N, int64_t len, double rescale_factor,
               double fc, double fs, int32_t normalize, void* stream);

/* toa_rmse (utils/metrics.py:9-41): gt[N, G], es[N, E] fp32 with 0/NaN/inf as padding ->
 * out[N, 7] = (rmse, precision, recall, jaccard, tp, fp, fn).                  */
int stof_toa_rmse(const float* gt, const float* es, int64_t N, int64_t G, int64_t E, float tol,
                  float* out, void* stream);

/* ------------------------------------------------------------------------- *
 * Training step building blocks (main.py:204-248), exact fp32.
 * Activations are channel-last [N][L][C] fp32 device buffers owned by the caller.
 * ------------------------------------------------------------------------- */
#define STOF_ACT_NONE 0
#define STOF_ACT_RELU 1
#define STOF_ACT_LRELU 2

/* Conv1d weights (cout, cin, K) -> tap-major [K][cout][cin] (transpose_flip = 0, forward) or the
 * data-gradient operand [K][cin][cout] with flipped taps (transpose_flip = 1).  precision = STOF_PREC_F16X3
 * writes the split-fp16 operand image instead ([K][A][B_pad/64][64 hi | 64 lo]); `out` must hold
 * stof_train_repack_floats() floats.  stof_train_conv must be called with the same precision.       */
size_t stof_train_repack_floats(int32_t cout, int32_t cin, int32_t K, int32_t transpose_flip, int32_t precision);
int stof_train_repack(const float* w, float* out, int32_t cout, int32_t cin, int32_t K, int32_t transpose_flip,
                      int32_t precision, void* stream);
/* y = act(bias + conv_same(x, w)) + residual                     (saved == NULL: F.conv1d forward,
 *                                                                   models/stofnet.py:45,56,62,65,100,106)
 * y = (conv_same(x, w) + residual) * act'(saved)                  (saved != NULL: autograd of the same ops)
 * x[N,L,cin], w tap-major [K][cout][cin], y/residual/saved [N,L,cout]; K odd <= 9.                  */
int stof_train_conv(const float* x, const float* w_tapmajor, const float* bias, const float* residual,
                    const float* saved, float* y, int64_t N, int64_t L, int32_t cin, int32_t cout,
                    int32_t K, int32_t act, int32_t precision, void* stream);
/* dw (cout,cin,K) = sum_t dy[t][o] x[t+d-pad][c];  db[cout] = sum_t dy[t][o]  (db may be NULL).  Overwrites
 * dw/db; partial sums go through `workspace` and are added in a fixed order (bitwise reproducible).   */
size_t stof_train_wgrad_workspace_bytes(int32_t cin, int32_t cout, int32_t K);
int stof_train_wgrad(const float* x, const float* dy, float* dw, float* db, int64_t N, int64_t L,
                     int32_t cin, int32_t cout, int32_t K, float out_scale, int32_t precision, void* workspace,
                     size_t workspace_bytes, void* stream);
/* r4: the weight / bias gradients of `count` (<= 12) 64 -> 64 layers of kernel size K in ONE launch pair (split-fp16 arithmetic):
 * the eleven k7 layers of the body (models/stofnet.py:31) in the training step.  x[i], dy[i] channel-last [N][L][64];
 * dw[i] (64,64,K), db[i] (64) are overwritten; partial sums are added in a fixed order (bitwise reproducible).          */
size_t stof_train_wgrad_batch_workspace_bytes(int32_t count, int32_t K);
int stof_train_wgrad_batch(const float* const* x, const float* const* dy, float* const* dw, float* const* db, int32_t count,
                           int64_t N, int64_t L, int32_t K, float out_scale, void* workspace, size_t workspace_bytes,
                           void* stream);
/* The same with operands stored as SPLIT ROWS (bit i of x_split / dy_split: operand i): a [N][L] tensor of 256-byte rows
 * [64 x fp16 hi | 64 x fp16 lo] with value = hi + lo -- what stof_train_sweep_split / stof_train_sweep_bwd_split dump (the halves the
 * sweeps compute anyway), so this kernel stages them into LDS without converting.  Split operands must be 16-byte aligned.  */
int stof_train_wgrad_batch_split(const float* const* x, const float* const* dy, float* const* dw, float* const* db, int32_t count,
                                 uint32_t x_split, uint32_t dy_split, int64_t N, int64_t L, int32_t K, float out_scale,
                                 void* workspace, size_t workspace_bytes, void* stream);
/* conv1 (1->64, k9) + ReLU forward to channel-last, and its weight gradient (g masked by relu').    */
int stof_train_conv1(const float* x, const float* w, const float* b, float* y, int64_t N, int64_t L, void* stream);
size_t stof_train_conv1_wgrad_workspace_bytes(void);
int stof_train_conv1_wgrad(const float* x, const float* g, const float* saved, float* dw, float* db,
                           int64_t N, int64_t L, float out_scale, void* workspace, size_t workspace_bytes,
                           void* stream);
/* Gradient with respect to the input frame, which the reference's autograd yields for free (models/stofnet.py:45):
 * dx[N][L] = out_scale * conv1^T(g * relu'(saved)), g and saved channel-last [N][L][64], w = conv1.weight (64,1,9).   */
int stof_train_conv1_dgrad(const float* g, const float* saved, const float* w, float* dx, int64_t N, int64_t L,
                           float out_scale, void* stream);
/* SemiGlobalBlock pieces (models/stofnet.py:103,108-115), channel-last: MaxPool1d(scale, scale) with arg-max
 * (scale = sample_scale <= 256, P = floor(L / scale) windows), its routing backward (times lrelu' of the pre-pool
 * activation), nearest upsample x scale + pad (rem_half = (L - P*scale) / 2 on each side) + add and its backward.
 * ABI 3: `scale` is an argument (ABI 2 fixed it at 80, the value of every shipped checkpoint).             */
int stof_train_pool(const float* c, float* pooled, uint8_t* arg, int64_t N, int64_t L, int64_t P, int32_t C,
                    int32_t scale, void* stream);
/* pool backward: `c` (the pre-pool activation) may be NULL when `pooled` is given -- the activation at the arg-max IS the
 * pooled value, which is all the leaky-ReLU derivative needs (the fused forward below never materialises c). */
int stof_train_pool_bwd(const float* gpool, const uint8_t* arg, const float* c, const float* pooled, float* gc, int64_t N,
                        int64_t L, int64_t P, int32_t C, int32_t scale, void* stream);

/* Data gradient of conv_last (64 -> r channels, k3) on the vector pipe, exact fp32: out[N, L, 64] from dz[N, L, r] and
 * conv_last.weight in torch layout [r][64][3] = what stof_train_conv computes with the repacked data-gradient weights.
 * r = 4 or 10; otherwise STOF_ERR_UNSUPPORTED.                                                                        */
int stof_train_conv_last_dgrad(const float* dz, const float* weight, float* out, int64_t N, int64_t L, int32_t r, void* stream);

/* SemiGlobalBlock backward, contract_conv's weight / bias gradient straight from the pool's SPARSE gradient (one non-zero
 * row per (waveform, window, channel)): dw[C][64][5], db[C] = out_scale * the gradient that stof_train_pool_bwd +
 * stof_train_wgrad(a1, gc, cin 64, cout C, K 5) produce, without the dense [N, L, C] tensor.  gpool / arg / pooled[N, P, C]
 * as for stof_train_pool_bwd, a1[N, L, 64] the convolution's input.  STOF_ERR_UNSUPPORTED when C is not a multiple of 128
 * or scale > 92 (the caller then takes the dense route).                                                            */
/* ... and its data gradient: out[N, L, 64] = resid (may be NULL) + conv_transpose(gc, weight[C][64][5]) = what
 * stof_train_pool_bwd + the data-gradient stof_train_conv produce, from the same sparse inputs, in a fixed summation
 * order.  workspace: stof_train_sgb_dgrad_workspace_bytes(C) bytes.  STOF_ERR_UNSUPPORTED when C is not a multiple of 64,
 * C > 512, scale > 88, scale < 4 or P == 0.                                                                          */
size_t stof_train_sgb_dgrad_workspace_bytes(int32_t C);
int stof_train_sgb_contract_dgrad(const float* gpool, const uint8_t* arg, const float* pooled, const float* weight,
                                  const float* resid, float* out, int64_t N, int64_t L, int64_t P, int32_t C, int32_t scale,
                                  void* workspace, size_t workspace_bytes, void* stream);
size_t stof_train_sgb_wgrad_workspace_bytes(int32_t C);
int stof_train_sgb_contract_wgrad(const float* gpool, const uint8_t* arg, const float* pooled, const float* a1, float* dw,
                                  float* db, int64_t N, int64_t L, int64_t P, int32_t C, int32_t scale, float out_scale,
                                  void* workspace, size_t workspace_bytes, void* stream);
/* Training forward of the SemiGlobalBlock's contracting path at sample_scale 80, split-fp16: relu(conv1) -> contract_conv ->
 * lrelu -> MaxPool1d(80) fused as in inference (models/stofnet.py:45,100-103; the [N, L, 512] activation never reaches HBM):
 * pooled[N][P][512] and arg[N][P][512] = row offset of each window's FIRST maximum (torch's max_pool1d backward routing).
 * The four parameter tensors are device pointers; blob_dev = stof_train_sgb_blob_bytes() bytes of scratch the call packs. */
size_t stof_train_sgb_blob_bytes(void);
int stof_train_sgb_contract_pool(const float* conv1_w, const float* conv1_b, const float* contract_w, const float* contract_b,
                                 void* blob_dev, const float* x, float* pooled, uint8_t* arg, int64_t N, int64_t L, void* stream);
int stof_train_upsample_add(const float* a, const float* e, float* out, int64_t N, int64_t L, int64_t P,
                            int32_t rem_half, int32_t scale, void* stream);
/* The same for rows of C channels (any C >= 1): the standalone SemiGlobalBlock (models/stofnet.py:80) takes any width. */
int stof_train_upsample_add_c(const float* a, const float* e, float* out, int64_t N, int64_t L, int64_t P,
                              int32_t rem_half, int32_t scale, int32_t C, void* stream);
int stof_train_upsample_bwd(const float* g, const float* e, float* ge, int64_t N, int64_t L, int64_t P,
                            int32_t rem_half, int32_t scale, void* stream);
/* Training forward on the fused sweep (split-fp16 mode; main.py:221 with the model in train mode): conv2 .. conv12 +
 * conv_last of models/stofnet.py:51-65 in ONE launch of the inference body sweep, which additionally writes every layer's
 * output to HBM for the backward pass -- instead of twelve stof_train_conv launches.
 *   stof_train_sweep_pack : device-side packing of the CURRENT parameters (26 device pointers in stof_pack_weights order:
 *                           conv1.weight, conv1.bias, conv2.weight, ..., conv12.bias, conv_last.weight, conv_last.bias) into
 *                           the sweep's operand blob (stof_train_sweep_blob_bytes); no host copy of the weights
 *   stof_train_sweep      : x[N][L] fp32; sgb_expand[N][floor(L/80)][64] = lrelu(expand_conv(pooled)) (stof_train_conv's
 *                           output on the pooled grid; NULL for semi_global_scale 1); dump = stof_train_sweep_dump_floats
 *                           floats: 12 channel-last tensors [N][L][64] (0: relu(conv1)+SemiGlobalBlock, 1..10: outputs of
 *                           conv2..conv11 after their leaky ReLU / residual add, 11: conv12's) + a scratch tail;
 *                           y[N][L*r] = the sample-shuffled prediction.  semi_global_scale 1 or 80 only.                  */
size_t stof_train_sweep_blob_bytes(const stof_net_desc* desc);
int stof_train_sweep_pack(const stof_net_desc* desc, const float* const* params_dev, void* blob_dev, void* stream);
size_t stof_train_sweep_dump_floats(int64_t N, int64_t L);
int stof_train_sweep(const stof_net_desc* desc, const void* blob_dev, const float* x, const float* sgb_expand, float* dump,
                     float* y, int64_t N, int64_t L, void* stream);
/* stof_train_sweep with dump tensors 0..10 written as SPLIT ROWS (see stof_train_wgrad_batch_split); tensor 11 (conv12's output,
 * read by conv_last's weight gradient) stays fp32.  STOF_ERR_UNSUPPORTED when the two-pass sweep kernel is switched off.  */
int stof_train_sweep_split(const stof_net_desc* desc, const void* blob_dev, const float* x, const float* sgb_expand, float* dump,
                           float* y, int64_t N, int64_t L, void* stream);
/* The data-gradient chain of the same step as ONE backward sweep (the mirror image of stof_train_sweep): from
 * g6[N][L][64] = dL/d(conv12 output) (= stof_train_conv of conv_last's transposed weights on dL/dpred) it runs conv12^T,
 * conv11^T .. conv2^T with the leaky-ReLU derivatives (read off the sign of the saved activations in fwd_dump = the dump
 * stof_train_sweep wrote) and the residual additions of models/stofnet.py:51-62 reversed, and writes every layer's output for the weight-gradient
 * kernels: dump tensor j (1..11; stof_train_sweep_dump_floats) = output of sweep layer j = conv(13-j)^T: j odd: dL/dx_k with
 * k = (11-j)/2 (before the long-skip term), j even: dL/d(conv(12-j) output before its leaky ReLU).  blob: the eleven
 * conv2..conv12 weights (device pointers, forward order) packed on the device by stof_train_sweep_bwd_pack into
 * stof_train_sweep_blob_bytes bytes.                                                                                   */
int stof_train_sweep_bwd_pack(const float* const* conv_weights_dev, void* blob_dev, void* stream);
int stof_train_sweep_bwd(const stof_net_desc* desc, const void* blob_dev, const float* g6, const float* fwd_dump,
                         float* dump, int64_t N, int64_t L, void* stream);
/* Loss of main.py:228-232: target = amplitude * blur7(coords2mask(gt)) / max, loss = MSE + lambda * mean|pred|;
 * writes target[N*M], tmax[1], dpred[N*M] = grad_scale * dloss/dpred and loss[1] (double).  grad_scale is a power
 * of two (loss scaling: keeps the back-propagated values inside the fp16 range of the f16x3 mode); the weight-
 * gradient entry points undo it with out_scale = 1/grad_scale, both exactly.                           */
int stof_train_loss(const float* pred, const int64_t* gt_idx, int64_t G, const float* taps7, int64_t N, int64_t M,
                    float amplitude, float lambda, float grad_scale, float* target, float* tmax, float* dpred,
                    double* loss, void* stream);
/* The two halves of stof_train_loss: target[N*M] = blur7(coords2mask(gt)) and its maximum tmax[0]; then the loss and
 * dpred from (pred, target, tmax).  A batch sharded over ranks MAX-all-reduces tmax in between, because the reference
 * normalises by the maximum over the WHOLE batch (main.py:230).                                                    */
int stof_train_loss_target(const int64_t* gt_idx, int64_t G, const float* taps7, int64_t N, int64_t M,
                           float* target, float* tmax, void* stream);
int stof_train_loss_grad(const float* pred, float* target, const float* tmax, int64_t N, int64_t M, float amplitude,
                         float lambda, float grad_scale, float* dpred, double* loss, void* stream);
/* stof_train_sweep_bwd on a forward dump written by stof_train_sweep_split (the leaky-ReLU derivative is read off the sign of the
 * saved activation's hi half: an activation with |y| < 2^-25 counts as not positive), all eleven output tensors as split rows;
 * g6 is a split-row tensor too (stof_train_conv_last_dgrad_split or stof_train_to_split_rows).                              */
int stof_train_sweep_bwd_split(const stof_net_desc* desc, const void* blob_dev, const float* g6, const float* fwd_dump,
                               float* dump, int64_t N, int64_t L, void* stream);
/* fp32 rows in[rows][64] -> split rows out (conv12's output gradient g6, so that every operand of
 * stof_train_wgrad_batch_split is split: that case runs on the global_load_lds kernel).                                   */
int stof_train_to_split_rows(const float* in, float* out, int64_t rows, void* stream);
/* out[rows][64] = (hi + lo of the split rows a_split) + b  (the long-skip join on the backward sweep's split-row dL/dx_0).  */
int stof_train_add_split(const float* a_split, const float* b, float* out, int64_t rows, void* stream);
/* The same with b a split-row tensor as well.                                                                              */
int stof_train_add_split2(const float* a_split, const float* b_split, float* out, int64_t rows, void* stream);
/* stof_train_conv_last_dgrad with out written as split rows: the g6 operand of stof_train_sweep_bwd_split (which expects its g6
 * as split rows) and of stof_train_wgrad_batch_split.                                                                      */
int stof_train_conv_last_dgrad_split(const float* dz, const float* weight, float* out, int64_t N, int64_t L, int32_t r, void* stream);
/* out = a + b (gradient joins of the residual / long-skip branches, models/stofnet.py:56,62).         */
int stof_train_add(const float* a, const float* b, float* out, int64_t n, void* stream);
/* torch.optim.AdamW step on one flat parameter vector (main.py:179,248).                             */
int stof_train_adamw(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                     float beta1, float beta2, float eps, float weight_decay, int64_t step, void* stream);
/* The same step behind the range guard of the split-fp16 training arithmetic (the reference trains in plain fp32,
 * main.py:221-248): one scan of the gradient bucket (and of *loss, if given) for non-finite values, then AdamW -- if the scan
 * found one, the gradients count as zero (and are zeroed) and guard_words[1] is raised (sticky; the caller reads and clears
 * it whenever it likes).  guard_words: two device ints owned by the caller, zero before the first step; guard_words[0] is
 * scratch (the number of the last bad step).  Two launches, no host read.                                              */
int stof_train_adamw_guarded(float* params, float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                             float beta1, float beta2, float eps, float weight_decay, int64_t step, const double* loss,
                             int32_t* guard_words, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* STOFNET_AMD_H */
