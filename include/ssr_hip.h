/* libssrhip - C ABI of the MI355X (gfx950) implementation of the ssr_eval DSP / metric hot path.
 *
 * This header is the drop-in boundary.  Every entry point names the reference interface
 * (haoheliu/ssr_eval v0.0.6, file:line) whose arithmetic it replaces; the Python classes in
 * ssr_eval_amd/ (AudioMetrics, FDomainHelper, lowpass, SSR_Eval_Helper) bind these symbols through
 * ctypes and nothing else (INTEGRATION.md shows the binding).
 *
 * Conventions
 *  - plain C, no exceptions, no ownership transfer; every function returns SSR_OK (0) or a negative
 *    error code, with a thread-local message available from ssr_last_error();
 *  - every `const float*`, `float*`, `double*`, offset and length ARRAY argument is a DEVICE pointer
 *    (HBM) valid on the current HIP device; scalar arguments are host values;
 *  - `stream` is a hipStream_t passed as void* (NULL = the null stream); all work is enqueued on it and
 *    nothing synchronises with the host;
 *  - batches are ragged: item i occupies elements [off[i], off[i] + len[i]) of its buffer; spectrogram
 *    rows of item i start at row frame_off[i] of a [total_rows, n_bins] float32 matrix;
 *  - a plan is immutable after creation and may be shared between host threads.
 */
#ifndef SSR_HIP_H_
#define SSR_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SSR_OK 0
#define SSR_ERR_INVALID_ARG (-1)
#define SSR_ERR_UNSUPPORTED (-2)
#define SSR_ERR_HIP (-3)
#define SSR_ERR_WORKSPACE (-4)

/* transform precision: SSR_F64 reproduces librosa's float64 pocketfft -> complex64 rounding (parity
 * mode, default); SSR_F32 is a float32 transform (faster, not parity-safe on band-limited input). */
#define SSR_F32 0
#define SSR_F64 1

/* metric_mask bits; outputs are always laid out [n_items][4] = lsd, log_sispec, sispec, ssim
 * (key order of ssr_eval/metrics.py:98-103); unrequested entries are NaN. */
#define SSR_METRIC_LSD 1u
#define SSR_METRIC_LOG_SISPEC 2u
#define SSR_METRIC_SISPEC 4u
#define SSR_METRIC_SSIM 8u
#define SSR_METRIC_ALL 15u

/* ssr_stft output kinds */
#define SSR_STFT_MAG 1     /* out_a = |X|                       (ssr_eval/metrics.py:27) */
#define SSR_STFT_COMPLEX 2 /* out_a = Re X, out_b = Im X        (ssr_eval/dsp.py:64,73,77) */

typedef struct ssr_plan ssr_plan;

const char* ssr_last_error(void);
int ssr_version(void);

/* STFT plan: centred, reflect-padded, periodic-Hann, win_length = n_fft.
 * Replaces the parameter choice of AudioMetrics.__init__ (ssr_eval/metrics.py:16-19: rate -> n_fft, hop),
 * FDomainHelper.__init__ (ssr_eval/dsp.py:7-59: 2048 / 441) and librosa.stft defaults (eval.py:29: 2048 / 512).
 * Any 2 <= n_fft <= 4096 is accepted: powers of two in [256, 4096] run a direct FFT; n_fft = R q with R in {1, 2, 3} and
 * q <= 768 - every AudioMetrics(rate) size: 2229 = 3 * 743, 1486 = 2 * 743, 1114 = 2 * 557, 743 - runs one radix-R step over R
 * chirp-z (Bluestein) transforms of length 1536 on float32 pairs (q up to 1024: length 2048); everything else, float64 signals
 * and single-signal mode run Bluestein over a power of two >= 2 n_fft - 1 (radix 3 over three 2048-point ones for 2229). */
int ssr_plan_create(int n_fft, int hop, int precision, ssr_plan** plan);
/* FDomainHelper(window_size, hop_size, center, pad_mode, window) with anything but the defaults (ssr_eval/dsp.py:7-59 passes the
 * five straight to torchlibrosa's STFT / ISTFT, win_length = n_fft).  window: HOST pointer to n_fft float64 values - what
 * librosa.filters.get_window(name, n_fft, fftbins=True) returns for the caller's window name - or NULL for the periodic Hann;
 * center: 0 = no padding, frames start at sample 0, T = 1 + (n - n_fft) / hop, ISTFT trims nothing at the front;
 * pad_mode: torch's F.pad mode when center != 0 (SSR_PAD_REFLECT, SSR_PAD_CONSTANT = zeros).
 * Such a plan has the SSR_LOWPASS_CONV engine only (n_fft = 32 m <= 4096; torchlibrosa's own construction - conv weights from any
 * window - is the only one defined for it) and serves ssr_stft(SSR_STFT_COMPLEX), ssr_istft, ssr_fft_lowpass, ssr_num_frames,
 * ssr_ola_workspace_bytes and ssr_plan_query; every other entry point returns SSR_ERR_UNSUPPORTED for it.
 * Per item: center + reflect needs len > n_fft / 2, center + constant len >= 1, center = 0 len >= n_fft (an item that cannot be
 * framed gets 0 rows and a zeroed output; the LONGEST item failing is SSR_ERR_INVALID_ARG).  With center = 0 the overlap-added signal
 * is (T - 1) hop + n_fft samples long; torch's slice ends there, the C ABI's fixed-size output is zero-filled beyond it.
 * Parity: bit-exact against oracle/tl_chain.c; the non-default branches of that oracle are restated from the published torchlibrosa
 * module without a reference-generated vector (the reference only ever constructs FDomainHelper(), lowpass.py:18): "parity unpinned". */
#define SSR_PAD_REFLECT 0
#define SSR_PAD_CONSTANT 1
int ssr_plan_create_ex(int n_fft, int hop, const double* window, int center, int pad_mode, ssr_plan** plan);
int ssr_plan_destroy(ssr_plan* plan);
int ssr_plan_query(const ssr_plan* plan, int* n_fft, int* hop, int* n_bins, int* fft_len, int* bluestein,
                   int* precision);
/* Engine of ssr_fft_lowpass / ssr_istft for this plan (set it before the plan is shared between threads):
 *   SSR_LOWPASS_SEGMENTS (default): the frame kernel writes frames / paired segments to the workspace, a second kernel overlap-adds;
 *   SSR_LOWPASS_FUSED: transforms AND overlap-add in one kernel, no workspace traffic (HBM bytes = signal in + signal out).
 *     Float64 2048-point plans with 228 <= hop <= 914 only (SSR_ERR_UNSUPPORTED otherwise).  Measured on MI355X: 3 % faster when
 *     the call runs alone, 8 % slower inside a pipeline of other FP64-heavy kernels (it needs more cycles at a lower power
 *     density) - hence not the default.  Both give the reference's result to float32 resolution; the
 *     fused engine rounds to float32 once (sums in float64 from the unrounded frames).
 *   SSR_LOWPASS_CONV: the REFERENCE'S ARITHMETIC CLASS.  torchlibrosa (ssr_eval/dsp.py:1,21-39) evaluates STFT / ISTFT as dense float32
 *     DFT convolutions (nn.Conv1d with DFT x Hann weights computed in float64, stored float32); a hard-low-passed signal's stop band is
 *     that transform's round-off floor, and LSD / log-SISpec of the degraded input take its logarithm: against the float64 FFT engines
 *     above they differ by 2-7 % (LSD).  This engine runs the same dense products on the fp32 matrix cores (v_mfma_f32_32x32x2_f32 =
 *     a float32 fused-multiply-add chain) IN THE ACCUMULATION ORDER OF TORCH-CPU'S F.conv1d (oneDNN, AVX-512, >= 2 threads,
 *     established bit for bit: tests/test_oracle.py::test_tl_chain_is_torch_conv1d_bit_for_bit): forward one chain of n_fft terms
 *     per output (what torch runs for signals of >= 55 frames), inverse one chain per block of 256 channels of the mirrored
 *     spectrum, the blocks added in float32 in order - oracle/tl_chain.c restates it, the kernels reproduce it bit for bit -
 *     followed by F.fold's float32 overlap-add and window-sum division.  With the reference's own weight tables
 *     (ssr_plan_set_tl_weights) a low-passed waveform of >= 55 frames IS the reference's, sample for sample.  ssr_fft_lowpass(_multi),
 *     ssr_istft and ssr_stft(SSR_STFT_COMPLEX) of such a plan all run it.  n_fft = 32 m.  Cost: 2 n_fft (2 cut) + 2 n_fft (4 cut)
 *     flops per frame instead of two FFTs (10-30x the time of the default engine), and ssr_ola_workspace_bytes grows to
 *     ~(2 n_fft + 2 n_bins + hop) floats per frame.  The Python mirror makes it the default of lowpass(_type="stft_hard") and
 *     FDomainHelper. */
#define SSR_LOWPASS_SEGMENTS 0
#define SSR_LOWPASS_FUSED 1
#define SSR_LOWPASS_CONV 2
int ssr_plan_set_lowpass_engine(ssr_plan* plan, int engine);
/* The float32 weight tables of SSR_LOWPASS_CONV as the library builds them (torchlibrosa's DFTBase.dft_matrix / idft_matrix x
 * periodic Hann: STFT.__init__, ISTFT.init_real_imag_conv), TRANSPOSED: fwd_*_t [n_fft][n_fft/2+1] (row = sample, column = bin),
 * inv_*_t [n_fft][n_fft] (row = channel of the mirrored spectrum, column = output sample), w2 = hann^2 [n_fft].
 * HOST pointers (the one exception to the device-pointer convention; any may be NULL): a host-only introspection call, needs no GPU;
 * the parity tests feed these tables to oracle/tl_chain.c. */
int ssr_tl_weights(int n_fft, float* fwd_re_t, float* fwd_im_t, float* inv_re_t, float* inv_im_t, float* w2);
/* Replace the SSR_LOWPASS_CONV tables of a plan by the caller's (HOST pointers, torchlibrosa's own module layout):
 *   fwd_re / fwd_im [n_bins][n_fft] = STFT.conv_real / conv_imag .weight[:, 0, :]   (torchlibrosa stft.py STFT.__init__)
 *   inv_re / inv_im [n_fft][n_fft]  = ISTFT.conv_real / conv_imag .weight[:, :, 0]  (row = output sample, column = channel)
 *   w2 [n_fft] = float32(window ** 2)                                              (ISTFT._get_ifft_window_sum_onnx / fold input)
 * The Python mirror evaluates DFTBase.dft_matrix / idft_matrix with the module's own numpy expressions (np.power(omega, x * y) in
 * complex128) and hands the float32 results over, so that the engine multiplies by the REFERENCE'S weights to the last bit (the
 * library's own tables - exact phase reduction in long double - differ from numpy's in 1 ulp of ~0.5 % of the entries).
 * Call it before the plan is shared between threads / before launches that should see the tables are enqueued.  The library's own
 * tables are only built by the first conv launch that finds none, so a caller that sets its tables first never pays for them; a set
 * that IS superseded (this call twice, or after such a launch) is released after a hipDeviceSynchronize(). */
int ssr_plan_set_tl_weights(ssr_plan* plan, const float* fwd_re, const float* fwd_im, const float* inv_re, const float* inv_im,
                            const float* w2);
/* The same for a caller-supplied window (HOST float64 [n_fft], NULL = periodic Hann): the tables of an ssr_plan_create_ex plan. */
int ssr_tl_weights_ex(int n_fft, const double* window, float* fwd_re_t, float* fwd_im_t, float* inv_re_t, float* inv_im_t, float* w2);
/* T = 1 + (n + 2 pad - n_fft) / hop, pad = n_fft / 2 (0 for a center = 0 plan; then T = 0 when n < n_fft)
 * (librosa / torchlibrosa frame count; bit-exact integer) */
int64_t ssr_num_frames(const ssr_plan* plan, int64_t n_samples);

/* K1+K2.  Batched STFT of ragged float32 waveforms -> [total_rows, n_bins] float32.
 * Replaces AudioMetrics.wav_to_spectrogram (ssr_eval/metrics.py:26-30 -> librosa.stft + abs + transpose)
 * and FDomainHelper.complex_spectrogram / spectrogram / spectrogram_phase (ssr_eval/dsp.py:61-81 ->
 * torchlibrosa STFT).  Reflect padding follows numpy.pad (what librosa calls): items shorter than n_fft/2 are reflected
 * repeatedly; every item needs len >= 1.  (torch's reflect padding, used by torchlibrosa, refuses such items - the Python
 * FDomainHelper mirror raises for them before calling in.) */
int ssr_stft(const ssr_plan* plan, const float* wav, const int64_t* wav_off, const int32_t* wav_len,
             const int64_t* frame_off, int n_items, int max_len, int out_kind, float* out_a, float* out_b,
             void* stream);

/* K2'.  mag = sqrt(max(re^2 + im^2, eps)), cos = re / mag, sin = im / mag   (ssr_eval/dsp.py:76-81). */
int ssr_magphase(const float* re, const float* im, int64_t n, float eps, float* mag, float* cosv, float* sinv,
                 void* stream);

/* K1-K5 fused driver: the four metrics of AudioMetrics.evaluation (ssr_eval/metrics.py:51-107) for a
 * ragged batch of (est, target) waveform pairs already truncated to a common length len[i]
 * (metrics.py:89-90).  out: double [n_items][4].  total_rows = sum_i T_i, frame_off = exclusive prefix
 * sum of T_i.  workspace: at least ssr_pair_metrics_workspace_bytes(...) bytes of device memory. */
size_t ssr_pair_metrics_workspace_bytes(const ssr_plan* plan, int n_items, int max_len, int64_t total_rows);
/* The same for a given metric_mask: without SSIM the two [total_rows, n_bins] magnitude images are never materialised and the
 * workspace shrinks from 8 bytes per bin of the batch to the partial records (a few hundred bytes per item).  A workspace
 * sized for a mask serves every call whose mask is a subset of it. */
size_t ssr_pair_metrics_workspace_bytes_for(const ssr_plan* plan, int n_items, int max_len, int64_t total_rows,
                                            unsigned metric_mask);
int ssr_pair_metrics(const ssr_plan* plan, const float* est, const int64_t* est_off, const float* tgt,
                     const int64_t* tgt_off, const int32_t* len, const int64_t* frame_off, int n_items, int max_len,
                     int64_t total_rows, unsigned metric_mask, double* out, void* workspace, size_t workspace_bytes,
                     void* stream);

/* The same for an estimate held as a FLOAT64 signal against a float32 target.  This is what
 * AudioMetrics.evaluation receives from SSR_Eval_Helper.evaluate_single (ssr_eval/eval.py:128-156) whenever the
 * degradation was an IIR low-pass: scipy.signal.sosfiltfilt returns float64 (ssr_eval/lowpass.py:54-131), the
 * testee output and librosa.resample keep it, librosa.stft then yields complex128, and torch promotes every
 * est-side operation of metrics.py:109-121 to float64.  Same workspace size, same output layout. */
int ssr_pair_metrics_est64(const ssr_plan* plan, const double* est, const int64_t* est_off, const float* tgt,
                           const int64_t* tgt_off, const int32_t* len, const int64_t* frame_off, int n_items,
                           int max_len, int64_t total_rows, unsigned metric_mask, double* out, void* workspace,
                           size_t workspace_bytes, void* stream);

/* Both signals float64 (e.g. arrays read with soundfile's default dtype handed to AudioMetrics.evaluation): two
 * complex128 spectra, every tensor of metrics.py:109-121 float64. */
int ssr_pair_metrics_f64(const ssr_plan* plan, const double* est, const int64_t* est_off, const double* tgt,
                         const int64_t* tgt_off, const int32_t* len, const int64_t* frame_off, int n_items, int max_len,
                         int64_t total_rows, unsigned metric_mask, double* out, void* workspace, size_t workspace_bytes,
                         void* stream);

/* K1-K5 for ONE target and K estimates per item: SSR_Eval_Helper.evaluate_single scores every degradation key of a file against the
 * same target (ssr_eval/eval.py:136-154).  est_off is key-major, [n_keys][n_items]: estimate k of item i starts at element
 * est_off[k * n_items + i] of est; all K estimates and the target of item i share the truncated length len[i] (metrics.py:89-90);
 * out: double [n_items][n_keys][4].  The target is transformed once (with estimate 0: that key is bit-identical to ssr_pair_metrics)
 * and its magnitude image stored once; estimates 1 .. K-1 are transformed two per complex transform and reduced against the stored
 * image - the same per-bin arithmetic on the same float32 magnitudes, so the metrics agree with K calls of ssr_pair_metrics to
 * <= 1e-6 relative (not bit for bit: an estimate separated from another estimate instead of the target keeps ~1e-16 of its
 * partner's spectrum, a last-bit flip of a float32 magnitude now and then; float64 partial sums are added in another order).
 * K + 1 real transforms and K + 1 magnitude images per item instead of 2 K and 2 K.  Plans whose pair transform runs a block engine
 * take K plain passes (bit-identical to ssr_pair_metrics). */
size_t ssr_pair_metrics_multi_workspace_bytes(const ssr_plan* plan, int n_items, int n_keys, int max_len, int64_t total_rows,
                                              unsigned metric_mask);
int ssr_pair_metrics_multi(const ssr_plan* plan, const float* est, const int64_t* est_off, const float* tgt, const int64_t* tgt_off,
                           const int32_t* len, const int64_t* frame_off, int n_items, int n_keys, int max_len, int64_t total_rows,
                           unsigned metric_mask, double* out, void* workspace, size_t workspace_bytes, void* stream);

/* The same for K FLOAT64 estimates per float32 target: what SSR_Eval_Helper.evaluate_single holds when `setting_lowpass_filtering` is on -
 * every IIR key of a file is a float64 signal (scipy.signal.sosfiltfilt, ssr_eval/lowpass.py:54-131) scored against the file's one
 * float32 target (ssr_eval/eval.py:136-154, 243-258).  est: double, key-major offsets as above.  Key 0 runs as ssr_pair_metrics_est64
 * (bit-identical to it) and stores the target's magnitude image; keys 1 .. K-1 are transformed two per complex float64 transform into
 * float32 magnitude rows - numpy.abs of the unrounded complex128 spectrum, rounded once - and reduced against the stored image (an odd
 * last key runs with the target again).  Against K calls of ssr_pair_metrics_est64: the estimate's magnitude enters the LSD / SISpec
 * terms rounded to float32 (6e-8 relative: <= 5e-8 absolute on an LSD term, the size of that entry point's own float32 logarithms)
 * and the terms are float32 sequences summed in float64 - metrics equal to <= 1e-6 relative, 1e-5 being the bar.  Plans without a
 * two-estimate wave kernel (today: every n_fft but 3 q <= 2304, i.e. every rate but AudioMetrics(48000)) take K plain passes,
 * bit-identical to ssr_pair_metrics_est64. */
size_t ssr_pair_metrics_multi_est64_workspace_bytes(const ssr_plan* plan, int n_items, int n_keys, int max_len, int64_t total_rows,
                                                    unsigned metric_mask);
int ssr_pair_metrics_multi_est64(const ssr_plan* plan, const double* est, const int64_t* est_off, const float* tgt, const int64_t* tgt_off,
                                 const int32_t* len, const int64_t* frame_off, int n_items, int n_keys, int max_len, int64_t total_rows,
                                 unsigned metric_mask, double* out, void* workspace, size_t workspace_bytes, void* stream);

/* Same, restricted to a subset of its three launches (bit 0: STFT + fused LSD/SISpec epilogue, bit 1:
 * SSIM, bit 2: finalisation) so that bench.py can time the dominant kernel on its own stream with
 * HIP events.  stages = 7 is ssr_pair_metrics. */
int ssr_pair_metrics_stages(const ssr_plan* plan, const float* est, const int64_t* est_off, const float* tgt,
                            const int64_t* tgt_off, const int32_t* len, const int64_t* frame_off, int n_items,
                            int max_len, int64_t total_rows, unsigned metric_mask, double* out, void* workspace,
                            size_t workspace_bytes, void* stream, int stages);

/* K3-K5 on precomputed [T_i, n_bins] magnitude spectrograms: AudioMetrics.lsd / .sispec (incl. the
 * to_log variant) / .ssim (ssr_eval/metrics.py:109-132, ssr_eval/utils.py:43-92).  est first, target second.
 * SSIM needs T_i >= 7 and n_bins >= 7. */
size_t ssr_spectrogram_metrics_workspace_bytes(int n_items, int max_rows, int n_bins);
int ssr_spectrogram_metrics(const float* est_sp, const float* tgt_sp, const int64_t* frame_off, const int32_t* n_rows,
                            int n_items, int max_rows, int n_bins, unsigned metric_mask, double* out, void* workspace,
                            size_t workspace_bytes, void* stream);

/* A6.  The tensor helpers of ssr_eval/utils.py as stand-alone calls (inside ssr_pair_metrics /
 * ssr_spectrogram_metrics they are fused; these back `from ssr_eval.utils import to_log, pow_p_norm, ...`).
 *   ssr_to_log      out = log10(x + 1e-12)                       utils.py:43-44
 *   ssr_from_log    out = 10 ** min(x, 5)                        utils.py:47-49
 *   ssr_energy_sums sums[i] = {sum a^2, sum b^2, sum a*b} over item i's per_item contiguous elements (float64):
 *                   pow_p_norm = (float)sqrt(sum)^2 (utils.py:68-76), pow_norm = sum a*b (utils.py:85-92)
 *   ssr_scale_items out[i][j] = (x[i][j] * mul[i]) / div[i]      the rescaled target of energy_unify, utils.py:79-82 */
int ssr_to_log(const float* x, int64_t n, float* out, void* stream);
int ssr_from_log(const float* x, int64_t n, float* out, void* stream);
int ssr_energy_sums(const float* a, const float* b, int n_items, int64_t per_item, double* sums, void* stream);
int ssr_scale_items(const float* x, const float* mul, const float* div, int n_items, int64_t per_item, float* out,
                    void* stream);

/* A5 on [B, C, T, F] tensors with C > 1: AudioMetrics.sispec / its to_log variant (ssr_eval/metrics.py:114-121 with
 * energy_unify, utils.py:79-82).  pow_norm (utils.py:85-92) is per (b, c), pow_p_norm (utils.py:68-76) over every dimension
 * but the batch, so the per-channel projections share one all-channel target energy and there is one ratio per batch item.
 * est / tgt: contiguous [n_batch, n_channels, per_image] float32; log_domain != 0 applies to_log (utils.py:43-44) to both
 * first.  out: [n_batch + 1] float64 = the per-item values, then their mean (metrics.py:121: sum / B).  Energies are
 * accumulated in float64 on the difference est - tgt (accurate at any SNR).  workspace: n_batch * n_channels * 24 bytes. */
int ssr_sispec_multichannel(const float* est, const float* tgt, int n_batch, int n_channels, int64_t per_image,
                            int log_domain, double* out, void* workspace, size_t workspace_bytes, void* stream);

/* K6.  STFT-domain hard low-pass: stft_hard_lowpass_v0 (ssr_eval/lowpass.py:17-28) through
 * FDomainHelper.wav_to_spectrogram_phase / spectrogram_phase_to_wav (ssr_eval/dsp.py:83-119).
 * cut[i] = first zeroed bin = int(n_bins * highcut / int(fs/2)) (lowpass.py:24,193-194; computed by the
 * caller, bit-exact integer).  Output has the input's layout (same off / len).  Power-of-two n_fft only.
 * workspace: ssr_ola_workspace_bytes(plan, total_rows).
 * Precondition, PER ITEM: len[i] > n_fft / 2 (torch's reflect padding inside torchlibrosa's STFT refuses anything
 * shorter, dsp.py:21-59).  max_len <= n_fft / 2 is SSR_ERR_INVALID_ARG; lengths live on the device, so a short item in a
 * batch whose max_len passes cannot be reported - it is skipped and its output samples are set to 0 (no out-of-range
 * access); the Python mirror checks every item on the host and raises ValueError.  Same for ssr_istft. */
size_t ssr_ola_workspace_bytes(const ssr_plan* plan, int64_t total_rows);
int ssr_fft_lowpass(const ssr_plan* plan, const float* in, const int64_t* off, const int32_t* len, const int32_t* cut,
                    const int64_t* frame_off, int n_items, int max_len, int64_t total_rows, float* out,
                    void* workspace, size_t workspace_bytes, void* stream);

/* K6 for ONE batch and K cut-offs: SSR_Eval_Helper.lowpass_stft_hard (ssr_eval/eval.py:401-410) loops stft_hard_lowpass_v0 over the
 * cutoffs of setting_fft on the same waveform.  cuts: HOST array of n_keys cut bins, each applied to every item of the batch
 * (int(n_bins * (cutoff // 2) / int(fs / 2)) with the evaluator's one fs).  Key k's output goes to out + k * key_stride (elements),
 * in the input's layout (same off / len).  On the SSR_LOWPASS_CONV engine the padded copy and the forward dense-DFT product are
 * computed ONCE, at the largest cut (spectrogram_phase and mag * cos / mag * sin of a bin do not depend on the cut: every output is
 * bit-identical to K ssr_fft_lowpass calls), followed by one inverse product per key over that key's channels; row tiles run over
 * the whole batch (one cut per launch).  The FFT engines run K plain calls.  workspace: ssr_ola_workspace_bytes(plan, total_rows). */
int ssr_fft_lowpass_multi(const ssr_plan* plan, const float* in, const int64_t* off, const int32_t* len, const int32_t* cuts,
                          int n_keys, const int64_t* frame_off, int n_items, int max_len, int64_t total_rows, float* out,
                          int64_t key_stride, void* workspace, size_t workspace_bytes, void* stream);

/* K6'.  Inverse STFT: torchlibrosa ISTFT.forward(real, imag, length) as used by
 * FDomainHelper.spectrogram_phase_to_wav / reverse_complex_spectrogram (ssr_eval/dsp.py:67-70,107-119).
 * re/im: [total_rows, n_bins]; item i has ssr_num_frames(len[i]) rows. */
int ssr_istft(const ssr_plan* plan, const float* re, const float* im, const int64_t* frame_off, const int32_t* len,
              const int64_t* out_off, int n_items, int max_len, int64_t total_rows, float* out, void* workspace,
              size_t workspace_bytes, void* stream);

/* K7.  Polyphase resampler = scipy.signal.resample_poly(x, up, down) for float32 x, i.e.
 * librosa.resample(..., res_type="polyphase") (ssr_eval/eval.py:145-150) and subsampling()
 * (ssr_eval/lowpass.py:134-144).  ssr_resample_plan reproduces SciPy's integer plan on the host;
 * `taps` is the DEVICE copy of zeros(n_pre_pad) ++ firwin(2*half_len+1, 1/max(up,down), kaiser 5.0)*up
 * as float32 (the caller designs the taps exactly as SciPy does).  Output bit-identical to SciPy. */
int ssr_resample_plan(int64_t n_in, int up, int down, int* up_reduced, int* down_reduced, int64_t* n_out,
                      int* half_len, int* n_pre_pad, int* n_pre_remove);
int ssr_resample_poly(const float* in, const int64_t* in_off, const int32_t* in_len, const int64_t* out_off,
                      const int32_t* out_len, int n_items, int max_out_len, int up, int down, const float* taps,
                      int n_taps, int n_pre_remove, float* out, void* stream);

/* K7 twice in one kernel (round 4): resample_poly(resample_poly(x, up1, down1), up2, down2) for float32 signals with the INTERMEDIATE
 * signal held in LDS only - BASELINE cfg-5's 16 kHz -> 44.1 kHz -> 48 kHz chain (librosa.resample(res_type="polyphase") at
 * ssr_eval/eval.py:144-150 behind a testee that itself up-sampled).  Both stages are the bit-exact kernel of ssr_resample_poly: the
 * intermediate has SciPy's bits and so has the output - bit-identical to two ssr_resample_poly calls, at 12.8 GB of HBM traffic per
 * 12,500 utterances of 4 s instead of 30.6 GB.  mid_len[i] (device) = the intermediate length of item i (ssr_resample_plan's n_out
 * for stage 1); out_len / max_out_len refer to the final signal.  Plans (reduced up / down, taps, n_pre_remove) as for
 * ssr_resample_poly.  SSR_ERR_UNSUPPORTED unless both plans have 21 taps per phase and 8 up1 = 24 down2 with up1 <= 448,
 * up2 <= 160 (441/160 then 160/147 and its multiples): the caller then runs the two stages through ssr_resample_poly. */
int ssr_resample_poly_chain(const float* in, const int64_t* in_off, const int32_t* in_len, const int32_t* mid_len,
                            const int64_t* out_off, const int32_t* out_len, int n_items, int max_out_len, int up1, int down1,
                            const float* taps1, int n_taps1, int n_pre_remove1, int up2, int down2, const float* taps2,
                            int n_taps2, int n_pre_remove2, float* out, void* stream);

/* K7 on the matrix cores (round 3): the same sums as ssr_resample_poly, evaluated as a dense (outputs x input window) by
 * (input window x 32 utterances) product on v_mfma_f32_32x32x2_f32 - float32 FUSED multiply-adds in ascending input index, one
 * rounding per tap where SciPy's upfirdn rounds product and sum separately.  NOT bit-identical to scipy.signal.resample_poly:
 * within ~1 ulp per tap of it.  Metrics of a resampled signal whose upper band is the resampler's own leakage move with those
 * ulps: LSD by up to 1.2e-5 relative over 12,500 white-noise utterances (typically 1e-6), log-SISpec by 2.5e-5 - a caller that needs
 * the 1e-5 bar against SciPy-resampled references uses ssr_resample_poly; 1.2-1.5x the rate of the bit-exact kernel (measured: 441/160 1.45 against 2.23 ms, 160/147
 * 1.93 against 2.27 ms per 4096 utterances of 4 s).  Same
 * arguments and sample indices.  SSR_ERR_UNSUPPORTED for plans whose tap table (> 96 KB) or 32-output window (> 446 samples)
 * do not fit - ssr_resample_poly serves every plan. */
int ssr_resample_poly_mfma(const float* in, const int64_t* in_off, const int32_t* in_len, const int64_t* out_off,
                           const int32_t* out_len, int n_items, int max_out_len, int up, int down, const float* taps,
                           int n_taps, int n_pre_remove, float* out, void* stream);

/* K7 for a float64 signal (what the reference has after an IIR degradation or a float64-returning testee):
 * SciPy then keeps the taps in float64 and accumulates in float64; same plan, same ordering, bit-identical. */
int ssr_resample_poly_f64(const double* in, const int64_t* in_off, const int32_t* in_len, const int64_t* out_off,
                          const int32_t* out_len, int n_items, int max_out_len, int up, int down, const double* taps,
                          int n_taps, int n_pre_remove, double* out, void* stream);

/* N2.  Band-limited windowed-sinc resampler: the arithmetic of resampy.resample(x, sr_orig, sr_new, filter="kaiser_best"),
 * i.e. of librosa.load(file, sr=...) / librosa.resample(res_type="kaiser_best") at the reference's ingest call sites
 * (ssr_eval/eval.py:242 preprocess -> librosa.load(file, sr=sr); ssr_eval/metrics.py:22-23 AudioMetrics.read; and in place of
 * the `sox -r` target resampling of eval.py:133-134).  The caller supplies, as DEVICE arrays, the interpolation filter
 * (interp_win, already scaled by the ratio when downsampling, and interp_delta = its forward differences), resampy's
 * `precision` as num_table = 2^precision, index_step = int(scale * num_table), scale = min(1, ratio), and the time register of
 * every output index (t / ratio, accumulated sequentially as resampy does; time_register_len entries, at least max_out_len -
 * checked).  out_len[i] = int(in_len[i] * ratio), ratio = sr_new / sr_orig.  phase_period: a of the reduced fraction
 * sr_new / sr_orig = a / b when the rates are integers (the filter phase repeats every a outputs: a wave then works on one
 * phase across 64 consecutive periods and its table reads coalesce), or 0 / 1 when unknown - a work-mapping hint only, the
 * result does not depend on it.  float64 weights, float32 running sum rounded after every tap, no fused multiply-add:
 * bit-identical to the NumPy restatement in oracle/resampy.py.  (resampy itself is absent from the reference tree and the
 * image: parity with the package is unpinned.)  n_items * ceil(max_out_len / block) must stay below 2^31 (checked). */
int ssr_resample_sinc(const float* in, const int64_t* in_off, const int32_t* in_len, const int64_t* out_off,
                      const int32_t* out_len, int n_items, int max_out_len, const double* time_register,
                      int64_t time_register_len, const double* interp_win, const double* interp_delta, int n_win,
                      int num_table, int index_step, double scale, double ratio, int phase_period, float* out, void* stream);

/* N2 (decode side).  16-bit PCM frames -> float32 mono, the values librosa.load(file, sr=None) returns for a PCM_16 file
 * (ssr_eval/eval.py:242, metrics.py:21-24: soundfile's float32 read = sample / 32768, then the channel mean): the host decoder
 * hands over the raw interleaved int16 frames (half the bytes across PCIe, no float pass on the host).
 * pcm_off[i]: element offset of item i in `pcm`; item i holds n_frames[i] * n_channels[i] samples (1 <= n_channels <= 8);
 * out_off[i]: element offset of its n_frames[i] float32 outputs. */
int ssr_pcm16_to_float(const int16_t* pcm, const int64_t* pcm_off, const int32_t* n_frames, const int32_t* n_channels,
                       int n_items, int max_frames, float* out, const int64_t* out_off, void* stream);

/* N2.  FLAC ingest: the decode half of librosa.load(file) for the format the reference's data set ships in
 * (ssr_eval/eval.py:158-169 lists ".wav" / ".flac"; eval.py:242, ssr_eval/metrics.py:21-24).  HOST entry points (paths and
 * buffers are host memory; no GPU needed): the decoder threads of the Python mirror call them with the pinned staging arena as
 * `out`, so 16-bit files cross PCIe as int16 exactly like PCM .wav files (ssr_pcm16_to_float then yields librosa's float32 values).
 * Full format support (RFC 9639: CONSTANT / VERBATIM / FIXED / LPC subframes, Rice partitions incl. escapes, wasted bits, the three
 * stereo decorrelations, 4-32 bits, CRC-8 / CRC-16 checked).  verify_md5 != 0: the MD5 of the decoded PCM is compared with the
 * signature STREAMINFO carries (when the encoder stored one) - a bit-exact self-check; mismatch = SSR_ERR_INVALID_ARG.
 * out: interleaved frames, capacity in SAMPLES (frames x channels); out == NULL only counts (and verifies); *frames_out = frames
 * per channel.  ssr_flac_decode_pcm16 takes streams of <= 16 bits per sample (SSR_ERR_UNSUPPORTED otherwise); _i32 takes any. */
int ssr_flac_info(const char* path, int* sample_rate, int* channels, int* bits, int64_t* total_samples, int* has_md5);
int ssr_flac_decode_pcm16(const char* path, int16_t* out, int64_t capacity, int verify_md5, int64_t* frames_out);
int ssr_flac_decode_i32(const char* path, int32_t* out, int64_t capacity, int verify_md5, int64_t* frames_out);

/* N4.  Position of the maximum of the full cross-correlation of two equal-length signals,
 *   z[k] = sum_l a[l] * b[l - k + n - 1],  k = 0 .. 2n-2   (scipy.signal.correlate(a, b, "full")),
 * first maximum on ties (numpy.argmax): the alignment step of SSR_Eval_Helper.mp3_encoding
 * (ssr_eval/eval.py:319  shift = argmax(correlate(decoded, x)) - len(x)).  argmax_out: int64 [n_items]. */
size_t ssr_xcorr_workspace_bytes(int n_items, int max_len);
int ssr_xcorr_argmax(const float* a, const int64_t* a_off, const float* b, const int64_t* b_off, const int32_t* len,
                     int n_items, int max_len, int64_t* argmax_out, void* workspace, size_t workspace_bytes,
                     void* stream);

/* N1.  Zero-phase IIR: scipy.signal.sosfiltfilt(sos, x) (padtype "odd", padlen 3*ntaps) for float32 x, float64
 * arithmetic and output - the arithmetic of lowpass_filter / bandpass_filter (ssr_eval/lowpass.py:54-131, called
 * from lowpass()/bandpass() :175-190,:215-254 and SSR_Eval_Helper.lowpass_butterworth/... eval.py:334-399).
 * sos: DEVICE [n_sections][6] float64 (b0 b1 b2 1 a1 a2, designed by the caller as the reference does with
 * scipy.signal.butter/cheby1/ellip/bessel(output="sos")); zi: DEVICE [n_sections][2] = scipy.signal.sosfilt_zi(sos);
 * edge = 3 * (2*n_sections + 1 - min(#(b2 == 0), #(a2 == 0))); every item needs len > edge.  n_sections <= 16.
 * y: float64, same ragged layout as x.  workspace: ssr_sosfiltfilt_workspace_bytes(total_len, n_items, edge).
 * Bit-identical to SciPy (same operation order, no fused multiply-add). */
size_t ssr_sosfiltfilt_workspace_bytes(int64_t total_len, int n_items, int edge);
int ssr_sosfiltfilt(const float* x, const int64_t* off, const int32_t* len, int n_items, int64_t total_len,
                    const double* sos, const double* zi, int n_sections, int edge, double* y, void* workspace,
                    size_t workspace_bytes, void* stream);
/* The same for a float64 signal (odd extension and filtering on the float64 values, as SciPy does). */
int ssr_sosfiltfilt_f64(const double* x, const int64_t* off, const int32_t* len, int n_items, int64_t total_len,
                    const double* sos, const double* zi, int n_sections, int edge, double* y, void* workspace,
                    size_t workspace_bytes, void* stream);

/* The same for n_designs filters over ONE batch in one launch (round 5): SSR_Eval_Helper.preprocess applies every
 * (filter type, cutoff, order) of setting_lowpass_filtering to the same waveform (ssr_eval/eval.py:243-258, three nested loops).
 * The recurrence is serial in time - a launch lasts as long as its longest utterance's chain (35-39 ns per sample step since round 6:
 * ssr_iir.h) whatever runs beside it - so 36 designs one after the other took 36 x that latency (94 % of an evaluate() pass with IIR
 * settings in round 4); side by side they take it once.  A design of S sections occupies the smallest power of two >= S lanes per utterance.
 * sos: DEVICE [n_designs][8][6] (rows >= n_sections[d] unused), zi: DEVICE [n_designs][8][2]; n_sections, edge: HOST [n_designs]
 * (n_sections <= 8, n_designs <= 48); y: [n_designs][y_stride] float64, design d's output in x's ragged layout from d * y_stride.
 * Every output bit-identical to the single-design call. */
size_t ssr_sosfiltfilt_multi_workspace_bytes(int64_t total_len, int n_items, const int32_t* edge, int n_designs);
int ssr_sosfiltfilt_multi(const float* x, const int64_t* off, const int32_t* len, int n_items, int64_t total_len,
                          const double* sos, const double* zi, const int32_t* n_sections, const int32_t* edge, int n_designs,
                          double* y, int64_t y_stride, void* workspace, size_t workspace_bytes, void* stream);

/* A12 / SURVEY 8(e).  The path's one collective: the per-speaker [metric sums ..., count] buffer summed over ranks in
 * float64 (what SSR_Eval_Helper.evaluate's mean-of-speaker-means needs from the other shards, ssr_eval/eval.py:200-216) -
 * ncclAllReduce(sum, double) over RCCL / xGMI, in place, on `stream`.  RCCL is resolved with dlopen at the first call
 * (SSR_ERR_UNSUPPORTED when it is absent); one communicator per rank / GPU:
 *   rank 0: ssr_comm_unique_id(uid) -> hand the 128 bytes to every rank (any side channel) -> all: ssr_comm_init_rank.
 * The Python mirror uses torch.distributed (the same RCCL) instead: ssr_eval_amd/dist.py. */
int ssr_comm_unique_id(void* uid128);
int ssr_comm_init_rank(const void* uid128, int n_ranks, int rank, void** comm);
int ssr_comm_destroy(void* comm);
int ssr_allreduce_sums(double* buf, int n, void* comm, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SSR_HIP_H_ */
