/*
 * mewzoom_hip.h — C ABI of libmewzoom_hip.so: the MI355X (gfx950) implementation of the
 * MewZoom upscale path of andrewdalpino/UltraZoom v0.3.0.
 *
 * The reference has no FFI / plugin interface of its own: its boundary for this path is the Python
 * API of `MewZoom` (src/ultrazoom/model.py:43-192).  Each entry point below names the reference
 * interface it stands in for; INTEGRATION.md shows the ctypes binding a maintainer would add.
 *
 * Conventions
 *   - plain C types only; every function returns 0 on success or a negative mz_status code and
 *     never throws; mz_last_error() returns a thread-local human-readable message.
 *   - all `dev` pointers are device (HBM) pointers owned by the CALLER (PyTorch-ROCm allocates
 *     them).  The library borrows them for the duration of the call and only keeps its own packed
 *     copy of the weights (allocated in mz_set_weight, freed in mz_destroy).
 *   - compute calls enqueue work on the given HIP stream and never synchronise; a handle must not
 *     be used from two threads at once.
 *   - images are NCHW, contiguous, element type = the handle's dtype, values nominally in [0, 1]
 *     (README.md:72-79 of the reference).
 */
#ifndef MEWZOOM_HIP_H
#define MEWZOOM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mz_handle mz_handle;

typedef enum mz_dtype {
    MZ_F32 = 0,  /* exact f32 MFMA (v_mfma_f32_32x32x2_f32): the 1e-3 max-abs verification mode */
    MZ_BF16 = 1, /* bf16 storage, v_mfma_f32_16x16x32_bf16 (3x3 convolutions, mixes) / 32x32x16 (image head, 1x1), f32 accumulate and epilogues */
    MZ_F16 = 2   /* fp16 storage, v_mfma_f32_16x16x32_f16 / 32x32x16, f32 accumulate and epilogues */
} mz_dtype;

typedef enum mz_status {
    MZ_OK = 0,
    MZ_ERR_INVALID_ARGUMENT = -1, /* the reference raises AssertionError for these (model.py:67-69, 218-222, 265-275, 738) */
    MZ_ERR_UNKNOWN_WEIGHT = -2,
    MZ_ERR_SHAPE_MISMATCH = -3,
    MZ_ERR_MISSING_WEIGHTS = -4,
    MZ_ERR_WORKSPACE_TOO_SMALL = -5,
    MZ_ERR_HIP = -6,
    MZ_ERR_NO_DEVICE = -7
} mz_status;

/* The 11 constructor kwargs of MewZoom.__init__ (model.py:51-64), verbatim. */
typedef struct mz_config {
    int32_t upscale_ratio;
    int32_t primary_channels;
    int32_t primary_layers;
    int32_t secondary_channels;
    int32_t secondary_layers;
    int32_t tertiary_channels;
    int32_t tertiary_layers;
    int32_t quaternary_channels;
    int32_t quaternary_layers;
    int32_t hidden_ratio;
    int32_t num_deg_features;
} mz_config;

/* ---- lifetime: replaces MewZoom.__init__ (model.py:51-92) -------------------------------- */

/* Validates the configuration exactly as the reference constructor does (same rejected values)
 * and creates a handle that will compute in `dtype`.  Touches no GPU. */
int mz_create(const mz_config* cfg, int dtype, mz_handle** out);
int mz_destroy(mz_handle* h);

/* ---- weights: replaces load_state_dict / PyTorchModelHubMixin.from_pretrained (model.py:37,43;
 *      SURVEY.md appendix B for the key names) ---------------------------------------------- */

/* Number of state_dict entries the model has, and the name / shape of entry i.
 * `shape` receives up to 4 dims; returns ndim (0 for the scalar `alpha`s). */
int mz_num_weights(const mz_handle* h);
int mz_weight_info(const mz_handle* h, int index, const char** name, int64_t shape[4]);

/* Hands one BAKED parameter (plain conv.weight / bias / alpha — no weight-norm or LoRA
 * parametrisation left, test_compare.py:32-45) to the library.  `dev_f32` points at a contiguous
 * float32 device tensor in the reference layout (OIHW for conv weights).  The library packs it
 * into MFMA-fragment order in the handle's dtype on the given stream; the caller may free its
 * tensor once the stream has passed this call. */
int mz_set_weight(mz_handle* h, const char* name, const float* dev_f32, const int64_t* shape, int ndim,
                  void* hip_stream);

/* 0 when every parameter has been set, else MZ_ERR_MISSING_WEIGHTS (message lists the first one). */
int mz_weights_complete(const mz_handle* h);

/* ---- the hot path: replaces MewZoom.forward / upscale / predict_degredation
 *      (model.py:149-164, 166-179, 181-192) -------------------------------------------------- */

/* Bytes of scratch HBM a call with this batch/shape needs.  The library processes the batch in
 * micro-batches of at most `max_images_in_flight` images (0 = library default), so memory does
 * not grow beyond that. */
int mz_workspace_bytes(const mz_handle* h, int B, int H, int W, int max_images_in_flight, size_t* bytes);

/* x        [B,3,H,W]      input, handle dtype
 * out_sr   [B,3,rH,rW]    s + head(unet(stem(x))) (model.py:162); clamped to [0,1] when clamp != 0
 *                         (model.py:177); must not be NULL
 * out_qa   [B,F] float32  degradation features z_qa (model.py:159,1026-1032), or NULL to skip the
 *                         quality head (upscale() discards it, model.py:175)
 */
int mz_forward(mz_handle* h, const void* x, void* out_sr, float* out_qa, int B, int H, int W, int clamp,
               void* workspace, size_t workspace_bytes, int max_images_in_flight, void* hip_stream);

/* The same path with uint8 images at both ends: every caller of the reference wraps
 * upscale() in `ToDtype(float32, scale=True)` and `save_image` (README.md:72-83, test_compare.py:53-57,89); here the
 * /255 happens in the stem's read and clamp -> *255 + 0.5 -> uint8 in the final store, so the two extra passes over
 * the largest tensors disappear.   x [B,3,H,W] uint8 -> out_sr [B,3,rH,rW] uint8. */
int mz_forward_u8(mz_handle* h, const uint8_t* x, uint8_t* out_sr, float* out_qa, int B, int H, int W,
                  void* workspace, size_t workspace_bytes, int max_images_in_flight, void* hip_stream);

/* ---- single operators, exported for the parity tests (tests/test_ops_gpu.py) ---------------
 * These run the SAME kernels mz_forward launches, on caller-provided tensors.
 * Activation tensors here are the library's internal layout: plane-major [B][P][H][W][16 bytes], channel count padded
 * to a multiple of 16, element type = dtype.  mz_padded_channels(c) gives that count. */
int mz_padded_channels(int c);

/* kind: 0 conv3x3 pad1 (+SiLU when silu!=0)            w: [cout,cin,3,3]    out [B,H,W,cout_p]
 *       1 conv3x3 + PixelShuffle(2) into [B,Hout,Wout,cout/4] (zero-filled beyond 2H,2W)
 *       2 PixelCrush conv2x2 stride2                   w: [cout,cin,2,2]    out [B,H/2,W/2,cout_p]
 *       3 AdaptiveResidualMix(in0, in1)                w: [c,2c,1,1], alpha out [B,H,W,c_p]
 */
int mz_op_conv(int dtype, int kind, const void* in0, const void* in1, const float* w_dev_f32, float alpha,
               void* out, int B, int H, int W, int cin, int cout, int Hout, int Wout, int silu,
               void* hip_stream);
/* stem: NCHW image -> NHWC features (model.py:239-242) */
int mz_op_stem(int dtype, const void* x, const float* w_dev_f32, const float* b_dev_f32, void* out, int B,
               int H, int W, int cout, void* hip_stream);
/* final: conv3x3 (cin -> 12) + PixelShuffle(2) + bicubic(img, R) + add [+ clamp] -> NCHW image
 * feat [B,H,W,cin_p]; img [B,3,H*2/R,W*2/R]; out [B,3,2H,2W]  (model.py:926-930, 156, 162, 177) */
int mz_op_final(int dtype, const void* feat, const void* img, const float* w_dev_f32, void* out, int B, int H,
                int W, int cin, int R, int clamp, void* hip_stream);

/* conv2 of an Encoder/DecoderBlock + AdaptiveResidualMix with the block input, ONE launch (reference model.py:773-778 second half and
 * 826-839: z = conv3x3(hid, w2); out = x + sigmoid(alpha) sigmoid(Wmix [x ; z]) (z - x)): the fused kernels of C <= 96.
 * hid [B,H,W,cin_p]; x, out [B,H,W,cout_p]; w2 [cout,cin,3,3]; wmix [cout,2 cout,1,1] (float32 on the device) */
int mz_op_conv_mix(int dtype, const void* hid, const void* x, const float* w2_dev_f32, const float* wmix_dev_f32, float alpha,
                   void* out, int B, int H, int W, int cin, int cout, void* hip_stream);

/* a17 of SURVEY.md section 8 -- NO reference counterpart: the snapshot (v0.3.0) has no ControlModule / FiLM (README.md:86-129
 * describes library version 0.2.x, whose source is absent), so this operator is "parity unpinned": it is checked against the
 * build's own CPU restatement (oracle.film_conv) only.   out = act(gamma[b, c] * conv3x3(in0, w)[b, c] + beta[b, c]),
 * act = SiLU when silu != 0.  gamma, beta: float32 [B][cout] on the device.  bf16 / fp16 only (the epilogue lives on the
 * 16x16x32 kernel); other configurations return MZ_ERR_INVALID_ARGUMENT. */
int mz_op_conv_film(int dtype, const void* in0, const float* w_dev_f32, const float* gamma_dev_f32,
                    const float* beta_dev_f32, void* out, int B, int H, int W, int cin, int cout, int silu,
                    void* hip_stream);

/* ---- introspection ------------------------------------------------------------------------ */
const char* mz_last_error(void);
const char* mz_version(void);
/* Algorithmic FLOPs (2 x conv MACs) of one forward on an H x W image. */
double mz_flops_per_image(const mz_handle* h, int H, int W);
/* Enables per-kernel HIP-event timing for bench.py's live roofline leg: after a forward, returns the
 * accumulated device time (ms) and FLOPs of all conv3x3 implicit-GEMM launches since the last reset. */
int mz_profile_enable(mz_handle* h, int on);
/* Writes one CSV row per profiled launch (layer shape, device ms, TFLOP/s) — tuning aid. */
int mz_profile_dump(mz_handle* h, const char* path);
int mz_profile_read(mz_handle* h, double* conv_ms, double* conv_flops, double* conv_launches,
                    double* other_ms, double* conv_bytes);

/* Diagnostics only: copies the in-kernel cycle-stamp buffer (16 x 64 x 8 uint64) to the host.  The buffer exists only
 * when the process was started with MZ_DEBUG_STAMPS=1 and is written only by -DMZ_STAMP builds of the kernels
 * (tools/stamp_probe*.py); returns -1 when it does not exist.  No reference counterpart. */
int mz_debug_read(unsigned long long* host_dst);

/* Kernel family of the calling thread's most recent convolution / mix launch ("conv3r", "conv3r_8x40", "conv3r_fused", "conv3t",
 * "conv3t_fused", "conv3r_ragged", "conv3s", "conv3s_fused", "conv3p", "conv3w", "conv3w_fused", "conv_kernel", "mix16", "mix16b",
 * "conv_kernel_mix"): lets a test that compares two kernels assert that it really ran both.  No reference counterpart. */
const char* mz_debug_last_kernel(void);

/* Host-only (no GPU): the tile list the role-alternating 3x3 kernels (conv3r / conv3t) walk -- B images of tiles_y x tiles_x pixel tiles
 * of th x tw pixels, ntiles N tiles, in groups of gm pixel tiles x gn N tiles (blk4 != 0: the tiles of an image in block rows of four
 * tile rows).  Entry i = out[2 i], out[2 i + 1] = {y0 | x0 << 16, image | N tile << 16}; at most `cap` entries are written.  Returns the
 * number of tiles listed (== B * tiles_y * tiles_x * ntiles), negative on bad arguments.  No reference counterpart. */
int mz_debug_tile_list(int B, int tiles_y, int tiles_x, int ntiles, int gm, int gn, int blk4, int th, int tw, unsigned int* out, int cap);

/* Hardware probe (ultrazoom_amd/csrc/mz_probe.hip; tests/test_store_hazard_gpu.py): on every CU, 16-byte buffer stores each followed --
 * `wait_states` (0, 1, 2) wait states later -- by a vector instruction that overwrites data register `dword` (0..3) of the store:
 * follower 0 v_mov_b32, 1 v_mul_f32, 2 v_cvt_pk_bf16_f32, 3 v_exp_f32, 4 v_pk_mul_f32, 5 v_mfma_f32_16x16x32_bf16;
 * form 0 = buffer_store_dwordx4 with soffset 0, 1 = ... with soffset in an SGPR, 2 = global_store_dwordx4 with a 64-bit vaddr, 3 = ... with
 * saddr.  `iters` (a multiple of 8) stores per wave,
 * `blocks` workgroups of four waves.  counts_out[0] = 16-byte entries that reached memory with anything but the register contents at
 * issue, counts_out[1..4] = per dword.  Returns 0, negative on bad arguments / HIP errors.  No reference counterpart. */
int mz_debug_store_hazard(int follower, int form, int wait_states, int dword, int iters, int blocks, unsigned int* counts_out);

#ifdef __cplusplus
}
#endif
#endif /* MEWZOOM_HIP_H */
