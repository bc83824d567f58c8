// Internal interface between the host runtime (mz_host.cpp) and the gfx950 kernels (mz_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace mz {

enum DType : int { DT_F32 = 0, DT_BF16 = 1, DT_F16 = 2 };

inline int dtype_size(int dt) { return dt == DT_F32 ? 4 : 2; }
// channels per 32-byte K-chunk
inline int chunk_channels(int dt) { return dt == DT_F32 ? 8 : 16; }
inline int pad16(int c) { return (c + 15) / 16 * 16; }

// ---- implicit-GEMM convolution ----------------------------------------------------------------
// A workgroup = 4 waves = 256 output pixels x BN = 32*NT output channels.
//   MODE_CONV3 : 3x3, pad 1, stride 1; pixel tile = 8 rows x 32 columns of one image.
//   MODE_GEMM1 : 1x1 over a gathered K axis; pixel tile = 256 consecutive output pixels
//                (row-major over B*Ho*Wo).  Sources: SRC_CONCAT ([in0 ; in1] along channels, the
//                AdaptiveResidualMix gate) or SRC_CRUSH (2x2 stride-2 patch of in0 = PixelCrush).
//   MODE_C3W16 / MODE_C3W8 : 3x3 on a 512-pixel tile (16 x 32 or 8 x 64) with 8 compute waves + 1 loader wave
enum ConvMode : int { MODE_CONV3 = 0, MODE_GEMM1 = 1, MODE_C3W16 = 2, MODE_C3W8 = 3 };
enum SrcKind : int { SRC_PLAIN = 0, SRC_CONCAT = 1, SRC_CRUSH = 2, SRC_MIXF = 3 /* pack-only: fused gate weights */ };
enum Epilogue : int {
    EPI_STORE = 0,  // NHWC store (optional SiLU)
    EPI_D2S = 1,    // PixelShuffle(2) store into a [B,Hout,Wout,cq] tensor
    EPI_MIX = 2,    // out = x + s*sigmoid(acc)*(z - x), x = in0, z = in1
    EPI_FINAL = 3,  // PixelShuffle(2) + bicubic(img) + add (+clamp) -> NCHW image
    EPI_FUSEDMIX = 4,  // conv2 of a block + AdaptiveResidualMix with the block input (in1) in one kernel
};

struct ConvArgs {
    const void* in0;
    const void* in1;
    const void* wpk;   // packed weights [ntile][kchunk][tap][nt][64 lanes][16 B]
    void* out;
    const void* zero;  // >= 16 bytes of zeros in HBM (source for halo / out-of-range pixels)
    const void* img;   // EPI_FINAL: NCHW low-resolution image
    const void* wmix;  // EPI_FUSEDMIX: gate weights packed with SRC_MIXF
    int mix_pieces;    // EPI_FUSEDMIX: KiB of gate weights
    int x_via_lds;     // EPI_FUSEDMIX: the block input's fragments are prefetched into LDS (room in ring slots 1-2)
    int B, H, W;       // input grid
    int Ho, Wo;        // output pixel grid of the GEMM (== H,W for CONV3; H/2,W/2 for CRUSH)
    int p0, p1;        // planes (16-byte channel groups) of in0 / in1: padded_channels * sizeof / 16
    int p_out;         // planes of the output tensor (D2S: of the shuffled target)
    int nchunks;       // K chunks in total (GEMM1: padded to a whole number of stages with zero weights)
    int nchunks_real;  // chunks that exist in the sources
    int nchunks0;      // CONCAT: chunks that come from in0; CRUSH: chunks per tap
    int src;           // SrcKind
    int ntiles;        // N tiles (of 32*NT channels)
    int mtiles;        // pixel tiles
    int tiles_x, tiles_y;  // CONV3: tiles per image
    int gm, gn;        // tile-group shape of the workgroup -> tile walk (map_tile)
    int grid;          // workgroups launched: whole groups, >= mtiles * ntiles
    int groups_m;      // ceil(mtiles / gm)
    float inv_gsz, inv_groups_m, inv_gn, inv_tpi, inv_tiles_x;  // 1.0f / divisor for fdiv() (all dividends < 2^24)
    int blk4;          // conv3s, the tile lists of conv3r / conv3t: tiles of an image are walked in block rows of four tile rows (tile_rc(), mz_device.h)
    float inv_bsz;     // 1.0f / (4 * tiles_x)
    // conv3r_kernel: the same divisors as floor(2^32 / d) for sdiv() (scalar-unit division, mz_device.h)
    uint32_t mg_gsz, mg_groups_m, mg_gn, mg_tpi, mg_tiles_x, mg_bsz;
    int epi;
    int silu;
    int cp_out;        // STORE/MIX: padded channels of out; D2S: channels per output pixel (cq_p)
    int Hout, Wout;    // D2S / FINAL target grid
    float mix_scale;   // sigmoid(alpha)
    float inv_mix_scale;  // 1 / sigmoid(alpha) = 1 + exp(-alpha): blend_() folds the scale into the reciprocal of the gate's sigmoid
    int R;             // FINAL: total upscale ratio of img -> out
    int Hi, Wi;        // FINAL: img size
    int clamp;
    int io_u8;         // EPI_FINAL: img and out are uint8 images (scaled by 1/255 on read, x255 + 0.5 on write)
    int persist;       // > 0: launch the persistent wide 3x3 kernel with this many workgroups (a multiple of 8)
    int s16;           // persistent launches of 16-bit types: use the 16x16x32-MFMA kernel (needs wpk16)
    const void* wpk16; // weights packed for it: [ntile][32-channel chunk][tap][2*nt][64 lanes][16 B]
    int nchunks16;     // 32-channel chunks
    const void* tile_tab;  // conv3r / conv3t: the launch's tiles in walk order, uint2 {y0 | x0 << 16, image | N tile << 16} each, padded with
                           // 4 * persist / 8 + 8 entries; a.grid = its length (mz_host.cpp: tile_table())
    int ragged_planes; // conv3r_kernel<.., RAG>: 16-byte planes that exist in the LAST 32-channel chunk (1..3; Cin = 48: 2): the pieces of the
                       // others are issued with every lane out of range (zeros into LDS); 0 = every chunk has its four planes
    int geo;           // conv3r_kernel: pixel-tile geometry, 0 = 8 x 48 (six pixel fragments per wave), 1 = 8 x 40 (five)
    const void* wmix16; // EPI_FUSEDMIX on the 16x16x32 kernel: gate weights packed [2*nt K-steps][2*nt][64 lanes][16 B]
    const float* film_gamma;  // EPI_STORE on conv3s_kernel only: per-image per-channel affine gamma * y + beta ahead of the SiLU
    const float* film_beta;   //   (float [B][cp_out], pad channels zero); nullptr = off.  No reference counterpart.
    int use_glds;      // stage through global_load_lds (1) or through registers (0)
    unsigned long long* dbg;  // -DMZ_DIAG builds (mz_diag.h): cycle counters of one workgroup of conv3r_kernel
};

size_t conv_lds_bytes(int mode, int nt);
// picks NT (32*NT output channels per workgroup) for a logical padded N
int choose_nt(int n_padded);
int gemm1_chunks_per_stage();
hipError_t launch_conv(int dtype, int mode, int nt, const ConvArgs& a, hipStream_t s);
// AdaptiveResidualMix for C = k * 192 on the 16x16x32 MFMA (16-bit types): a.wpk16 / a.nchunks16 = K steps over [x ; z]
hipError_t launch_mix16(int dtype, const ConvArgs& a, hipStream_t s);
// C = 192 without the second read of x and z: a.wpk16 = gate weights packed with PackArgs::frag16 = 3 (accumulator rows in B-operand
// order); the gate matrix stays in LDS, every wave walks its own 32-pixel units
hipError_t launch_mix16b(int dtype, const ConvArgs& a, hipStream_t s, int workgroups);  // persistent: at most `workgroups` (one per CU)
hipError_t init_kernels();  // raises the dynamic-LDS limits (per device)
// conv3r_kernel (mz_conv3r.h): 3x3 convolution, 16-bit types, 96-channel N tiles (NT = 3), 8 x 48 / 8 x 40 pixel tiles, a.persist workgroups of
// 512 threads; a.wpk16 / a.nchunks16 as for conv3s_kernel.  The two waves of every SIMD alternate between the compute
// and the loader + epilogue role from tile to tile.  >= 3 chunks of 32 channels (odd counts included; a.ragged_planes != 0: exactly
// two, the second with a.ragged_planes real planes -- Cin = 48 --, EPI_STORE + SiLU only); EPI_STORE / EPI_D2S
// (32-bit store offsets: 12 planes of the output, resp. one whole D2S target image, must stay below 4 GiB); EPI_FUSEDMIX: >= 6
// chunks, a.wmix16 = gate weights packed with PackArgs::frag16 = 2, a.in1 / a.p1 = the block input.
hipError_t launch_conv3r(int dtype, const ConvArgs& a, hipStream_t s);
// conv3t_kernel (mz_conv3t.h): conv3r's role-alternating structure for ONE N tile of <= 48 channels (three 16-channel fragments x twelve
// pixel fragments per wave, 12 x 64 pixel tiles).  a.wpk16 = weights packed with PackArgs::nfr = 3; a.nchunks16 = 3 or >= 6; EPI_STORE
// (plain / SiLU) or EPI_FUSEDMIX (a.wmix16 = gate weights packed with PackArgs::frag16 = 4, a.in1 / a.p1 = the block input).
hipError_t launch_conv3t(int dtype, const ConvArgs& a, hipStream_t s);

// ---- weight packing ---------------------------------------------------------------------------
enum OutMap : int { OUT_PLAIN = 0, OUT_D2S = 1, OUT_FINAL = 2 };
struct PackArgs {
    const float* w;    // [cout][cin][kh][kw] float32
    void* dst;
    int dtype;
    int cout, cin, kh, kw;
    int taps;          // packed taps (9 for conv3, 1 for GEMM1)
    int nt, ntiles, nchunks;
    int out_map;       // OutMap
    int cq, cq_p;      // D2S: real / padded channels per output pixel
    int in_map;        // SrcKind
    int frag16;        // 1: fragments of the 16x16x32 MFMA (16 channels x 32 K; 16-bit types); nchunks counts 32-channel chunks.  2 (SRC_MIXF): ... with
                       // the x half of the gate weights in accumulator-row order as well (conv3r_kernel's fused variant)
                       // 3 (SRC_CONCAT, nt = 6): mix16b_kernel's row and K-step order
                       // 4 (SRC_MIXF, nfr = 3, nchunks = 3): conv3t_kernel's gate: K step s = fragments 2 s, 2 s + 1 of [x0 x1 x2 z0 z1 z2]
    int c0, cp0, c1;   // CONCAT: real/padded channels of in0, real channels of in1;  PLAIN/CRUSH: c0 = cin, cp0 = padded cin
    int nfr;           // frag16 packings: 16-channel fragments per tap and N tile; 0 = 2 * nt (conv3t_kernel: 3)
};
size_t packed_bytes(int taps, int nt, int ntiles, int nchunks);
size_t conv16_lds_bytes(int mode, int nt, bool fuse);
hipError_t launch_pack(const PackArgs& a, hipStream_t s);

// ---- small kernels ----------------------------------------------------------------------------
// stem weights: float [cp][4] = {w0, w1, w2, bias}
hipError_t launch_pack_stem(const float* w, const float* b, float* dst, int c, int cp, hipStream_t s);
hipError_t launch_stem(int dtype, const void* x, const float* w4, void* out, int B, int H, int W, int cp,
                       hipStream_t s, int u8 = 0);
// zero rows >= Hv and columns >= Wv of an NHWC tensor [B,Hout,Wout,cp]
hipError_t launch_zero_border(int dtype, void* t, int B, int Hout, int Wout, int cp, int Hv, int Wv,
                              hipStream_t s);
// qa[b][f] = bias[f] + mean_p feat[b][p][f]
hipError_t launch_qa_reduce(int dtype, const void* feat, const float* bias, float* qa, int B, int P, int cp,
                            int F, hipStream_t s);
// layout helpers used by the operator-level tests
hipError_t launch_fill_zero(void* p, size_t bytes, hipStream_t s);

}  // namespace mz
