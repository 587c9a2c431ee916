// Shared constants of the gfx950 StofNet kernels and the packed-weight blob.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include "../../include/stofnet_amd.h"

namespace stof {

constexpr int NF = 64;            // num_features (models/stofnet.py:11)
constexpr int NF_SGB = 512;       // feat_scale * out_channels = 8 * 64 (models/stofnet.py:85,88)
constexpr int SGB_SCALE = 80;     // semi_global_scale of every shipped checkpoint
constexpr int GAP = 4;            // zero rows between waveforms in the body sweep (= conv1 padding)

// LDS geometry.  An activation row is 256 B of payload padded to 272 B (68 floats) so that the
// 16 lanes of a ds_read_b128 group (consecutive time rows, same channel offset) fall on 16
// distinct 4-bank groups.  Payload per precision mode:
//   fp32  : 64 channels x fp32
//   f16x3 : 64 channels x fp16 "hi" | 64 channels x fp16 "lo"   (x = hi + lo to ~2^-22)
constexpr int ROWF = 68;

// Weights are stored in MFMA-fragment order so that a wave fetches one operand fragment with a
// single fully coalesced 1-KiB load (lane l gets bytes [16 l, 16 l + 16)) straight into the
// registers the MFMA reads -- no LDS staging.  A "chunk" is one (layer, tap, 32-input-channel
// half); it holds FRAGS_PER_CHUNK fragments for each 32-wide output-channel tile:
//   fp32  : fragment q (0..3)        : lane (m = l&31, h = l>>5) -> W[m][32*half + 8q + 4h + 0..3]   (4 x fp32)
//   f16x3 : fragment 2*ks + part     : lane (m, h) -> part(W[m][32*half + 16ks + 8h + 0..7])         (8 x fp16)
//           part 0 = hi, 1 = lo
//   f16x3 body, 16x16x32 (r3, the default; STOF_BODY16=0 in the environment of BOTH the packing and the launching process selects
//           the 32x32x16 form above for A/B runs): the body sweep multiplies on v_mfma_f32_16x16x32_f16 -- on this part the
//           16x16x32 shape sustains a ~12 % higher clock than 32x32x16 under the power cap at equal cycles per flop
//           (tools/micro/mfma_shape_lds.hip, profiles/r03_mfma_shape_lds.jsonl).  A chunk is the whole K = 32 of one MFMA:
//           fragment 2*m + part : lane (i = l&15, q = l>>4) -> part(W[body16_out_channel(block, m, i)][32*half + 8q + 0..7])
//           The output-channel order inside a 32-channel block is permuted so that the accumulator lane (column = time row,
//           q) of M-tiles m = 0, 1 holds the 8 CONSECUTIVE channels 8q .. 8q+7: one 16-byte LDS store per lane and row.
//   f16x3 SemiGlobalBlock chunks, 16x16x32 (with the body): the weights are the B operand (N = output channel), time is on M:
//           fragment 2*nt + part of wave tile `tile` : lane (j = l&15, q = l>>4) -> part(W[32 tile + 16 nt + j][32*half + 8q + 0..7])
constexpr int FRAGS_PER_CHUNK = 4;
constexpr int FRAG_F = 256;                                   // floats per fragment (1 KiB)
constexpr int BODY_CHUNK_F = FRAGS_PER_CHUNK * 2 * FRAG_F;    // 2 output tiles (64 channels)
constexpr int SGB_CHUNK_F = FRAGS_PER_CHUNK * 4 * FRAG_F;     // 4 output tiles (128-channel block)
constexpr int BODY_CHUNKS_K7 = 14;             // 7 taps x 2 halves
constexpr int BODY_CHUNKS_LAST = 6;            // 3 taps x 2 halves
constexpr int BODY_NCHUNK = 11 * BODY_CHUNKS_K7 + BODY_CHUNKS_LAST;   // 160 per sweep step
constexpr int SGB_NCHUNK = 4 * 5 * 2;          // 4 oc blocks x 5 taps x 2 halves

// Packed blob: a 256-byte header followed by float sections (offsets in floats
// from the start of the blob).
struct PackedHeader {
    uint32_t magic;               // 'STOF'
    uint32_t abi;
    int32_t r, sgs, precision, pad0;
    uint64_t off_c1;              // [64][10]: conv1 taps 0..8 + bias per channel
    uint64_t off_bias;            // [13][64]: row j = bias of sweep layer j (1..11 conv2..12, 12 conv_last)
    uint64_t off_body;            // BODY_NCHUNK chunks of BODY_CHUNK_F floats
    uint64_t off_cbias;           // [512] contract_conv bias
    uint64_t off_cchunks;         // SGB_NCHUNK chunks of SGB_CHUNK_F floats
    uint64_t off_ew;              // expand_conv operand of conv_cl_kernel: fp32 [5 taps][64 oc][512 ch];
                                  // f16x3 [5][64][8 blocks][64 hi | 64 lo] fp16 (same float count)
    uint64_t off_ebias;           // [64]
    uint64_t total_floats;
    uint64_t off_last16;          // f16x3 and r <= 16 only (else 0): conv_last as 16x16x32 MFMA operands,
                                  // [6 chunks (tap, 32-ch half)][hi | lo][64 lanes][8 fp16] = LAST16_F floats
    uint8_t reserved[256 - 4 * 2 - 4 * 4 - 8 * 9];
};
static_assert(sizeof(PackedHeader) == 256, "header must be 256 bytes");
constexpr uint32_t PACK_MAGIC = 0x464F5453u;

// 16x16x32 body: output channel held by row i (0..15) of M-tile m (0, 1) of the 32-channel block `block`:
// accumulator element e of lane (col, q) is row 4q + e, i.e. channel 32 block + 8q + 4m + e.
constexpr int body16_out_channel(int block, int m, int i) { return 32 * block + 8 * (i >> 2) + 4 * m + (i & 3); }
inline bool body16_enabled() {
    const char* e = getenv("STOF_BODY16");
    return e == nullptr || e[0] != '0';
}
// r4: the 16x16x32 body sweeps its k7 layers two-pass tile-major (body_p2.h); STOF_BODY_P2=0 selects the r3 chunk-major kernel
// on the same packed blob (A/B runs)
inline bool body_p2_enabled() {
    const char* e = getenv("STOF_BODY_P2");
    return e == nullptr || e[0] != '0';
}
constexpr int LAST16_F = 6 * 2 * 64 * 4;      // floats of the off_last16 section

}  // namespace stof
