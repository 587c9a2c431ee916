// Shared constants of the gfx950 StofNet kernels and the packed-weight blob.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include "../../include/stofnet_amd.h"

namespace stof {

constexpr int NF = 64;            // num_features (models/stofnet.py:11)
constexpr int NF_SGB = 512;       // feat_scale * out_channels = 8 * 64 (models/stofnet.py:85,88)
constexpr int SGB_SCALE = 80;     // semi_global_scale of every shipped checkpoint
constexpr int GAP = 4;            // zero rows between waveforms in the body sweep (= conv1 padding)

// LDS geometry (floats).  Activation rows hold 64 channels (256 B) padded to 272 B so
// that the 16 lanes of a ds_read_b128 group (consecutive time rows, same channel
// offset) fall on 16 distinct 4-bank groups; weight-chunk rows hold 32 input channels
// (128 B) padded to 144 B for the same reason.
constexpr int ROWF = 68;
constexpr int WROWF = 36;
constexpr int BODY_CHUNK_F = 64 * WROWF;       // one (layer, tap, 32-channel half): [64 out][36]
constexpr int SGB_CHUNK_F = 128 * WROWF;       // one (oc block of 128, tap, half): [128 out][36]
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
    uint64_t off_ew;              // expand_conv weights as [5 taps][512 ch][64 oc]
    uint64_t off_ebias;           // [64]
    uint64_t total_floats;
    uint8_t reserved[256 - 4 * 2 - 4 * 4 - 8 * 8];
};
static_assert(sizeof(PackedHeader) == 256, "header must be 256 bytes");
constexpr uint32_t PACK_MAGIC = 0x464F5453u;

}  // namespace stof
