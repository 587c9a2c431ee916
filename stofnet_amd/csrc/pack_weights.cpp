// Host-side repack of the reference's state_dict (models/stofnet.py:23-31,88,94) into
// the streaming layout the gfx950 kernels read.  Pure CPU code: no HIP calls.
#include <stdlib.h>
#include <string.h>
#include "stof_common.h"

using namespace stof;

static void layout(const stof_net_desc* d, PackedHeader* h) {
    memset(h, 0, sizeof(*h));
    h->magic = PACK_MAGIC;
    h->abi = STOF_ABI_VERSION;
    h->r = d->upsample_factor;
    h->sgs = d->semi_global_scale;
    h->precision = d->precision;
    h->pad0 = (d->precision == STOF_PREC_F16X3 && body16_enabled()) ? 1 : 0;     // body chunk layout: 1 = 16x16x32 fragments
    uint64_t off = sizeof(PackedHeader) / sizeof(float);
    h->off_c1 = off;      off += 64 * 10;
    h->off_bias = off;    off += 13 * 64;
    h->off_body = off;    off += (uint64_t)BODY_NCHUNK * BODY_CHUNK_F;
    if (d->precision == STOF_PREC_F16X3 && d->upsample_factor <= 16) { h->off_last16 = off; off += LAST16_F; }
    if (d->semi_global_scale != 1) {
        h->off_cbias = off;   off += NF_SGB;
        h->off_cchunks = off; off += (uint64_t)SGB_NCHUNK * SGB_CHUNK_F;
        h->off_ew = off;      off += 5ull * NF_SGB * NF;
        h->off_ebias = off;   off += NF;
    }
    h->total_floats = off;
}

static int check_desc(const stof_net_desc* d) {
    if (!d) return STOF_ERR_BAD_ARG;
    if (d->upsample_factor < 1 || d->upsample_factor > 64) return STOF_ERR_UNSUPPORTED;
    if (d->semi_global_scale != 1 && d->semi_global_scale != SGB_SCALE) return STOF_ERR_UNSUPPORTED;
    if (d->precision != STOF_PREC_FP32 && d->precision != STOF_PREC_F16X3) return STOF_ERR_UNSUPPORTED;
    if (d->seg_policy < 0 || d->seg_policy > 6) return STOF_ERR_UNSUPPORTED;
    return STOF_OK;
}

// One chunk = FRAGS_PER_CHUNK fragments x `tiles` output tiles of 32 channels, fragment-major:
// float offset ((frag * tiles + tile) * 64 + lane) * 4.  Rows >= co are zero (conv_last padding).
static void pack_chunk(float* dst, int tiles, int row0, const float* w, int co, int ci, int K, int tap, int hh,
                       int precision) {
    for (int frag = 0; frag < FRAGS_PER_CHUNK; ++frag)
        for (int tile = 0; tile < tiles; ++tile)
            for (int lane = 0; lane < 64; ++lane) {
                const int m = row0 + 32 * tile + (lane & 31), hl = lane >> 5;
                float* out = dst + ((size_t)(frag * tiles + tile) * 64 + lane) * 4;
                if (precision == STOF_PREC_FP32) {
                    for (int e = 0; e < 4; ++e) {
                        const int c = 32 * hh + 8 * frag + 4 * hl + e;
                        out[e] = m < co ? w[((size_t)m * ci + c) * K + tap] : 0.f;
                    }
                } else {
                    const int ks = frag >> 1, part = frag & 1;
                    _Float16* oh = reinterpret_cast<_Float16*>(out);
                    for (int e = 0; e < 8; ++e) {
                        const int c = 32 * hh + 16 * ks + 8 * hl + e;
                        const float v = m < co ? w[((size_t)m * ci + c) * K + tap] : 0.f;
                        const _Float16 hi = (_Float16)v;
                        oh[e] = part == 0 ? hi : (_Float16)(v - (float)hi);
                    }
                }
            }
}

// Body chunk of the split-fp16 sweep on v_mfma_f32_16x16x32_f16 (stof_common.h, "f16x3 body, 16x16x32"): 2 output blocks
// of 32 channels x 4 fragments (M-tile m = 0, 1 x hi | lo), float offset ((frag * 2 + block) * 64 + lane) * 4.
static void pack_chunk16(float* dst, const float* w, int co, int ci, int K, int tap, int hh) {
    for (int frag = 0; frag < FRAGS_PER_CHUNK; ++frag)
        for (int blk = 0; blk < 2; ++blk)
            for (int lane = 0; lane < 64; ++lane) {
                const int m = frag >> 1, part = frag & 1, i = lane & 15, q = lane >> 4;
                const int o = body16_out_channel(blk, m, i);
                _Float16* oh = reinterpret_cast<_Float16*>(dst + ((size_t)(frag * 2 + blk) * 64 + lane) * 4);
                for (int e = 0; e < 8; ++e) {
                    const int c = 32 * hh + 8 * q + e;
                    const float v = o < co ? w[((size_t)o * ci + c) * K + tap] : 0.f;
                    const _Float16 hi = (_Float16)v;
                    oh[e] = part == 0 ? hi : (_Float16)(v - (float)hi);
                }
            }
}

// SemiGlobalBlock contract chunk for the 16x16x32 form: the weights are the MFMA's B operand (N = output channel); per wave
// tile of 32 channels 4 fragments (N-tile nt = 0, 1 x hi | lo), float offset ((frag * 4 + tile) * 64 + lane) * 4; lane
// (j = lane & 15, q = lane >> 4) holds W[row0 + 32 tile + 16 nt + j][32 hh + 8 q + 0..7][tap].
static void pack_chunk16_sgb(float* dst, int row0, const float* w, int ci, int K, int tap, int hh) {
    for (int frag = 0; frag < FRAGS_PER_CHUNK; ++frag)
        for (int tile = 0; tile < 4; ++tile)
            for (int lane = 0; lane < 64; ++lane) {
                const int nt = frag >> 1, part = frag & 1, j = lane & 15, q = lane >> 4;
                const int o = row0 + 32 * tile + 16 * nt + j;
                _Float16* oh = reinterpret_cast<_Float16*>(dst + ((size_t)(frag * 4 + tile) * 64 + lane) * 4);
                for (int e = 0; e < 8; ++e) {
                    const float v = w[((size_t)o * ci + 32 * hh + 8 * q + e) * K + tap];
                    const _Float16 hi = (_Float16)v;
                    oh[e] = part == 0 ? hi : (_Float16)(v - (float)hi);
                }
            }
}

extern "C" size_t stof_packed_weights_bytes(const stof_net_desc* desc) {
    if (check_desc(desc) != STOF_OK) return 0;
    PackedHeader h;
    layout(desc, &h);
    return (size_t)h.total_floats * sizeof(float);
}

extern "C" int stof_pack_weights(const stof_net_desc* desc, const float* const* params,
                                 void* packed_host, size_t packed_bytes) {
    int st = check_desc(desc);
    if (st != STOF_OK) return st;
    if (!params || !packed_host) return STOF_ERR_BAD_ARG;
    const int nparams = desc->semi_global_scale != 1 ? STOF_NUM_PARAMS : 26;
    for (int i = 0; i < nparams; ++i)
        if (!params[i]) return STOF_ERR_BAD_ARG;
    PackedHeader h;
    layout(desc, &h);
    if (packed_bytes < (size_t)h.total_floats * sizeof(float)) return STOF_ERR_WORKSPACE;
    memset(packed_host, 0, (size_t)h.total_floats * sizeof(float));
    memcpy(packed_host, &h, sizeof(h));
    float* base = static_cast<float*>(packed_host);
    const int r = desc->upsample_factor;

    // conv1: [ch][tap 0..8, bias]
    for (int c = 0; c < NF; ++c) {
        for (int t = 0; t < 9; ++t) base[h.off_c1 + c * 10 + t] = params[0][c * 9 + t];
        base[h.off_c1 + c * 10 + 9] = params[1][c];
    }
    // biases of sweep layers 1..11 (conv2..conv12) and 12 (conv_last, zero padded)
    for (int j = 1; j <= 11; ++j)
        for (int c = 0; c < NF; ++c) base[h.off_bias + j * 64 + c] = params[3 + 2 * (j - 1)][c];
    for (int c = 0; c < r; ++c) base[h.off_bias + 12 * 64 + c] = params[25][c];

    // body chunks in streaming order: (layer, tap, 32-channel half) -> fragments for 2 output tiles
    float* ck = base + h.off_body;
    for (int j = 1; j <= 12; ++j) {
        const bool last = (j == 12);
        const float* w = last ? params[24] : params[2 + 2 * (j - 1)];   // (co, 64, K)
        const int K = last ? 3 : 7;
        const int co = last ? r : NF;
        for (int t = 0; t < K; ++t)
            for (int hh = 0; hh < 2; ++hh) {
                if (desc->precision == STOF_PREC_F16X3 && body16_enabled()) pack_chunk16(ck, w, co, NF, K, t, hh);
                else pack_chunk(ck, 2, 0, w, co, NF, K, t, hh, desc->precision);
                ck += BODY_CHUNK_F;
            }
    }
    if (h.off_last16) {
        // conv_last (r <= 16 output channels) as the A operand of v_mfma_f32_16x16x32_f16: lane (i = lane & 15,
        // kg = lane >> 4) holds output channel i, input channels 32 hh + 8 kg .. + 7 of tap t; rows >= r are zero
        const float* w = params[24];                                    // (r, 64, 3)
        _Float16* dst = reinterpret_cast<_Float16*>(base + h.off_last16);
        for (int t = 0; t < 3; ++t)
            for (int hh = 0; hh < 2; ++hh)
                for (int lane = 0; lane < 64; ++lane)
                    for (int e = 0; e < 8; ++e) {
                        const int o = lane & 15, ch = 32 * hh + 8 * (lane >> 4) + e;
                        const float v = o < r ? w[((size_t)o * NF + ch) * 3 + t] : 0.f;
                        const _Float16 hi = (_Float16)v;
                        const size_t cc = (size_t)t * 2 + hh;
                        dst[((cc * 2 + 0) * 64 + lane) * 8 + e] = hi;
                        dst[((cc * 2 + 1) * 64 + lane) * 8 + e] = (_Float16)(v - (float)hi);
                    }
    }
    if (desc->semi_global_scale != 1) {
        for (int c = 0; c < NF_SGB; ++c) base[h.off_cbias + c] = params[27][c];
        const float* wc = params[26];                                   // (512, 64, 5)
        float* cc = base + h.off_cchunks;
        for (int ocb = 0; ocb < 4; ++ocb)
            for (int t = 0; t < 5; ++t)
                for (int hh = 0; hh < 2; ++hh) {
                    if (desc->precision == STOF_PREC_F16X3 && body16_enabled()) pack_chunk16_sgb(cc, 128 * ocb, wc, NF, 5, t, hh);
                    else pack_chunk(cc, 4, 128 * ocb, wc, NF_SGB, NF, 5, t, hh, desc->precision);
                    cc += SGB_CHUNK_F;
                }
        const float* we = params[28];                                   // (64, 512, 5)
        // operand image of the channel-last MFMA conv (train.hip conv_cl_kernel): tap-major rows per output channel
        for (int t = 0; t < 5; ++t)
            for (int o = 0; o < NF; ++o)
                for (int c = 0; c < NF_SGB; ++c) {
                    const float v = we[(o * NF_SGB + c) * 5 + t];
                    if (desc->precision == STOF_PREC_FP32) {
                        base[h.off_ew + ((uint64_t)t * NF + o) * NF_SGB + c] = v;
                    } else {                                            // [t][o][c/64][64 hi | 64 lo] fp16
                        _Float16* row = reinterpret_cast<_Float16*>(base + h.off_ew) +
                                        (((uint64_t)t * NF + o) * (NF_SGB / 64) + c / 64) * 128;
                        const _Float16 hi = (_Float16)v;
                        row[c % 64] = hi;
                        row[64 + c % 64] = (_Float16)(v - (float)hi);
                    }
                }
        for (int o = 0; o < NF; ++o) base[h.off_ebias + o] = params[29][o];
    }
    return STOF_OK;
}

extern "C" const char* stof_status_string(int status) {
    switch (status) {
        case STOF_OK: return "ok";
        case STOF_ERR_BAD_ARG: return "bad argument (null pointer or negative size)";
        case STOF_ERR_ODD_SGB_REMAINDER:
            return "The size of tensor a must match the size of tensor b at non-singleton dimension 2 "
                   "(SemiGlobalBlock: L - 80*floor(L/80) is odd)";
        case STOF_ERR_UNSUPPORTED: return "unsupported shape or mode";
        case STOF_ERR_WORKSPACE: return "workspace or packed-weight buffer too small";
        case STOF_ERR_HIP: return "HIP runtime error";
        case STOF_ERR_CHANNELS: return "input channels not divisible by upsample_factor";
        case STOF_ERR_POOL_EMPTY: return "max_pool1d() Invalid computed output size: 0";
        default: return "unknown status";
    }
}

extern "C" int stof_abi_version(void) { return STOF_ABI_VERSION; }
