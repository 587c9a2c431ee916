"""CPU checks of the C-ABI library: it loads, exports every symbol the header
declares, and the pure-host weight packer lays the state_dict out as documented."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT, load_weights
from stofnet_amd import synth
from stofnet_amd import _lib
from stofnet_amd import build as sbuild


@pytest.fixture(scope='module')
def lib():
    sbuild.build(verbose=False)
    return _lib.lib()


def test_exports_match_header(lib):
    hdr = open(os.path.join(ROOT, 'include', 'stofnet_amd.h')).read()
    declared = set(re.findall(r'^(?:int|int64_t|size_t|const char\*)\s+(stof_[a-z0-9_]+)\s*\(', hdr, flags=re.M))
    assert declared, 'no declarations found'
    for sym in sorted(declared):
        assert hasattr(lib, sym), f'{sym} declared in include/stofnet_amd.h but not exported'
    assert set(_lib.EXPORTED_SYMBOLS) == declared
    assert lib.stof_abi_version() == 3
    assert _lib.status_string(0) == 'ok'
    assert 'must match the size of tensor b' in _lib.status_string(_lib.STOF_ERR_ODD_SGB_REMAINDER)


ORDER = (['conv1'] + [f'conv{i}' for i in range(2, 13)] + ['conv_last',
         'semi_global_block.contract_conv', 'semi_global_block.expand_conv'])


def pack(lib, sd, r, sgs, prec=0):
    desc = _lib.NetDesc(r, sgs, prec, 0)
    n = lib.stof_packed_weights_bytes(ctypes.byref(desc))
    arr = (ctypes.c_void_p * 30)()
    keep = []
    names = ORDER if sgs != 1 else ORDER[:13]
    for i, nm in enumerate(names):
        for j, suf in enumerate(['.weight', '.bias']):
            a = np.ascontiguousarray(sd[nm + suf], dtype=np.float32)
            keep.append(a)
            arr[2 * i + j] = a.ctypes.data
    blob = np.zeros(n, np.uint8)
    assert lib.stof_pack_weights(ctypes.byref(desc), arr, blob.ctypes.data, n) == 0
    return blob.view(np.float32), desc


def unpack_chunk(chunk, tiles, prec):
    """Invert the fragment order documented in stofnet_amd/csrc/stof_common.h:
    chunk [4 frags][tiles][64 lanes][16 bytes] -> dense [32*tiles rows][32 channels] (fp32 value)."""
    raw = chunk.reshape(4, tiles, 64, 4)
    out = np.zeros((32 * tiles, 32), np.float64)
    lane = np.arange(64)
    m, hl = lane & 31, lane >> 5
    for tile in range(tiles):
        if prec == 0:
            for q in range(4):
                for e in range(4):
                    out[32 * tile + m, 8 * q + 4 * hl + e] = raw[q, tile, :, e]
        else:
            halves = raw.view(np.float16).reshape(4, tiles, 64, 8).astype(np.float64)
            for ks in range(2):
                for e in range(8):
                    out[32 * tile + m, 16 * ks + 8 * hl + e] = halves[2 * ks, tile, :, e] + halves[2 * ks + 1, tile, :, e]
    return out


def unpack_chunk16(chunk):
    """Body chunk of the split-fp16 sweep on v_mfma_f32_16x16x32_f16 (stof_common.h): [4 frags = (M-tile m, hi | lo)][2 blocks]
    [64 lanes = (i, q)][8 fp16] -> dense [64 out-channels][32 channels]; row i of M-tile m of block b is output channel
    32 b + 8 (i >> 2) + 4 m + (i & 3), lane q holds input channels 8 q .. 8 q + 7."""
    halves = chunk.view(np.float16).reshape(4, 2, 64, 8).astype(np.float64)
    out = np.zeros((64, 32), np.float64)
    for m in range(2):
        for blk in range(2):
            for lane in range(64):
                i, q = lane & 15, lane >> 4
                o = 32 * blk + 8 * (i >> 2) + 4 * m + (i & 3)
                out[o, 8 * q:8 * q + 8] = halves[2 * m, blk, lane] + halves[2 * m + 1, blk, lane]
    return out


def unpack_chunk16_sgb(chunk):
    """SemiGlobalBlock contract chunk, 16x16x32 form: [4 frags = (N-tile nt, hi | lo)][4 wave tiles][64 lanes = (j, q)][8 fp16]
    -> dense [128 out-channels][32 channels]; lane (j, q) of N-tile nt of tile t is channel 32 t + 16 nt + j, inputs 8 q .. + 7."""
    halves = chunk.view(np.float16).reshape(4, 4, 64, 8).astype(np.float64)
    out = np.zeros((128, 32), np.float64)
    for nt in range(2):
        for tile in range(4):
            for lane in range(64):
                j, q = lane & 15, lane >> 4
                out[32 * tile + 16 * nt + j, 8 * q:8 * q + 8] = halves[2 * nt, tile, lane] + halves[2 * nt + 1, tile, lane]
    return out


@pytest.mark.parametrize('r,sgs,prec', [(4, 80, 0), (10, 80, 0), (4, 1, 0), (10, 80, 1), (20, 1, 1)])
def test_pack_layout(lib, r, sgs, prec):
    sd = synth.synth_state_dict(r, seed=1, semi_global_scale=sgs)
    f, _ = pack(lib, sd, r, sgs, prec)
    # hi + lo reproduces an fp32 weight to 2^-22 relative, or to half an fp16 subnormal step (2^-25)
    # absolute when the lo part is subnormal (weights below ~0.06)
    def close(dense, ref):
        if prec == 0:
            return np.array_equal(dense, ref)
        return bool(np.all(np.abs(dense - ref) <= np.maximum(2.0 ** -21 * np.abs(ref), 2.0 ** -25)))
    hdr = f[:64].view(np.uint32)
    assert hdr[0] == 0x464F5453 and int(hdr[2].view(np.int32)) == r and int(hdr[4].view(np.int32)) == prec
    body16 = int(hdr[5]) == 1                  # header word pad0: body chunks in 16x16x32 fragment order (split-fp16, the default)
    assert body16 == (prec == 1 and os.environ.get('STOF_BODY16', '1') != '0')
    off = 64
    c1 = f[off:off + 640].reshape(64, 10); off += 640
    assert np.array_equal(c1[:, :9], sd['conv1.weight'][:, 0, :]) and np.array_equal(c1[:, 9], sd['conv1.bias'])
    bias = f[off:off + 13 * 64].reshape(13, 64); off += 13 * 64
    for j in range(1, 12):
        assert np.array_equal(bias[j], sd[f'conv{j + 1}.bias'])
    assert np.array_equal(bias[12, :r], sd['conv_last.bias']) and not bias[12, r:].any()
    chunks = f[off:off + 160 * 2048].reshape(160, 2048); off += 160 * 2048
    c = 0
    for j in range(1, 13):
        w = sd['conv_last.weight'] if j == 12 else sd[f'conv{j + 1}.weight']
        for t in range(w.shape[2]):
            for hh in range(2):
                dense = unpack_chunk16(chunks[c]) if body16 else unpack_chunk(chunks[c], 2, prec)
                ref = np.zeros((64, 32))
                ref[:w.shape[0]] = w[:, 32 * hh:32 * hh + 32, t]
                assert close(dense, ref)
                c += 1
    assert c == 160
    if prec == 1 and r <= 16:
        # conv_last as v_mfma_f32_16x16x32_f16 A operands: [tap*2 + half][hi | lo][lane][8 fp16], lane = (out-ch, k-group)
        sec = f[off:off + 6 * 2 * 64 * 4].view(np.float16).reshape(6, 2, 64, 8).astype(np.float64); off += 6 * 2 * 64 * 4
        w = sd['conv_last.weight']
        dense = sec[:, 0] + sec[:, 1]                                          # [chunk][lane][8]
        for cc in range(6):
            t, hh = cc // 2, cc % 2
            for lane in range(64):
                o, kg = lane & 15, lane >> 4
                ref = w[o, 32 * hh + 8 * kg:32 * hh + 8 * kg + 8, t] if o < r else np.zeros(8)
                assert close(dense[cc, lane], ref)
    if sgs != 1:
        assert np.array_equal(f[off:off + 512], sd['semi_global_block.contract_conv.bias']); off += 512
        cc = f[off:off + 40 * 4096].reshape(4, 5, 2, 4096); off += 40 * 4096
        wc = sd['semi_global_block.contract_conv.weight']
        for ocb in range(4):
            for t in range(5):
                for hh in range(2):
                    dense = unpack_chunk16_sgb(cc[ocb, t, hh]) if body16 else unpack_chunk(cc[ocb, t, hh], 4, prec)
                    ref = wc[128 * ocb:128 * ocb + 128, 32 * hh:32 * hh + 32, t]
                    assert close(dense, ref)
        we = sd['semi_global_block.expand_conv.weight']                      # (64, 512, 5)
        if prec == 0:       # fp32 operand of the channel-last MFMA conv: [tap][oc][ch]
            ew = f[off:off + 5 * 512 * 64].reshape(5, 64, 512)
            assert np.array_equal(ew, we.transpose(2, 0, 1))
        else:               # f16x3: [tap][oc][ch/64][64 hi | 64 lo] fp16
            hl = f[off:off + 5 * 512 * 64].view(np.float16).reshape(5, 64, 8, 2, 64).astype(np.float64)
            dense = (hl[:, :, :, 0] + hl[:, :, :, 1]).reshape(5, 64, 512)
            assert close(dense, we.transpose(2, 0, 1))
        off += 5 * 512 * 64
        assert np.array_equal(f[off:off + 64], sd['semi_global_block.expand_conv.bias']); off += 64
    assert off == f.size


def test_pack_rejects_bad_args(lib):
    desc = _lib.NetDesc(0, 80, 0, 0)
    assert lib.stof_packed_weights_bytes(ctypes.byref(desc)) == 0
    desc = _lib.NetDesc(4, 40, 0, 0)
    assert lib.stof_packed_weights_bytes(ctypes.byref(desc)) == 0
    desc = _lib.NetDesc(4, 80, 0, 0)
    arr = (ctypes.c_void_p * 30)()
    blob = np.zeros(16, np.uint8)
    assert lib.stof_pack_weights(ctypes.byref(desc), arr, blob.ctypes.data, 16) == _lib.STOF_ERR_BAD_ARG


def test_module_state_dict_names_match_reference():
    import torch
    from stofnet_amd import StofNet
    m = StofNet(upsample_factor=4)
    sd = load_weights('different-armadillo')
    assert set(m.state_dict().keys()) == set(sd.keys())
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    assert sum(p.numel() for p in m.parameters()) == 645764
    m2 = StofNet(upsample_factor=4, semi_global_scale=1)
    m2.load_state_dict({k: torch.from_numpy(v) for k, v in load_weights('clean-serenity').items()}, strict=True)
    assert sum(p.numel() for p in m2.parameters()) == 317508
    with pytest.raises(RuntimeError):       # strict mismatch, as in the reference
        m.load_state_dict({k: torch.from_numpy(v) for k, v in load_weights('clean-serenity').items()}, strict=True)


def test_no_cpu_fallback():
    import torch
    from stofnet_amd import StofNet, mask2coords, SampleShuffle1D
    with pytest.raises(RuntimeError, match='ROCm device only'):
        StofNet()(torch.zeros(1, 1, 160))
    with pytest.raises(RuntimeError, match='ROCm device only'):
        mask2coords(torch.zeros(1, 1, 16), 20)
    with pytest.raises(RuntimeError, match='ROCm device only'):
        SampleShuffle1D(4)(torch.zeros(1, 4, 16))
