"""HIP execution engine for the per-plane convolutions of the MPI producer network.

The S-times-batched parts of the network - the feature-mask UNet (reference model/CPN/unet.py:18-69) and the gated
decoder (model/CPN/decoder.py:74-174) - are run as 20 launches of `mpf_conv3x3_f16` (mpiflow_amd/csrc/mpf_conv.hip):
every launch is one 3x3 convolution over all S planes whose loader synthesises the layer's input (the expand / cat /
upsample / reflection-pad tensors of the reference are never written) and whose epilogue applies BatchNorm, activation and
the gate.  The single-image parts (ResNet-18 encoder model/CPN/encoder.py:20-101, the 1x1 / 3x3 bottleneck at 1/32 .. 1/128 resolution
model/CPN/decoder.py:131-138) are 24 launches of `mpf_conv2d_f32` (mpiflow_amd/csrc/mpf_encoder.hip, fp32 MFMA) + 3 max-pools on a side
stream underneath the feature-mask network: 1 % of the work, no MIOpen / ATen kernel left in the forward.

Precision: fp16 storage and MFMA inputs, fp32 accumulation and epilogue - the reference's own GPU configuration
(`.half()`, gen_3dphoto_dynamic_v2.py:46,59,82-84).  `MPIPredictor.forward` (fp32 torch) remains the bit-exact mirror of
the reference model; this engine is the fast path of `gen_3dphoto_dynamic.py --model-engine hip`.

This module only PACKS parameters (host side, torch CPU) and sequences launches; all arithmetic is in the HIP kernels.
"""
import ctypes
import os

import torch

from .. import _lib

LD_FMN_INPUT, LD_DIRECT, LD_BILINEAR_CAT, LD_NEAREST_PLANE, LD_FMN_SYNTH, LD_BILINEAR_SYNTH, LD_NEAREST_PHASE = 0, 1, 2, 3, 4, 5, 6
EP_AFFINE_RELU, EP_AFFINE_RELU_F32, EP_GATED_ELU, EP_GATED_PLANAR_F32, EP_AFFINE_F32_NHWC, EP_GATED_PLANAR_F32_PAIRED, EP_GATED_ELU_PAIRED = 0, 1, 2, 3, 4, 5, 6


def _env_override(var, layer):
    """The integer a tuning variable of the form "l7=32,up1_1=16" assigns to `layer`, or None.  Empty items (an unset variable, a trailing comma),
    unnamed layers and values that are not integers are ignored."""
    if not layer:
        return None
    for item in os.environ.get(var, "").split(","):
        key, _, val = item.partition("=")
        if key.strip() == layer and val.strip().lstrip("-").isdigit():
            return int(val)
    return None


def _ct(layer, default):
    """Channels staged per tap and chunk for a layer; MPIFLOW_CT="l7=32,up1_1=16" overrides (tuning aid)."""
    v = _env_override("MPIFLOW_CT", layer)
    return default if v is None else v


# layers whose weights go through LDS once per workgroup (by LDS-DMA) - all but three, measured per layer at 64x384x1280
# (profiles/r2/engine_glds_layers.txt): the stride-2 encoder layers of the feature-mask network load their fragments per wave
_PER_WAVE_LAYERS = {"l2", "l3", "l4"}
_WLDS_LAYERS = {"l1", "l5", "l6", "l7", "l8", "l9", "up0_4", "up1_4", "up0_3", "up1_3", "up0_2", "up1_2", "up0_1", "up1_1", "up0_0", "up1_0", "disp0"}


# planes per workgroup of the few-block full-resolution layers (MpfConvArgs.pw): the workgroup walks that many consecutive planes at its tile position and
# computes the pixel-only part of the loader once.  MPIFLOW_PW="l8s=2,disp0=4" overrides (tuning aid; 1 = one plane per workgroup).
_PW_LAYERS = {"l7": 4, "l8s": 4, "l9": 4, "up0_0": 4, "up1_0": 4, "disp0": 4}


def _pw(layer, S, nb):
    v = _env_override("MPIFLOW_PW", layer)
    pw = _PW_LAYERS.get(layer, 1) if v is None else v
    return pw if pw > 1 and nb <= 2 and S % pw == 0 else 1


def _wlds(layer, default):
    """MPIFLOW_WLDS="all" | "none" | "l8,up1_1" overrides the per-layer choice (tuning aid)."""
    v = os.environ.get("MPIFLOW_WLDS")
    if v is None:
        return default
    return v == "all" or layer in v.split(",")


def pad8(c):
    return (int(c) + 7) // 8 * 8


def ksteps(ct):
    return (9 * ct + 31) // 32


def pack_weights(w_rows, vmap, ct, device=None):
    """Weights in MFMA A-fragment order for mpf_conv3x3_f16.

    w_rows [R, Cin, 3, 3] fp32 (R = 16 * nblk rows in packed order, zero rows for padding);  vmap: LongTensor [Cv], virtual
    input channel -> real input channel or -1 (Cv = nchunk * ct).
    Returns fp16 [nchunk, ksteps, nblk, 64, 8]: lane l of fragment (chunk, ks, blk) holds, for j = 0..7,
    W[row = 16 blk + (l & 15)][tap = ks * (32/ct) + (l >> 4) // (ct/8)][virtual channel = chunk * ct + ((l >> 4) % (ct/8)) * 8 + j]
    (zero when the tap index exceeds 8) - the operand layout of v_mfma_f32_16x16x32_f16.
    """
    R = w_rows.shape[0]
    assert R % 16 == 0 and vmap.numel() % ct == 0
    nblk, nchunk, KS, vpp, tps = R // 16, vmap.numel() // ct, ksteps(ct), ct // 8, 32 // ct
    # pure data movement (gathers of fp32 values, one rounding to fp16 at the end): done on `device` when given - the 20 layers of the network
    # take 0.6 s of host time on the CPU, a few milliseconds on the GPU, same bytes
    dev = torch.device(device) if device is not None else w_rows.device
    w_rows, vmap = w_rows.to(dev), vmap.to(dev)
    wv = torch.zeros(R, vmap.numel(), 10, dtype=torch.float32, device=dev)          # tap 9 = the all-zero tap
    valid = vmap >= 0
    wv[:, valid, :9] = w_rows.float().reshape(R, w_rows.shape[1], 9)[:, vmap[valid], :]
    q = torch.arange(4)
    ks = torch.arange(KS)
    slot = (ks[:, None] * tps + (q[None, :] // vpp)).clamp(max=9)       # [KS, 4]
    ch = (((q % vpp) * 8)[:, None] + torch.arange(8)[None, :]).to(dev)   # [4, 8] channel inside the chunk
    out = torch.empty(nchunk, KS, nblk, 4, 16, 8, dtype=torch.float32, device=dev)
    wv = wv.reshape(nblk, 16, nchunk, ct, 10)
    for k in range(KS):
        for qq in range(4):
            out[:, k, :, qq] = wv[:, :, :, ch[qq], int(slot[k, qq])].permute(2, 0, 1, 3)
    return out.reshape(nchunk, KS, nblk, 64, 8).to(torch.float16).contiguous()


def pack_weights_taps(w_RCT, vmap, ct, device=None):
    """pack_weights for any tap count: w_RCT [R, Cin, T] fp32 -> fp16 [nchunk, ksteps(T), nblk, 64, 8] in MFMA A-fragment order: lane l of fragment (chunk, ks, blk)
    holds W[row 16 blk + (l & 15)][tap ks * (32/ct) + (l >> 4) // (ct/8)][virtual channel chunk * ct + ((l >> 4) % (ct/8)) * 8 + j], zero past the last tap."""
    R, T = w_RCT.shape[0], w_RCT.shape[2]
    assert R % 16 == 0 and vmap.numel() % ct == 0
    nblk, nchunk, KS, vpp, tps = R // 16, vmap.numel() // ct, (T * ct + 31) // 32, ct // 8, 32 // ct
    dev = torch.device(device) if device is not None else w_RCT.device
    w_RCT, vmap = w_RCT.to(dev), vmap.to(dev)
    wv = torch.zeros(R, vmap.numel(), T + 1, dtype=torch.float32, device=dev)      # tap T = the all-zero tap
    valid = vmap >= 0
    wv[:, valid, :T] = w_RCT.float()[:, vmap[valid], :]
    q = torch.arange(4)
    slot = (torch.arange(KS)[:, None] * tps + (q[None, :] // vpp)).clamp(max=T)
    ch = (((q % vpp) * 8)[:, None] + torch.arange(8)[None, :]).to(dev)
    out = torch.empty(nchunk, KS, nblk, 4, 16, 8, dtype=torch.float32, device=dev)
    wv = wv.reshape(nblk, 16, nchunk, ct, T + 1)
    for k in range(KS):
        for qq in range(4):
            out[:, k, :, qq] = wv[:, :, :, ch[qq], int(slot[k, qq])].permute(2, 0, 1, 3)
    return out.reshape(nchunk, KS, nblk, 64, 8).to(torch.float16).contiguous()


# which of the three taps of one axis fall on the first / second of the two distinct low-resolution pixels a 3-tap window of phase p covers (k_conv3x3_up)
_PHASE_TAPS = {0: ((0,), (1, 2)), 1: ((0, 1), (2,))}


def pack_weights_up(w_rows, n_up, vmap_skip, ct, device=None):
    """Weights of a phase-decomposed x2-nearest layer (MPF_CONV_LD_NEAREST_PHASE).  w_rows [R, Cin, 3, 3]; the first `n_up` input channels are the upsampled
    source, the rest the skip (vmap_skip: its virtual -> real map, real indices counted from n_up).  The upsampled part becomes, per phase (py, px), a 2 x 2
    kernel of SUMS of the nine weights (summed in fp32, rounded to fp16 once): [chunkA][phase 2 py + px][ksteps(4)][nblk][64][8]; the skip part keeps its nine
    taps: [chunkB][ksteps(9)][nblk][64][8].  -> one flat fp16 tensor, A part first."""
    R = w_rows.shape[0]
    dev = torch.device(device) if device is not None else w_rows.device
    w = w_rows.float().to(dev)
    nchunkA = (n_up + ct - 1) // ct
    vmapA = torch.full((nchunkA * ct,), -1, dtype=torch.long)
    vmapA[:n_up] = torch.arange(n_up)
    parts = []
    phases = []
    for py in (0, 1):
        for px in (0, 1):
            w4 = torch.zeros(R, n_up, 2, 2, device=dev)
            for ty in (0, 1):
                for tx in (0, 1):
                    for ky in _PHASE_TAPS[py][ty]:
                        for kx in _PHASE_TAPS[px][tx]:
                            w4[:, :, ty, tx] += w[:, :n_up, ky, kx]
            phases.append(pack_weights_taps(w4.reshape(R, n_up, 4), vmapA, ct, device=dev))          # [chunkA, KSA, nblk, 64, 8]
    parts.append(torch.stack(phases, dim=1).reshape(-1))                                               # [chunkA, phase, KSA, nblk, 64, 8]
    if vmap_skip is not None and vmap_skip.numel():
        parts.append(pack_weights_taps(w[:, n_up:].reshape(R, w.shape[1] - n_up, 9), vmap_skip, ct, device=dev).reshape(-1))
    return torch.cat(parts).contiguous()


def _bn_affine(bn):
    scale = bn.weight.detach().float() / torch.sqrt(bn.running_var.detach().float() + bn.eps)
    return scale, bn.bias.detach().float() - bn.running_mean.detach().float() * scale


class ConvLayer:
    """One packed layer + its launch."""

    def __init__(self, device, *, loader, epi, stride, pad_mode, ct, vmap, rows_w, ep, nblk, ncg, Cst, CA, CB, name="", plane_major=False, n_up=0):
        self.name, self.wlds_default, self.plane_major = name, name.rstrip("sp") in _WLDS_LAYERS or name in _WLDS_LAYERS, plane_major
        self.loader, self.epi, self.stride, self.pad_mode, self.ct = loader, epi, stride, pad_mode, ct
        self.nchunk = vmap.numel() // ct
        self.nblk, self.ncg, self.Cst, self.CA, self.CB = nblk, ncg, Cst, CA, CB
        self.vmap_real = vmap
        self.rows_real = int((rows_w.reshape(rows_w.shape[0], -1).abs().sum(1) > 0).sum())     # output rows that are not padding
        if loader == LD_NEAREST_PHASE:
            # n_up real channels of the upsampled source (CA physical, padded to 8), then the skip's CB virtual channels: whole chunks per source
            skip = vmap[CA:] - n_up if CB else None                      # virtual -> real map of the skip part, real indices counted from n_up
            if skip is not None:
                skip = torch.where(vmap[CA:] >= 0, skip, torch.full_like(skip, -1))
                skip = torch.cat([skip, torch.full(((-skip.numel()) % ct,), -1, dtype=torch.long)])
            self.nchunk = (CA + ct - 1) // ct + (0 if skip is None else skip.numel() // ct)
            self.wpack = pack_weights_up(rows_w, n_up, skip, ct, device=device).to(device)
        else:
            self.wpack = pack_weights(rows_w, vmap, ct, device=device).to(device)
        self.ep = ep.float().contiguous().to(device)
        assert self.ep.shape == (3, nblk * 16)

    # -- builders ---------------------------------------------------------------------------------------------------
    @staticmethod
    def _vmap(segments, ct):
        """segments: list of (padded, real) channel counts in concatenation order -> virtual->real map padded to ct."""
        idx, real0 = [], 0
        for padded, real in segments:
            idx += list(range(real0, real0 + real)) + [-1] * (padded - real)
            real0 += real
        idx += [-1] * (-len(idx) % ct)
        return torch.tensor(idx, dtype=torch.long)

    @classmethod
    def affine_relu(cls, device, cbr, segments, *, loader, stride, ct, f32_out=False, name="", pre_activation=False, plane_major=False):
        """Conv2d(bias) + BatchNorm(eval) + ReLU (ConvBNReLU, model/CPN/unet.py:6-15), zero padding.  pre_activation: no ReLU, every channel
        as fp32 NHWC (EP_AFFINE_F32_NHWC) - the factorised first layer's maps."""
        conv, bn = cbr.layer[0], cbr.layer[1]
        cout = conv.out_channels
        nblk = (cout + 15) // 16
        rows_w = torch.zeros(nblk * 16, conv.in_channels, 3, 3)
        rows_w[:cout] = conv.weight.detach().float().cpu()
        scale, shift = _bn_affine(bn)
        ep = torch.zeros(3, nblk * 16)
        ep[0, :cout] = scale.cpu()
        ep[1, :cout] = (shift + conv.bias.detach().float() * scale).cpu()
        nb = nblk if nblk <= 8 else 8
        if nb == 8 and stride == 1:
            nb = 4              # l5: the stride-1 8-block kernel runs ONE wave per SIMD (128 accumulator registers); 4 blocks: 0.21 -> 0.17 ms.
                                # (the stride-2 l4 is the other way round: 0.20 ms with 8 blocks, 0.27 with 4)
        v = _env_override("MPIFLOW_NB", name)                             # tuning aid: MPIFLOW_NB="l4=4,l5=4"
        if v is not None and v > 0 and nblk % v == 0:
            nb = v
        assert nblk % nb == 0
        CA, CB = segments[0][0], (segments[1][0] if len(segments) > 1 else 0)
        epi = EP_AFFINE_F32_NHWC if pre_activation else (EP_AFFINE_RELU_F32 if f32_out else EP_AFFINE_RELU)
        return cls(device, loader=loader, epi=epi, stride=stride, pad_mode=0, ct=ct,
                   vmap=cls._vmap(segments, ct), rows_w=rows_w, ep=ep, nblk=nblk, ncg=nblk // nb, Cst=1 if f32_out else pad8(cout),
                   CA=CA, CB=CB, name=name, plane_major=plane_major)

    @classmethod
    def gated(cls, device, gconv, bn, segments, *, loader, ct, planar=False, name=""):
        """GatedConv (+ BatchNorm + ELU when bn is given), reflection padding (model/CPN/decoder.py:10-71)."""
        cf, cmk = gconv.conv2d, gconv.mask_conv2d
        cout = cf.out_channels
        if planar and bn is None and cout <= 8 and os.environ.get("MPIFLOW_DISP_PAIRED", "1") != "0":
            # the 4-channel output layer: feature and gate rows interleaved in ONE 16-row block (row 2c / 2c + 1) instead of one block each
            rows_w, ep = torch.zeros(16, cf.in_channels, 3, 3), torch.zeros(3, 16)
            rows_w[0:2 * cout:2], rows_w[1:2 * cout:2] = cf.weight.detach().float().cpu(), cmk.weight.detach().float().cpu()
            ep[0, 0:2 * cout:2], ep[0, 1:2 * cout:2] = cf.bias.detach().float().cpu(), cmk.bias.detach().float().cpu()
            return cls(device, loader=loader, epi=EP_GATED_PLANAR_F32_PAIRED, stride=1, pad_mode=1, ct=ct, vmap=cls._vmap(segments, ct), rows_w=rows_w, ep=ep,
                       nblk=1, ncg=1, Cst=cout, CA=segments[0][0], CB=(segments[1][0] if len(segments) > 1 else 0), name=name)
        nf_total = (cout + 15) // 16
        if not planar and bn is not None and (2 * cout + 15) // 16 == 3 and os.environ.get("MPIFLOW_GATED_PAIRED", "1") != "0":
            # 24 output channels: feature / gate rows interleaved fill 3 blocks, the block-per-half layout needs 2 + 2 half-empty ones
            rows_w, ep = torch.zeros(48, cf.in_channels, 3, 3), torch.zeros(3, 48)
            rows_w[0:2 * cout:2], rows_w[1:2 * cout:2] = cf.weight.detach().float().cpu(), cmk.weight.detach().float().cpu()
            ep[0, 0:2 * cout:2], ep[0, 1:2 * cout:2] = cf.bias.detach().float().cpu(), cmk.bias.detach().float().cpu()
            scale, shift = _bn_affine(bn)
            ep[1, :cout], ep[2, :cout] = scale.cpu(), shift.cpu()
            return cls(device, loader=loader, epi=EP_GATED_ELU_PAIRED, stride=1, pad_mode=1, ct=ct, vmap=cls._vmap(segments, ct), rows_w=rows_w, ep=ep,
                       nblk=3, ncg=1, Cst=pad8(cout), CA=segments[0][0], CB=(segments[1][0] if len(segments) > 1 else 0), name=name, n_up=segments[0][1])
        nf = max(d for d in (4, 3, 2, 1) if nf_total % d == 0)            # feature blocks per workgroup (NB = 2 nf in {2,4,6,8})
        if nf == 4:
            # the 8-block kernels hold 128 accumulator + 158 other registers: ONE wave per SIMD, nothing to hide a load behind.  With 4 blocks
            # (3 waves per SIMD) the 192-channel layers run 0.25 -> 0.215 ms (up0_4) and 0.48 -> 0.42 ms (up1_4) at 64 x 384 x 1280
            nf = 2
        if loader == LD_NEAREST_PHASE and name == "up1_3":
            nf = 2              # phase form: 4 blocks at three waves per SIMD beat 6 at two (0.315 vs 0.326 ms, profiles/r6/engine_up_layers_sweep.txt)
        v = _env_override("MPIFLOW_NF", name)                             # tuning aid: MPIFLOW_NF="up0_4=2,up1_4=3"
        if v is not None and v > 0 and nf_total % v == 0:
            nf = v
        ncg = nf_total // nf
        nblk = 2 * nf_total
        rows_w = torch.zeros(nblk * 16, cf.in_channels, 3, 3)
        ep = torch.zeros(3, nblk * 16)
        if bn is not None:
            scale, shift = _bn_affine(bn)
        for cg in range(ncg):
            for b in range(nf):
                c0 = (cg * nf + b) * 16
                n = max(0, min(16, cout - c0))
                rf = (cg * 2 * nf + b) * 16
                rm = (cg * 2 * nf + nf + b) * 16
                rows_w[rf:rf + n] = cf.weight.detach().float().cpu()[c0:c0 + n]
                rows_w[rm:rm + n] = cmk.weight.detach().float().cpu()[c0:c0 + n]
                ep[0, rf:rf + n] = cf.bias.detach().float().cpu()[c0:c0 + n]
                ep[0, rm:rm + n] = cmk.bias.detach().float().cpu()[c0:c0 + n]
                if bn is not None:
                    ep[1, rf:rf + n] = scale.cpu()[c0:c0 + n]
                    ep[2, rf:rf + n] = shift.cpu()[c0:c0 + n]
        CA, CB = segments[0][0], (segments[1][0] if len(segments) > 1 else 0)
        return cls(device, loader=loader, epi=EP_GATED_PLANAR_F32 if planar else EP_GATED_ELU, stride=1, pad_mode=1, ct=ct,
                   vmap=cls._vmap(segments, ct), rows_w=rows_w, ep=ep, nblk=nblk, ncg=ncg, Cst=cout if planar else pad8(cout),
                   CA=CA, CB=CB, name=name, n_up=segments[0][1])

    # -- launch -------------------------------------------------------------------------------------------------------
    def __call__(self, S, Hin, Win, srcA=None, srcB=None, cm=None, fm=None, plane_vals=None, HA=None, WA=None, out=None, bprime_table=False):
        Hout, Wout = (Hin - 1) // self.stride + 1, (Win - 1) // self.stride + 1
        dev = self.wpack.device
        if out is None:
            if self.epi == EP_AFFINE_RELU_F32:
                out = torch.empty(S, Hout, Wout, dtype=torch.float32, device=dev)
            elif self.epi == EP_AFFINE_F32_NHWC:
                out = torch.empty(S, Hout, Wout, self.Cst, dtype=torch.float32, device=dev)
            elif self.epi in (EP_GATED_PLANAR_F32, EP_GATED_PLANAR_F32_PAIRED):
                out = torch.empty(S, self.Cst, Hout, Wout, dtype=torch.float32, device=dev)
            else:
                out = torch.empty(S, Hout, Wout, self.Cst, dtype=torch.float16, device=dev)
        a = _lib.MpfConvArgs()
        p = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())      # noqa: E731
        a.srcA, a.srcB, a.cm, a.fm, a.plane_vals = p(srcA), p(srcB), p(cm), p(fm), p(plane_vals)
        a.wpack, a.ep, a.out = p(self.wpack), p(self.ep), p(out)
        a.S, a.Hin, a.Win, a.Hout, a.Wout = S, Hin, Win, Hout, Wout
        a.CA, a.CB = self.CA, self.CB
        a.HA, a.WA = (HA if HA is not None else Hin), (WA if WA is not None else Win)
        a.ct, a.nchunk, a.nblk, a.ncg, a.Cst = self.ct, self.nchunk, self.nblk, self.ncg, self.Cst
        a.loader, a.epi, a.stride, a.pad_mode = self.loader, self.epi, self.stride, self.pad_mode
        a.wlds = int(_wlds(self.name, self.wlds_default))
        a.plane_major = int(self.plane_major)
        a.bprime_table = int(bool(bprime_table))
        # (of the phase-decomposed layers only the single-chunk one without a skip source - upconv(0,1) - has a plane-walking kernel)
        a.pw = _pw(self.name, S, self.nblk // self.ncg) if (self.loader != LD_NEAREST_PHASE or (self.CB == 0 and self.CA <= 16)) else 1
        self.last_call = dict(S=S, Hin=Hin, Win=Win, Hout=Hout, Wout=Wout, HA=a.HA, WA=a.WA)
        if self.loader in (LD_BILINEAR_CAT, LD_BILINEAR_SYNTH):
            a.fparams[0] = (a.HA - 1) / (Hin - 1) if Hin > 1 else 0.0
            a.fparams[1] = (a.WA - 1) / (Win - 1) if Win > 1 else 0.0
        with torch.cuda.device(dev):
            stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
            _lib.check(_lib.load().mpf_conv3x3_f16(ctypes.byref(a), stream), "mpf_conv3x3_f16")
        return out


def layer_accounting(layer):
    """Algorithmic work of ONE launch of `layer` (shapes of its last call): flops of the convolution on the REAL channels (the reference's
    own count: 2 * S * Hout * Wout * Cout * Cin * 9, gated layers have two such convolutions), and the bytes that have to cross HBM once -
    every source tensor read once (fp16 activations; the shared skip features and fp32 masks of the per-plane loader once per image, not per
    plane), the output written once, the packed weights once."""
    c = layer.last_call
    S, Hin, Win, Hout, Wout = c["S"], c["Hin"], c["Win"], c["Hout"], c["Wout"]
    cin_real = int((layer.vmap_real >= 0).sum())
    rows_real = layer.rows_real
    flops = 2.0 * S * Hout * Wout * rows_real * cin_real * 9
    if layer.loader == LD_FMN_INPUT:
        rd = Hin * Win * (3 + 1) * 4                                    # image + disparity, fp32, shared by the S planes
    elif layer.loader == LD_FMN_SYNTH:
        rd = 2 * Hin * Win * 16 * 4                                     # the two fp32 first-layer maps, shared by the S planes
    elif layer.loader == LD_BILINEAR_SYNTH:
        rd = S * c["HA"] * c["WA"] * layer.CA * 2 + 2 * Hin * Win * 16 * 4
    elif layer.loader == LD_DIRECT:
        rd = S * Hin * Win * layer.CA * 2
    elif layer.loader == LD_BILINEAR_CAT:
        rd = S * c["HA"] * c["WA"] * layer.CA * 2 + S * Hin * Win * layer.CB * 2
    else:                                                               # nearest-upsampled planes (either loader form) + shared skip features + the two fp32 masks
        rd = S * c["HA"] * c["WA"] * layer.CA * 2 + (Hin * Win * (layer.CB - 8) * 2 + 2 * S * Hin * Win * 4 if layer.CB else 0)
    if layer.epi == EP_AFFINE_RELU_F32:
        wr = S * Hout * Wout * 4
    elif layer.epi == EP_AFFINE_F32_NHWC:
        wr = S * Hout * Wout * layer.Cst * 4
    elif layer.epi in (EP_GATED_PLANAR_F32, EP_GATED_PLANAR_F32_PAIRED):
        wr = S * layer.Cst * Hout * Wout * 4
    else:
        wr = S * Hout * Wout * layer.Cst * 2
    return dict(name=layer.name, flops=flops, bytes=float(rd + wr + layer.wpack.numel() * 2), read_bytes=float(rd), write_bytes=float(wr))


def _nhwc16(t_1CHW):
    return t_1CHW[0].permute(1, 2, 0).contiguous().to(torch.float16)


class FeatMaskEngine:
    """FeatMaskNetwork.forward (model/CPN/unet.py:44-69) in 9 launches + one softmax over the planes."""

    def __init__(self, fmn, device):
        A = ConvLayer.affine_relu
        self._fmn, self._device = fmn, device
        self.factor = os.environ.get("MPIFLOW_FMN_FACTOR", "1") != "0"
        self.l3 = A(device, fmn.conv3, [(32, 32)], loader=LD_DIRECT, stride=2, ct=32, name="l3")
        self.l4 = A(device, fmn.conv4, [(64, 64)], loader=LD_DIRECT, stride=2, ct=32, name="l4")
        self.l5 = A(device, fmn.conv5, [(128, 128)], loader=LD_DIRECT, stride=1, ct=_ct("l5", 32), name="l5")
        self.l6 = A(device, fmn.conv6, [(128, 128), (64, 64)], loader=LD_BILINEAR_CAT, stride=1, ct=_ct("l6", 16), name="l6")
        self.l7 = A(device, fmn.conv7, [(64, 64), (32, 32)], loader=LD_BILINEAR_CAT, stride=1, ct=_ct("l7", 16), name="l7")
        self.l9 = A(device, fmn.conv9, [(16, 16)], loader=LD_DIRECT, stride=1, ct=16, f32_out=True, name="l9")
        # The first layer factorised (model/CPN/unet.py:44-50): the 64 plane-images differ only in the constant plane channel d_s and the layer is
        # affine in it up to the ReLU, c1[s] = relu(A' + d_s * B').  A' (per image) and B' (per size: zero padding makes it position-dependent at
        # the border) are fp32 [H,W,16] maps from ONE plane's worth of the layer-1 kernel without its ReLU; layers 2 and 8 synthesise c1 in their
        # loaders, plane index fastest in the grid so that the planes of a tile share the maps in L2.  The 1 GB activation is never written / read
        # twice, the 0.32 ms launch is gone.  MPIFLOW_FMN_FACTOR=0 keeps the materialised form (A/B, per-layer tests).
        # Only the set the mode uses is packed and uploaded; the other one is built on first access (per-layer tests, A/B runs).
        self._plane_map, self._zeros, self._plane_table = {}, {}, {}

    _LAZY = {"l1": lambda A, f, d: A(d, f.conv1, [(8, 5)], loader=LD_FMN_INPUT, stride=1, ct=8, name="l1"),
             "l2": lambda A, f, d: A(d, f.conv2, [(16, 16)], loader=LD_DIRECT, stride=2, ct=16, name="l2"),
             "l8": lambda A, f, d: A(d, f.conv8, [(32, 32), (16, 16)], loader=LD_BILINEAR_CAT, stride=1, ct=_ct("l8", 16), name="l8"),
             "l1p": lambda A, f, d: A(d, f.conv1, [(8, 5)], loader=LD_FMN_INPUT, stride=1, ct=8, name="l1p", pre_activation=True),
             "l2s": lambda A, f, d: A(d, f.conv2, [(16, 16)], loader=LD_FMN_SYNTH, stride=2, ct=16, name="l2s", plane_major=True),
             "l8s": lambda A, f, d: A(d, f.conv8, [(32, 32), (16, 16)], loader=LD_BILINEAR_SYNTH, stride=1, ct=_ct("l8", 16), name="l8s", plane_major=True)}

    def __getattr__(self, name):                       # only reached when the attribute does not exist yet
        build = FeatMaskEngine._LAZY.get(name)
        if build is None or "_fmn" not in self.__dict__:
            raise AttributeError(name)
        layer = build(ConvLayer.affine_relu, self._fmn, self._device)
        setattr(self, name, layer)
        return layer

    def first_layer_maps(self, image_3HW, disp_HW):
        """-> (A' [H,W,16] fp32 for this image, B' [H,W,16] fp32 for this size)"""
        H, W = disp_HW.shape
        dev = disp_HW.device
        key = (H, W)
        if key not in self._plane_map:
            z3, z1 = torch.zeros(3, H, W, device=dev), torch.zeros(H, W, device=dev)
            pre = self.l1p(2, H, W, srcA=z3, srcB=z1, plane_vals=torch.tensor([0.0, 1.0], device=dev))
            self._plane_map[key] = (pre[1] - pre[0]).contiguous()        # BN-scale * conv(0, 0, 0, 0, 1): the bias / shift cancel
            self._zeros[key] = torch.zeros(1, device=dev)
        return self.l1p(1, H, W, srcA=image_3HW, srcB=disp_HW, plane_vals=self._zeros[key])[0], self._plane_map[key]

    def plane_table(self, H, W):
        """B' as the [3,3,16] table of its border classes, or None.  B' depends on the pixel only through WHICH of the nine taps fall inside the image (zero
        padding): top / inner / bottom row x left / inner / right column.  Layers 2 and 8 read 576 bytes from L1 instead of 64 bytes per pixel and plane from L2
        (MpfConvArgs.bprime_table) - the SAME values, checked here against the whole map once per size; a size where that does not hold keeps the map."""
        key = (H, W)
        if key not in self._plane_table:
            Bm, t = self._plane_map[key], None
            if H >= 3 and W >= 3 and os.environ.get("MPIFLOW_BPRIME_TABLE", "1") != "0":
                rows, cols = torch.tensor([0, 1, H - 1], device=Bm.device), torch.tensor([0, 1, W - 1], device=Bm.device)
                t = Bm[rows][:, cols].contiguous()                        # [3,3,16]
                ry = torch.ones(H, dtype=torch.long, device=Bm.device)
                cx = torch.ones(W, dtype=torch.long, device=Bm.device)
                ry[0], ry[-1], cx[0], cx[-1] = 0, 2, 0, 2
                if not torch.equal(t[ry][:, cx], Bm):
                    t = None
            self._plane_table[key] = t
        return self._plane_table[key]

    def logits(self, image_3HW, disp_HW, plane_disp_S):
        S = plane_disp_S.numel()
        H, W = disp_HW.shape
        if H % 8 or W % 8:
            raise ValueError("feature-mask network needs H and W divisible by 8 (three stride-2 stages and x2 upsampling back)")
        img, dsp, pd = image_3HW.float().contiguous(), disp_HW.float().contiguous(), plane_disp_S.float().contiguous()
        if self.factor:
            A1, B1 = self.first_layer_maps(img, dsp)
            Bt = self.plane_table(H, W)
            c2 = self.l2s(S, H, W, srcA=A1, srcB=B1 if Bt is None else Bt, plane_vals=pd, bprime_table=Bt is not None)
        else:
            c1 = self.l1(S, H, W, srcA=img, srcB=dsp, plane_vals=pd)
            c2 = self.l2(S, H, W, srcA=c1)
        c3 = self.l3(S, H // 2, W // 2, srcA=c2)
        c4 = self.l4(S, H // 4, W // 4, srcA=c3)
        c5 = self.l5(S, H // 8, W // 8, srcA=c4)
        c6 = self.l6(S, H // 4, W // 4, srcA=c5, srcB=c3, HA=H // 8, WA=W // 8)
        c7 = self.l7(S, H // 2, W // 2, srcA=c6, srcB=c2, HA=H // 4, WA=W // 4)
        if self.factor:
            c8 = self.l8s(S, H, W, srcA=c7, srcB=A1, cm=B1 if Bt is None else Bt, plane_vals=pd, HA=H // 2, WA=W // 2, bprime_table=Bt is not None)
        else:
            c8 = self.l8(S, H, W, srcA=c7, srcB=c1, HA=H // 2, WA=W // 2)
        return self.l9(S, H, W, srcA=c8)

    def __call__(self, image_3HW, disp_HW, plane_disp_S):
        """-> feature mask [S,H,W] fp32 (softmax over the planes, model/CPN/unet.py:68-69)"""
        return torch.softmax(self.logits(image_3HW, disp_HW, plane_disp_S), dim=0)


def pack_weights_f32(w, device=None):
    """Conv2d weight [Cout, Cin, k, k] fp32 -> [Cout/16, nsteps, 64, 4]: the A-operand order of mpf_conv2d_f32 (include/mpiflow_hip.h).
    K runs over 4-channel vectors v = tap * (Cin/4) + c4; step s covers v = 4s .. 4s+3 (one per 16-lane group g), zero past the last tap."""
    Cout, Cin, k, _ = w.shape
    assert Cout % 16 == 0 and Cin % 4 == 0
    dev = torch.device(device) if device is not None else w.device
    nv = k * k * (Cin // 4)
    nsteps = (nv + 3) // 4
    wv = w.detach().float().to(dev).permute(0, 2, 3, 1).reshape(Cout, nv, 4)
    wv = torch.cat([wv, torch.zeros(Cout, nsteps * 4 - nv, 4, device=dev)], dim=1).reshape(Cout // 16, 16, nsteps, 4, 4)    # [blk, m, s, g, j]
    return wv.permute(0, 2, 3, 1, 4).reshape(Cout // 16, nsteps, 64, 4).contiguous()


class Conv2dF32:
    """One single-image fp32 convolution (no bias) + folded BatchNorm [+ residual] + activation: a launch of mpf_conv2d_f32."""

    ACT = {None: 0, "relu": 1, "leaky": 2}

    def __init__(self, device, conv, bn, *, act, up=0, slope=0.0, name=""):
        assert conv.bias is None and conv.kernel_size[0] == conv.kernel_size[1] and conv.stride[0] == conv.stride[1]
        self.name = name
        self.k, self.stride, self.pad, self.up = conv.kernel_size[0], conv.stride[0], conv.padding[0], up
        self.cin, self.cout = conv.in_channels, conv.out_channels
        self.act, self.slope = self.ACT[act], float(slope)
        self.wpack = pack_weights_f32(conv.weight, device=device)
        scale, shift = _bn_affine(bn)
        self.scale, self.shift = scale.contiguous().to(device), shift.contiguous().to(device)

    def __call__(self, src_HWC, residual=None, f32=True, f16=False):
        """src_HWC f32 [h,w,Cin] (the source BEFORE the x2 nearest upsampling when up = 1) -> (f32 [Hout,Wout,Cout] or None, f16 same or None)"""
        hs, ws, c = src_HWC.shape
        assert c == self.cin and src_HWC.dtype == torch.float32 and src_HWC.is_contiguous()
        Hin, Win = hs << self.up, ws << self.up
        Hout, Wout = (Hin + 2 * self.pad - self.k) // self.stride + 1, (Win + 2 * self.pad - self.k) // self.stride + 1
        dev = src_HWC.device
        out = torch.empty(Hout, Wout, self.cout, dtype=torch.float32, device=dev) if f32 else None
        out16 = torch.empty(Hout, Wout, self.cout, dtype=torch.float16, device=dev) if f16 else None
        a = _lib.MpfConv2dArgs()
        p = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())      # noqa: E731
        a.src, a.wpack, a.scale, a.shift, a.residual, a.out, a.out_f16 = p(src_HWC), p(self.wpack), p(self.scale), p(self.shift), p(residual), p(out), p(out16)
        a.Hin, a.Win, a.Cin, a.Hout, a.Wout, a.Cout = Hin, Win, self.cin, Hout, Wout, self.cout
        a.ksize, a.stride, a.pad, a.up, a.act, a.slope = self.k, self.stride, self.pad, self.up, self.act, self.slope
        with torch.cuda.device(dev):
            stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
            _lib.check(_lib.load().mpf_conv2d_f32(ctypes.byref(a), stream), "mpf_conv2d_f32")
        self.last_call = dict(Hout=Hout, Wout=Wout, bytes=float(src_HWC.numel() * 4 + self.wpack.numel() * 4 + (residual.numel() * 4 if residual is not None else 0)
                                                                + Hout * Wout * self.cout * ((4 if f32 else 0) + (2 if f16 else 0))))
        return out, out16

    def flops(self):
        c = self.last_call
        return 2.0 * c["Hout"] * c["Wout"] * self.cout * self.cin * self.k * self.k


def maxpool3x3s2(src_HWC):
    """nn.MaxPool2d(3, 2, 1) on an NHWC fp32 image (mpf_maxpool3x3s2_f32)"""
    h, w, c = src_HWC.shape
    out = torch.empty((h - 1) // 2 + 1, (w - 1) // 2 + 1, c, dtype=torch.float32, device=src_HWC.device)
    with torch.cuda.device(src_HWC.device):
        stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        _lib.check(_lib.load().mpf_maxpool3x3s2_f32(ctypes.c_void_p(src_HWC.data_ptr()), h, w, c, ctypes.c_void_p(out.data_ptr()), stream), "mpf_maxpool3x3s2_f32")
    return out


class EncoderEngine:
    """The single-image part of the producer in HIP, fp32: ResnetEncoder.forward (model/CPN/encoder.py:86-101) and the bottleneck at the head
    of DepthDecoder.forward (model/CPN/decoder.py:131-138) - 24 convolutions + 3 max-pools + the input normalisation, 28 launches, no
    MIOpen / ATen kernel.  Hands the per-plane decoder exactly what DecoderEngine.shared_inputs did: the bottleneck output and the four
    skip features as NHWC fp16 (written by the producing convolution's epilogue beside its fp32 output)."""

    def __init__(self, encoder, decoder, device):
        e = encoder.encoder
        C = Conv2dF32
        self.conv1 = C(device, e.conv1, e.bn1, act="relu", name="enc.conv1")
        self.blocks = []
        for li in range(1, 5):
            for bi, blk in enumerate(getattr(e, "layer%d" % li)):
                n = "enc.layer%d.%d" % (li, bi)
                ds = None if blk.downsample is None else C(device, blk.downsample[0], blk.downsample[1], act=None, name=n + ".downsample")
                self.blocks.append((C(device, blk.conv1, blk.bn1, act="relu", name=n + ".conv1"), C(device, blk.conv2, blk.bn2, act="relu", name=n + ".conv2"), ds))
        L = lambda seq, up, n: C(device, seq[0], seq[1], act="leaky", slope=seq[2].negative_slope, up=up, name=n)      # noqa: E731
        self.down1, self.down2 = L(decoder.conv_down1, 0, "dec.conv_down1"), L(decoder.conv_down2, 0, "dec.conv_down2")
        self.up1, self.up2 = L(decoder.conv_up1, 1, "dec.conv_up1"), L(decoder.conv_up2, 1, "dec.conv_up2")

    def convs(self):
        out = [self.conv1]
        for c1, c2, ds in self.blocks:
            out += [c for c in (ds, c1, c2) if c is not None]
        return out + [self.down1, self.down2, self.up1, self.up2]

    def forward(self, image_3HW, disp_HW, keep_f32=False):
        """-> (top f16 [H/32,W/32,512], [c1, b1, b2, b3] f16 NHWC); keep_f32: also the fp32 NHWC tensors [c1, b1, b2, b3, b4, top] (tests)"""
        H, W = disp_HW.shape
        dev = disp_HW.device
        img, dsp = image_3HW.float().contiguous(), disp_HW.float().contiguous()
        x = torch.empty(H, W, 4, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
            _lib.check(_lib.load().mpf_encoder_input(ctypes.c_void_p(img.data_ptr()), ctypes.c_void_p(dsp.data_ptr()), H, W, ctypes.c_void_p(x.data_ptr()), stream),
                       "mpf_encoder_input")
        c1, c1h = self.conv1(x, f16=True)
        f32, f16 = [c1], [c1h]
        x = maxpool3x3s2(c1)
        for i, (conv_a, conv_b, ds) in enumerate(self.blocks):
            identity = x if ds is None else ds(x)[0]
            y, _ = conv_a(x)
            last_of_stage = i % 2 == 1
            x, xh = conv_b(y, residual=identity, f16=last_of_stage and i < 7)
            if last_of_stage:
                f32.append(x)
                if xh is not None:
                    f16.append(xh)
        t, _ = self.down1(maxpool3x3s2(x))
        t, _ = self.down2(maxpool3x3s2(t))
        t, _ = self.up1(t)
        top, toph = self.up2(t, f32=keep_f32, f16=True)
        if keep_f32:
            return toph, f16, f32 + [top]
        return toph, f16

    __call__ = forward


class DecoderEngine:
    """DepthDecoder.forward (model/CPN/decoder.py:124-174) for B = 1: bottleneck on torch, the 11 gated convolutions over
    S planes in HIP.  Returns the raw last-layer output [S,4,H,W] fp32 and the cumulative mask [S,H,W] fp32 - the hand-off
    mpf_src_blend_flow(..., d_cum_mask) finishes in registers."""

    DEC = [12, 24, 48, 96, 192]

    def __init__(self, decoder, num_ch_enc, device, amp_dtype=torch.float16):
        self.decoder = decoder
        self.amp_dtype = amp_dtype
        enc = [int(c) for c in num_ch_enc]
        G = ConvLayer.gated
        key = lambda *t: "-".join(str(tuple(t)))                     # noqa: E731
        self.up0, self.up1 = {}, {}
        dec = self.DEC
        # upconv(i, 1): the 3x3 window over the x2-nearest source covers 2 x 2 distinct low-resolution pixels per output phase - four 2x2 convolutions with
        # host-summed weights (k_conv3x3_up, MPF_CONV_LD_NEAREST_PHASE) instead of nine taps on a gathered full-resolution tile.  MPIFLOW_UP_PHASE=0 (or a
        # list of layer names to keep on the gather form, "up1_0,up1_4") selects the round-5 kernels (A/B, per-layer tests).
        off = os.environ.get("MPIFLOW_UP_PHASE", "1")
        up_loader = lambda n: LD_NEAREST_PLANE if off == "0" or n in off.split(",") else LD_NEAREST_PHASE      # noqa: E731
        for i in range(4, -1, -1):
            blk0, blk1 = decoder.convs[key("upconv", i, 0)], decoder.convs[key("upconv", i, 1)]
            if i == 4:
                self.up0[i] = G(device, blk0.gated_conv, blk0.bn, [(0, 0), (enc[4] + 8, enc[4] + 2)], loader=LD_NEAREST_PLANE, ct=_ct("up0_4", 32), name="up0_4")
            else:
                cin = dec[i + 1]
                self.up0[i] = G(device, blk0.gated_conv, blk0.bn, [(pad8(cin), cin)], loader=LD_DIRECT, ct=_ct("up0_%d" % i, 16), name="up0_%d" % i)
            cx = dec[i]
            if i > 0:
                self.up1[i] = G(device, blk1.gated_conv, blk1.bn, [(pad8(cx), cx), (enc[i - 1] + 8, enc[i - 1] + 2)],
                                loader=up_loader("up1_%d" % i), ct=_ct("up1_%d" % i, 16 if (i in (1, 2, 3) or up_loader("up1_%d" % i) == LD_NEAREST_PHASE) else 32),
                                name="up1_%d" % i)       # (up1_4 as a phase layer: 16 channels per tap, 0.348 vs 0.377 ms with 32)
            else:
                self.up1[i] = G(device, blk1.gated_conv, blk1.bn, [(pad8(cx), cx)], loader=up_loader("up1_0"), ct=16, name="up1_0")
        self.disp0 = G(device, decoder.convs[key("dispconv", 0)], None, [(16, dec[0])], loader=LD_DIRECT, ct=16, planar=True, name="disp0")

    def shared_inputs(self, feats):
        """The batch-1 part: bottleneck on the 1/32 feature map (torch) and the NHWC fp16 images of everything the per-plane
        layers read.  -> (top [h,w,512], [skip feature maps NHWC fp16 for the 1/2 .. 1/16 scales])"""
        d = self.decoder
        with torch.autocast("cuda", dtype=self.amp_dtype, enabled=self.amp_dtype is not None):
            top = d.conv_up2(d.upsample(d.conv_up1(d.upsample(d.conv_down2(d.downsample(d.conv_down1(d.downsample(feats[-1]))))))))
        return _nhwc16(top), [_nhwc16(f) for f in feats[:4]]

    def __call__(self, feats, masks, shared=None):
        """feats: the encoder's five feature maps [1,C,h,w]; masks: plane_masks(logits) -> (raw [S,4,H,W] fp32, cum_mask)"""
        S, H, W = masks["cum"].shape
        top, skips = shared if shared is not None else self.shared_inputs(feats)
        h, w = top.shape[:2]
        if (h * 32, w * 32) != (H, W):
            raise ValueError("bottleneck output is %s, expected %s" % ((h, w), (H // 32, W // 32)))
        x = self.up0[4](S, h, w, srcB=top, cm=masks["cm"][4], fm=masks["fm"][4])
        for i in range(4, -1, -1):
            if i < 4:
                x = self.up0[i](S, h, w, srcA=x)
            ha, wa = h, w
            h, w = 2 * h, 2 * w
            if i > 0:
                f = skips[i - 1]
                if tuple(f.shape[:2]) != (h, w):
                    raise ValueError("encoder feature %d is %s, decoder expects %s" % (i - 1, tuple(f.shape[:2]), (h, w)))
                x = self.up1[i](S, h, w, srcA=x, srcB=f, cm=masks["cm"][i - 1], fm=masks["fm"][i - 1], HA=ha, WA=wa)
            else:
                x = self.up1[i](S, h, w, srcA=x, HA=ha, WA=wa)
        return self.disp0(S, h, w, srcA=x), masks["cum"]


def plane_masks(logits_SHW, want_feature_mask=False):
    """mpf_plane_masks: softmax over the planes + cumulative mask + the context / feature mask pyramid (H/2 .. H/32) in one
    pass over the logits.  -> dict(cum [S,H,W], cm [5 x [S,H/k,W/k]], fm [...], fmask [S,H,W] or None)"""
    S, H, W = logits_SHW.shape
    dev = logits_SHW.device
    lg = logits_SHW.float().contiguous()
    cum = torch.empty_like(lg)
    fmask = torch.empty_like(lg) if want_feature_mask else None
    cm = [torch.empty(S, H >> k, W >> k, dtype=torch.float32, device=dev) for k in range(1, 6)]
    fm = [torch.empty_like(t) for t in cm]
    arr = lambda ts: (ctypes.c_void_p * 5)(*[t.data_ptr() for t in ts])            # noqa: E731
    with torch.cuda.device(dev):
        stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        _lib.check(_lib.load().mpf_plane_masks(ctypes.c_void_p(lg.data_ptr()), S, H, W,
                                               ctypes.c_void_p(fmask.data_ptr()) if fmask is not None else None,
                                               ctypes.c_void_p(cum.data_ptr()), arr(cm), arr(fm), stream), "mpf_plane_masks")
    return dict(cum=cum, cm=cm, fm=fm, fmask=fmask)


def pad16(c):
    return (int(c) + 15) // 16 * 16


class HipPredictor:
    """MPIPredictor.forward(raw=True) (model/AdaMPI.py:55-78) for one image with the per-plane networks on the HIP engine.

    encoder: "hip" (default; env MPIFLOW_ENCODER) = EncoderEngine, the single-image part as fp32 HIP kernels (0.64 ms alone, forward 7.8 ms);
    "torch" = the torch modules on MIOpen / ATen (0.99 ms alone, forward 8.4 ms: its kernels take more from the feature-mask network they
    run beside), kept for A/B.  encoder_dtype: autocast dtype of that torch variant.  Default None = fp32: fp16 is not faster there and
    costs 3x the end-to-end error (mean |rgb| error vs the fp32 model 3.2e-3 with an fp32 encoder, 1.1e-2 with fp16; torch fp16: 1.2e-2) -
    which is why the HIP encoder computes in fp32 too.

    graph=True captures the whole forward (28 single-image + 21 per-plane HIP launches) into one hipGraph per input size and replays
    it: the forward is ~30 launches of a few hundred microseconds each, so Python/launch overhead would otherwise be as long
    as the GPU work.  With a graph the returned tensors are STATIC buffers, overwritten by the next call."""

    def __init__(self, model, encoder_dtype=None, graph=False, encoder=None):
        import os
        dev = next(model.parameters()).device
        if dev.type != "cuda":
            raise _lib.MpiFlowHipError("HipPredictor needs the model on the GPU; there is no CPU path")
        self.model = model.eval()
        self.encoder_dtype = encoder_dtype
        self.fmn = FeatMaskEngine(model.fmn, dev)
        self.dec = DecoderEngine(model.decoder, model.encoder.num_ch_enc, dev, amp_dtype=encoder_dtype)
        # the single-image part: "hip" (default) = EncoderEngine, fp32 HIP kernels; "torch" = the torch modules (MIOpen / ATen), kept for A/B
        self.encoder_kind = encoder or os.environ.get("MPIFLOW_ENCODER", "hip")
        if self.encoder_kind not in ("hip", "torch"):
            raise ValueError("encoder must be 'hip' or 'torch', not %r" % (self.encoder_kind,))
        self.enc = EncoderEngine(model.encoder, model.decoder, dev) if self.encoder_kind == "hip" else None
        self.graph = graph
        self._graphs = {}
        self._side, self._fork, self._join = torch.cuda.Stream(device=dev), torch.cuda.Event(), torch.cuda.Event()
        # constants the torch parts would otherwise copy from the host on every call (not allowed while capturing a graph)
        model.encoder.img_mean = model.encoder.img_mean.to(dev)
        model.encoder.img_std = model.encoder.img_std.to(dev)
        self._plane_disp = model.plane_disparities(torch.zeros(1, device=dev))[0].contiguous()

    @torch.no_grad()
    def _forward(self, src_imgs, src_depths):
        m = self.model
        disp = self._plane_disp
        # the single-image part (encoder, bottleneck: 28 small launches) runs on a side stream underneath the feature-mask
        # network, which fills the GPU on its own; the decoder joins the two
        main = torch.cuda.current_stream()
        self._fork.record(main)
        with torch.cuda.stream(self._side):
            self._side.wait_event(self._fork)
            if self.enc is not None:
                feats, shared = None, self.enc(src_imgs[0], src_depths[0, 0])       # deterministic by construction (fixed split-K order)
            else:
                # MIOpen's default algorithm choice for these batch-1 convolutions is not run-to-run reproducible (1e-4 on the 1/32 feature
                # map, amplified to ~1 % of the output range by a random-weight decoder); its deterministic algorithms are, and the
                # encoder is hidden underneath the feature-mask network either way - so a replayed graph equals an eager run bit for bit
                det = torch.backends.cudnn.deterministic
                torch.backends.cudnn.deterministic = True
                try:
                    with torch.autocast("cuda", dtype=self.encoder_dtype, enabled=self.encoder_dtype is not None):
                        feats = m.encoder(src_imgs, src_depths)
                    shared = self.dec.shared_inputs(feats)
                finally:
                    torch.backends.cudnn.deterministic = det
            self._join.record(self._side)
        masks = plane_masks(self.fmn.logits(src_imgs[0], src_depths[0, 0], disp))
        main.wait_event(self._join)
        for t in [shared[0]] + shared[1]:
            t.record_stream(main)
        raw, cum = self.dec(feats, masks, shared=shared)
        return raw, cum, disp

    def layers(self):
        f, d = self.fmn, self.dec
        out = ([f.l1p, f.l2s, f.l3, f.l4, f.l5, f.l6, f.l7, f.l8s, f.l9] if f.factor else [f.l1, f.l2, f.l3, f.l4, f.l5, f.l6, f.l7, f.l8, f.l9]) + [d.up0[4]]
        for i in range(4, -1, -1):
            if i < 4:
                out.append(d.up0[i])
            out.append(d.up1[i])
        return out + [d.disp0]

    def accounting(self):
        """Per-layer and total algorithmic flops / HBM bytes of the 20 per-plane convolution launches of the last forward (layer_accounting), the
        plane-mask pass (logits read twice, cumulative mask + pyramid written) and - one row - the single-image part on the HIP encoder (its 24
        convolutions: real flops; source, residual, weights read once, outputs written once)."""
        rows = [layer_accounting(L) for L in self.layers() if getattr(L, "last_call", None)]
        c = self.fmn.l9.last_call
        n = c["S"] * c["Hout"] * c["Wout"] * 4
        rows.append(dict(name="plane_masks", flops=0.0, bytes=float(2 * n + n + 2 * n * (1 / 4 + 1 / 16 + 1 / 64 + 1 / 256 + 1 / 1024)), read_bytes=float(2 * n),
                         write_bytes=float(n + 2 * n * (1 / 4 + 1 / 16 + 1 / 64 + 1 / 256 + 1 / 1024))))
        if self.enc is not None:
            convs = [c for c in self.enc.convs() if getattr(c, "last_call", None)]
            rows.append(dict(name="single_image_part", flops=sum(c.flops() for c in convs), bytes=sum(c.last_call["bytes"] for c in convs), launches=len(convs) + 4))
        return rows, dict(flops=sum(r["flops"] for r in rows), bytes=sum(r["bytes"] for r in rows))

    @torch.no_grad()
    def __call__(self, src_imgs, src_depths):
        """(image [1,3,H,W], disparity [1,1,H,W]) -> (raw [S,4,H,W] fp32, cum_mask [S,H,W] fp32, plane disparities [S])"""
        if src_imgs.shape[0] != 1:
            raise ValueError("HipPredictor runs one image at a time (the S planes are the batch)")
        if not self.graph:
            return self._forward(src_imgs, src_depths)
        key = (tuple(src_imgs.shape), src_imgs.device.index)
        if key not in self._graphs:
            with torch.cuda.device(src_imgs.device):
                s_img, s_dsp = src_imgs.float().clone(), src_depths.float().clone()
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):                 # warm-up off the capture: MIOpen picks its kernels, caches fill
                    for _ in range(2):
                        self._forward(s_img, s_dsp)
                torch.cuda.current_stream().wait_stream(side)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    out = self._forward(s_img, s_dsp)
            self._graphs[key] = (g, s_img, s_dsp, out)
        g, s_img, s_dsp, out = self._graphs[key]
        s_img.copy_(src_imgs)
        s_dsp.copy_(src_depths)
        g.replay()
        return out
