"""The producer network's PARITY-GRADE engine: MPIPredictor.forward (reference model/AdaMPI.py:55-78) on this repo's HIP kernels in the
arithmetic class of the reference's CPU path - fp32 storage, fp32-grade products, fp32 accumulation in blocks of 64 products carried in fp64
(`dtype=torch.float32`; `x3=True`, the default: every product from the three bf16 pieces each fp32 factor is exactly the sum of, on
v_mfma_f32_16x16x32_bf16; `x3=False`: products on v_mfma_f32_16x16x4_f32) - or in fp64 throughout (`dtype=torch.float64`, v_mfma_f64_16x16x4_f64), the mode the tests use to show that the engine computes the
reference's network itself (error ~1e-12 against the torch modules run in double) and not an approximation of it.

Every convolution is one launch of `mpf_pconv` (mpiflow_amd/csrc/mpf_pconv.hip): the RGBD ResNet-18 encoder and the decoder's bottleneck
(model/CPN/encoder.py:20-101, model/CPN/decoder.py:131-138; S = 1), the feature-mask UNet (model/CPN/unet.py:18-69) and the gated decoder
(model/CPN/decoder.py:10-71, :124-174) over the S plane-images.  The tensors the reference builds around them (expand / cat / bilinear
Upsample / adaptive_avg_pool2d / softmax / cumsum) are small HIP kernels of the same file; x2 nearest up-sampling, concatenation and
padding happen in the convolution's loader.  No MIOpen / ATen kernel runs in the forward.

This is the accuracy mode (`gen_3dphoto_dynamic.py --model-engine hip --model-dtype fp32|fp64`); the fast mode is engine.HipPredictor
(fp16 storage, the precision of the reference's own GPU run).  This module only PACKS parameters (torch CPU, in float64) and sequences
launches; all arithmetic is in the HIP kernels.
"""
import ctypes
import os

import torch

from .. import _lib

EP_AFFINE, EP_AFFINE_MAP, EP_GATED, EP_GATED_PLANAR = 0, 1, 2, 3
DTYPE_CODE = {torch.float32: 0, torch.float64: 1}
X3_CODE, X3_TILE_CODE, X3_CHUNK_CODE = 2, 3, 4      # MPF_DTYPE_F32X3 / _TILE / _CHUNK: fp32 tensors, products from bf16 pieces on the matrix cores
X3_TILE = True                    # tools/bench_precise.py switches the LDS-tile forms off for A/B timings
X3_CHUNK = True
X3_TILE_MAXC = 40                 # input channels up to which the one-pass tile kernel is used (it takes 56; l8's 48 are 10 % faster as two 32-channel chunks)
ACT = {None: 0, "relu": 1, "leaky": 2}


def pad4(c):
    return (int(c) + 3) // 4 * 4


def pad16(c):
    return (int(c) + 15) // 16 * 16


def physical_rows(R, dtype):
    """Physical (packed) row of every LOGICAL row L = 0 .. R-1, block by 16-row block.  fp32: L itself (the C/D layout of
    v_mfma_f32_16x16x4_f32 is row = 4 g + i); fp64: (L >> 2) + 4 (L & 3) inside the block (v_mfma_f64_16x16x4_f64: row = g + 4 i)."""
    L = torch.arange(R)
    if dtype == torch.float64:
        return (L // 16) * 16 + ((L % 16) >> 2) + 4 * (L % 4)
    return L


def pack_weights(w_rows, dtype, device=None, CA=None):
    """[R, Cv, k, k] float64 in LOGICAL row order (R a multiple of 16, Cv a multiple of 4; the first CA virtual channels are source A's, the rest source
    B's; default: one source) -> [R/16, nstA + nstB, 64, 4] of `dtype`: the A-operand order of mpf_pconv (include/mpiflow_hip.h: MpfPConvArgs) - K runs over
    source A's (tap, 4-channel vector) pairs, four per step, then over source B's."""
    R, Cv, k, _ = w_rows.shape
    CA = Cv if CA is None else CA
    assert R % 16 == 0 and Cv % 4 == 0 and CA % 4 == 0 and 0 < CA <= Cv
    phys = torch.empty_like(w_rows)
    phys[physical_rows(R, dtype)] = w_rows
    parts = []
    for c0, c1 in ((0, CA), (CA, Cv)):
        if c1 == c0:
            continue
        nv = k * k * ((c1 - c0) // 4)
        nst = (nv + 3) // 4
        wv = phys[:, c0:c1].permute(0, 2, 3, 1).reshape(R, nv, 4)                                    # [row, v = tap * Vs + c4, j]
        parts.append(torch.cat([wv, torch.zeros(R, nst * 4 - nv, 4, dtype=wv.dtype)], dim=1).reshape(R // 16, 16, nst, 4, 4))   # [blk, m, s, g, j]
    wv = torch.cat(parts, dim=2)
    out = wv.permute(0, 2, 3, 1, 4).reshape(R // 16, wv.shape[2], 64, 4).to(dtype).contiguous()
    return out.to(device) if device is not None else out


def pack_weights_x3(w_rows, device=None, CA=None):
    """The A operand of mpf_pconv's MPF_DTYPE_F32X3 kernels (v_mfma_f32_16x16x32_bf16): the fp32 packing with every source padded to an EVEN number of
    K-steps, two K-steps per instruction, every fp32 weight as the three bf16 numbers it is exactly the sum of ->
    [R/16, steps, 3 pieces, 64 lanes, 8] bfloat16."""
    R, Cv, k, _ = w_rows.shape
    CA = Cv if CA is None else CA
    parts = []
    for c0, c1 in ((0, CA), (CA, Cv)):
        if c1 == c0:
            continue
        w32 = pack_weights(w_rows[:, c0:c1], torch.float32)                                         # [blk, nst, 64, 4], rounded to fp32 as the engine's weights are
        if w32.shape[1] % 2:
            w32 = torch.cat([w32, torch.zeros(w32.shape[0], 1, 64, 4)], dim=1)
        parts.append(w32.reshape(w32.shape[0], -1, 2, 64, 4).permute(0, 1, 3, 2, 4).reshape(w32.shape[0], -1, 64, 8))
    out = _split_bf16x3(torch.cat(parts, dim=1)).permute(1, 2, 0, 3, 4).contiguous()
    return out.to(device) if device is not None else out


def _split_bf16x3(w):
    """fp32 tensor -> [3, ...] bfloat16 with p1 + p2 + p3 == w exactly."""
    p1 = w.to(torch.bfloat16)
    r1 = w - p1.float()
    p2 = r1.to(torch.bfloat16)
    r2 = r1 - p2.float()
    p3 = r2.to(torch.bfloat16)
    assert bool((p1.double() + p2.double() + p3.double() == w.double()).all()), "an fp32 weight is not the sum of its three bf16 pieces"
    return torch.stack([p1, p2, p3])


def pack_weights_x3_tile(w_rows, device=None):
    """The A operand of k_pconv_x3_tile (MPF_DTYPE_F32X3_TILE): 3 x 3 weights [R, Cv, 3, 3] over the CONCATENATED channels of both sources (Cv a multiple
    of 4, zero-padded to a multiple of 8 here); K-vector 4 t + g = (tap, 8-channel vector), tap-major -> [R/16, steps, 3 pieces, 64 lanes, 8] bfloat16."""
    R, Cv, k, _ = w_rows.shape
    assert k == 3 and R % 16 == 0 and Cv % 4 == 0
    V8 = (Cv // 4 + 1) // 2
    w = torch.zeros(R, V8 * 8, 3, 3, dtype=torch.float64)
    w[:, :Cv] = w_rows
    nkv = 9 * V8
    nst = (nkv + 3) // 4
    kv = w.permute(0, 2, 3, 1).reshape(R, nkv, 8)                                                   # [row, tap * V8 + c8, j]
    kv = torch.cat([kv, torch.zeros(R, nst * 4 - nkv, 8, dtype=kv.dtype)], dim=1).reshape(R // 16, 16, nst, 4, 8)   # [blk, m, t, g, j]
    w32 = kv.permute(0, 2, 3, 1, 4).reshape(R // 16, nst, 64, 8).float()
    out = _split_bf16x3(w32).permute(1, 2, 0, 3, 4).contiguous()
    return out.to(device) if device is not None else out


# which of the three taps of one axis fall on the first / second of the two distinct low-resolution pixels a 3-tap window of output phase p covers (x2 nearest)
_PHASE_TAPS = {0: ((0,), (1, 2)), 1: ((0, 1), (2,))}


def pack_weights_x3_tile_phase(w_rows, device=None):
    """k_pconv_x3_tile in phase mode (MpfPConvArgs.up == 2): 3 x 3 weights [R, Cv, 3, 3] of a layer whose ONLY source is x2-nearest up-sampled -> per output phase
    (py, px) the 2 x 2 kernel of SUMS of the nine weights (float64 sums, split into the three bf16 pieces like every other weight); K-vector 4 t + g = (tap 2 ty + tx,
    8-channel vector), tap-major -> [4 phases, R/16, steps, 3 pieces, 64 lanes, 8] bfloat16."""
    R, Cv, k, _ = w_rows.shape
    assert k == 3 and R % 16 == 0 and Cv % 4 == 0
    V8 = (Cv // 4 + 1) // 2
    w = torch.zeros(R, V8 * 8, 3, 3, dtype=torch.float64)
    w[:, :Cv] = w_rows
    nkv = 4 * V8
    nst = (nkv + 3) // 4
    out = []
    for py in (0, 1):
        for px in (0, 1):
            w4 = torch.zeros(R, V8 * 8, 2, 2, dtype=torch.float64)
            for ty in (0, 1):
                for tx in (0, 1):
                    for ky in _PHASE_TAPS[py][ty]:
                        for kx in _PHASE_TAPS[px][tx]:
                            w4[:, :, ty, tx] += w[:, :, ky, kx]
            kv = w4.permute(0, 2, 3, 1).reshape(R, nkv, 8)                                              # [row, tap * V8 + c8, j]
            kv = torch.cat([kv, torch.zeros(R, nst * 4 - nkv, 8, dtype=kv.dtype)], dim=1).reshape(R // 16, 16, nst, 4, 8)
            w32 = kv.permute(0, 2, 3, 1, 4).reshape(R // 16, nst, 64, 8).float()
            out.append(_split_bf16x3(w32).permute(1, 2, 0, 3, 4).contiguous())                          # [R/16, steps, 3, 64, 8]
    out = torch.stack(out).contiguous()
    return out.to(device) if device is not None else out


def pack_weights_x3_chunk(w_rows, device=None):
    """The A operand of k_pconv_x3_chunk (MPF_DTYPE_F32X3_CHUNK): 3 x 3 weights [R, Cv, 3, 3] over the CONCATENATED channels of both sources (Cv a multiple of 4,
    zero-padded to a multiple of 32 here); step = chunk * 9 + tap, lane (m, g) holds row m, channels 32 chunk + 8 g .. + 7 -> [R/16, steps, 3, 64, 8] bfloat16."""
    R, Cv, k, _ = w_rows.shape
    assert k == 3 and R % 16 == 0 and Cv % 4 == 0
    nch = (Cv + 31) // 32
    w = torch.zeros(R, nch * 32, 3, 3, dtype=torch.float64)
    w[:, :Cv] = w_rows
    kv = w.reshape(R // 16, 16, nch, 4, 8, 9).permute(0, 2, 5, 3, 1, 4)                             # [blk, chunk, tap, g, m, j]
    w32 = kv.reshape(R // 16, nch * 9, 64, 8).float()
    out = _split_bf16x3(w32).permute(1, 2, 0, 3, 4).contiguous()
    return out.to(device) if device is not None else out


def _virtual_weights(w, segments):
    """conv weight [Cout, Cin_real, k, k] -> [Cout, sum(padded), k, k] with zero columns at the padding channels of every segment
    (segments: (padded, real) channel counts in concatenation order)."""
    cols, r0 = [], 0
    for padded, real in segments:
        cols.append(w[:, r0:r0 + real])
        if padded > real:
            cols.append(torch.zeros(w.shape[0], padded - real, *w.shape[2:], dtype=w.dtype))
        r0 += real
    assert r0 == w.shape[1], (r0, w.shape)
    return torch.cat(cols, dim=1)


def _bn_affine64(bn):
    scale = bn.weight.detach().double().cpu() / torch.sqrt(bn.running_var.detach().double().cpu() + bn.eps)
    return scale, bn.bias.detach().double().cpu() - bn.running_mean.detach().double().cpu() * scale


class PConv:
    """One packed convolution of the precise engine + its launch."""

    def __init__(self, device, dtype, w_rows, *, epi, scale=None, shift=None, bias=None, ksize=3, stride=1, pad=1, pad_mode=0, up=0, Cst, act=None, slope=0.0,
                 CA, CB=0, name="", rows_real=0, cin_real=0, x3=False):
        self.name, self.dtype, self.epi = name, dtype, epi
        self.ksize, self.stride, self.pad, self.pad_mode, self.up = ksize, stride, pad, pad_mode, up
        self.CA, self.CB, self.Cst, self.act, self.slope = CA, CB, Cst, ACT[act], float(slope)
        assert w_rows.shape[1] == CA + CB and w_rows.shape[0] % 16 == 0
        self.nblk = w_rows.shape[0] // 16
        self.rows_real, self.cin_real = rows_real, cin_real
        if x3 and dtype != torch.float32:
            raise ValueError("the split-bf16 kernels compute on fp32 tensors")
        if ksize > 3:
            x3 = False                                                   # the 7 x 7 stem (one launch, S = 1) stays on the fp32 instruction
        # the few-channel 3 x 3 layers: the input tile split once into LDS (k_pconv_x3_tile) instead of once per tap
        self.tile = bool(x3 and ksize == 3 and stride == 1 and pad == 1 and CA + CB <= X3_TILE_MAXC and self.nblk <= 3 and X3_TILE)
        # every other 3 x 3 / stride 1 layer: the tile split once per 32-channel chunk (k_pconv_x3_chunk) - where the launch has the pixels for it (decided
        # per call, `_form`; both packings are built on first use); strides and 1 x 1 kernels stay on k_pconv_x3
        self.chunk = bool(x3 and not self.tile and ksize == 3 and stride == 1 and pad == 1 and X3_CHUNK)
        self.code = X3_TILE_CODE if self.tile else X3_CODE if x3 else DTYPE_CODE[dtype]
        self._device, self._packs, self.force_chunk = device, {}, None       # force_chunk: True / False overrides the per-call choice (tests)
        # the x2-nearest layer without a second source on the tile form: phase-decomposed (MpfPConvArgs.up == 2; MPIFLOW_PRECISE_PHASE=0 keeps the gather form)
        self.phase = bool(self.tile and up == 1 and CB == 0 and pad_mode == 1 and os.environ.get("MPIFLOW_PRECISE_PHASE", "1") != "0")
        if self.phase:
            self.wpack = pack_weights_x3_tile_phase(w_rows.double(), device)
        elif self.tile:
            self.wpack = pack_weights_x3_tile(w_rows.double(), device)
        else:
            self.wpack = pack_weights_x3(w_rows.double(), device, CA=CA) if x3 else pack_weights(w_rows.double(), dtype, device, CA=CA)
        if self.chunk:
            self._w_rows = w_rows.double().clone()
            self._packs[X3_CODE] = self.wpack
        put = lambda t: None if t is None else t.to(dtype).contiguous().to(device)      # noqa: E731
        self.scale, self.shift, self.bias = put(scale), put(shift), put(bias)

    # -- builders (parameters folded in float64 on the host, rounded once to the engine's dtype) -----------------------------------------
    @classmethod
    def affine(cls, device, dtype, conv, bn, segments, *, act, slope=0.0, up=0, as_map=False, name="", x3=False):
        """Conv2d [+ bias] + BatchNorm(eval) + activation, zero padding: ConvBNReLU (model/CPN/unet.py:6-15), the encoder's conv + bn
        (model/CPN/encoder.py via torchvision's BasicBlock) and the bottleneck's conv + bn + LeakyReLU (model/CPN/decoder.py:85-88)."""
        cout, k = conv.out_channels, conv.kernel_size[0]
        R = pad16(cout)
        w = torch.zeros(R, sum(p for p, _ in segments), k, k, dtype=torch.float64)
        w[:cout] = _virtual_weights(conv.weight.detach().double().cpu(), segments)
        sc, sh = _bn_affine64(bn)
        if conv.bias is not None:
            sh = sh + conv.bias.detach().double().cpu() * sc
        scale, shift = torch.zeros(R, dtype=torch.float64), torch.zeros(R, dtype=torch.float64)
        scale[:cout], shift[:cout] = sc, sh
        return cls(device, dtype, w, epi=EP_AFFINE_MAP if as_map else EP_AFFINE, scale=scale, shift=shift, ksize=k, stride=conv.stride[0], pad=conv.padding[0],
                   pad_mode=0, up=up, Cst=1 if as_map else pad4(cout), act=act, slope=slope, CA=segments[0][0], CB=segments[1][0] if len(segments) > 1 else 0,
                   name=name, rows_real=cout, cin_real=conv.in_channels, x3=x3)

    @classmethod
    def gated(cls, device, dtype, gconv, bn, segments, *, up=0, planar=False, name="", x3=False):
        """GatedConv (+ BatchNorm + ELU when bn is given), reflection padding (model/CPN/decoder.py:10-71): logical rows (2c, 2c+1) = (feature, gate) of
        channel c."""
        cf, cm = gconv.conv2d, gconv.mask_conv2d
        cout = cf.out_channels
        R = pad16(2 * cout)
        w = torch.zeros(R, sum(p for p, _ in segments), 3, 3, dtype=torch.float64)
        w[0:2 * cout:2] = _virtual_weights(cf.weight.detach().double().cpu(), segments)
        w[1:2 * cout:2] = _virtual_weights(cm.weight.detach().double().cpu(), segments)
        bias = torch.zeros(R, dtype=torch.float64)
        bias[0:2 * cout:2], bias[1:2 * cout:2] = cf.bias.detach().double().cpu(), cm.bias.detach().double().cpu()
        scale = shift = None
        if bn is not None:
            scale, shift = torch.zeros(R // 2, dtype=torch.float64), torch.zeros(R // 2, dtype=torch.float64)
            scale[:cout], shift[:cout] = _bn_affine64(bn)
        return cls(device, dtype, w, epi=EP_GATED_PLANAR if planar else EP_GATED, scale=scale, shift=shift, bias=bias, ksize=3, stride=1, pad=1, pad_mode=1, up=up,
                   Cst=cout if planar else pad4(cout), CA=segments[0][0], CB=segments[1][0] if len(segments) > 1 else 0, name=name, rows_real=2 * cout,
                   cin_real=cf.in_channels, x3=x3)

    def _form(self, S, Hout, Wout):
        """(dtype code, packed weights) of this launch.  The chunked LDS-tile kernel works on 8 x 16 pixel tiles, one 32-channel chunk after the other: it wins
        (-10 ... -40 %) where there are tiles enough to fill the GPU and the plane is not mostly tile padding - not on the single-image encoder (a 12 x 40 plane
        of 512 channels is 6 tiles x 16 sequential chunks) and not at 1/32 resolution (up0_4: 0.94 vs 0.66 ms)."""
        if not self.chunk:
            return self.code, self.wpack
        tiles = ((Hout + 7) // 8) * ((Wout + 15) // 16)
        auto = S * tiles * (self.nblk / 4.0) >= 1024 and Hout * Wout >= 0.8 * tiles * 128
        code = X3_CHUNK_CODE if (auto if self.force_chunk is None else self.force_chunk) else X3_CODE
        if code not in self._packs:
            self._packs[code] = pack_weights_x3_chunk(self._w_rows, self._device)
        return code, self._packs[code]

    # -- launch -----------------------------------------------------------------------------------------------------------------------------
    def __call__(self, S, Hin, Win, srcA, srcB=None, residual=None, shareA=False, shareB=False):
        """Hin x Win: the conv's (virtual) input size, after the x2 nearest up-sampling of srcA when up = 1."""
        dev = self.wpack.device
        Hout = (Hin + 2 * self.pad - self.ksize) // self.stride + 1
        Wout = (Win + 2 * self.pad - self.ksize) // self.stride + 1
        HA, WA = Hin >> self.up, Win >> self.up
        assert srcA.dtype == self.dtype and srcA.is_contiguous() and srcA.numel() == (1 if shareA else S) * HA * WA * self.CA, (self.name, tuple(srcA.shape))
        assert (srcB is None) == (self.CB == 0)
        if srcB is not None:
            assert srcB.dtype == self.dtype and srcB.is_contiguous() and srcB.numel() == (1 if shareB else S) * Hin * Win * self.CB, (self.name, tuple(srcB.shape))
        if self.epi == EP_AFFINE_MAP:
            out = torch.empty(S, Hout, Wout, dtype=self.dtype, device=dev)
        elif self.epi == EP_GATED_PLANAR:
            out = torch.empty(S, self.Cst, Hout, Wout, dtype=self.dtype, device=dev)
        else:
            out = torch.empty(S, Hout, Wout, self.Cst, dtype=self.dtype, device=dev)
        a = _lib.MpfPConvArgs()
        p = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())      # noqa: E731
        code, wpack = self._form(S, Hout, Wout)
        a.srcA, a.srcB, a.wpack, a.scale, a.shift, a.bias, a.residual, a.out = p(srcA), p(srcB), p(wpack), p(self.scale), p(self.shift), p(self.bias), p(residual), p(out)
        a.dtype = self.last_code = code
        a.S, a.Hin, a.Win, a.Hout, a.Wout = S, Hin, Win, Hout, Wout
        a.HA, a.WA, a.CA, a.CB = HA, WA, self.CA, self.CB
        a.up, a.shareA, a.shareB = (2 if getattr(self, "phase", False) else self.up), int(shareA), int(shareB)
        a.ksize, a.stride, a.pad, a.pad_mode = self.ksize, self.stride, self.pad, self.pad_mode
        a.nblk, a.Cst, a.epi, a.act, a.slope = self.nblk, self.Cst, self.epi, self.act, self.slope
        with torch.cuda.device(dev):
            stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
            _lib.check(_lib.load().mpf_pconv(ctypes.byref(a), stream), "mpf_pconv(%s)" % self.name)
        es = srcA.element_size()
        self.last_call = dict(flops=2.0 * S * Hout * Wout * self.rows_real * self.cin_real * self.ksize ** 2,
                              bytes=float(es * (srcA.numel() + (srcB.numel() if srcB is not None else 0) + (residual.numel() if residual is not None else 0)
                                                + out.numel() + self.wpack.numel())))
        return out


def _stream(dev):
    return ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


class PrecisePredictor:
    """MPIPredictor.forward(raw=True) (model/AdaMPI.py:55-78) for one image, every layer on mpf_pconv in `dtype` (torch.float32 or torch.float64).

    __call__(image [1,3,H,W], disparity [1,1,H,W]) -> (raw [S,4,H,W], cum_mask [S,H,W], plane disparities [S]) - the hand-off
    mpf_src_blend_flow(..., d_cum_mask) finishes in registers (sigmoid / relu(x * cum) + 1e-4, model/CPN/decoder.py:166-173).  The returned raw /
    cum tensors are fp32 unless `keep_dtype` (tests compare the fp64 engine in double)."""

    DEC = [12, 24, 48, 96, 192]

    def __init__(self, model, dtype=torch.float32, keep_dtype=False, x3=None):
        dev = next(model.parameters()).device
        if dev.type != "cuda":
            raise _lib.MpiFlowHipError("PrecisePredictor needs the model on the GPU; there is no CPU path")
        if dtype not in DTYPE_CODE:
            raise ValueError("dtype must be torch.float32 or torch.float64")
        x3 = (dtype == torch.float32) if x3 is None else bool(x3)
        if x3 and dtype != torch.float32:
            raise ValueError("x3 (products from bf16 pieces on the matrix cores) goes with dtype torch.float32")
        self.model, self.dtype, self.dev, self.keep_dtype, self.x3 = model.eval(), dtype, dev, keep_dtype, x3
        self.code = DTYPE_CODE[dtype]
        A = lambda *a, **k: PConv.affine(dev, dtype, *a, x3=x3, **k)          # noqa: E731
        G = lambda *a, **k: PConv.gated(dev, dtype, *a, x3=x3, **k)           # noqa: E731
        # ---- single-image part: ResNet-18 encoder + bottleneck
        e = model.encoder.encoder
        self.e_conv1 = A(e.conv1, e.bn1, [(4, 4)], act="relu", name="enc.conv1")
        self.e_blocks = []
        for li in range(1, 5):
            for bi, blk in enumerate(getattr(e, "layer%d" % li)):
                n = "enc.layer%d.%d" % (li, bi)
                cin, cout = blk.conv1.in_channels, blk.conv1.out_channels
                ds = None if blk.downsample is None else A(blk.downsample[0], blk.downsample[1], [(cin, cin)], act=None, name=n + ".downsample")
                self.e_blocks.append((A(blk.conv1, blk.bn1, [(cin, cin)], act="relu", name=n + ".conv1"),
                                      A(blk.conv2, blk.bn2, [(cout, cout)], act="relu", name=n + ".conv2"), ds))
        d = model.decoder
        L = lambda seq, up, n: A(seq[0], seq[1], [(seq[0].in_channels, seq[0].in_channels)], act="leaky", slope=seq[2].negative_slope, up=up, name=n)   # noqa: E731
        self.d_down1, self.d_down2 = L(d.conv_down1, 0, "dec.conv_down1"), L(d.conv_down2, 0, "dec.conv_down2")
        self.d_up1, self.d_up2 = L(d.conv_up1, 1, "dec.conv_up1"), L(d.conv_up2, 1, "dec.conv_up2")
        # ---- feature-mask UNet
        f = model.fmn
        C = lambda cbr, segs, name, **k: A(cbr.layer[0], cbr.layer[1], segs, act="relu", name=name, **k)       # noqa: E731
        self.l1 = C(f.conv1, [(8, 5)], "l1")
        self.l2 = C(f.conv2, [(16, 16)], "l2")
        self.l3 = C(f.conv3, [(32, 32)], "l3")
        self.l4 = C(f.conv4, [(64, 64)], "l4")
        self.l5 = C(f.conv5, [(128, 128)], "l5")
        self.l6 = C(f.conv6, [(128, 128), (64, 64)], "l6")
        self.l7 = C(f.conv7, [(64, 64), (32, 32)], "l7")
        self.l8 = C(f.conv8, [(32, 32), (16, 16)], "l8")
        self.l9 = C(f.conv9, [(16, 16)], "l9", as_map=True)
        # ---- gated decoder
        enc = [int(c) for c in model.encoder.num_ch_enc]
        key = lambda *t: "-".join(str(tuple(t)))                       # noqa: E731
        dec = self.DEC
        self.up0, self.up1 = {}, {}
        for i in range(4, -1, -1):
            b0, b1 = d.convs[key("upconv", i, 0)], d.convs[key("upconv", i, 1)]
            if i == 4:
                self.up0[i] = G(b0.gated_conv, b0.bn, [(enc[4] + 4, enc[4] + 2)], name="up0_4")
            else:
                self.up0[i] = G(b0.gated_conv, b0.bn, [(dec[i + 1], dec[i + 1])], name="up0_%d" % i)
            if i > 0:
                self.up1[i] = G(b1.gated_conv, b1.bn, [(dec[i], dec[i]), (enc[i - 1] + 4, enc[i - 1] + 2)], up=1, name="up1_%d" % i)
            else:
                self.up1[i] = G(b1.gated_conv, b1.bn, [(dec[0], dec[0])], up=1, name="up1_0")
        self.disp0 = G(d.convs[key("dispconv", 0)], None, [(dec[0], dec[0])], planar=True, name="disp0")
        self._plane_disp = model.plane_disparities(torch.zeros(1, device=dev))[0].contiguous()
        self._side = torch.cuda.Stream(device=dev)
        self.debug = None                                              # set to a dict to keep intermediate tensors (tests)

    # -- small kernels ----------------------------------------------------------------------------------------------------------------------
    def _empty(self, *shape):
        return torch.empty(*shape, dtype=self.dtype, device=self.dev)

    def _maxpool(self, x):
        h, w, c = x.shape
        out = self._empty((h - 1) // 2 + 1, (w - 1) // 2 + 1, c)
        _lib.check(_lib.load().mpf_pmaxpool3x3s2(ctypes.c_void_p(x.data_ptr()), h, w, c, ctypes.c_void_p(out.data_ptr()), self.code, _stream(self.dev)), "mpf_pmaxpool3x3s2")
        return out

    def _bilinear2x(self, x):
        S, h, w, c = x.shape
        out = self._empty(S, 2 * h, 2 * w, c)
        _lib.check(_lib.load().mpf_pbilinear2x(ctypes.c_void_p(x.data_ptr()), S, h, w, c, ctypes.c_void_p(out.data_ptr()), self.code, _stream(self.dev)), "mpf_pbilinear2x")
        return out

    def _per_plane(self, feat, cm, fm):
        h, w, c = feat.shape
        S = cm.shape[0]
        assert tuple(cm.shape) == (S, h, w) and tuple(fm.shape) == (S, h, w), (tuple(cm.shape), (S, h, w))
        out = self._empty(S, h, w, c + 4)
        _lib.check(_lib.load().mpf_pper_plane(ctypes.c_void_p(feat.data_ptr()), ctypes.c_void_p(cm.data_ptr()), ctypes.c_void_p(fm.data_ptr()), S, h, w, c,
                                              ctypes.c_void_p(out.data_ptr()), self.code, _stream(self.dev)), "mpf_pper_plane")
        return out

    def plane_masks(self, logits):
        S, H, W = logits.shape
        fmask, cum, ctx = self._empty(S, H, W), self._empty(S, H, W), self._empty(S, H, W)
        cm = [self._empty(S, H >> k, W >> k) for k in range(1, 6)]
        fm = [self._empty(S, H >> k, W >> k) for k in range(1, 6)]
        arr = lambda ts: (ctypes.c_void_p * 5)(*[t.data_ptr() for t in ts])            # noqa: E731
        _lib.check(_lib.load().mpf_pplane_masks(ctypes.c_void_p(logits.data_ptr()), S, H, W, ctypes.c_void_p(fmask.data_ptr()), ctypes.c_void_p(cum.data_ptr()),
                                                ctypes.c_void_p(ctx.data_ptr()), arr(cm), arr(fm), self.code, _stream(self.dev)), "mpf_pplane_masks")
        return dict(fmask=fmask, cum=cum, ctx=ctx, cm=cm, fm=fm)

    # -- the three parts ------------------------------------------------------------------------------------------------------------------------
    def encoder(self, img_3HW, dsp_HW):
        """-> ([c1, b1, b2, b3, b4] NHWC, top [H/32,W/32,512])   model/CPN/encoder.py:86-101, model/CPN/decoder.py:131-138"""
        H, W = dsp_HW.shape
        x = self._empty(H, W, 4)
        _lib.check(_lib.load().mpf_pencoder_input(ctypes.c_void_p(img_3HW.data_ptr()), ctypes.c_void_p(dsp_HW.data_ptr()), H, W, ctypes.c_void_p(x.data_ptr()), self.code,
                                                  _stream(self.dev)), "mpf_pencoder_input")
        c1 = self.e_conv1(1, H, W, x)[0]
        feats = [c1]
        x = self._maxpool(c1)
        for i, (ca, cb, ds) in enumerate(self.e_blocks):
            h, w, _ = x.shape
            identity = x if ds is None else ds(1, h, w, x)
            y = ca(1, h, w, x)[0]
            x = cb(1, y.shape[0], y.shape[1], y, residual=identity)[0]
            if i % 2 == 1:
                feats.append(x)
        t = self._maxpool(x)
        t = self.d_down1(1, t.shape[0], t.shape[1], t)[0]
        t = self._maxpool(t)
        t = self.d_down2(1, t.shape[0], t.shape[1], t)[0]
        t = self.d_up1(1, 2 * t.shape[0], 2 * t.shape[1], t)[0]
        top = self.d_up2(1, 2 * t.shape[0], 2 * t.shape[1], t)[0]
        return feats, top

    def logits(self, img_3HW, dsp_HW, pd):
        """FeatMaskNetwork.forward up to the softmax (model/CPN/unet.py:44-67) -> [S,H,W]"""
        S = pd.numel()
        H, W = dsp_HW.shape
        x0 = self._empty(S, H, W, 8)
        _lib.check(_lib.load().mpf_pfmn_input(ctypes.c_void_p(img_3HW.data_ptr()), ctypes.c_void_p(dsp_HW.data_ptr()), ctypes.c_void_p(pd.data_ptr()), S, H, W,
                                              ctypes.c_void_p(x0.data_ptr()), self.code, _stream(self.dev)), "mpf_pfmn_input")
        c1 = self.l1(S, H, W, x0)
        del x0
        c2 = self.l2(S, H, W, c1)
        c3 = self.l3(S, H // 2, W // 2, c2)
        c4 = self.l4(S, H // 4, W // 4, c3)
        c5 = self.l5(S, H // 8, W // 8, c4)
        del c4
        c6 = self.l6(S, H // 4, W // 4, self._bilinear2x(c5), c3)
        del c5, c3
        c7 = self.l7(S, H // 2, W // 2, self._bilinear2x(c6), c2)
        del c6, c2
        c8 = self.l8(S, H, W, self._bilinear2x(c7), c1)
        del c7, c1
        return self.l9(S, H, W, c8)

    def decoder(self, feats, top, masks):
        """DepthDecoder.forward from the per-plane expansion on (model/CPN/decoder.py:140-165) -> raw [S,4,H,W]"""
        S, H, W = masks["cum"].shape
        h, w = top.shape[:2]
        if (h * 32, w * 32) != (H, W):
            raise ValueError("bottleneck output is %s, expected %s" % ((h, w), (H // 32, W // 32)))
        x = self.up0[4](S, h, w, self._per_plane(top, masks["cm"][4], masks["fm"][4]))
        for i in range(4, -1, -1):
            if i < 4:
                x = self.up0[i](S, h, w, x)
            h, w = 2 * h, 2 * w
            if i > 0:
                x = self.up1[i](S, h, w, x, self._per_plane(feats[i - 1], masks["cm"][i - 1], masks["fm"][i - 1]))
            else:
                x = self.up1[i](S, h, w, x)
        return self.disp0(S, h, w, x)

    @torch.no_grad()
    def __call__(self, src_imgs, src_depths):
        if src_imgs.shape[0] != 1:
            raise ValueError("PrecisePredictor runs one image at a time (the S planes are the batch)")
        H, W = src_imgs.shape[-2:]
        for n in (H, W):
            # five x2 scales down to 1/32, then the bottleneck's two stride-2 pools and two x2 up-samplings must return to that size - the
            # reference's own constraint (its torch.cat of the decoder would fail otherwise)
            pooled = ((n // 32 - 1) // 2 + 1 - 1) // 2 + 1
            if n % 32 or pooled * 4 != n // 32:
                raise ValueError("H and W must be multiples of 32 whose 1/32 size survives two stride-2 pools and two x2 up-samplings, got %d" % n)
        with torch.cuda.device(self.dev):
            img, dsp = src_imgs[0].float().contiguous(), src_depths[0, 0].float().contiguous()
            pd = self._plane_disp
            # the single-image part (ResNet-18 + bottleneck: ~30 launches of a few dozen workgroups) runs beside the S-plane UNet on a second stream
            main = torch.cuda.current_stream(self.dev)
            self._side.wait_stream(main)
            with torch.cuda.stream(self._side):
                feats, top = self.encoder(img, dsp)
            lg = self.logits(img, dsp, pd)
            masks = self.plane_masks(lg)
            main.wait_stream(self._side)
            for t in feats + [top]:
                t.record_stream(main)
            raw = self.decoder(feats, top, masks)
        if self.debug is not None:
            self.debug.update(feats=feats, top=top, logits=lg, masks=masks)
        cum = masks["cum"]
        if not self.keep_dtype and self.dtype != torch.float32:
            raw, cum = raw.float(), cum.float()
        return raw, cum, pd

    def layers(self):
        out = [self.e_conv1]
        for ca, cb, ds in self.e_blocks:
            out += [c for c in (ds, ca, cb) if c is not None]
        out += [self.d_down1, self.d_down2, self.d_up1, self.d_up2, self.l1, self.l2, self.l3, self.l4, self.l5, self.l6, self.l7, self.l8, self.l9, self.up0[4]]
        for i in range(4, -1, -1):
            if i < 4:
                out.append(self.up0[i])
            out.append(self.up1[i])
        return out + [self.disp0]

    def accounting(self):
        """Per-launch and total algorithmic flops (real channels) and bytes (materialised sources read once, output written once) of the last forward."""
        rows = [dict(name=L.name, **L.last_call) for L in self.layers() if getattr(L, "last_call", None)]
        return rows, dict(flops=sum(r["flops"] for r in rows), bytes=sum(r["bytes"] for r in rows))
