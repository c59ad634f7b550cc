"""AdaMPI predictor: RGBD ResNet-18 encoder, per-plane feature-mask UNet, gated-conv decoder.

Architecture and parameter names follow the reference so its checkpoints load with strict=True:
  MPIPredictor            model/AdaMPI.py:7-78        encoder / fmn / dpn / decoder
  RGBD ResNet-18 encoder  model/CPN/encoder.py:20-101 (torchvision ResNet layout: conv1, bn1, layer1-4, fc)
  FeatMaskNetwork         model/CPN/unet.py:18-69
  DepthDecoder            model/CPN/decoder.py:74-174 (ModuleDict keys are '-'.join(str(tuple)), :75-77)
  DepthPredictionNetwork  model/PAN.py:80-109         present for checkpoint compatibility; bypassed in forward, as in the
                                                      reference (model/AdaMPI.py:70-71): plane disparities are always
                                                      linspace(1, 0.001, S+2)[1:-1]
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


# ---- encoder ------------------------------------------------------------------------------------------------------

class _BasicBlock(nn.Module):
    def __init__(self, c_in, c_out, stride):
        super().__init__()
        self.conv1 = nn.Conv2d(c_in, c_out, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(c_out)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(c_out, c_out, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(c_out)
        self.downsample = None
        if stride != 1 or c_in != c_out:
            self.downsample = nn.Sequential(nn.Conv2d(c_in, c_out, 1, stride, bias=False), nn.BatchNorm2d(c_out))

    def forward(self, x):
        y = self.relu(self.bn1(self.conv1(x)))
        y = self.bn2(self.conv2(y))
        return self.relu(y + (x if self.downsample is None else self.downsample(x)))


class _ResNet18RGBD(nn.Module):
    """torchvision-style ResNet-18 whose first conv takes 4 channels (rgb + disparity)."""

    def __init__(self):
        super().__init__()
        self.conv1 = nn.Conv2d(4, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        widths = [64, 128, 256, 512]
        c_in = 64
        for i, w in enumerate(widths):
            stride = 1 if i == 0 else 2
            setattr(self, "layer%d" % (i + 1), nn.Sequential(_BasicBlock(c_in, w, stride), _BasicBlock(w, w, 1)))
            c_in = w
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(512, 1000)       # unused; exists in the reference's state dict (inherits torchvision ResNet)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)


class ResnetEncoder(nn.Module):
    def __init__(self):
        super().__init__()
        self.num_ch_enc = np.array([64, 64, 128, 256, 512])
        self.encoder = _ResNet18RGBD()
        self.img_mean = torch.tensor([0.485, 0.456, 0.406], dtype=torch.float32).view(1, 3, 1, 1)
        self.img_std = torch.tensor([0.229, 0.224, 0.225], dtype=torch.float32).view(1, 3, 1, 1)

    def forward(self, image, disp):
        e = self.encoder
        x = torch.cat([(image - self.img_mean.to(image)) / self.img_std.to(image), disp], dim=1)
        c1 = e.relu(e.bn1(e.conv1(x)))
        b1 = e.layer1(e.maxpool(c1))
        b2 = e.layer2(b1)
        b3 = e.layer3(b2)
        b4 = e.layer4(b3)
        return [c1, b1, b2, b3, b4]


# ---- feature-mask UNet -------------------------------------------------------------------------------------------------

class _CBR(nn.Module):
    def __init__(self, c_in, c_out, stride):
        super().__init__()
        self.layer = nn.Sequential(nn.Conv2d(c_in, c_out, 3, stride, 1), nn.BatchNorm2d(c_out), nn.ReLU())

    def forward(self, x):
        return self.layer(x)


class FeatMaskNetwork(nn.Module):
    def __init__(self):
        super().__init__()
        spec = [(5, 16, 1), (16, 32, 2), (32, 64, 2), (64, 128, 2), (128, 128, 1), (192, 64, 1), (96, 32, 1), (48, 16, 1), (16, 1, 1)]
        for i, (a, b, s) in enumerate(spec):
            setattr(self, "conv%d" % (i + 1), _CBR(a, b, s))
        self.upsample = nn.Upsample(scale_factor=2, mode="bilinear", align_corners=True)

    def forward(self, image, disp, plane_disp):
        """image [b,3,h,w], disp [b,1,h,w], plane_disp [b,s] -> softmax-over-planes feature mask [b,s,h,w]"""
        _, _, h, w = image.shape
        b, s = plane_disp.shape
        x = torch.cat([image.unsqueeze(1).expand(b, s, 3, h, w), disp.unsqueeze(1).expand(b, s, 1, h, w),
                       plane_disp[:, :, None, None, None].expand(b, s, 1, h, w)], dim=2).reshape(b * s, 5, h, w)
        c1 = self.conv1(x)
        c2 = self.conv2(c1)
        c3 = self.conv3(c2)
        c5 = self.conv5(self.conv4(c3))
        c6 = self.conv6(torch.cat([self.upsample(c5), c3], dim=1))
        c7 = self.conv7(torch.cat([self.upsample(c6), c2], dim=1))
        c8 = self.conv8(torch.cat([self.upsample(c7), c1], dim=1))
        return torch.softmax(self.conv9(c8).reshape(b, s, h, w), dim=1)


# ---- decoder -----------------------------------------------------------------------------------------------------------

class GatedConv(nn.Module):
    def __init__(self, c_in, c_out):
        super().__init__()
        self.pad = nn.ReflectionPad2d(1)
        self.conv2d = nn.Conv2d(c_in, c_out, 3)
        self.mask_conv2d = nn.Conv2d(c_in, c_out, 3)
        self.sigmoid = nn.Sigmoid()

    def forward(self, x):
        x = self.pad(x)
        return self.conv2d(x) * self.sigmoid(self.mask_conv2d(x))


class GatedConvBlock(nn.Module):
    def __init__(self, c_in, c_out):
        super().__init__()
        self.gated_conv = GatedConv(c_in, c_out)
        self.nonlin = nn.ELU(inplace=True)
        self.bn = nn.BatchNorm2d(c_out)

    def forward(self, x):
        return self.nonlin(self.bn(self.gated_conv(x)))


def _conv_bn_lrelu(c_in, c_out, k):
    return nn.Sequential(nn.Conv2d(c_in, c_out, k, 1, (k - 1) // 2, bias=False), nn.BatchNorm2d(c_out), nn.LeakyReLU(0.1, inplace=True))


def _key(*t):
    return "-".join(str(tuple(t)))           # the reference's ModuleDict key scheme (model/CPN/decoder.py:75-77)


class DepthDecoder(nn.Module):
    def __init__(self, num_ch_enc):
        super().__init__()
        top = int(num_ch_enc[-1])
        self.downsample = nn.MaxPool2d(3, stride=2, padding=1)
        self.upsample = nn.UpsamplingNearest2d(scale_factor=2)
        self.conv_down1 = _conv_bn_lrelu(top, 512, 1)
        self.conv_down2 = _conv_bn_lrelu(512, 256, 3)
        self.conv_up1 = _conv_bn_lrelu(256, 256, 3)
        self.conv_up2 = _conv_bn_lrelu(256, top, 1)
        enc = [int(c) + 2 for c in num_ch_enc]           # every feature map gets (context mask, feature mask) appended
        dec = [12, 24, 48, 96, 192]
        self.convs = nn.ModuleDict()
        for i in range(4, -1, -1):
            self.convs[_key("upconv", i, 0)] = GatedConvBlock(enc[-1] if i == 4 else dec[i + 1], dec[i])
            self.convs[_key("upconv", i, 1)] = GatedConvBlock(dec[i] + (enc[i - 1] if i > 0 else 0), dec[i])
        for s in range(4):
            self.convs[_key("dispconv", s)] = GatedConv(dec[s], 4)
        self.sigmoid = nn.Sigmoid()

    @staticmethod
    def _per_plane(feat, context_mask, feature_mask):
        """[B,C,h,w] features -> [B*S,C+2,h,w]: features gated by the context mask, plus both masks at that resolution."""
        B, S = feature_mask.shape[:2]
        _, C, h, w = feat.shape
        cm = F.adaptive_avg_pool2d(context_mask, (h, w)).unsqueeze(2)
        fm = F.adaptive_avg_pool2d(feature_mask, (h, w)).unsqueeze(2)
        f = feat.unsqueeze(1).expand(B, S, C, h, w)
        return torch.cat([f * cm, cm, fm], dim=2).reshape(B * S, C + 2, h, w)

    def forward(self, feats, feature_mask, raw=False):
        """-> mpi [B,S,4,H,W] (rgb = sigmoid, sigma = relu(x*cum_mask)+1e-4); raw=True returns (pre-activation [B,S,4,H,W],
        cum_mask [B,S,H,W]) instead, for the fused epilogue in Stage A+C."""
        B, S = feature_mask.shape[:2]
        top = self.conv_up2(self.upsample(self.conv_up1(self.upsample(self.conv_down2(self.downsample(
            self.conv_down1(self.downsample(feats[-1]))))))))
        cum_mask = torch.cumsum(feature_mask, dim=1)
        context_mask = 1 - torch.cat([torch.zeros_like(cum_mask[:, -1:]), cum_mask[:, :-1]], dim=1)
        x = self._per_plane(top, context_mask, feature_mask)
        skips = [self._per_plane(f, context_mask, feature_mask) for f in feats]
        for i in range(4, -1, -1):
            x = self.convs[_key("upconv", i, 0)](x)
            x = F.interpolate(x, scale_factor=2, mode="nearest")
            if i > 0:
                x = torch.cat([x, skips[i - 1]], dim=1)
            x = self.convs[_key("upconv", i, 1)](x)
        out = self.convs[_key("dispconv", 0)](x)           # only scale 0 is consumed (model/AdaMPI.py:78)
        H, W = out.shape[-2:]
        mpi = out.view(B, S, 4, H, W)
        cur_mask = F.adaptive_avg_pool2d(cum_mask, (H, W))
        if raw:
            return mpi, cur_mask
        rgb = self.sigmoid(mpi[:, :, 0:3])
        sigma = torch.relu(mpi[:, :, 3:] * cur_mask.unsqueeze(2)) + 1e-4
        return torch.cat((rgb, sigma), dim=2)


# ---- plane-adjust network (kept for checkpoint compatibility; bypassed) ---------------------------------------------

class _ResBlock(nn.Module):
    def __init__(self, c_in, c_out, c_hid):
        super().__init__()
        self.conv1 = nn.Conv2d(c_in, c_hid, 3, padding=1)
        self.conv2 = nn.Conv2d(c_hid, c_out, 3, padding=1)
        self.conv3 = nn.Conv2d(c_in, c_out, 1)
        self.activation = nn.ReLU()
        self.bn = nn.BatchNorm2d(c_hid)

    def forward(self, x):
        return self.activation(self.conv3(x) + self.conv2(self.bn(self.activation(self.conv1(x)))))


class _DownsizeEncoder(nn.Module):
    def __init__(self, n, c_in, c_out):
        super().__init__()
        blocks = []
        for i in range(n):
            a = c_in if i == 0 else max(c_in, c_out // (2 ** (n - i)))
            b = max(c_in, c_out // (2 ** (n - i - 1)))
            blocks.append(_ResBlock(a, b, b))
        self.res_blocks = nn.ModuleList(blocks)

    def forward(self, x):
        for blk in self.res_blocks:
            x = F.avg_pool2d(blk(x), 2)
        return x


class _MHSA(nn.Module):
    def __init__(self, heads, c_in, c_qk, c_v):
        super().__init__()
        self.wQs = nn.ModuleList([nn.Linear(c_in, c_qk) for _ in range(heads)])
        self.wKs = nn.ModuleList([nn.Linear(c_in, c_qk) for _ in range(heads)])
        self.wVs = nn.ModuleList([nn.Linear(c_in, c_v // heads) for _ in range(heads)])
        self.fusion = nn.Linear(c_v, c_v)
        self.norm = c_qk ** 0.5

    def forward(self, f):
        out = []
        for q, k, v in zip(self.wQs, self.wKs, self.wVs):
            att = torch.softmax(torch.einsum("bik,bjk->bij", q(f), k(f)) / self.norm, dim=2)
            out.append(torch.einsum("bij,bjc->bic", att, v(f)))
        return self.fusion(torch.cat(out, dim=-1))


class _LinearSigmoid(nn.Module):
    def __init__(self, c_in):
        super().__init__()
        self.linear = nn.Linear(c_in, 1)

    def forward(self, feat, init_disp):
        return init_disp + self.linear(feat).squeeze(-1) * 1.0 / init_disp.shape[1]


class DepthPredictionNetwork(nn.Module):
    def __init__(self):
        super().__init__()
        self.context_encoder = _DownsizeEncoder(5, 5, 128)
        self.self_attention = _MHSA(4, 128, 32, 128)
        self.embed = nn.Sequential(nn.Linear(128, 32), nn.ReLU())
        self.to_disp = _LinearSigmoid(32)

    def forward(self, init_disp, rgb_low, disp_low):
        B, S = init_disp.shape
        h, w = rgb_low.shape[-2:]
        x = torch.cat([rgb_low[:, None].expand(B, S, 3, h, w), disp_low[:, None].expand(B, S, 1, h, w),
                       init_disp[:, :, None, None, None].expand(B, S, 1, h, w)], dim=2).reshape(B * S, 5, h, w)
        ctx = F.adaptive_avg_pool2d(self.context_encoder(x), (1, 1)).reshape(B, S, -1)
        return self.to_disp(self.embed(self.self_attention(ctx)), init_disp)


# ---- top level ----------------------------------------------------------------------------------------------------------

class MPIPredictor(nn.Module):
    """(src_imgs [B,3,H,W], src_depths [B,1,H,W]) -> (mpi [B,S,4,H,W], plane disparities [B,S])   model/AdaMPI.py:55-78"""

    def __init__(self, width=384, height=256, num_planes=64):
        super().__init__()
        self.num_planes = num_planes
        self.far, self.near = 0.001, 1
        self.low_res_size = (int(height / 4), int(width / 4))
        self.encoder = ResnetEncoder()
        self.fmn = FeatMaskNetwork()
        self.dpn = DepthPredictionNetwork()
        self.decoder = DepthDecoder(self.encoder.num_ch_enc)

    def plane_disparities(self, like):
        d = torch.linspace(self.near, self.far, self.num_planes + 2)[1:-1]
        return d.to(like.dtype).to(like.device).unsqueeze(0).repeat(like.shape[0], 1)

    def forward(self, src_imgs, src_depths, raw=False):
        disp = self.plane_disparities(src_imgs)
        feature_mask = self.fmn(src_imgs, src_depths, disp)
        feats = self.encoder(src_imgs, src_depths)
        out = self.decoder(feats, feature_mask, raw=raw)
        if raw:
            return out[0], out[1], disp
        return out, disp

    def randomize_(self, seed=0):
        """Deterministic non-trivial parameters AND BatchNorm statistics (tests / benches; there are no weights offline)."""
        g = torch.Generator().manual_seed(seed)
        with torch.no_grad():
            for n, p in sorted(self.named_parameters()):
                if p.ndim == 1:
                    p.copy_(torch.randn(p.shape, generator=g) * 0.1 + (1.0 if n.endswith("weight") else 0.0))
                else:
                    fan_in = p[0].numel()
                    p.copy_(torch.randn(p.shape, generator=g) * (2.0 / fan_in) ** 0.5)
            for n, b in sorted(self.named_buffers()):
                if n.endswith("running_mean"):
                    b.copy_(torch.randn(b.shape, generator=g) * 0.1)
                elif n.endswith("running_var"):
                    b.copy_(torch.rand(b.shape, generator=g) + 0.5)
        return self

    @classmethod
    def from_checkpoint(cls, path, width, height, map_location="cpu"):
        """Load a reference checkpoint: {'num_planes': S, 'weight': state_dict}   (gen_3dphoto_dynamic_v2.py:52-58)"""
        ckpt = torch.load(path, map_location=map_location)
        model = cls(width=width, height=height, num_planes=ckpt["num_planes"])
        model.load_state_dict(ckpt["weight"])
        return model.eval()
