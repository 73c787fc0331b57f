"""Oracle (test infrastructure): CRAFT detector and CRNN recogniser, torch fp32 on CPU.

Module / parameter names are the upstream ones so a real EasyOCR state-dict
(``craft_mlt_25k.pth``, ``english_g2.pth``) loads with ``strict=True``:
  CRAFT : ``easyocr/craft.py::CRAFT`` + ``easyocr/model/modules.py::vgg16_bn`` /
          ``double_conv``  (keys ``basenet.slice{1..5}.N.*``, ``upconv{1..4}.conv.N.*``,
          ``conv_cls.N.*``)
  CRNN  : ``easyocr/model/vgg_model.py::Model`` + ``model/modules.py::
          {VGG_FeatureExtractor,BidirectionalLSTM}`` (keys ``FeatureExtraction.ConvNet.N.*``,
          ``SequenceModeling.{0,1}.{rnn,linear}.*``, ``Prediction.*``)
The reference reaches them through ``easyocr.Reader(["en"], gpu=...)``
(``pipeline_demo/extractor/enhanced_extractor.py:153``).  PARITY UNPINNED.

``precision="bf16"`` emulates the device data path (operands and every stored
activation rounded to bf16, fp32 accumulation) so tests can separate "bf16
storage" error from kernel bugs; ``precision="fp32"`` is the reference semantics.
"""
from __future__ import annotations

from collections import OrderedDict

import torch
import torch.nn as nn
import torch.nn.functional as F

# torchvision.models.vgg16_bn(...).features layout, indices 0..38 (conv5_3 and the last pool are unused)
_VGG_CFG = [64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512]


def _vgg_features():
    layers, cin = [], 3
    for v in _VGG_CFG:
        if v == "M":
            layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
        else:
            layers += [nn.Conv2d(cin, v, kernel_size=3, padding=1), nn.BatchNorm2d(v), nn.ReLU(inplace=False)]
            cin = v
    return layers


class _Vgg16BN(nn.Module):
    """model/modules.py::vgg16_bn: slices end on a BatchNorm (no ReLU); ReLU opens the next slice."""

    def __init__(self):
        super().__init__()
        feats = _vgg_features()
        self.slice1 = nn.Sequential(OrderedDict((str(i), feats[i]) for i in range(0, 12)))
        self.slice2 = nn.Sequential(OrderedDict((str(i), feats[i]) for i in range(12, 19)))
        self.slice3 = nn.Sequential(OrderedDict((str(i), feats[i]) for i in range(19, 29)))
        self.slice4 = nn.Sequential(OrderedDict((str(i), feats[i]) for i in range(29, 39)))
        self.slice5 = nn.Sequential(
            nn.MaxPool2d(kernel_size=3, stride=1, padding=1),
            nn.Conv2d(512, 1024, kernel_size=3, padding=6, dilation=6),
            nn.Conv2d(1024, 1024, kernel_size=1),
        )

    def forward(self, x):
        h = self.slice1(x); s_relu2_2 = h
        h = self.slice2(h); s_relu3_2 = h
        h = self.slice3(h); s_relu4_3 = h
        h = self.slice4(h); s_relu5_3 = h
        h = self.slice5(h)
        return h, s_relu5_3, s_relu4_3, s_relu3_2, s_relu2_2


class _DoubleConv(nn.Module):
    def __init__(self, in_ch, mid_ch, out_ch):
        super().__init__()
        self.conv = nn.Sequential(
            nn.Conv2d(in_ch + mid_ch, mid_ch, kernel_size=1), nn.BatchNorm2d(mid_ch), nn.ReLU(inplace=False),
            nn.Conv2d(mid_ch, out_ch, kernel_size=3, padding=1), nn.BatchNorm2d(out_ch), nn.ReLU(inplace=False),
        )

    def forward(self, x):
        return self.conv(x)


class CRAFT(nn.Module):
    """easyocr/craft.py::CRAFT.forward -> (y [B,H/2,W/2,2], feature [B,32,H/2,W/2])."""

    def __init__(self):
        super().__init__()
        self.basenet = _Vgg16BN()
        self.upconv1 = _DoubleConv(1024, 512, 256)
        self.upconv2 = _DoubleConv(512, 256, 128)
        self.upconv3 = _DoubleConv(256, 128, 64)
        self.upconv4 = _DoubleConv(128, 64, 32)
        self.conv_cls = nn.Sequential(
            nn.Conv2d(32, 32, kernel_size=3, padding=1), nn.ReLU(inplace=False),
            nn.Conv2d(32, 32, kernel_size=3, padding=1), nn.ReLU(inplace=False),
            nn.Conv2d(32, 16, kernel_size=3, padding=1), nn.ReLU(inplace=False),
            nn.Conv2d(16, 16, kernel_size=1), nn.ReLU(inplace=False),
            nn.Conv2d(16, 2, kernel_size=1),
        )

    def forward(self, x):
        sources = self.basenet(x)
        y = torch.cat([sources[0], sources[1]], dim=1)
        y = self.upconv1(y)
        y = F.interpolate(y, size=sources[2].size()[2:], mode="bilinear", align_corners=False)
        y = torch.cat([y, sources[2]], dim=1)
        y = self.upconv2(y)
        y = F.interpolate(y, size=sources[3].size()[2:], mode="bilinear", align_corners=False)
        y = torch.cat([y, sources[3]], dim=1)
        y = self.upconv3(y)
        y = F.interpolate(y, size=sources[4].size()[2:], mode="bilinear", align_corners=False)
        y = torch.cat([y, sources[4]], dim=1)
        feature = self.upconv4(y)
        y = self.conv_cls(feature)
        return y.permute(0, 2, 3, 1), feature


class _VGGFeatureExtractor(nn.Module):
    def __init__(self, input_channel=1, output_channel=256):
        super().__init__()
        oc = [output_channel // 8, output_channel // 4, output_channel // 2, output_channel]
        self.ConvNet = nn.Sequential(
            nn.Conv2d(input_channel, oc[0], 3, 1, 1), nn.ReLU(False),
            nn.MaxPool2d(2, 2),
            nn.Conv2d(oc[0], oc[1], 3, 1, 1), nn.ReLU(False),
            nn.MaxPool2d(2, 2),
            nn.Conv2d(oc[1], oc[2], 3, 1, 1), nn.ReLU(False),
            nn.Conv2d(oc[2], oc[2], 3, 1, 1), nn.ReLU(False),
            nn.MaxPool2d((2, 1), (2, 1)),
            nn.Conv2d(oc[2], oc[3], 3, 1, 1, bias=False), nn.BatchNorm2d(oc[3]), nn.ReLU(False),
            nn.Conv2d(oc[3], oc[3], 3, 1, 1, bias=False), nn.BatchNorm2d(oc[3]), nn.ReLU(False),
            nn.MaxPool2d((2, 1), (2, 1)),
            nn.Conv2d(oc[3], oc[3], 2, 1, 0), nn.ReLU(False),
        )

    def forward(self, x):
        return self.ConvNet(x)


class _BidirectionalLSTM(nn.Module):
    def __init__(self, input_size, hidden_size, output_size):
        super().__init__()
        self.rnn = nn.LSTM(input_size, hidden_size, bidirectional=True, batch_first=True)
        self.linear = nn.Linear(hidden_size * 2, output_size)

    def forward(self, x):
        rec, _ = self.rnn(x)
        return self.linear(rec)


class CRNN(nn.Module):
    """easyocr/model/vgg_model.py::Model (english_g2: input_channel=1, output_channel=256, hidden=256, 97 classes)."""

    def __init__(self, input_channel=1, output_channel=256, hidden_size=256, num_class=97):
        super().__init__()
        self.FeatureExtraction = _VGGFeatureExtractor(input_channel, output_channel)
        self.AdaptiveAvgPool = nn.AdaptiveAvgPool2d((None, 1))
        self.SequenceModeling = nn.Sequential(
            _BidirectionalLSTM(output_channel, hidden_size, hidden_size),
            _BidirectionalLSTM(hidden_size, hidden_size, hidden_size),
        )
        self.Prediction = nn.Linear(hidden_size, num_class)

    def forward(self, x, text=None):
        v = self.FeatureExtraction(x)
        v = self.AdaptiveAvgPool(v.permute(0, 3, 1, 2)).squeeze(3)
        c = self.SequenceModeling(v)
        return self.Prediction(c.contiguous())


def load_state_dict_any(model: nn.Module, sd: dict):
    """Accept upstream checkpoints with or without the DataParallel ``module.`` prefix."""
    new = OrderedDict()
    for k, v in sd.items():
        new[k[7:] if k.startswith("module.") else k] = v
    model.load_state_dict(new, strict=True)
    return model
