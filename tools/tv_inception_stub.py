"""Stand-in for `torchvision.models.inception` (torchvision 0.8.2, pinned by the reference's requirements.txt:97-104; NOT installed in this image
and not part of /root/reference) -- used ONLY by tools/make_golden_inception.py to let the reference's own `metric/inception.py` construct
its InceptionV3 (whose FIDInceptionA / C / E_1 / E_2 classes derive from torchvision's blocks and whose `fid_inception_v3()` starts from
`models.inception_v3(num_classes=1008, aux_logits=False, pretrained=False)`).

It restates the PUBLISHED architecture (Szegedy et al., "Rethinking the Inception Architecture for Computer Vision", 2015; layer names and
shapes as the pt_inception-2015-12-05 checkpoint the reference downloads spells them) -- constructor shapes for every block, and the forward
of the blocks the reference does NOT override (BasicConv2d, InceptionB, InceptionD; A / C / E also carry torchvision's forward for
completeness, the reference replaces them).  What a fixture made through this stub pins: the reference's own code (its FID block forwards,
block grouping, input resize / normalisation, output selection, get_activations_from_ims) on top of this restatement of the third-party
constructors.  Nothing under cat_amd/, oracle/, tests/ or bench.py imports this file."""
import torch
import torch.nn as nn
import torch.nn.functional as F


class BasicConv2d(nn.Module):
    def __init__(self, in_channels, out_channels, **kwargs):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, out_channels, bias=False, **kwargs)
        self.bn = nn.BatchNorm2d(out_channels, eps=0.001)

    def forward(self, x):
        return F.relu(self.bn(self.conv(x)), inplace=True)


class InceptionA(nn.Module):
    def __init__(self, in_channels, pool_features, conv_block=None):
        super().__init__()
        cb = conv_block or BasicConv2d
        self.branch1x1 = cb(in_channels, 64, kernel_size=1)
        self.branch5x5_1 = cb(in_channels, 48, kernel_size=1)
        self.branch5x5_2 = cb(48, 64, kernel_size=5, padding=2)
        self.branch3x3dbl_1 = cb(in_channels, 64, kernel_size=1)
        self.branch3x3dbl_2 = cb(64, 96, kernel_size=3, padding=1)
        self.branch3x3dbl_3 = cb(96, 96, kernel_size=3, padding=1)
        self.branch_pool = cb(in_channels, pool_features, kernel_size=1)

    def forward(self, x):
        b1 = self.branch1x1(x)
        b5 = self.branch5x5_2(self.branch5x5_1(x))
        b3 = self.branch3x3dbl_3(self.branch3x3dbl_2(self.branch3x3dbl_1(x)))
        bp = self.branch_pool(F.avg_pool2d(x, kernel_size=3, stride=1, padding=1))
        return torch.cat([b1, b5, b3, bp], 1)


class InceptionB(nn.Module):
    def __init__(self, in_channels, conv_block=None):
        super().__init__()
        cb = conv_block or BasicConv2d
        self.branch3x3 = cb(in_channels, 384, kernel_size=3, stride=2)
        self.branch3x3dbl_1 = cb(in_channels, 64, kernel_size=1)
        self.branch3x3dbl_2 = cb(64, 96, kernel_size=3, padding=1)
        self.branch3x3dbl_3 = cb(96, 96, kernel_size=3, stride=2)

    def forward(self, x):
        b3 = self.branch3x3(x)
        bd = self.branch3x3dbl_3(self.branch3x3dbl_2(self.branch3x3dbl_1(x)))
        bp = F.max_pool2d(x, kernel_size=3, stride=2)
        return torch.cat([b3, bd, bp], 1)


class InceptionC(nn.Module):
    def __init__(self, in_channels, channels_7x7, conv_block=None):
        super().__init__()
        cb = conv_block or BasicConv2d
        c7 = channels_7x7
        self.branch1x1 = cb(in_channels, 192, kernel_size=1)
        self.branch7x7_1 = cb(in_channels, c7, kernel_size=1)
        self.branch7x7_2 = cb(c7, c7, kernel_size=(1, 7), padding=(0, 3))
        self.branch7x7_3 = cb(c7, 192, kernel_size=(7, 1), padding=(3, 0))
        self.branch7x7dbl_1 = cb(in_channels, c7, kernel_size=1)
        self.branch7x7dbl_2 = cb(c7, c7, kernel_size=(7, 1), padding=(3, 0))
        self.branch7x7dbl_3 = cb(c7, c7, kernel_size=(1, 7), padding=(0, 3))
        self.branch7x7dbl_4 = cb(c7, c7, kernel_size=(7, 1), padding=(3, 0))
        self.branch7x7dbl_5 = cb(c7, 192, kernel_size=(1, 7), padding=(0, 3))
        self.branch_pool = cb(in_channels, 192, kernel_size=1)

    def forward(self, x):
        b1 = self.branch1x1(x)
        b7 = self.branch7x7_3(self.branch7x7_2(self.branch7x7_1(x)))
        bd = self.branch7x7dbl_5(self.branch7x7dbl_4(self.branch7x7dbl_3(self.branch7x7dbl_2(self.branch7x7dbl_1(x)))))
        bp = self.branch_pool(F.avg_pool2d(x, kernel_size=3, stride=1, padding=1))
        return torch.cat([b1, b7, bd, bp], 1)


class InceptionD(nn.Module):
    def __init__(self, in_channels, conv_block=None):
        super().__init__()
        cb = conv_block or BasicConv2d
        self.branch3x3_1 = cb(in_channels, 192, kernel_size=1)
        self.branch3x3_2 = cb(192, 320, kernel_size=3, stride=2)
        self.branch7x7x3_1 = cb(in_channels, 192, kernel_size=1)
        self.branch7x7x3_2 = cb(192, 192, kernel_size=(1, 7), padding=(0, 3))
        self.branch7x7x3_3 = cb(192, 192, kernel_size=(7, 1), padding=(3, 0))
        self.branch7x7x3_4 = cb(192, 192, kernel_size=3, stride=2)

    def forward(self, x):
        b3 = self.branch3x3_2(self.branch3x3_1(x))
        b7 = self.branch7x7x3_4(self.branch7x7x3_3(self.branch7x7x3_2(self.branch7x7x3_1(x))))
        bp = F.max_pool2d(x, kernel_size=3, stride=2)
        return torch.cat([b3, b7, bp], 1)


class InceptionE(nn.Module):
    def __init__(self, in_channels, conv_block=None):
        super().__init__()
        cb = conv_block or BasicConv2d
        self.branch1x1 = cb(in_channels, 320, kernel_size=1)
        self.branch3x3_1 = cb(in_channels, 384, kernel_size=1)
        self.branch3x3_2a = cb(384, 384, kernel_size=(1, 3), padding=(0, 1))
        self.branch3x3_2b = cb(384, 384, kernel_size=(3, 1), padding=(1, 0))
        self.branch3x3dbl_1 = cb(in_channels, 448, kernel_size=1)
        self.branch3x3dbl_2 = cb(448, 384, kernel_size=3, padding=1)
        self.branch3x3dbl_3a = cb(384, 384, kernel_size=(1, 3), padding=(0, 1))
        self.branch3x3dbl_3b = cb(384, 384, kernel_size=(3, 1), padding=(1, 0))
        self.branch_pool = cb(in_channels, 192, kernel_size=1)

    def forward(self, x):
        b1 = self.branch1x1(x)
        t = self.branch3x3_1(x)
        b3 = torch.cat([self.branch3x3_2a(t), self.branch3x3_2b(t)], 1)
        t = self.branch3x3dbl_2(self.branch3x3dbl_1(x))
        bd = torch.cat([self.branch3x3dbl_3a(t), self.branch3x3dbl_3b(t)], 1)
        bp = self.branch_pool(F.avg_pool2d(x, kernel_size=3, stride=1, padding=1))
        return torch.cat([b1, b3, bd, bp], 1)


class Inception3(nn.Module):
    def __init__(self, num_classes=1000, aux_logits=True, transform_input=False, **kw):
        super().__init__()
        if aux_logits:
            raise NotImplementedError('the FID network is built with aux_logits=False (metric/inception.py:162-164)')
        self.aux_logits, self.transform_input = aux_logits, transform_input
        self.Conv2d_1a_3x3 = BasicConv2d(3, 32, kernel_size=3, stride=2)
        self.Conv2d_2a_3x3 = BasicConv2d(32, 32, kernel_size=3)
        self.Conv2d_2b_3x3 = BasicConv2d(32, 64, kernel_size=3, padding=1)
        self.maxpool1 = nn.MaxPool2d(kernel_size=3, stride=2)
        self.Conv2d_3b_1x1 = BasicConv2d(64, 80, kernel_size=1)
        self.Conv2d_4a_3x3 = BasicConv2d(80, 192, kernel_size=3)
        self.maxpool2 = nn.MaxPool2d(kernel_size=3, stride=2)
        self.Mixed_5b = InceptionA(192, pool_features=32)
        self.Mixed_5c = InceptionA(256, pool_features=64)
        self.Mixed_5d = InceptionA(288, pool_features=64)
        self.Mixed_6a = InceptionB(288)
        self.Mixed_6b = InceptionC(768, channels_7x7=128)
        self.Mixed_6c = InceptionC(768, channels_7x7=160)
        self.Mixed_6d = InceptionC(768, channels_7x7=160)
        self.Mixed_6e = InceptionC(768, channels_7x7=192)
        self.Mixed_7a = InceptionD(768)
        self.Mixed_7b = InceptionE(1280)
        self.Mixed_7c = InceptionE(2048)
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.dropout = nn.Dropout()
        self.fc = nn.Linear(2048, num_classes)


def inception_v3(pretrained=False, progress=True, **kwargs):
    if pretrained:
        raise RuntimeError('no network in this container')
    return Inception3(**kwargs)
