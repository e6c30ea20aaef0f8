"""ORACLE (test infrastructure only — never imported by the product path).

Network- and step-level CPU restatement of the reference hot path in plain torch fp32 (torch.nn + autograd):
  * Resnet2D            ganslate/nn/generators/resnet/resnet2d.py:14-93
  * PatchGAN2D          ganslate/nn/discriminators/patchgan/patchgan2d.py:17-66
  * Resnet3D/PatchGAN3D ganslate/nn/generators/resnet/resnet3d.py:14-92, .../patchgan/patchgan3d.py:17-65
  * AdversarialLoss     ganslate/nn/losses/adversarial_loss.py:7-98 (lsgan / vanilla / wgangp)
  * CycleGAN losses     ganslate/nn/losses/cyclegan_losses.py:7-101
  * ImagePool           ganslate/data/utils/image_pool.py:5-60
  * CycleGAN step       ganslate/nn/gans/unpaired/cyclegan.py:92-214 (+ base.py:155-170, nn/utils.py:83-99)
  * training metrics    ganslate/utils/metrics/train_metrics.py:5-67
Module/parameter names equal the reference's, so state_dicts are interchangeable.

PINNING: tests/test_oracle_pinned.py checks this file against golden vectors produced by the REAL reference code
imported in the build container (oracle/gen_golden.py -> tests/golden/*.json). The reference's own test-suite
holds no numeric fixtures for this path (tests/test_first_run.py:24-28 only asserts run() returns None).
"""
import random
from collections import OrderedDict

import torch
from torch import nn

from .ops_ref import ssim_distance


# ---- networks ------------------------------------------------------------------------------------------------------
class _Residual(nn.Module):
    def __init__(self, ch):
        super().__init__()
        self.conv_block = nn.Sequential(
            nn.ReflectionPad2d(1), nn.Conv2d(ch, ch, 3), nn.InstanceNorm2d(ch), nn.ReLU(True),
            nn.ReflectionPad2d(1), nn.Conv2d(ch, ch, 3), nn.InstanceNorm2d(ch))

    def forward(self, x):
        return x + self.conv_block(x)


class Resnet2D(nn.Module):
    def __init__(self, in_channels, out_channels, n_residual_blocks=9):
        super().__init__()
        layers = [nn.ReflectionPad2d(3), nn.Conv2d(in_channels, 64, 7), nn.InstanceNorm2d(64), nn.ReLU(True)]
        ch = 64
        for _ in range(2):
            layers += [nn.Conv2d(ch, ch * 2, 3, stride=2, padding=1), nn.InstanceNorm2d(ch * 2), nn.ReLU(True)]
            ch *= 2
        layers += [_Residual(ch) for _ in range(n_residual_blocks)]
        self.encoder = nn.ModuleList(layers)       # aliases the modules above (duplicated state_dict keys)
        for _ in range(2):
            layers += [nn.ConvTranspose2d(ch, ch // 2, 3, stride=2, padding=1, output_padding=1),
                       nn.InstanceNorm2d(ch // 2), nn.ReLU(True)]
            ch //= 2
        layers += [nn.ReflectionPad2d(3), nn.Conv2d(64, out_channels, 7), nn.Tanh()]
        self.model = nn.Sequential(*layers)

    def forward(self, x):
        return self.model(x)


class PatchGAN2D(nn.Module):
    def __init__(self, in_channels, ndf=64, n_layers=3, kernel_size=4):
        super().__init__()
        kw = kernel_size
        seq = [nn.Conv2d(in_channels, ndf, kw, 2, 1), nn.LeakyReLU(0.2, True)]
        mult = 1
        for n in range(1, n_layers):
            prev, mult = mult, min(2 ** n, 8)
            seq += [nn.Conv2d(ndf * prev, ndf * mult, kw, 2, 1), nn.InstanceNorm2d(ndf * mult), nn.LeakyReLU(0.2, True)]
        prev, mult = mult, min(2 ** n_layers, 8)
        seq += [nn.Conv2d(ndf * prev, ndf * mult, kw, 1, 1), nn.InstanceNorm2d(ndf * mult), nn.LeakyReLU(0.2, True)]
        seq += [nn.Conv2d(ndf * mult, 1, kw, 1, 1)]
        self.model = nn.Sequential(*seq)

    def forward(self, x):
        return self.model(x)


class _Residual3D(nn.Module):
    """ganslate/nn/generators/resnet/resnet3d.py:70-92 restated"""

    def __init__(self, ch):
        super().__init__()
        self.conv_block = nn.Sequential(
            nn.ReplicationPad3d(1), nn.Conv3d(ch, ch, 3), nn.InstanceNorm3d(ch), nn.ReLU(True),
            nn.ReplicationPad3d(1), nn.Conv3d(ch, ch, 3), nn.InstanceNorm3d(ch))

    def forward(self, x):
        return x + self.conv_block(x)


class Resnet3D(nn.Module):
    """ganslate/nn/generators/resnet/resnet3d.py:14-67 restated (replication padding, no `encoder` alias)"""

    def __init__(self, in_channels, out_channels, n_residual_blocks=9):
        super().__init__()
        layers = [nn.ReplicationPad3d(3), nn.Conv3d(in_channels, 64, 7), nn.InstanceNorm3d(64), nn.ReLU(True)]
        ch = 64
        for _ in range(2):
            layers += [nn.Conv3d(ch, ch * 2, 3, stride=2, padding=1), nn.InstanceNorm3d(ch * 2), nn.ReLU(True)]
            ch *= 2
        layers += [_Residual3D(ch) for _ in range(n_residual_blocks)]
        for _ in range(2):
            layers += [nn.ConvTranspose3d(ch, ch // 2, 3, stride=2, padding=1, output_padding=1),
                       nn.InstanceNorm3d(ch // 2), nn.ReLU(True)]
            ch //= 2
        layers += [nn.ReplicationPad3d(3), nn.Conv3d(64, out_channels, 7), nn.Tanh()]
        self.model = nn.Sequential(*layers)

    def forward(self, x):
        return self.model(x)


class PatchGAN3D(nn.Module):
    """ganslate/nn/discriminators/patchgan/patchgan3d.py:17-65 restated"""

    def __init__(self, in_channels, ndf=64, n_layers=3, kernel_size=4):
        super().__init__()
        kw = kernel_size
        seq = [nn.Conv3d(in_channels, ndf, kw, 2, 1), nn.LeakyReLU(0.2, True)]
        mult = 1
        for n in range(1, n_layers):
            prev, mult = mult, min(2 ** n, 8)
            seq += [nn.Conv3d(ndf * prev, ndf * mult, kw, 2, 1), nn.InstanceNorm3d(ndf * mult), nn.LeakyReLU(0.2, True)]
        prev, mult = mult, min(2 ** n_layers, 8)
        seq += [nn.Conv3d(ndf * prev, ndf * mult, kw, 1, 1), nn.InstanceNorm3d(ndf * mult), nn.LeakyReLU(0.2, True)]
        seq += [nn.Conv3d(ndf * mult, 1, kw, 1, 1)]
        self.model = nn.Sequential(*seq)

    def forward(self, x):
        return self.model(x)


class SelfAttentionBlock(nn.Module):
    """ganslate/nn/attention.py:12-47 restated (Self-Attention GAN layer on all D*W*H voxels)"""

    def __init__(self, in_dim):
        super().__init__()
        self.query_conv = nn.Conv3d(in_dim, in_dim // 8, 1)
        self.key_conv = nn.Conv3d(in_dim, in_dim // 8, 1)
        self.value_conv = nn.Conv3d(in_dim, in_dim, 1)
        self.gamma = nn.Parameter(torch.zeros(1))

    def forward(self, x):
        B, Cc = x.shape[:2]
        q = self.query_conv(x).view(B, -1, x[0, 0].numel()).permute(0, 2, 1)
        k = self.key_conv(x).view(B, -1, x[0, 0].numel())
        attention = torch.softmax(torch.bmm(q, k), dim=-1)
        v = self.value_conv(x).view(B, -1, x[0, 0].numel())
        out = torch.bmm(v, attention.permute(0, 2, 1)).view(x.shape)
        return self.gamma * out + x


class SelfAttentionPatchGAN3D(nn.Module):
    """ganslate/nn/discriminators/patchgan/selfattention_patchgan3d.py:18-79 restated"""

    def __init__(self, in_channels, ndf=64, n_layers=3, kernel_size=4):
        super().__init__()
        kw = kernel_size
        seq = [nn.Conv3d(in_channels, ndf, kw, 3, 1), nn.LeakyReLU(0.2, True)]
        mult = 1
        for n in range(1, n_layers):
            prev, mult = mult, min(2 ** n, 8)
            seq += [nn.Conv3d(ndf * prev, ndf * mult, kw, 2, 1), nn.InstanceNorm3d(ndf * mult), nn.LeakyReLU(0.2, True)]
        seq += [SelfAttentionBlock(ndf * mult)]
        prev, mult = mult, min(2 ** n_layers, 8)
        seq += [nn.Conv3d(ndf * prev, ndf * mult, kw, 1, 1), nn.InstanceNorm3d(ndf * mult), nn.LeakyReLU(0.2, True)]
        seq += [SelfAttentionBlock(ndf * mult)]
        seq += [nn.Conv3d(ndf * mult, 1, kw, 1, 1)]
        self.model = nn.Sequential(*seq)

    def forward(self, x):
        return self.model(x)


class MultiScalePatchGAN3D(nn.Module):
    """ganslate/nn/discriminators/patchgan/multiscale_patchgan3d.py:44-60 restated: one PatchGAN3D per scale in an
    nn.ModuleDict keyed "1".."scales"; scale s sees a random window of the input with spatial extents // s (one window per
    call, shared by the batch). The window start is drawn with Python's `random` (one randint per shrinking axis, D-H-W
    order) — see oracle/ref_stubs/monai/transforms for why the reference's own draws cannot be pinned."""

    def __init__(self, in_channels, ndf=64, n_layers=3, kernel_size=4, scales=2):
        super().__init__()
        self.model = nn.ModuleDict({str(s): PatchGAN3D(in_channels, ndf, n_layers, kernel_size)
                                    for s in range(1, scales + 1)})

    @staticmethod
    def crop(x, scale):
        import random
        if scale == 1:
            return x
        sizes = [x.shape[a] // scale for a in (2, 3, 4)]
        st = [random.randint(0, x.shape[a] - n) if x.shape[a] > n else 0 for a, n in zip((2, 3, 4), sizes)]
        return x[:, :, st[0]:st[0] + sizes[0], st[1]:st[1] + sizes[1], st[2]:st[2] + sizes[2]]

    def forward(self, x):
        return {s: m(self.crop(x, int(s))) for s, m in self.model.items()}


class _UnetBlock(nn.Module):
    """ganslate/nn/generators/unet/unet2d.py:80-157 restated"""

    def __init__(self, outer, inner, in_ch=None, sub=None, outermost=False, innermost=False, dropout=False):
        super().__init__()
        self.outermost = outermost
        in_ch = outer if in_ch is None else in_ch
        down = nn.Conv2d(in_ch, inner, 4, 2, 1)
        if outermost:
            mods = [down, sub, nn.ReLU(), nn.ConvTranspose2d(inner * 2, outer, 4, 2, 1), nn.Tanh()]
        elif innermost:
            mods = [nn.LeakyReLU(0.2), down, nn.ReLU(), nn.ConvTranspose2d(inner, outer, 4, 2, 1),
                    nn.InstanceNorm2d(outer)]
        else:
            mods = [nn.LeakyReLU(0.2), down, nn.InstanceNorm2d(inner), sub, nn.ReLU(),
                    nn.ConvTranspose2d(inner * 2, outer, 4, 2, 1), nn.InstanceNorm2d(outer)]
            if dropout:
                mods.append(nn.Dropout(0.5))
        self.model = nn.Sequential(*mods)

    def forward(self, x):
        return self.model(x) if self.outermost else torch.cat([x, self.model(x)], 1)


class Unet2D(nn.Module):
    """ganslate/nn/generators/unet/unet2d.py:17-76 restated"""

    def __init__(self, in_channels, out_channels, num_downs=7, ngf=64, use_dropout=False):
        super().__init__()
        blk = _UnetBlock(ngf * 8, ngf * 8, innermost=True)
        for _ in range(num_downs - 5):
            blk = _UnetBlock(ngf * 8, ngf * 8, sub=blk, dropout=use_dropout)
        blk = _UnetBlock(ngf * 4, ngf * 8, sub=blk)
        blk = _UnetBlock(ngf * 2, ngf * 4, sub=blk)
        blk = _UnetBlock(ngf, ngf * 2, sub=blk)
        self.model = _UnetBlock(out_channels, ngf, in_ch=in_channels, sub=blk, outermost=True)

    def forward(self, x):
        return self.model(x)


class _UnetBlock3D(nn.Module):
    """ganslate/nn/generators/unet/unet3d.py:75-156 restated"""

    def __init__(self, outer, inner, in_ch=None, sub=None, outermost=False, innermost=False, dropout=False):
        super().__init__()
        self.outermost = outermost
        in_ch = outer if in_ch is None else in_ch
        down = nn.Conv3d(in_ch, inner, 4, 2, 1)
        if outermost:
            mods = [down, sub, nn.ReLU(), nn.ConvTranspose3d(inner * 2, outer, 4, 2, 1), nn.Tanh()]
        elif innermost:
            mods = [nn.LeakyReLU(0.2), down, nn.ReLU(), nn.ConvTranspose3d(inner, outer, 4, 2, 1),
                    nn.InstanceNorm3d(outer)]
        else:
            mods = [nn.LeakyReLU(0.2), down, nn.InstanceNorm3d(inner), sub, nn.ReLU(),
                    nn.ConvTranspose3d(inner * 2, outer, 4, 2, 1), nn.InstanceNorm3d(outer)]
            if dropout:
                mods.append(nn.Dropout(0.5))
        self.model = nn.Sequential(*mods)

    def forward(self, x):
        return self.model(x) if self.outermost else torch.cat([x, self.model(x)], 1)


class Unet3D(nn.Module):
    """ganslate/nn/generators/unet/unet3d.py:17-72 restated"""

    def __init__(self, in_channels, out_channels, num_downs=7, ngf=64, use_dropout=False):
        super().__init__()
        blk = _UnetBlock3D(ngf * 8, ngf * 8, innermost=True)
        for _ in range(num_downs - 5):
            blk = _UnetBlock3D(ngf * 8, ngf * 8, sub=blk, dropout=use_dropout)
        blk = _UnetBlock3D(ngf * 4, ngf * 8, sub=blk)
        blk = _UnetBlock3D(ngf * 2, ngf * 4, sub=blk)
        blk = _UnetBlock3D(ngf, ngf * 2, sub=blk)
        self.model = _UnetBlock3D(out_channels, ngf, in_ch=in_channels, sub=blk, outermost=True)

    def forward(self, x):
        return self.model(x)


# ---- Vnet3D (ganslate/nn/generators/vnet/vnet3d.py:27-267 + ganslate/nn/invertible.py:8-48) --------------------------
# memcnn (third-party, unpinned in the reference: setup.cfg:31, absent from the container) is restated from its
# published algorithm: AdditiveCoupling splits channels in two halves, y1 = x1 + Fm(x2), y2 = x2 + Gm(y1), Gm a deep
# copy of Fm; InvertibleModuleWrapper(disable=True) is a plain call. Module names follow memcnn's (`_fn`, `Fm`, `Gm`).
class _AdditiveCoupling(nn.Module):
    def __init__(self, Fm, Gm):
        super().__init__()
        self.Fm, self.Gm = Fm, Gm

    def forward(self, x):
        x1, x2 = torch.chunk(x, 2, dim=1)
        y1 = x1 + self.Fm(x2)
        y2 = x2 + self.Gm(y1)
        return torch.cat([y1, y2], dim=1)

    def inverse(self, y):
        y1, y2 = torch.chunk(y, 2, dim=1)
        x2 = y2 - self.Gm(y1)
        x1 = y1 - self.Fm(x2)
        return torch.cat([x1, x2], dim=1)


class _Wrapper(nn.Module):
    def __init__(self, fn):
        super().__init__()
        self._fn = fn

    def forward(self, x):
        return self._fn(x)

    def inverse(self, y):
        return self._fn.inverse(y)


def _vconv(dims):
    return (nn.Conv2d, nn.ConvTranspose2d, nn.InstanceNorm2d) if dims == 2 else (nn.Conv3d, nn.ConvTranspose3d, nn.InstanceNorm3d)


class _InvertibleBlock(nn.Module):
    def __init__(self, h, dims=3):
        super().__init__()
        Conv, _, Norm = _vconv(dims)
        mk = lambda: nn.Sequential(Conv(h, h, 5, padding=2), Norm(h), nn.PReLU(h))
        self.invertible_block = _Wrapper(_AdditiveCoupling(mk(), mk()))

    def forward(self, x, inverse=False):
        return self.invertible_block.inverse(x) if inverse else self.invertible_block(x)


class _InvertibleSequence(nn.Module):
    def __init__(self, h, n, dims=3):
        super().__init__()
        self.sequence = nn.Sequential(*[_InvertibleBlock(h, dims) for _ in range(n)])

    def forward(self, x, inverse=False):          # invertible.py:36-48
        for block in (reversed(self.sequence) if inverse else self.sequence):
            x = block(x, inverse)
        return x


class _VInput(nn.Module):
    def __init__(self, cin, cout, dims=3):
        super().__init__()
        Conv, _, Norm = _vconv(dims)
        self.n_repeats = cout // cin
        self.conv1 = Conv(cin, cout, 5, padding=2)
        self.bn1 = Norm(cout)
        self.relu = nn.PReLU(cout)

    def forward(self, x):
        return self.relu(self.bn1(self.conv1(x)) + x.repeat(1, self.n_repeats, *([1] * (x.dim() - 2))))


class _VDown(nn.Module):
    def __init__(self, cin, n, dims=3, use_inverse=False):
        super().__init__()
        Conv, _, Norm = _vconv(dims)
        cout = 2 * cin
        self.down_conv_ab = nn.Sequential(Conv(cin, cout, 2, stride=2), Norm(cout), nn.PReLU(cout))
        if use_inverse:
            self.down_conv_ba = nn.Sequential(Conv(cin, cout, 2, stride=2), Norm(cout), nn.PReLU(cout))
        self.core = _InvertibleSequence(cout // 2, n, dims)
        self.relu = nn.PReLU(cout)

    def forward(self, x, inverse=False):
        down = (self.down_conv_ba if inverse else self.down_conv_ab)(x)
        return self.relu(self.core(down, inverse) + down)


class _VUp(nn.Module):
    def __init__(self, cin, cout, n, dims=3, use_inverse=False):
        super().__init__()
        _, ConvT, Norm = _vconv(dims)
        self.up_conv_ab = nn.Sequential(ConvT(cin, cout // 2, 2, stride=2), Norm(cout // 2), nn.PReLU(cout // 2))
        if use_inverse:
            self.up_conv_ba = nn.Sequential(ConvT(cin, cout // 2, 2, stride=2), Norm(cout // 2), nn.PReLU(cout // 2))
        self.core = _InvertibleSequence(cout // 2, n, dims)
        self.relu = nn.PReLU(cout)

    def forward(self, x, skip, inverse=False):
        xcat = torch.cat(((self.up_conv_ba if inverse else self.up_conv_ab)(x), skip), 1)
        return self.relu(self.core(xcat, inverse) + xcat)


class _VOut(nn.Module):
    def __init__(self, cin, cout, dims=3):
        super().__init__()
        Conv, _, Norm = _vconv(dims)
        self.conv1 = Conv(cin, cin, 5, padding=2)
        self.bn1 = Norm(cin)
        self.relu1 = nn.PReLU(cin)
        self.conv2 = Conv(cin, cout, 1)

    def forward(self, x):
        return torch.tanh(self.conv2(self.relu1(self.bn1(self.conv1(x)))))


class Vnet3D(nn.Module):
    """is_separable=False; use_inverse=False is the brats yaml's network, use_inverse=True adds the B -> A copies of the
    non-invertible layers and forward(x, inverse=True) (vnet3d.py:55-150) — memory saving does not change values"""
    dims = 3

    def __init__(self, in_channels, out_channels, first_layer_channels=16, down_blocks=(1, 2, 3, 2),
                 up_blocks=(2, 2, 1, 1), use_inverse=False):
        super().__init__()
        c, dims, inv = first_layer_channels, type(self).dims, use_inverse
        self.use_inverse = inv
        self.in_ab = _VInput(in_channels, c, dims)
        if inv:
            self.in_ba = _VInput(in_channels, c, dims)
        self.out_ab = _VOut(2 * c, out_channels, dims)
        if inv:
            self.out_ba = _VOut(2 * c, out_channels, dims)
        self.downs = nn.ModuleList([_VDown(c * 2 ** i, n, dims, inv) for i, n in enumerate(down_blocks)])
        self.encoder = nn.ModuleList([self.in_ab]).extend(self.downs)
        ucf = [2 * 2 ** i for i in reversed(range(len(down_blocks)))]
        ups = [_VUp(c * ucf[0], c * ucf[0], up_blocks[0], dims, inv)]
        for i, n in enumerate(up_blocks[1:]):
            ups.append(_VUp(c * ucf[i], c * ucf[i + 1], n, dims, inv))
        self.ups = nn.ModuleList(ups)

    def forward(self, x, inverse=False):
        out1 = (self.in_ba if inverse else self.in_ab)(x)
        downs = []
        for i, d in enumerate(self.downs):
            downs.append(d(out1 if i == 0 else downs[-1], inverse))
        rev = list(reversed(downs))
        out = rev[0]
        for i, up in enumerate(self.ups):
            out = up(out, out1 if i == len(self.ups) - 1 else rev[i + 1], inverse)
        return (self.out_ba if inverse else self.out_ab)(out)


class SelfAttentionVnet3D(Vnet3D):
    """ganslate/nn/generators/vnet/selfattention_vnet3d.py:44-181 restated: Vnet3D with a SelfAttentionBlock (or Identity) on
    the output of every down block, feeding both the next down block and the skip connection"""

    def __init__(self, in_channels, out_channels, first_layer_channels=16, down_blocks=(1, 2, 3, 2), up_blocks=(2, 2, 1, 1),
                 use_inverse=False, enable_attention_block=(True, True, True, True)):
        super().__init__(in_channels, out_channels, first_layer_channels, down_blocks, up_blocks, use_inverse)
        ups, enc = self.ups, self.encoder   # registration order of the reference: downs, attn_blocks, encoder, ups
        del self.ups, self.encoder
        self.attn_blocks = nn.ModuleList([SelfAttentionBlock(first_layer_channels * 2 ** i * 2) if on else nn.Identity()
                                          for i, on in enumerate(enable_attention_block)])
        self.encoder = enc
        self.ups = ups

    def forward(self, x, inverse=False):
        out1 = (self.in_ba if inverse else self.in_ab)(x)
        downs = []
        for i, (d, attn) in enumerate(zip(self.downs, self.attn_blocks)):
            downs.append(attn(d(out1 if i == 0 else downs[-1], inverse)))
        rev = list(reversed(downs))
        out = rev[0]
        for i, up in enumerate(self.ups):
            out = up(out, out1 if i == len(self.ups) - 1 else rev[i + 1], inverse)
        return (self.out_ba if inverse else self.out_ab)(out)


class Vnet2D(Vnet3D):
    """ganslate/nn/generators/vnet/vnet2d.py:22-248 with use_inverse=False, use_memory_saving=False: the same network on
    Conv2d / ConvTranspose2d / InstanceNorm2d"""
    dims = 2


class _PiInvBlock(nn.Module):
    def __init__(self, h):
        super().__init__()
        mk = lambda: nn.Sequential(nn.InstanceNorm3d(h), nn.ReplicationPad3d(1), nn.Conv3d(h, h, 3), nn.InstanceNorm3d(h),
                                   nn.ReLU())
        self.invertible_block = _Wrapper(_AdditiveCoupling(mk(), mk()))

    def forward(self, x, inverse=False):
        return self.invertible_block.inverse(x) if inverse else self.invertible_block(x)


class _PiSequence(nn.Module):
    def __init__(self, h, n):
        super().__init__()
        self.sequence = nn.Sequential(*[_PiInvBlock(h) for _ in range(n)])

    def forward(self, x, inverse=False):
        for block in (reversed(self.sequence) if inverse else self.sequence):
            x = block(x, inverse)
        return x


class Piresnet3D(nn.Module):
    """ganslate/nn/generators/resnet/piresnet3d.py:29-108 (memory saving does not change values)"""

    def __init__(self, in_channels, out_channels, depth, first_layer_channels=64, use_inverse=True):
        super().__init__()
        c = first_layer_channels
        down = lambda: nn.Sequential(nn.ReplicationPad3d(2), nn.Conv3d(in_channels, c, 5), nn.InstanceNorm3d(c), nn.ReLU(),
                                     nn.Conv3d(c, 2 * c, 3, stride=2, padding=1), nn.InstanceNorm3d(2 * c), nn.ReLU())
        up = lambda: nn.Sequential(nn.ConvTranspose3d(2 * c, c, 3, stride=2, padding=1, output_padding=1),
                                   nn.InstanceNorm3d(c), nn.ReLU(), nn.ReplicationPad3d(2), nn.Conv3d(c, out_channels, 5),
                                   nn.Tanh())
        self.use_inverse = use_inverse
        self.downconv_ab, self.upconv_ab = down(), up()
        if use_inverse:
            self.downconv_ba, self.upconv_ba = down(), up()
        self.core = _PiSequence(c, depth)

    def forward(self, x, inverse=False):
        d, u = (self.downconv_ba, self.upconv_ba) if inverse else (self.downconv_ab, self.upconv_ab)
        return u(self.core(d(x), inverse))


def seeded_state_dict(module: nn.Module, seed: int, gain=0.02, bias_gain=0.01):
    """Deterministic weights independent of module construction order / torch's default init RNG use:
    every tensor of the state_dict (in key order, aliases share one draw) ~ N(0, gain) (biases N(0, bias_gain))
    from its own torch.Generator. Used for the reference AND the restatement AND the HIP nets."""
    sd, seen = OrderedDict(), {}
    for k, (name, t) in enumerate(module.state_dict().items()):
        if t.data_ptr() in seen:
            sd[name] = seen[t.data_ptr()]
            continue
        g = torch.Generator().manual_seed(seed * 1000 + k)
        v = torch.randn(t.shape, generator=g) * (bias_gain if name.endswith("bias") else gain)
        sd[name] = v
        seen[t.data_ptr()] = v
    return sd


# ---- losses ----------------------------------------------------------------------------------------------------------
def adversarial_loss(pred, target_is_real, mode="lsgan"):
    if mode == "lsgan":
        return ((pred - (1.0 if target_is_real else 0.0)) ** 2).mean()
    if mode == "vanilla":
        t = torch.full_like(pred, 1.0 if target_is_real else 0.0)
        return nn.functional.binary_cross_entropy_with_logits(pred, t)
    if mode == "wgangp":
        return -pred.mean() if target_is_real else pred.mean()
    raise NotImplementedError(mode)


def cycle_loss(real, rec, proportion_ssim):
    l1 = (rec - real).abs().mean()
    if proportion_ssim > 0:
        return proportion_ssim * ssim_distance(rec, real) + (1 - proportion_ssim) * l1
    return l1


class ImagePool:
    def __init__(self, size):
        self.size, self.images = size, []

    def query(self, images):
        if self.size == 0:
            return images
        out = []
        for img in images:
            img = img.detach().unsqueeze(0)
            if len(self.images) < self.size:
                self.images.append(img)
                out.append(img)
            elif random.uniform(0, 1) > 0.5:
                i = random.randint(0, self.size - 1)
                out.append(self.images[i].clone())
                self.images[i] = img
            else:
                out.append(img)
        return torch.cat(out, 0)


# ---- the CycleGAN training step ----------------------------------------------------------------------------
class CycleGANStep:
    """fp32 restatement of CycleGAN.optimize_parameters with the reference's defaults."""

    def __init__(self, in_ch=3, out_ch=3, n_blocks=9, ndf=64, n_layers=3, lr_G=2e-4, lr_D=2e-4, beta1=0.5,
                 beta2=0.999, lambda_AB=10.0, lambda_BA=10.0, lambda_identity=0.0, proportion_ssim=0.0,
                 pool_size=50, adv="lsgan", n_iters=100, n_iters_decay=100, metrics_ssim=True, metrics_D=True,
                 seed=0, dims=2, vnet=None, make_G=None, make_D=None):
        G, D = (Resnet2D, PatchGAN2D) if dims == 2 else (Resnet3D, PatchGAN3D)
        if vnet is not None:          # brats yaml generator: Vnet3D(first_layer_channels, down_blocks, up_blocks)
            G = lambda i, o, _n: Vnet3D(i, o, vnet["first_layer_channels"], tuple(vnet["down_blocks"]),
                                        tuple(vnet["up_blocks"]))
        if make_G is not None:        # any other generator / discriminator pair (the self-attention networks)
            G = lambda i, o, _n: make_G(i, o)
        if make_D is not None:
            D = lambda i, _ndf, _nl: make_D(i)
        self.nets = OrderedDict(G_AB=G(in_ch, out_ch, n_blocks), G_BA=G(out_ch, in_ch, n_blocks),
                                D_B=D(out_ch, ndf, n_layers), D_A=D(in_ch, ndf, n_layers))
        for k, (name, net) in enumerate(self.nets.items()):
            net.load_state_dict(seeded_state_dict(net, seed + k))
        self.hp = dict(lambda_AB=lambda_AB, lambda_BA=lambda_BA, lambda_identity=lambda_identity,
                       proportion_ssim=proportion_ssim, adv=adv, metrics_ssim=metrics_ssim, metrics_D=metrics_D)
        pG = list(self.nets["G_AB"].parameters()) + list(self.nets["G_BA"].parameters())
        pD = list(self.nets["D_B"].parameters()) + list(self.nets["D_A"].parameters())
        self.opt_G = torch.optim.Adam(pG, lr=lr_G, betas=(beta1, beta2))
        self.opt_D = torch.optim.Adam(pD, lr=lr_D, betas=(beta1, beta2))
        rule = lambda it: 1.0 - max(0, it + 1 - n_iters) / float(n_iters_decay + 1)
        self.sched = [torch.optim.lr_scheduler.LambdaLR(o, rule) for o in (self.opt_G, self.opt_D)]
        self.pool_A, self.pool_B = ImagePool(pool_size), ImagePool(pool_size)
        self.visuals = {}

    def _set_D_grad(self, flag):
        for n in ("D_B", "D_A"):
            for p in self.nets[n].parameters():
                p.requires_grad = flag

    def step(self, real_A, real_B):
        hp, nets = self.hp, self.nets
        losses, metrics = {}, {}
        fake_B = nets["G_AB"](real_A); rec_A = nets["G_BA"](fake_B)
        fake_A = nets["G_BA"](real_B); rec_B = nets["G_AB"](fake_A)
        idt_A = idt_B = None
        if hp["lambda_identity"] > 0:
            idt_B = nets["G_AB"](real_B); idt_A = nets["G_BA"](real_A)
        self.visuals = dict(real_A=real_A, real_B=real_B, fake_A=fake_A, fake_B=fake_B, rec_A=rec_A, rec_B=rec_B,
                            idt_A=idt_A, idt_B=idt_B)
        if hp["metrics_ssim"]:
            with torch.no_grad():
                metrics["ssim_A"] = 1 - ssim_distance(real_A, rec_A)
                metrics["ssim_B"] = 1 - ssim_distance(real_B, rec_B)
        # ---- generators ----
        self._set_D_grad(False)
        self.opt_G.zero_grad(set_to_none=True)
        losses["G_AB"] = adversarial_loss(nets["D_B"](fake_B), True, hp["adv"])
        losses["G_BA"] = adversarial_loss(nets["D_A"](fake_A), True, hp["adv"])
        losses["cycle_A"] = hp["lambda_AB"] * cycle_loss(real_A, rec_A, hp["proportion_ssim"])
        losses["cycle_B"] = hp["lambda_BA"] * cycle_loss(real_B, rec_B, hp["proportion_ssim"])
        total = losses["cycle_A"] + losses["cycle_B"]
        if hp["lambda_identity"] > 0:
            losses["idt_B"] = hp["lambda_AB"] * (idt_B - real_B).abs().mean() * hp["lambda_identity"]
            losses["idt_A"] = hp["lambda_BA"] * (idt_A - real_A).abs().mean() * hp["lambda_identity"]
            total = total + losses["idt_B"] + losses["idt_A"]
        (total + losses["G_AB"] + losses["G_BA"]).backward()
        self.opt_G.step()
        # ---- discriminators ----
        self._set_D_grad(True)
        self.opt_D.zero_grad(set_to_none=True)
        for name, real, fake, pool in (("D_B", real_B, fake_B, self.pool_B), ("D_A", real_A, fake_A, self.pool_A)):
            fake = pool.query(fake)
            pred_real = nets[name](real)
            pred_fake = nets[name](fake.detach())
            losses[name] = adversarial_loss(pred_real, True, hp["adv"]) + adversarial_loss(pred_fake, False, hp["adv"])
            losses[name].backward()
            if hp["metrics_D"]:
                metrics[f"{name}_real"] = pred_real.detach().mean()
                metrics[f"{name}_fake"] = pred_fake.detach().mean()
        self.opt_D.step()
        return {k: float(v.detach()) for k, v in losses.items()}, {k: float(v) for k, v in metrics.items()}

    def update_learning_rate(self):
        for s in self.sched:
            s.step()

    def lrs(self):
        return {"lr_G": self.opt_G.param_groups[0]["lr"], "lr_D": self.opt_D.param_groups[0]["lr"]}


class RevGANStep(CycleGANStep):
    """fp32 restatement of RevGAN.optimize_parameters (ganslate/nn/gans/unpaired/revgan.py:89-206): CycleGAN's losses and
    update order with ONE generator used in both directions — G(x) and G(x, inverse=True) — networks G, D_B, D_A (:50-51),
    one Adam over G (:71-76). The generator-side adversarial terms are taken as written in the reference: D_B judges fake_A
    and D_A judges fake_B (:187-193), unlike CycleGAN."""

    def __init__(self, ch=1, ndf=64, n_layers=2, lr_G=2e-4, lr_D=2e-4, beta1=0.5, beta2=0.999, lambda_AB=10.0,
                 lambda_BA=10.0, lambda_identity=0.0, proportion_ssim=0.0, pool_size=50, adv="lsgan", n_iters=100,
                 n_iters_decay=100, metrics_ssim=False, metrics_D=True, seed=0, dims=3, vnet=None, piresnet=None):
        V, D = (Vnet2D, PatchGAN2D) if dims == 2 else (Vnet3D, PatchGAN3D)
        if piresnet is not None:
            G = Piresnet3D(ch, ch, piresnet["depth"], piresnet["first_layer_channels"], use_inverse=True)
        else:
            kw = dict(first_layer_channels=vnet["first_layer_channels"])
            if "down_blocks" in vnet:
                kw.update(down_blocks=tuple(vnet["down_blocks"]), up_blocks=tuple(vnet["up_blocks"]))
            G = V(ch, ch, use_inverse=True, **kw)
        self.nets = OrderedDict(G=G, D_B=D(ch, ndf, n_layers), D_A=D(ch, ndf, n_layers))
        for k, (name, net) in enumerate(self.nets.items()):
            net.load_state_dict(seeded_state_dict(net, seed + k))
        self.hp = dict(lambda_AB=lambda_AB, lambda_BA=lambda_BA, lambda_identity=lambda_identity,
                       proportion_ssim=proportion_ssim, adv=adv, metrics_ssim=metrics_ssim, metrics_D=metrics_D)
        pD = list(self.nets["D_B"].parameters()) + list(self.nets["D_A"].parameters())
        self.opt_G = torch.optim.Adam(self.nets["G"].parameters(), lr=lr_G, betas=(beta1, beta2))
        self.opt_D = torch.optim.Adam(pD, lr=lr_D, betas=(beta1, beta2))
        rule = lambda it: 1.0 - max(0, it + 1 - n_iters) / float(n_iters_decay + 1)
        self.sched = [torch.optim.lr_scheduler.LambdaLR(o, rule) for o in (self.opt_G, self.opt_D)]
        self.pool_A, self.pool_B = ImagePool(pool_size), ImagePool(pool_size)
        self.visuals = {}

    def step(self, real_A, real_B):
        hp, nets = self.hp, self.nets
        G = nets["G"]
        losses, metrics = {}, {}
        fake_B = G(real_A); rec_A = G(fake_B, inverse=True)
        fake_A = G(real_B, inverse=True); rec_B = G(fake_A)
        idt_A = idt_B = None
        if hp["lambda_identity"] > 0:
            idt_B = G(real_B); idt_A = G(real_A, inverse=True)
        self.visuals = dict(real_A=real_A, real_B=real_B, fake_A=fake_A, fake_B=fake_B, rec_A=rec_A, rec_B=rec_B,
                            idt_A=idt_A, idt_B=idt_B)
        if hp["metrics_ssim"]:
            with torch.no_grad():
                metrics["ssim_A"] = 1 - ssim_distance(real_A, rec_A)
                metrics["ssim_B"] = 1 - ssim_distance(real_B, rec_B)
        self._set_D_grad(False)
        self.opt_G.zero_grad(set_to_none=True)
        losses["G_AB"] = adversarial_loss(nets["D_B"](fake_A), True, hp["adv"])      # revgan.py:187,191 (as written)
        losses["G_BA"] = adversarial_loss(nets["D_A"](fake_B), True, hp["adv"])      # revgan.py:188,193
        losses["cycle_A"] = hp["lambda_AB"] * cycle_loss(real_A, rec_A, hp["proportion_ssim"])
        losses["cycle_B"] = hp["lambda_BA"] * cycle_loss(real_B, rec_B, hp["proportion_ssim"])
        total = losses["cycle_A"] + losses["cycle_B"]
        if hp["lambda_identity"] > 0:
            losses["idt_B"] = hp["lambda_AB"] * (idt_B - real_B).abs().mean() * hp["lambda_identity"]
            losses["idt_A"] = hp["lambda_BA"] * (idt_A - real_A).abs().mean() * hp["lambda_identity"]
            total = total + losses["idt_B"] + losses["idt_A"]
        (total + losses["G_AB"] + losses["G_BA"]).backward()
        self.opt_G.step()
        self._set_D_grad(True)
        self.opt_D.zero_grad(set_to_none=True)
        for name, real, fake, pool in (("D_B", real_B, fake_B, self.pool_B), ("D_A", real_A, fake_A, self.pool_A)):
            fake = pool.query(fake)
            pred_real = nets[name](real)
            pred_fake = nets[name](fake.detach())
            losses[name] = adversarial_loss(pred_real, True, hp["adv"]) + adversarial_loss(pred_fake, False, hp["adv"])
            losses[name].backward()
            if hp["metrics_D"]:
                metrics[f"{name}_real"] = pred_real.detach().mean()
                metrics[f"{name}_fake"] = pred_fake.detach().mean()
        self.opt_D.step()
        return {k: float(v.detach()) for k, v in losses.items()}, {k: float(v) for k, v in metrics.items()}


class Pix2PixStep:
    """fp32 restatement of Pix2PixConditionalGAN.optimize_parameters (ganslate/nn/gans/paired/pix2pix.py:76-152):
    G then D; D sees cat([real_A, fake_B], 1); loss_G = LSGAN(D(A, G(A)), 1) + lambda * L1(G(A), B)."""

    def __init__(self, num_downs=7, ngf=64, use_dropout=False, n_layers=3, lr_G=2e-4, lr_D=1e-4, beta1=0.5,
                 beta2=0.999, lambda_pix2pix=100.0, adv="lsgan", n_iters=100, n_iters_decay=100, seed=0):
        self.nets = OrderedDict(G=Unet2D(3, 3, num_downs, ngf, use_dropout), D=PatchGAN2D(6, 64, n_layers))
        for k, (name, net) in enumerate(self.nets.items()):
            net.load_state_dict(seeded_state_dict(net, seed + k))
        self.lam, self.adv = lambda_pix2pix, adv
        self.opt_G = torch.optim.Adam(self.nets["G"].parameters(), lr=lr_G, betas=(beta1, beta2))
        self.opt_D = torch.optim.Adam(self.nets["D"].parameters(), lr=lr_D, betas=(beta1, beta2))
        rule = lambda it: 1.0 - max(0, it + 1 - n_iters) / float(n_iters_decay + 1)
        self.sched = [torch.optim.lr_scheduler.LambdaLR(o, rule) for o in (self.opt_G, self.opt_D)]

    def step(self, real_A, real_B):
        G, Dn = self.nets["G"], self.nets["D"]
        losses, metrics = {}, {}
        fake_B = G(real_A)
        for p in Dn.parameters():
            p.requires_grad = False
        self.opt_G.zero_grad(set_to_none=True)
        losses["G"] = adversarial_loss(Dn(torch.cat([real_A, fake_B], 1)), True, self.adv)
        losses["pix2pix"] = self.lam * (fake_B - real_B).abs().mean()
        (losses["G"] + losses["pix2pix"]).backward()
        self.opt_G.step()
        for p in Dn.parameters():
            p.requires_grad = True
        self.opt_D.zero_grad(set_to_none=True)
        pred_real = Dn(torch.cat([real_A, real_B], 1))
        pred_fake = Dn(torch.cat([real_A, fake_B.detach()], 1))
        losses["D"] = adversarial_loss(pred_real, True, self.adv) + adversarial_loss(pred_fake, False, self.adv)
        losses["D"].backward()
        metrics["D_real"], metrics["D_fake"] = pred_real.detach().mean(), pred_fake.detach().mean()
        self.opt_D.step()
        return {k: float(v.detach()) for k, v in losses.items()}, {k: float(v) for k, v in metrics.items()}

    def update_learning_rate(self):
        for s_ in self.sched:
            s_.step()

    def lrs(self):
        return {"lr_G": self.opt_G.param_groups[0]["lr"], "lr_D": self.opt_D.param_groups[0]["lr"]}


class _PatchMLP(nn.Module):
    """FeaturePatchMLP restated (ganslate/nn/gans/unpaired/cut.py:229-294): same module names (mlps.N.0 / mlps.N.2)"""

    def __init__(self, channels, num_patches=256, nc=256):
        super().__init__()
        self.num_patches = num_patches
        self.mlps = nn.ModuleList(nn.Sequential(nn.Linear(c, nc), nn.ReLU(), nn.Linear(nc, nc)) for c in channels)

    def forward(self, feats, patch_ids=None):
        out, ids = [], []
        for i, feat in enumerate(feats):
            feat = feat.permute(0, 2, 3, 1).flatten(1, 2)
            if patch_ids is not None:
                pid = patch_ids[i]
            else:
                pid = torch.randperm(feat.shape[1], device=feat.device)
                pid = pid[:int(min(self.num_patches, len(pid)))]
            f = self.mlps[i](feat[:, pid, :].flatten(0, 1))
            f = f.div(f.pow(2).sum(1, keepdim=True).pow(0.5) + 1e-7)
            out.append(f)
            ids.append(pid)
        return out, ids


def patch_nce(feat_q, feat_k, batch, T=0.07):
    """PatchNCELoss.forward restated (ganslate/nn/losses/cut_losses.py:14-43)"""
    dim = feat_q.shape[1]
    feat_k = feat_k.detach()
    l_pos = torch.bmm(feat_q.view(-1, 1, dim), feat_k.view(-1, dim, 1)).view(-1, 1)
    q, k = feat_q.view(batch, -1, dim), feat_k.view(batch, -1, dim)
    n = q.size(1)
    l_neg = torch.bmm(q, k.transpose(2, 1))
    l_neg.masked_fill_(torch.eye(n, dtype=torch.bool)[None], -10.0)
    out = torch.cat((l_pos, l_neg.view(-1, n)), 1) / T
    return nn.functional.cross_entropy(out, torch.zeros(out.size(0), dtype=torch.long), reduction="none")


class CUTStep:
    """fp32 restatement of CUT.optimize_parameters (ganslate/nn/gans/unpaired/cut.py:113-226): D first, then G + mlp;
    NCE over encoder features of layers (0, 4, 8, 12, 16) with 256 random patches per level; identity NCE weighted
    by lambda_nce_idt."""

    def __init__(self, batch, n_blocks=9, nce_layers=(0, 4, 8, 12, 16), num_patches=256, mlp_nc=256, lr=2e-4,
                 beta1=0.5, beta2=0.999, lambda_adv=1.0, lambda_nce=1.0, lambda_nce_idt=0.5, nce_T=0.07,
                 n_iters=100, n_iters_decay=100, seed=0):
        self.G, self.D = Resnet2D(3, 3, n_blocks), PatchGAN2D(3, 64, 3)
        self.layers, self.batch, self.T = list(nce_layers), batch, nce_T
        with torch.no_grad():
            f, ch = torch.zeros(1, 3, 64, 64), []
            for i, layer in enumerate(self.G.encoder):
                f = layer(f)
                if i in self.layers:
                    ch.append(f.shape[1])
        self.mlp = _PatchMLP(ch, num_patches, mlp_nc)
        self.nets = OrderedDict(G=self.G, D=self.D, mlp=self.mlp)
        for k, (name, net) in enumerate(self.nets.items()):
            net.load_state_dict(seeded_state_dict(net, seed + k))
        self.lam = (lambda_adv, lambda_nce, lambda_nce_idt)
        self.opts = OrderedDict(G=torch.optim.Adam(self.G.parameters(), lr=lr, betas=(beta1, beta2)),
                                D=torch.optim.Adam(self.D.parameters(), lr=lr, betas=(beta1, beta2)),
                                mlp=torch.optim.Adam(self.mlp.parameters(), lr=lr, betas=(beta1, beta2)))
        rule = lambda it: 1.0 - max(0, it + 1 - n_iters) / float(n_iters_decay + 1)
        self.sched = [torch.optim.lr_scheduler.LambdaLR(o, rule) for o in self.opts.values()]

    def _features(self, x):
        feats, f = [], x
        for i, layer in enumerate(self.G.encoder):
            f = layer(f)
            if i in self.layers:
                feats.append(f)
        return feats

    def _nce(self, source, target):
        sp, ids = self.mlp(self._features(source))
        tp, _ = self.mlp(self._features(target), ids)
        loss = 0
        for t, s_ in zip(tp, sp):
            loss = loss + (patch_nce(t, s_, self.batch, self.T) * self.lam[1]).mean()
        return loss / len(self.layers)

    def step(self, real_A, real_B):
        lam_adv, lam_nce, lam_idt = self.lam
        losses = {}
        fake_B = self.G(real_A)
        idt_B = self.G(real_B) if lam_idt > 0 else None
        for p in self.D.parameters():
            p.requires_grad = True
        self.opts["D"].zero_grad(set_to_none=True)
        losses["D"] = adversarial_loss(self.D(real_B), True) + adversarial_loss(self.D(fake_B.detach()), False)
        losses["D"].backward()
        self.opts["D"].step()
        for p in self.D.parameters():
            p.requires_grad = False
        self.opts["G"].zero_grad(set_to_none=True)
        self.opts["mlp"].zero_grad(set_to_none=True)
        losses["G"] = adversarial_loss(self.D(fake_B), True) * lam_adv
        nce = self._nce(real_A, fake_B)
        losses["NCE"] = nce
        if lam_idt > 0:
            losses["NCE_idt"] = lam_idt * self._nce(real_B, idt_B)
            nce = (1 - lam_idt) * nce + losses["NCE_idt"]
        (losses["G"] + nce).backward()
        self.opts["G"].step()
        self.opts["mlp"].step()
        return {k: float(v.detach()) for k, v in losses.items()}

    def update_learning_rate(self):
        for s_ in self.sched:
            s_.step()
