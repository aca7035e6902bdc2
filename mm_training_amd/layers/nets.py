"""Dense building blocks the reference imports from mmdet / mmdet3d / mmcv
(README.md:14-27; none vendored), restated in plain PyTorch so PyTorch-ROCm runs them
on MIOpen / hipBLASLt (MFMA).  They are the CALLERS' side of the hot path, not
hand-written kernels (out of scope for HIP).

  ResNet        mmdet ResNet (depth 18/50, base_channels, num_stages, strides, out_indices)
                -- lss_fpn.py:293 (image backbone), bev_depth_head.py:79 (BEV trunk)
  SECONDFPN     mmdet3d SECONDFPN -- lss_fpn.py:294, bev_depth_head.py:81
  DeformConv2dPack  mmcv 'DCN' (lss_fpn.py:189-197) as bilinear grid_sample + grouped GEMM
"""
import torch
import torch.nn.functional as F
from torch import nn

from ..ops.bn_relu import ConvBNAct, bn_act


class BasicBlock(nn.Module):
    expansion = 1
    fork_output = False        # True (set by ResNet for a block that another block of its stage follows): the output as a pair of aliases

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        x, x_res = x[:2] if isinstance(x, tuple) else (x, x)      # (a forked block output: see ResNet.forward)
        identity = x_res if self.downsample is None else self.downsample(x_res)
        out = bn_act(self.bn1, self.conv1(x))
        return bn_act(self.bn2, self.conv2(out), residual=identity, fork=self.fork_output)


class Bottleneck(nn.Module):
    expansion = 4
    fork_output = False        # (see BasicBlock)

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        # mmdet style='pytorch': the stride sits on the 3x3 conv
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        x, x_res = x[:2] if isinstance(x, tuple) else (x, x)      # (a forked block output: see ResNet.forward)
        identity = x_res if self.downsample is None else self.downsample(x_res)
        out = bn_act(self.bn1, self.conv1(x))
        out = bn_act(self.bn2, self.conv2(out))
        return bn_act(self.bn3, self.conv3(out), residual=identity, fork=self.fork_output)


class ResNet(nn.Module):
    ARCH = {18: (BasicBlock, (2, 2, 2, 2)), 34: (BasicBlock, (3, 4, 6, 3)), 50: (Bottleneck, (3, 4, 6, 3))}

    def __init__(self, depth=50, in_channels=3, base_channels=64, num_stages=4, strides=(1, 2, 2, 2),
                 dilations=(1, 1, 1, 1), out_indices=(0, 1, 2, 3), frozen_stages=-1, norm_eval=False, init_cfg=None, pretrained=None, **unused):
        super().__init__()
        # whether weights from a checkpoint have reached this module (load_state_dict / init_cfg / pretrained): a FROZEN stem
        # (frozen_stages >= 0, the reference's configuration) that still holds its random initialisation is a model-quality trap --
        # conv1 / bn1 never train and bn1 normalises with identity statistics -- so the first training forward warns about it
        self._weights_loaded = False
        self._warned_frozen_random = False
        block, blocks = self.ARCH[depth]
        self.out_indices = tuple(out_indices)
        # mmdet ResNet (the reference's image backbone is built with frozen_stages=0, norm_eval=False: exps/conf_aim.py:57-59):
        # frozen_stages >= 0 freezes the stem (conv1 + norm1: no gradients, norm1 in eval mode), >= i also stage i; norm_eval keeps
        # every BatchNorm in eval mode while training.  `train()` re-applies both, as mmdet's does.
        self.frozen_stages = int(frozen_stages)
        self.norm_eval = bool(norm_eval)
        self.conv1 = nn.Conv2d(in_channels, base_channels, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(base_channels)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        self.stages = nn.ModuleList()
        inplanes = base_channels
        for i in range(num_stages):
            planes = base_channels * 2 ** i
            stride = strides[i]
            layers = []
            for j in range(blocks[i]):
                s = stride if j == 0 else 1
                down = None
                if s != 1 or inplanes != planes * block.expansion:
                    down = ConvBNAct(nn.Conv2d(inplanes, planes * block.expansion, 1, s, bias=False),
                                         nn.BatchNorm2d(planes * block.expansion))
                layers.append(block(inplanes, planes, s, down))
                # another block reads the output twice (first convolution; identity or downsample), the neck a stage's output once more
                layers[-1].fork_output = 2 if j + 1 < blocks[i] else ((3 if i in self.out_indices else 2) if i + 1 < num_stages else False)
                inplanes = planes * block.expansion
            self.stages.append(nn.Sequential(*layers))
        self.init_weights()
        self._freeze_stages()
        ckpt = pretrained
        if ckpt is None and isinstance(init_cfg, dict) and init_cfg.get("type") == "Pretrained":
            ckpt = init_cfg.get("checkpoint")            # exps/conf_aim.py:60: dict(type='Pretrained', checkpoint='torchvision://resnet50')
        if ckpt:
            self.load_pretrained(ckpt)

    def init_weights(self):
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.ones_(m.weight)
                nn.init.zeros_(m.bias)

    # torchvision / mmdet name the stages layer1..layer4; this module keeps them in `stages` (0-based).  Everything below a stage has
    # the same names (N.conv1, N.bn1, N.downsample.0 / .1), so a checkpoint of either family loads after this one substitution.
    @staticmethod
    def _checkpoint_key(key):
        import re
        m = re.match(r"layer([1-9])\.(.*)", key)
        return f"stages.{int(m.group(1)) - 1}.{m.group(2)}" if m else key

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        for key in [k for k in state_dict if k.startswith(prefix)]:
            local = key[len(prefix):]
            if local.startswith("fc."):                  # the classifier of an ImageNet checkpoint: not part of a backbone
                state_dict.pop(key)
                continue
            mapped = self._checkpoint_key(local)
            if mapped != local:
                state_dict[prefix + mapped] = state_dict.pop(key)
        if any(k.startswith(prefix) for k in state_dict):
            self._weights_loaded = True
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs)

    def load_pretrained(self, checkpoint):
        """Weights of a torchvision / mmdet ResNet checkpoint file.  `torchvision://resnet50`-style names (what the reference's
        configuration carries) cannot be fetched here: they are looked up as $MMT_PRETRAINED_DIR/<name>.pth; a name that resolves
        to no file leaves the random initialisation and says so."""
        import os
        import warnings
        path = checkpoint
        if "://" in checkpoint:
            root = os.environ.get("MMT_PRETRAINED_DIR")
            path = os.path.join(root, checkpoint.split("://", 1)[1] + ".pth") if root else None
        if not path or not os.path.exists(path):
            warnings.warn(f"ResNet: pretrained checkpoint {checkpoint!r} not found"
                          + ("" if path else " (set MMT_PRETRAINED_DIR to a directory holding <name>.pth)") + "; keeping the random initialisation")
            return False
        sd = torch.load(path, map_location="cpu")
        sd = sd.get("state_dict", sd) if isinstance(sd, dict) else sd
        sd = {(k[len("backbone."):] if k.startswith("backbone.") else k): v for k, v in sd.items()}
        missing, unexpected = self.load_state_dict(sd, strict=False)
        if missing:
            warnings.warn(f"ResNet: {len(missing)} parameters not in {checkpoint!r} (e.g. {missing[:3]})")
        return True

    def _freeze_stages(self):
        if self.frozen_stages >= 0:
            self.bn1.eval()
            for m in (self.conv1, self.bn1):
                for p in m.parameters():
                    p.requires_grad = False
        for i in range(1, self.frozen_stages + 1):
            if i <= len(self.stages):
                self.stages[i - 1].eval()
                for p in self.stages[i - 1].parameters():
                    p.requires_grad = False

    def train(self, mode=True):
        super().train(mode)
        self._freeze_stages()
        if mode and self.norm_eval:
            for m in self.modules():
                if isinstance(m, nn.BatchNorm2d):
                    m.eval()
        return self

    def forward(self, x):
        if self.training and self.frozen_stages >= 0 and not self._weights_loaded and not self._warned_frozen_random:
            import warnings
            self._warned_frozen_random = True
            warnings.warn(f"ResNet: frozen_stages={self.frozen_stages} freezes the stem"
                          + (f" and stages 1..{self.frozen_stages}" if self.frozen_stages > 0 else "")
                          + ", but no checkpoint has been loaded: the frozen layers keep their RANDOM initialisation and identity BatchNorm "
                          "statistics for the whole training (the reference freezes a pretrained stem: exps/conf_aim.py:57-60).  Load weights "
                          "(load_state_dict / init_cfg / pretrained=) or build with frozen_stages=-1; throughput benchmarks are unaffected.")
        x = self.maxpool(bn_act(self.bn1, self.conv1(x)))
        outs = []
        # A block hands its output on as a tuple of ALIASES of one buffer (bn_act(..., fork=2|3)): the next block reads the first with
        # its first convolution and the second as its identity (or through its downsample convolution), the neck takes the third
        # of a stage's output -- and the gradients of the uses meet inside the fused BatchNorm backward (added while loading)
        # instead of in accumulation passes of autograd's.  The last block of the last stage returns a plain tensor.
        for i, stage in enumerate(self.stages):
            x = stage(x)
            if i in self.out_indices:
                outs.append(x[-1] if isinstance(x, tuple) else x)
        return tuple(outs)


class SECONDFPN(nn.Module):
    def __init__(self, in_channels, upsample_strides, out_channels, **unused):
        super().__init__()
        blocks = []
        for cin, s, cout in zip(in_channels, upsample_strides, out_channels):
            if s > 1 or s == 1:
                up = nn.ConvTranspose2d(cin, cout, int(s), int(s), bias=False)
            else:
                k = int(round(1 / s))
                up = nn.Conv2d(cin, cout, k, k, bias=False)
            blocks.append(ConvBNAct(up, nn.BatchNorm2d(cout, eps=1e-3, momentum=0.01), nn.ReLU(inplace=True)))
        self.deblocks = nn.ModuleList(blocks)
        self.init_weights()

    def init_weights(self):
        for m in self.modules():
            if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)):
                nn.init.kaiming_normal_(m.weight)
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.ones_(m.weight)
                nn.init.zeros_(m.bias)

    def forward(self, xs):
        ups = [blk(x) for blk, x in zip(self.deblocks, xs)]
        return [torch.cat(ups, 1) if len(ups) > 1 else ups[0]]


class DeformConv2dPack(nn.Module):
    """mmcv DCN v1 pack: offsets from a zero-initialised 3x3 conv, then a deformable
    3x3 convolution = bilinear sampling (zeros outside) of each kernel tap + grouped GEMM."""

    def __init__(self, in_channels, out_channels, kernel_size=3, padding=1, groups=1, deform_groups=1, **unused):
        super().__init__()
        assert kernel_size == 3 and padding == 1 and deform_groups == 1
        self.groups = groups
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels // groups, 3, 3))
        nn.init.kaiming_uniform_(self.weight, nonlinearity="relu")
        self.conv_offset = nn.Conv2d(in_channels, 18, 3, 1, 1, bias=True)
        nn.init.zeros_(self.conv_offset.weight)
        nn.init.zeros_(self.conv_offset.bias)

    def forward(self, x):
        offset = self.conv_offset(x)
        from ..ops.deform_conv import deform_conv3x3              # implicit GEMMs on the matrix cores (csrc/deform_conv_mfma.hip)
        self.last_shape = (x.shape[0], x.shape[1], x.shape[2], x.shape[3], self.weight.shape[0], self.groups)   # (bench.py's roofline_dcn)
        with torch.autocast("cuda", enabled=False):
            return deform_conv3x3(x, offset, self.weight, self.groups)   # raises for CPU tensors: no fallback

    def forward_reference(self, x, offset):
        """The same operator with plain torch ops (bilinear grid_sample per tap + grouped GEMM).
        TEST REFERENCE ONLY (tests/test_geometry_gpu.py calls it explicitly); `forward` never does."""
        B, C, H, W = x.shape
        ys = torch.arange(H, device=x.device, dtype=x.dtype).view(1, H, 1)
        xs = torch.arange(W, device=x.device, dtype=x.dtype).view(1, 1, W)
        cols = []
        for k in range(9):
            ky, kx = divmod(k, 3)
            py = ys + (ky - 1) + offset[:, 2 * k]
            px = xs + (kx - 1) + offset[:, 2 * k + 1]
            grid = torch.stack((2 * px / max(W - 1, 1) - 1, 2 * py / max(H - 1, 1) - 1), -1)
            cols.append(F.grid_sample(x, grid, mode="bilinear", padding_mode="zeros", align_corners=True))
        col = torch.stack(cols, 2)                                     # [B, C, 9, H, W]
        g = self.groups
        col = col.reshape(B, g, (C // g) * 9, H * W)
        w = self.weight.reshape(g, -1, (C // g) * 9)
        out = torch.einsum("gok,bgkn->bgon", w, col)
        return out.reshape(B, -1, H, W)
