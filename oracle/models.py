"""ORACLE (test infrastructure, not product): CPU restatement of the reference models.

Stock torch CPU ops only.  Restates /root/reference/Models/models.py:
  FiLM                    :58-77
  UNet blocks / UNet      :108-306
  BasicBlock / Bottleneck :327-397
  HighResolutionModule    :400-544
  HighResolutionNet       :554-802
with the reference's module nesting, so state_dict keys and shapes are the
reference's and the name-keyed weight recipe (utils/synth.py) fills both
identically.  BN flavour: plain batch-statistics BatchNorm2d (bn_helper.py:4-11
picks SyncBatchNorm, which without a process group is F.batch_norm on local
statistics -- SURVEY.md D7).

Parity: pinned by tests/golden/*.npz, generated from the imported reference by
tests/golden/gen_golden.py (tests/test_oracle_golden.py).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from hrseg_amd.utils.hierarchy import build_hierarchy_indices, get_level_classes, child_groups

BN_MOMENTUM = 0.1
EPS_GATE = 1e-6


class FiLM(nn.Module):
    def __init__(self, feat_ch, cond_ch):
        super().__init__()
        self.cond_pool = nn.AdaptiveAvgPool2d(1)
        self.mlp = nn.Sequential(nn.Flatten(), nn.Linear(cond_ch, 2 * feat_ch))

    def forward(self, feats, cond_map):
        cond = cond_map.mean(dim=(2, 3)) if cond_map.dim() == 4 else cond_map
        gb = self.mlp[1](cond)
        c = feats.shape[1]
        return feats * gb[:, :c, None, None] + gb[:, c:, None, None]


def compose_levels(logits_fn, levels, groups_per_level, n_levels):
    """Level loop shared by UNet and HRNet (models.py:263-306, :760-802).

    logits_fn(L, probs_prev, logits_prev) -> z_L [B,C_L,H,W]."""
    probs, logits = [], []
    z0 = logits_fn(0, None, None)
    probs.append(torch.sigmoid(z0))
    logits.append(z0)
    for L in range(1, n_levels):
        z = logits_fn(L, probs[L - 1], logits[L - 1])
        groups = groups_per_level[L - 1]
        logits.append(z)
        if not groups:
            probs.append(torch.zeros_like(z))
            continue
        parts, start = [], 0
        for pname, ch in groups:
            g = len(ch)
            p_idx = levels[L - 1].index(pname)
            pp = probs[L - 1][:, p_idx:p_idx + 1]
            q = torch.softmax(z[:, start:start + g] + torch.log(pp + EPS_GATE), dim=1)
            parts.append(pp * q)
            start += g
        probs.append(torch.cat(parts, dim=1))
    return probs, logits


# ----------------------------------------------------------------------------- UNet
class _DoubleConv(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.conv = nn.Sequential(
            nn.Conv2d(cin, cout, 3, padding=1), nn.BatchNorm2d(cout), nn.ReLU(inplace=True),
            nn.Conv2d(cout, cout, 3, padding=1), nn.BatchNorm2d(cout), nn.ReLU(inplace=True))

    def forward(self, x):
        return self.conv(x)


class _In(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.conv = _DoubleConv(cin, cout)

    def forward(self, x):
        return self.conv(x)


class _Down(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.mpconv = nn.Sequential(nn.MaxPool2d(2), _DoubleConv(cin, cout))

    def forward(self, x):
        return self.mpconv(x)


class _Up(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.up = nn.Upsample(scale_factor=2, mode="bilinear", align_corners=True)
        self.conv = _DoubleConv(cin, cout)

    def forward(self, low, skip):
        low = self.up(low)
        dy, dx = skip.shape[2] - low.shape[2], skip.shape[3] - low.shape[3]
        low = F.pad(low, (dx // 2, dx - dx // 2, dy // 2, dy - dy // 2))
        return self.conv(torch.cat([skip, low], dim=1))


class _Out(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, 1)

    def forward(self, x):
        return self.conv(x)


class UNet(nn.Module):
    def __init__(self, size=620, n_channels=1, hierarchy={}, model_type=0, concat_prev_logits=False):
        """concat_prev_logits (opt-in extension, SURVEY 8(f4) / north_star wording; the reference re-runs on the image
        only, models.py:267,277): level L >= 1 re-encodes cat(image, logits_{L-1}) -- its first convolution is
        `cond_stems[L-1]` (n_channels + C_{L-1} inputs), every other layer is shared with level 0."""
        super().__init__()
        self.model_type, self.hierarchy = model_type, hierarchy
        self.concat_prev_logits = bool(concat_prev_logits) and model_type != 0
        self.inc0 = _In(n_channels, 64)
        self.down1, self.down2 = _Down(64, 128), _Down(128, 256)
        self.down3, self.down4 = _Down(256, 512), _Down(512, 512)
        self.up1, self.up2 = _Up(1024, 256), _Up(512, 128)
        self.up3, self.up4 = _Up(256, 64), _Up(128, 64)
        if model_type == 0:
            n_leaves = sum(len(v) for v in get_level_classes(hierarchy, inc_parent=False).values())
            self.out_flat = _Out(64, n_leaves)
        else:
            self.levels, self.parent_of, self.children_of = build_hierarchy_indices(hierarchy)
            self.child_groups = child_groups(self.levels, self.children_of)
            self.heads = nn.ModuleList([_Out(64, len(self.levels[0]))])
            for groups in self.child_groups:
                n = sum(len(ch) for _, ch in groups)
                self.heads.append(_Out(64, n if n > 0 else 1))
            self.films = nn.ModuleList([FiLM(64, len(self.levels[L - 1]))
                                        for L in range(1, len(self.levels))])
            if self.concat_prev_logits:
                self.cond_stems = nn.ModuleList([nn.Conv2d(n_channels + self.heads[L - 1].conv.out_channels, 64, 3, padding=1)
                                                 for L in range(1, len(self.levels))])

    def _run_unet(self, x, first=None):
        if first is None:
            x1 = self.inc0(x)
        else:                                   # the level's own first convolution, then the shared rest of inc0
            x1 = first(x)
            for mod in list(self.inc0.conv.conv)[1:]:
                x1 = mod(x1)
        x2 = self.down1(x1)
        x3 = self.down2(x2)
        x4 = self.down3(x3)
        x5 = self.down4(x4)
        d = self.up1(x5, x4)
        d = self.up2(d, x3)
        d = self.up3(d, x2)
        return self.up4(d, x1)

    def forward(self, x, type=0, hierarchy={}, threshold=0.5):
        if self.model_type == 0 or type == 0:
            return [], self.out_flat(self._run_unet(x))

        def logits_fn(L, prev, zprev):
            if L > 0 and self.concat_prev_logits:
                d = self._run_unet(torch.cat([x, zprev], dim=1), self.cond_stems[L - 1])
            else:
                d = self._run_unet(x)
            if L > 0:
                d = self.films[L - 1](d, prev)
            return self.heads[L](d)

        return compose_levels(logits_fn, self.levels, self.child_groups, len(self.levels))


# ----------------------------------------------------------------------------- HRNet
def _bn(c):
    return nn.BatchNorm2d(c, momentum=BN_MOMENTUM)


def _c3(cin, cout, stride=1):
    return nn.Conv2d(cin, cout, 3, stride, 1, bias=False)


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, cin, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1, self.bn1 = _c3(cin, planes, stride), _bn(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2, self.bn2 = _c3(planes, planes), _bn(planes)
        self.downsample = downsample

    def forward(self, x):
        r = x if self.downsample is None else self.downsample(x)
        y = self.relu(self.bn1(self.conv1(x)))
        y = self.bn2(self.conv2(y))
        return self.relu(y + r)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, cin, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1, self.bn1 = nn.Conv2d(cin, planes, 1, bias=False), _bn(planes)
        self.conv2, self.bn2 = nn.Conv2d(planes, planes, 3, stride, 1, bias=False), _bn(planes)
        self.conv3, self.bn3 = nn.Conv2d(planes, planes * 4, 1, bias=False), _bn(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        r = x if self.downsample is None else self.downsample(x)
        y = self.relu(self.bn1(self.conv1(x)))
        y = self.relu(self.bn2(self.conv2(y)))
        y = self.bn3(self.conv3(y))
        return self.relu(y + r)


_BLOCKS = {"BASIC": BasicBlock, "BOTTLENECK": Bottleneck}


def _make_layer(block, cin, planes, n, stride=1):
    ds = None
    if stride != 1 or cin != planes * block.expansion:
        ds = nn.Sequential(nn.Conv2d(cin, planes * block.expansion, 1, stride, bias=False),
                           _bn(planes * block.expansion))
    layers = [block(cin, planes, stride, ds)]
    layers += [block(planes * block.expansion, planes) for _ in range(1, n)]
    return nn.Sequential(*layers)


class HighResolutionModule(nn.Module):
    def __init__(self, num_branches, block, num_blocks, num_inchannels, num_channels,
                 multi_scale_output=True, align_corners=True):
        super().__init__()
        self.num_branches = num_branches
        self.align_corners = align_corners
        self.num_inchannels = list(num_inchannels)
        self.branches = nn.ModuleList()
        for b in range(num_branches):
            self.branches.append(_make_layer(block, self.num_inchannels[b], num_channels[b], num_blocks[b]))
            self.num_inchannels[b] = num_channels[b] * block.expansion
        self.fuse_layers = self._fuse(multi_scale_output) if num_branches > 1 else None
        self.relu = nn.ReLU(inplace=True)

    def _fuse(self, multi):
        ch, nb = self.num_inchannels, self.num_branches
        rows = []
        for i in range(nb if multi else 1):
            row = []
            for j in range(nb):
                if j > i:
                    row.append(nn.Sequential(nn.Conv2d(ch[j], ch[i], 1, 1, 0, bias=False), _bn(ch[i])))
                elif j == i:
                    row.append(None)
                else:
                    steps = []
                    for k in range(i - j):
                        last = k == i - j - 1
                        cout = ch[i] if last else ch[j]
                        mods = [nn.Conv2d(ch[j], cout, 3, 2, 1, bias=False), _bn(cout)]
                        if not last:
                            mods.append(nn.ReLU(inplace=True))
                        steps.append(nn.Sequential(*mods))
                    row.append(nn.Sequential(*steps))
            rows.append(nn.ModuleList(row))
        return nn.ModuleList(rows)

    def forward(self, xs):
        if self.num_branches == 1:
            return [self.branches[0](xs[0])]
        xs = [self.branches[b](xs[b]) for b in range(self.num_branches)]
        outs = []
        for i, row in enumerate(self.fuse_layers):
            y = xs[0] if i == 0 else row[0](xs[0])
            for j in range(1, self.num_branches):
                if j == i:
                    y = y + xs[j]
                elif j > i:
                    y = y + F.interpolate(row[j](xs[j]), size=xs[i].shape[-2:], mode="bilinear",
                                          align_corners=self.align_corners)
                else:
                    y = y + row[j](xs[j])
            outs.append(self.relu(y))
        return outs


class HighResolutionNet(nn.Module):
    def __init__(self, config, hierarchy={}, model_type=0, concat_prev_logits=False, **kwargs):
        super().__init__()
        self.concat_prev_logits = bool(concat_prev_logits) and model_type != 0     # see UNet
        extra = config.MODEL.EXTRA
        self.align_corners = config.MODEL.ALIGN_CORNERS
        self.model_type, self.hierarchy = model_type, hierarchy
        self.relu = nn.ReLU(inplace=True)
        self.stem = nn.Sequential(_c3(3, 64, 2), _bn(64), nn.ReLU(inplace=True),
                                  _c3(64, 64, 2), _bn(64), nn.ReLU(inplace=True))
        s1 = extra["STAGE1"]
        blk = _BLOCKS[s1["BLOCK"]]
        self.layer1 = _make_layer(blk, 64, s1["NUM_CHANNELS"][0], s1["NUM_BLOCKS"][0])
        pre = [blk.expansion * s1["NUM_CHANNELS"][0]]
        self.stage_cfgs = []
        for idx in (2, 3, 4):
            cfg = extra[f"STAGE{idx}"]
            blk = _BLOCKS[cfg["BLOCK"]]
            chans = [c * blk.expansion for c in cfg["NUM_CHANNELS"]]
            setattr(self, f"transition{idx - 1}", self._transition(pre, chans))
            stage, pre = self._stage(cfg, chans)
            setattr(self, f"stage{idx}", stage)
            self.stage_cfgs.append(cfg)
        last = int(sum(pre))
        self.shared_head = nn.Sequential(nn.Conv2d(last, last, 1, 1, 0, bias=True), _bn(last),
                                         nn.ReLU(inplace=True))
        k = extra["FINAL_CONV_KERNEL"]
        pad = 1 if k == 3 else 0
        if model_type == 0:
            n_leaves = sum(len(v) for v in get_level_classes(hierarchy, inc_parent=False).values())
            self.classifier = nn.Conv2d(last, n_leaves, k, 1, pad)
        else:
            self.levels, self.parent_of, self.children_of = build_hierarchy_indices(hierarchy)
            self.child_groups = child_groups(self.levels, self.children_of)
            self.classifiers = nn.ModuleList([nn.Conv2d(last, len(self.levels[0]), k, 1, pad)])
            for groups in self.child_groups:
                n = sum(len(ch) for _, ch in groups)
                self.classifiers.append(nn.Conv2d(last, n if n > 0 else 1, k, 1, pad))
            self.films = nn.ModuleList([FiLM(last, len(self.levels[L - 1]))
                                        for L in range(1, len(self.levels))])
            if self.concat_prev_logits:
                self.cond_stems = nn.ModuleList([_c3(3 + self.classifiers[L - 1].out_channels, 64, 2)
                                                 for L in range(1, len(self.levels))])

    @staticmethod
    def _transition(pre, cur):
        layers = []
        for i, c in enumerate(cur):
            if i < len(pre):
                if c != pre[i]:
                    layers.append(nn.Sequential(_c3(pre[i], c), _bn(c), nn.ReLU(inplace=True)))
                else:
                    layers.append(None)
            else:
                steps = []
                for j in range(i + 1 - len(pre)):
                    cout = c if j == i - len(pre) else pre[-1]
                    steps.append(nn.Sequential(_c3(pre[-1], cout, 2), _bn(cout), nn.ReLU(inplace=True)))
                layers.append(nn.Sequential(*steps))
        return nn.ModuleList(layers)

    def _stage(self, cfg, num_in, multi_scale_output=True):
        mods = []
        blk = _BLOCKS[cfg["BLOCK"]]
        for m in range(cfg["NUM_MODULES"]):
            multi = multi_scale_output or m != cfg["NUM_MODULES"] - 1
            mods.append(HighResolutionModule(cfg["NUM_BRANCHES"], blk, cfg["NUM_BLOCKS"], num_in,
                                             cfg["NUM_CHANNELS"], multi, self.align_corners))
            num_in = mods[-1].num_inchannels
        return nn.Sequential(*mods), num_in

    def _forward_backbone(self, x, first=None):
        if first is None:
            x = self.stem(x)
        else:
            x = first(x)
            for mod in list(self.stem)[1:]:
                x = mod(x)
        x = self.layer1(x)
        ys = [x]
        for t_idx, cfg in zip((1, 2, 3), self.stage_cfgs):
            trans = getattr(self, f"transition{t_idx}")
            xs = []
            for i in range(cfg["NUM_BRANCHES"]):
                if trans[i] is None:
                    xs.append(ys[i])
                else:
                    xs.append(trans[i](ys[i] if i < len(ys) else ys[-1]))
            ys = getattr(self, f"stage{t_idx + 1}")(xs)
        h, w = ys[0].shape[-2:]
        ups = [ys[0]] + [F.interpolate(y, size=(h, w), mode="bilinear", align_corners=self.align_corners)
                         for y in ys[1:]]
        return self.shared_head(torch.cat(ups, dim=1))

    def forward(self, x):
        size = x.shape[-2:]

        def up(z):
            return F.interpolate(z, size=size, mode="bilinear", align_corners=self.align_corners)

        if self.model_type == 0:
            return [], up(self.classifier(self._forward_backbone(x)))

        def logits_fn(L, prev, zprev):
            if L > 0 and self.concat_prev_logits:
                f = self._forward_backbone(torch.cat([x, zprev], dim=1), self.cond_stems[L - 1])
            else:
                f = self._forward_backbone(x)
            if L > 0:
                f = self.films[L - 1](f, prev)
            return up(self.classifiers[L](f))

        return compose_levels(logits_fn, self.levels, self.child_groups, len(self.levels))
