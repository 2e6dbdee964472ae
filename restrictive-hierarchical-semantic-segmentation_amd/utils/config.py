"""HRNet topology config without yacs.

The reference reads ``config.MODEL.EXTRA[...]`` / ``config.MODEL.ALIGN_CORNERS``
from a yacs CfgNode (config/default.py:38-40, merged from
config/seg_hrnet_w48_*.yaml:13-66).  yacs is not required here: any object
with that attribute / item access works, and ``hrnet_w48_config()`` returns
the W48 topology the reference ships.
"""
from __future__ import annotations


class AttrDict(dict):
    """dict with attribute access, recursively."""

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        for k, v in list(self.items()):
            if isinstance(v, dict) and not isinstance(v, AttrDict):
                self[k] = AttrDict(v)

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


def _stage(modules, blocks, channels, block="BASIC"):
    return dict(NUM_MODULES=modules, NUM_BRANCHES=len(channels), BLOCK=block,
                NUM_BLOCKS=list(blocks), NUM_CHANNELS=list(channels), FUSE_METHOD="SUM")


def hrnet_config(width=48, align_corners=True, final_conv_kernel=1):
    """HRNetV2-W<width> segmentation topology (W48: 48/96/192/384)."""
    w = width
    return AttrDict(MODEL=dict(
        NAME="seg_hrnet", ALIGN_CORNERS=align_corners, NUM_OUTPUTS=1,
        EXTRA=dict(
            FINAL_CONV_KERNEL=final_conv_kernel,
            STAGE1=_stage(1, [4], [64], "BOTTLENECK"),
            STAGE2=_stage(1, [4, 4], [w, 2 * w]),
            STAGE3=_stage(4, [4, 4, 4], [w, 2 * w, 4 * w]),
            STAGE4=_stage(3, [4, 4, 4, 4], [w, 2 * w, 4 * w, 8 * w]),
        )))


def hrnet_w48_config():
    return hrnet_config(48)


def load_yaml_config(path):
    """Read a reference-style yaml (MODEL.EXTRA...) into an AttrDict."""
    import yaml
    with open(path) as f:
        cfg = yaml.safe_load(f)
    cfg.setdefault("MODEL", {}).setdefault("ALIGN_CORNERS", True)
    return AttrDict(cfg)
