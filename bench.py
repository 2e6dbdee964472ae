#!/usr/bin/env python
"""Headline benchmark: training images/sec of hierarchical HRNet-W48 at 620x620 on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--model hrnet|unet] [--batch B] [--size S]

One "step" = the reference's train-loop batch body (train.py:179-248) on one synthetic batch: forward
of the L level passes, prediction prep + metrics, CE+Dice+consistency loss, backward, gradient
all-reduce, AdamW, the per-class metric vectors and the ONE device->host readback of loss + metrics;
the batch is resident in HBM when the timed region starts (`value`).  The same body fed from pinned
host memory (H2D copy inside the timed step) is measured next to it (`batch_body_from_host`).
N>1: one rank per GPU over RCCL, every rank on its own 4-image shard (weak scaling), rank 0 prints ONE
JSON line.  `python bench.py --gpus N` starts its own N ranks (torch.distributed.run as a child
process, before this process touches the GPU); under torchrun (WORLD_SIZE set) it is a rank itself.

Extra objects in that line:
  roofline     -- the dominant kernel (wave-specialised halo-patch 3x3 convolution of the parallel HRNet branches,
                  fp16x2 on the 16-bit matrix pipe): algorithmic FLOPs per launch / average launch time measured
                  here with events on the launch stream; `roofline_other` the next kernel families
  cpu_baseline -- the CPU oracle (oracle/, a torch-CPU port of the reference path) timed on this host's
                  cores on a bounded sample (one batch-1 step); reported, never the target
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

FP32_MFMA_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 / 32x32x2, dense
BF16_MFMA_PEAK_TFLOPS = 2500.0     # MI355X_MICROARCH.md: dense bf16 MFMA
# model.conv_dtype -> what the convolution contractions execute
CONV_ARITHMETIC = {
    "f32": "fp32 operands on v_mfma_f32_16x16x4_f32 (exact fp32 products)",
    "bf16x3": "fp32 operands split exactly into 3 bf16 pieces, 6 products per tile on v_mfma_f32_16x16x32_bf16, "
              "fp32 accumulate (fp32-grade: the dropped cross terms are below 2^-24 relative)",
    "bf16x2": "fp32 operands split into 2 bf16 pieces, 3 products per tile on v_mfma_f32_16x16x32_bf16, fp32 "
              "accumulate (2^-16 relative operand error)",
    "bf16": "operands rounded to bf16, one product on v_mfma_f32_16x16x32_bf16, fp32 accumulate",
    "fp16x2": "fp32 operands split into 2 fp16 pieces (22 significant bits, power-of-two operand scaling), 3 products "
              "per tile on v_mfma_f32_16x16x32_f16, fp32 accumulate (fp32-grade: ~1e-6 of the fp32 kernels' results)",
    "auto": "fp32-grade throughout: fp16x2 (2 fp16 pieces per fp32 operand, 3 products on v_mfma_f32_16x16x32_f16, "
            "fp32 accumulate) where it is the faster kernel family, exact-fp32 v_mfma_f32_16x16x4_f32 kernels on the "
            "small problems; all weight gradients fp16x2"}
# the bench line's "dtype": the arithmetic type the convolution contractions compute in, spelled out
DTYPE_LABEL = {"f32": "f32 (exact fp32 MFMA)",
               "auto": "f32 (fp16x2 split on fp16 MFMA, 22-bit operands, fp32 accumulate; exact-fp32 MFMA on the small problems)",
               "fp16x2": "f32 (fp16x2 split on fp16 MFMA, 22-bit operands, fp32 accumulate)",
               "bf16x3": "f32 (bf16x3 split on bf16 MFMA, 24-bit operands, fp32 accumulate)",
               "bf16x2": "bf16x2 split (16-bit operands, fp32 accumulate)", "bf16": "bf16 operands, fp32 accumulate"}
CONV_PRODUCTS = {"f32": 1, "bf16x3": 6, "bf16x2": 3, "bf16": 1, "fp16x2": 3, "auto": 3}
TRAIN_GFLOP_PER_IMAGE = {          # BASELINE.md section 2 (conv+linear MACs x2, fwd+dgrad+wgrad = 3x fwd), 620x620
    ("hrnet", True): 1662.0, ("hrnet", False): 831.0, ("unet", True): 2168.0, ("unet", False): 1084.0}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--model", default="hrnet", choices=["hrnet", "unet"])
    ap.add_argument("--batch", type=int, default=4, help="images per GPU")
    ap.add_argument("--size", type=int, default=620)
    ap.add_argument("--flat", action="store_true", help="non-hierarchical (model_type 0)")
    ap.add_argument("--tree", default="class_tree_tl.json")
    ap.add_argument("--graph", action="store_true", help="replay the step as one captured hipGraph (N=1 only) instead "
                                                          "of issuing the kernels from Python; measured slightly slower "
                                                          "than eager issue + weight-gradient side stream on MI355X")
    ap.add_argument("--no-graph", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--eager", action="store_true",
                    help="issue every step through the Python engine (train.train_step) instead of replaying the recorded "
                         "launch tape of the step (train.TapedTrainStep, the default: same kernels, same arguments, same "
                         "streams, a fraction of the host time)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-probe", action="store_true")
    ap.add_argument("--host-time", action="store_true", help="also log the host-side issue time of one step")
    ap.add_argument("--no-bf16-line", action="store_true",
                    help="skip the extra measurement of the opt-in bf16-input convolutions (N=1)")
    ap.add_argument("--no-dedup-line", action="store_true",
                    help="skip the extra measurement of the opt-in de-duplicated level passes (N=1, hierarchical)")
    ap.add_argument("--no-f32-line", action="store_true",
                    help="skip the extra measurement with the exact-fp32 MFMA convolutions (N=1)")
    ap.add_argument("--comm-only", action="store_true",
                    help="time ONLY the gradient exchange of a step (the bucketed all-reduce of the flat gradient buffer, "
                         "as GradSync issues it) on N ranks: compute and exchange can then be read apart in a scaling run")
    return ap.parse_args()


def build(args, device):
    from hrseg_amd.Metrics import losses
    from hrseg_amd.Models import models
    from hrseg_amd import train as T
    from hrseg_amd.utils import synth
    from hrseg_amd.utils.config import hrnet_w48_config
    from hrseg_amd.utils.hierarchy import get_classes
    tree = json.load(open(os.path.join(ROOT, "restrictive-hierarchical-semantic-segmentation_amd", "data", args.tree)))
    hier = not args.flat
    torch.manual_seed(0)
    if args.model == "unet":
        model = models.UNet(size=args.size, n_channels=3, hierarchy=tree, model_type=1 if hier else 0)
    else:
        model = models.HighResolutionNet(hrnet_w48_config(), hierarchy=tree, model_type=1 if hier else 0)
    model.to(device)
    num_classes = get_classes(tree, full=hier)
    if hier:
        weights = synth.README_LEVEL_WEIGHTS_TL if args.tree == "class_tree_tl.json" else \
            [[1.0] * n for n in num_classes]
    else:
        weights = synth.README_LEVEL_WEIGHTS_FLAT
        num_classes = [sum(num_classes)]
    ns = argparse.Namespace(model_type=1 if hier else 0, model_select=0 if args.model == "unet" else 1,
                            num_classes=num_classes, level_weights=weights, level0_pretrain_epochs=None,
                            batch_size=args.batch)
    loss_fns = [[losses.CrossEntropyLoss(), losses.SoftDiceLoss(num_classes=n)] for n in num_classes]
    opt = T.FusedAdamW(model, lr=[1e-4])
    return tree, model, ns, loss_fns, opt


def _branch_tensors(device, batch, size):
    s4 = ((size + 1) // 2 + 1) // 2
    sizes = [s4]
    for _ in range(3):
        sizes.append((sizes[-1] + 1) // 2)
    chans = [48, 96, 192, 384]
    xs = [torch.randn(batch, h, h, c, device=device) for c, h in zip(chans, sizes)]
    ws = [torch.randn(c, 9, c, device=device) * 0.05 for c in chans]
    fl = [2.0 * batch * h * h * c * c * 9 for c, h in zip(chans, sizes)]
    return sizes, chans, xs, ws, fl


def _timed(fn, reps=20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


def probe_dominant_kernel(device, batch, size, conv_dtype="auto"):
    """The kernel with the largest share of the step.
    conv_dtype 'auto' / 'fp16x2' (default): igemm_patch_ws_group_kernel<fp16x2>, the wave-specialised halo-patch
    launch of the parallel HRNet branches (3x3 convs, 48 ch at size/4 ... 384 ch at size/32) -- every BasicBlock conv
    of stages 2-4 issues one forward and one data-gradient launch of it (64 + 64 per backbone pass).  Timed here with
    events on the launch stream.  `achieved` = ALGORITHMIC FLOPs (2*M*N*K per branch) / launch time; every fp32-grade
    product costs three fp16 MFMA products, so `peak` = dense fp16 MFMA peak / 3 and `frac` = executed MFMA rate / 2.5 PFLOP/s.
    conv_dtype 'f32': igemm_group_kernel on v_mfma_f32_16x16x4_f32 (stage-2/3/4 forward mix), peak 157.3."""
    from hrseg_amd import _lib, ops
    sizes, chans, xs, ws, fl = _branch_tensors(device, batch, size)
    if conv_dtype == "f32":
        mix = [(2, 8), (3, 32), (4, 24)]

        def one_pass():
            for n, reps in mix:
                for _ in range(reps):
                    ops.conv_fwd_group(xs[:n], ws[:n], [None] * n, 3, 1, chans[:n])
        launches = sum(r for _, r in mix)
        t = _timed(one_pass, 3) / launches
        flops = sum(sum(fl[:n]) * r for n, r in mix) / launches
        ach = flops / t / 1e12
        traffic, src = pmc_value("igemm_group_kernel<1, 3, 3, 1, true>", "traffic")
        return {"bound": "mfma", "kernel": "igemm_group_kernel<1,3,3,1,true> (3x3 branch convs 48/96/192/384 ch at %s, B=%d; "
                                           "stage-2/3/4 mix of one backbone pass)" % ("/".join(str(h) for h in sizes), batch),
                "achieved": round(ach, 2), "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": round(ach / FP32_MFMA_PEAK_TFLOPS, 4), "traffic": traffic, "traffic_unit": "bytes/launch (HBM read+write)",
                "traffic_source": src, "avg_launch_us": round(t * 1e6, 2), "flop_per_launch": flops}
    # auto / fp16x2: the wave-specialised halo-patch group launch.  One backbone pass issues it once per BasicBlock conv
    # of stages 2-4: 8 launches on the two high-resolution branches, 32 on three, 24 on all four -- timed here as that
    # mix, as the engine issues it (`auto`), each launch preceded by its weight-image kernel (about 5 us, included)
    pr = _lib.CONV_PRECISION["auto"]
    mix = [(2, 8), (3, 32), (4, 24)]

    def one_pass():
        for n, reps in mix:
            for _ in range(reps):
                ops.conv_fwd_group(xs[:n], ws[:n], [None] * n, 3, 1, chans[:n], prec=pr)
    launches = sum(r for _, r in mix)
    t = _timed(one_pass, 5) / launches
    flops = sum(sum(fl[:n]) * r for n, r in mix) / launches
    ach = flops / t / 1e12
    peak = BF16_MFMA_PEAK_TFLOPS / 3.0
    name = "igemm_patch_ws_group_kernel<4, 0>"          # <arithmetic (4 = fp16x2), tap geometry (0 = forward)>
    traffic, src = pmc_value(name, "traffic")
    busy, bsrc = pmc_value(name, "mfma_busy")
    return {"bound": "mfma",
            "kernel": "igemm_patch_ws_group_kernel<fp16x2> (wave-specialised halo-patch 3x3 convs of the parallel branches: "
                      "%s ch at %s, B=%d; forward launches of one backbone pass -- 8 on two branches, 32 on three, 24 on four; "
                      "the data gradient runs the same kernel)" % ("/".join(str(c) for c in chans), "/".join(str(h) for h in sizes), batch),
            "achieved": round(ach, 2), "peak": round(peak, 1), "unit": "TFLOP/s", "frac": round(ach / peak, 4),
            "peak_note": "dense fp16 MFMA peak 2500 TFLOP/s / 3 MFMA products per fp32-grade product (fp16x2 split)",
            "executed_mfma_tflops": round(3 * ach, 1), "algorithmic_vs_fp32_mfma_peak": round(ach / FP32_MFMA_PEAK_TFLOPS, 3),
            "mfma_busy": busy, "mfma_busy_source": bsrc,
            "traffic": traffic, "traffic_unit": "bytes/launch (HBM read+write)", "traffic_source": src,
            "avg_launch_us": round(t * 1e6, 2),
            "avg_launch_note": "launch = sp_weight_image_kernel + igemm_patch_ws_group_kernel (the probe's weights are scratch tensors; in the "
                               "train step the images are persistent and the launch is the convolution kernel alone: see profiles/%s_*)" % PMC_PROFILE,
            "flop_per_launch": flops}


def probe_secondary_kernels(device, batch, size, conv_dtype="auto"):
    """The next kernel families by time, measured the same way (events on the launch stream, algorithmic work over the
    measured duration): the grouped weight gradient of the four branch convs (auto: wgrad9_sp_group_kernel3<fp16x2> +
    its ordered reduce), the full four-branch forward group as the model issues it (auto: one wave-specialised fp16x2
    halo-patch launch for all four branches), and the grouped BatchNorm
    forward + backward of the four branches (HBM-bound; algorithmic bytes per element: statistics 4, apply 8, backward
    reduce 8, backward apply 12 -- no residual, ReLU mask recomputed from y)."""
    from hrseg_amd import _lib, ops
    sizes, chans, xs, ws, fl = _branch_tensors(device, batch, size)
    pr = _lib.CONV_PRECISION.get(conv_dtype, 0)
    f16 = conv_dtype in ("auto", "fp16x2")
    dys = [torch.randn(batch, h, h, c, device=device) * 1e-4 for c, h in zip(chans, sizes)]
    dws = [torch.zeros(c, 9, c, device=device) for c in chans]
    gms = [d.abs().max().reshape(1).repeat(64) for d in dys] if f16 else None
    t_w = _timed(lambda: ops.conv_wgrad_group(xs, dys, dws, 3, 1, prec=pr, gmaxs=gms))
    t_g = _timed(lambda: ops.conv_fwd_group(xs, ws, [None] * 4, 3, 1, chans, prec=pr))
    items = [dict(y=x, gamma=torch.ones(c, device=device), beta=torch.zeros(c, device=device),
                  rm=torch.zeros(c, device=device), rv=torch.ones(c, device=device),
                  nbt=torch.zeros((), dtype=torch.int64, device=device), momentum=0.1, eps=1e-5, residual=None, relu=True)
             for x, c in zip(xs, chans)]
    zc = ops.bn_fwd_group(items, True)
    t_f = _timed(lambda: ops.bn_fwd_group(items, True))
    bw = [dict(dz=d.clone(), z=None, relu=True, y=x, coef=c_, dgamma=torch.zeros(c, device=device),
               dbeta=torch.zeros(c, device=device), dres=None, dres_accumulate=False)
          for d, x, (_, c_), c in zip(dys, xs, zc, chans)]
    t_b = _timed(lambda: ops.bn_bwd_group(bw, False))
    elems = sum(x.numel() for x in xs)
    gbs = elems * (12 + 20) / (t_f + t_b) / 1e9
    peak = BF16_MFMA_PEAK_TFLOPS / 3.0 if f16 else FP32_MFMA_PEAK_TFLOPS
    total = sum(fl)
    return [
        {"bound": "mfma", "kernel": ("wgrad9_sp_group_kernel3<fp16x2> + wgrad9_reduce_kernel" if f16 else "wgrad_group_kernel<3,3,64,1>") +
         " (weight gradient of the four 3x3 branch convs, B=%d)" % batch,
         "achieved": round(total / t_w / 1e12, 2), "peak": round(peak, 1), "unit": "TFLOP/s",
         "frac": round(total / t_w / 1e12 / peak, 4), "avg_launch_us": round(t_w * 1e6, 2)},
        {"bound": "mfma", "kernel": "four-branch forward group as issued (%s), B=%d" % (
            "one wave-specialised fp16x2 halo-patch launch" if f16 else "igemm_group_kernel", batch),
         "achieved": round(total / t_g / 1e12, 2), "peak": round(peak, 1), "unit": "TFLOP/s",
         "frac": round(total / t_g / 1e12 / peak, 4), "avg_us": round(t_g * 1e6, 2)},
        {"bound": "hbm", "kernel": "bn_{stats,finalize,apply}_group + bn_bwd_{reduce,finalize,apply}_group (four branches, B=%d)" % batch,
         "achieved": round(gbs, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(gbs / 8000.0, 4),
         "algorithmic_bytes_per_element": 32, "fwd_us": round(t_f * 1e6, 2), "bwd_us": round(t_b * 1e6, 2)},
    ]


def _csrc_digest():
    """sha256 over the kernel sources: recorded counters are only quoted for the sources they were collected on"""
    import hashlib
    d = os.path.join(ROOT, "restrictive-hierarchical-semantic-segmentation_amd", "csrc")
    h = hashlib.sha256()
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


PMC_PROFILE = "r04"


def pmc_value(kernel, what):
    """RECORDED rocprofv3 PMC results for `kernel` (counters cannot be collected from inside this process):
    'traffic'   -- HBM bytes per launch: FETCH_SIZE (doubled, the gfx950 correction of MI355X_MICROARCH.md) + WRITE_SIZE,
                   separate --pmc passes over full train steps (profiles/README.md)
    'mfma_busy' -- SQ_VALU_MFMA_BUSY_CYCLES / (4 x SQ_BUSY_CU_CYCLES) as recorded in the csv's mfma_busy column
    -> (value, source) or (None, reason).  profiles/<PMC_PROFILE>_meta.json names the kernel-source digest the counters
    were collected on (tools/collect_profiles.sh writes it); with other sources the recorded values are NOT quoted."""
    import csv
    root = os.path.join(ROOT, "profiles")
    fname = {"traffic": PMC_PROFILE + "_pmc_traffic_per_launch.csv", "mfma_busy": PMC_PROFILE + "_pmc_mfma_busy.csv"}[what]
    try:
        meta = json.load(open(os.path.join(root, PMC_PROFILE + "_meta.json")))
        if meta.get("csrc_digest") != _csrc_digest():
            return None, "profiles/%s was recorded on other kernel sources (digest %s, now %s): not quoted" % (
                fname, meta.get("csrc_digest"), _csrc_digest())
        for row in csv.DictReader(open(os.path.join(root, fname))):
            if row["kernel"].startswith(kernel):
                tag = "RECORDED in profiles/%s on these kernel sources (commit %s)" % (fname, meta.get("commit", "?"))
                if what == "traffic":
                    mb = float(row["FETCH_bytes_MB_corrected_x2"]) + float(row["WRITE_MB"])
                    return round(mb * 2**20), tag + ": rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over full train steps"
                return round(float(row["mfma_busy"]), 4), tag + ": SQ_VALU_MFMA_BUSY_CYCLES / (4 x SQ_BUSY_CU_CYCLES)"
    except (OSError, KeyError, ValueError):
        pass
    return None, "no recorded counters under profiles/%s_*" % PMC_PROFILE


def host_cores():
    """CPUs this process may actually use: affinity mask capped by the cgroup quota
    (os.cpu_count() reports the whole host, 256 on the GPU boxes with a 16-CPU share)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(args, tree):
    """the CPU port (oracle) on this host: ONE batch-1 train step at the benchmark resolution"""
    from oracle import models as OM
    from oracle import train_step as OT
    from hrseg_amd.utils import synth
    from hrseg_amd.utils.config import hrnet_w48_config
    from hrseg_amd.utils.hierarchy import get_classes
    hier = not args.flat
    cores = host_cores()
    torch.set_num_threads(cores)
    if args.model == "unet":
        m = OM.UNet(size=args.size, n_channels=3, hierarchy=tree, model_type=1 if hier else 0)
    else:
        m = OM.HighResolutionNet(hrnet_w48_config(), hierarchy=tree, model_type=1 if hier else 0)
    opt = torch.optim.AdamW(m.parameters(), lr=1e-4)
    bs = 4
    x, t = synth.synthetic_batch(tree, bs, args.size, seed=1, hierarchical=hier)
    nc = get_classes(tree, full=hier)
    w = synth.README_LEVEL_WEIGHTS_TL if hier else synth.README_LEVEL_WEIGHTS_FLAT
    xt, tt = torch.from_numpy(x), torch.from_numpy(t)
    t0 = time.time()
    steps = 0
    while steps < 4 and (steps == 0 or time.time() - t0 < 10.0):     # about 10-30 s of CPU work
        OT.train_step(m, opt, xt, tt, nc if hier else [sum(nc)], w, hierarchical=hier, is_unet=(args.model == "unet"))
        steps += 1
    dt = time.time() - t0
    out = {"value": round(bs * steps / dt, 4), "unit": "images/s", "cores": cores, "kind": "port",
           "sample": "%d full train step(s), batch %d, %dx%d, %s %s, torch-CPU oracle on %d threads (%.1f s)" % (
               steps, bs, args.size, args.size, "hierarchical" if hier else "flat", args.model, cores, dt)}
    # BASELINE.json configs[0], exactly as written: UNet donor, non-hierarchical, batch 2, 128x128, 7 classes
    m0 = OM.UNet(size=128, n_channels=3, hierarchy=tree, model_type=0)
    opt0 = torch.optim.AdamW(m0.parameters(), lr=1e-4)
    x0, t0_ = synth.synthetic_batch(tree, 2, 128, seed=2, hierarchical=False)
    x0, t0_ = torch.from_numpy(x0), torch.from_numpy(t0_)
    nleaf = [sum(get_classes(tree, full=False))]
    OT.train_step(m0, opt0, x0, t0_, nleaf, synth.README_LEVEL_WEIGHTS_FLAT, hierarchical=False, is_unet=True)   # warm-up
    tc, n0 = time.time(), 0
    while n0 < 20 and (n0 < 3 or time.time() - tc < 4.0):
        OT.train_step(m0, opt0, x0, t0_, nleaf, synth.README_LEVEL_WEIGHTS_FLAT, hierarchical=False, is_unet=True)
        n0 += 1
    d0 = time.time() - tc
    out["configs0"] = {"value": round(2 * n0 / d0, 3), "unit": "images/s", "ms_per_step": round(1e3 * d0 / n0, 1),
                       "sample": "%d train steps, UNet flat (model_type 0), batch 2, 128x128, 7 classes, torch-CPU oracle "
                                 "on %d threads" % (n0, cores)}
    return out


def self_launch(args):
    """`python bench.py --gpus N` without torchrun's environment: start the N ranks as a CHILD
    torch.distributed.run job (this process has not touched the GPU and never will), relay rank 0's JSON
    line and the job's exit code.  Reference analogue: train.py:509-510 (nn.DataParallel over all GPUs)."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
    for ln in proc.stdout.splitlines():
        if ln not in lines:
            print(ln, file=sys.stderr)
    if lines:
        print(lines[-1], flush=True)
    return proc.returncode if proc.returncode != 0 or lines else 1


def measure_exchange(args, model, sync, world, device):
    """the gradient exchange of one step in isolation: the buckets GradSync issues during the last backward level
    (same boundaries, same side stream, same backend), on the model's own gradient buffer, nothing else running.
    -> dict(ms per exchange, bytes, buckets, algorithm / ring bus bandwidth, backend); every rank must call it
    (reference analogue: train.py:509-510, nn.DataParallel's reduce-add)."""
    import torch.distributed as dist
    from hrseg_amd.parallel import bucket_offsets
    flat = model.flatten_parameters(device)
    offs = bucket_offsets(flat, sync.marks)
    marks = sorted(offs, key=lambda m: -offs[m])          # the order the reverse pass crosses them

    def exchange():
        for m in marks:
            sync(m)
        sync("end")

    for _ in range(max(1, args.warmup)):
        exchange()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        exchange()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt)
    nbytes = flat.numel * 4
    algbw = nbytes / (dt / args.steps) / 1e9
    return {"ms": round(1e3 * dt / args.steps, 3), "bytes": nbytes, "buckets_bytes": [4 * (hi - lo) for lo, hi in sync.launched],
            "algbw_GBps": round(algbw, 1),
            "busbw_GBps": round(algbw * 2 * (world - 1) / max(world, 1), 1) if world > 1 else None,
            "backend": sync.backend + ":" + (dist.get_backend() if dist.is_initialized() else "-"),
            "note": "exchange only, no compute beside it; in a train step it overlaps pass 0 of the backward"}


def comm_only(args, model, sync, rank, world, device):
    """`--comm-only`: ONE JSON line from rank 0 with the exchange alone (measure_exchange), so that a scaling run can tell
    compute from exchange"""
    import torch.distributed as dist
    from hrseg_amd.parallel import GradSync
    flat = model.flatten_parameters(device)
    if sync is None:
        os.environ["HRSEG_FORCE_SYNC"] = "1"
        if not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29534")
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=device)
        sync = GradSync(model)
    flat.grad.normal_()
    r = measure_exchange(args, model, sync, world, device)
    if rank == 0:
        print(json.dumps({
            "metric": "gradient exchange per step (bucketed all-reduce of the flat fp32 gradient)", "value": r["ms"],
            "unit": "ms", "higher_is_better": False, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "bytes": r["bytes"], "buckets": [[lo, hi] for lo, hi in sync.launched], "algbw_GBps": r["algbw_GBps"],
            "busbw_GBps": r["busbw_GBps"], "backend": r["backend"], "note": r["note"]}), flush=True)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
    import torch.distributed as dist
    from hrseg_amd.parallel import GradSync, init_distributed
    from hrseg_amd import train as T
    from hrseg_amd.Metrics.performance_metrics import METRIC_NAMES
    from hrseg_amd.utils import synth
    rank, local, world = init_distributed()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    device = torch.device("cuda", local)
    torch.cuda.set_device(device)
    tree, model, ns, loss_fns, opt = build(args, device)
    sync = None
    if world > 1 or os.environ.get("HRSEG_FORCE_SYNC", "0") == "1":
        if os.environ.get("HRSEG_COMM", "torch") == "rccl":       # the library's own RCCL wrappers (hrseg_comm_*)
            from hrseg_amd.parallel import RcclComm
            sync = GradSync(model, backend="rccl", comm=RcclComm(rank, world, device))
        else:                                                     # default: torch.distributed, backend nccl = RCCL
            sync = GradSync(model)
        opt.grad_scale = 1.0 / world
    hier = not args.flat
    if args.comm_only:
        comm_only(args, model, sync, rank, world, device)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return
    x_np, t_np = synth.synthetic_batch(tree, args.batch, args.size, seed=100 + rank, hierarchical=hier)
    x_host, t_host = torch.from_numpy(x_np).pin_memory(), torch.from_numpy(t_np).pin_memory()
    x, t = x_host.to(device), t_host.to(device)
    model.train()
    level_loss = []

    from hrseg_amd import _lib
    FAMILIES = ("ws", "ws_group", "patch_sp", "sp_im2col", "sp_pgroup", "sp_group", "sp_wide", "f32", "f32_group", "wgrad9",
                "wgrad_sp", "wgrad_sp_group", "wgrad_sp_wide", "wgrad_f32", "wgrad_f32_group", "small_cin")
    state = {"graphed": None, "taped": None}
    if world == 1 and args.graph:
        state["graphed"] = T.GraphedTrainStep(model, opt, loss_fns, ns, tree, x, t, warmup=1)

    def launch_mode():
        return "hipGraph replay" if state["graphed"] is not None else ("eager" if args.eager else "launch tape replay")

    def retape():
        """(re)record the launch tape of the step for the model's current configuration; the old tape's pool is released"""
        if args.eager or state["graphed"] is not None:
            return
        state["taped"] = None
        torch.cuda.empty_cache()
        state["taped"] = T.TapedTrainStep(model, opt, loss_fns, ns, tree, x, t)

    def body(xd, td):
        """the batch body of train_epoch (train.py:179-248): step, metric vectors, ONE device->host copy"""
        if state["taped"] is not None:
            packed, _ = state["taped"](xd, td)
            return state["taped"].unpack(packed.tolist())[0]
        if state["graphed"] is not None:
            loss, cms = state["graphed"](xd, td)
        else:
            loss, cms = T.train_step(model, opt, xd, td, loss_fns, ns, tree, level_loss)
        vec = T._metric_vectors(cms)
        host = torch.cat([loss.reshape(1)] + [vec[k] for k in METRIC_NAMES]).tolist()
        return host[0]

    def step():
        return body(x, t)

    def step_from_host():
        return body(x_host.to(device, non_blocking=True), t_host.to(device, non_blocking=True))

    def step_async():
        if state["taped"] is not None:
            return state["taped"](x, t)
        if state["graphed"] is not None:
            return state["graphed"](x, t)
        return T.train_step(model, opt, x, t, loss_fns, ns, tree, level_loss)

    def host_issue():
        """host time to ISSUE one step (no readback, nothing waited for) and the library's conv launches per step by kernel
        family: says whether two lines of this report ran the same kernels, and whether a step is host- or GPU-bound"""
        ts = []
        for _ in range(3):
            torch.cuda.synchronize()
            _lib.launch_count(None, reset=True)
            th = time.perf_counter()
            step_async()
            ts.append(time.perf_counter() - th)
            torch.cuda.synchronize()
        counts = {f: _lib.launch_count(f, reset=True) for f in FAMILIES}
        return round(1e3 * sorted(ts)[1], 2), {k: v for k, v in counts.items() if v}

    def log(msg):
        if rank == 0:
            print("[bench] " + msg, file=sys.stderr, flush=True)

    def timed(fn, n):
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            out = fn()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([dt], device=device, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt)
        return dt, out

    log("model on %s, %d params; warmup %d, steps %d" % (device, sum(p.numel() for p in model.parameters()),
                                                          args.warmup, args.steps))
    retape()                                  # (the recording run is a real train step on this batch)
    for i in range(args.warmup):
        tw = time.perf_counter()
        loss = step()
        torch.cuda.synchronize()
        log("warmup step %d: %.3f s, loss %.5f, peak mem %.1f GB" % (i, time.perf_counter() - tw, float(loss),
                                                                    torch.cuda.max_memory_allocated() / 2**30))
    dt, final_loss = timed(step, args.steps)
    dt_host, _ = timed(step_from_host, args.steps)
    dt_async, _ = timed(step_async, args.steps)
    host_ms, launches = host_issue()
    log("host issue %.1f ms per step (%s); conv launches per step %s" % (host_ms, launch_mode(), launches))
    exchange = None
    if sync is not None:
        # what a scaling run needs to be read: the exchange alone (same job, same buckets), each bucket's overlap with the
        # reverse pass (events), and the slowest rank's host issue time
        sync.timing = True
        step()
        step()
        buckets = sync.bucket_report()
        sync.timing = False
        hmax = torch.tensor([host_ms], device=device, dtype=torch.float64)
        if world > 1:
            dist.all_reduce(hmax, op=dist.ReduceOp.MAX)
        exchange = measure_exchange(args, model, sync, world, device)
        exchange.update(buckets_in_step=buckets, host_issue_ms_slowest_rank=round(float(hmax), 2),
                        exposed_ms_in_step=round(sum(b.get("exposed_ms") or 0.0 for b in buckets), 3))
        log("exchange alone %.2f ms; in the step: %s" % (exchange["ms"], json.dumps(buckets)))
    if rank == 0:
        ips = world * args.batch * args.steps / dt
        gf = TRAIN_GFLOP_PER_IMAGE.get((args.model, hier))
        conv_dtype = getattr(model, "conv_dtype", "f32")
        line = {
            "metric": "train images/sec (620x620, hier-HRNet-W48)" if (args.model == "hrnet" and hier and args.size == 620) else
            "train images/sec (%dx%d, %s%s)" % (args.size, args.size, "hier-" if hier else "flat-", args.model),
            "value": round(ips, 3), "unit": "images/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 2),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": DTYPE_LABEL.get(conv_dtype, conv_dtype),
            "data": "synthetic",
            "config": {"workload": "%s %s (%s), %dx%d, batch %d per GPU, train_epoch batch body (fwd L passes, prediction prep "
                                   "+ metrics, CE+Dice+consistency, bwd, grad all-reduce, AdamW, metric vectors, one D2H "
                                   "readback of loss+metrics)" % (
                                       "HRNet-W48" if args.model == "hrnet" else "UNet",
                                       "hierarchical" if hier else "flat", args.tree, args.size, args.size, args.batch),
                       "global_batch": world * args.batch, "parallelism": "dp%d" % world, "final_loss": final_loss,
                       "launch": launch_mode(),
                       "conv_arithmetic": CONV_ARITHMETIC.get(conv_dtype, conv_dtype),
                       "rccl_ranks": (dist.get_world_size() if dist.is_initialized() else 1),
                       "dist_backend": (dist.get_backend() if dist.is_initialized() else None),
                       "comm": (None if sync is None else
                                "hrseg_comm_* (the library's own RCCL wrappers, HRSEG_COMM=rccl)" if sync.backend == "rccl" else
                                "torch.distributed all_reduce (HRSEG_COMM=torch, default)")},
            "batch_body_from_host": {"value": round(world * args.batch * args.steps / dt_host, 3), "unit": "images/s",
                                     "ms_per_step": round(1e3 * dt_host / args.steps, 2),
                                     "note": "same body with the batch copied from pinned host memory inside the timed "
                                             "step (train.py:181); never the headline value"},
            "ms_per_step_no_readback": round(1e3 * dt_async / args.steps, 2),
            "host_issue_ms": host_ms, "conv_launches_per_step": launches,
        }
        if exchange is not None:
            line["gradient_exchange"] = exchange
        if gf is not None and args.size == 620:
            tf = ips * gf / 1e3
            line["step_conv_roofline"] = {"train_gflop_per_image": gf, "achieved_tflops": round(tf, 2),
                                          "frac_of_fp32_mfma_peak": round(tf / (world * FP32_MFMA_PEAK_TFLOPS), 4)}
        log("timed: %.3f s for %d steps (resident), %.3f s (from host), %.3f s (no readback)" % (dt, args.steps, dt_host, dt_async))
        line["config"]["level_passes"] = ("sequential" if getattr(model, "sequential_passes", False) else
                                          "batched (one launch per layer for all L passes)") if hier else "n/a"
        graphed = state["graphed"]
        if world == 1 and hier and graphed is None and not args.no_dedup_line:
            # NOT the headline number: the opt-in mode that runs the L bit-identical level passes once
            # (Models/models.py:_run).  Same result, 1/L of the backbone FLOPs executed -- reported apart.
            model.dedup_passes = True
            retape()
            for _ in range(2):
                step()
            td, _ = timed(step, args.steps)
            hi, lc = host_issue()
            model.dedup_passes = False
            line["opt_in_dedup_passes"] = {
                "value": round(args.batch * args.steps / td, 3), "unit": "images/s",
                "ms_per_step": round(1e3 * td / args.steps, 2), "executed_backbone_passes_per_step": 1,
                "host_issue_ms": hi, "conv_launches_per_step": lc,
                "note": "explicit opt-in (model.dedup_passes / HRSEG_DEDUP_PASSES=1); the default and headline value "
                        "execute all L passes"}
            log("opt-in dedup passes: %.1f ms/step" % (1e3 * td / args.steps))
        if world == 1 and graphed is None and not args.no_bf16_line and hasattr(model, "conv_dtype"):
            # NOT the headline: bf16-input / fp32-accumulate convolutions (BASELINE configs[4] arithmetic), opt-in
            prev = model.conv_dtype
            model.conv_dtype = "bf16"
            retape()
            for _ in range(2):
                step()
            tb, _ = timed(step, args.steps)
            hi, lc = host_issue()
            model.conv_dtype = prev
            line["opt_in_bf16_convs"] = {
                "value": round(args.batch * args.steps / tb, 3), "unit": "images/s", "ms_per_step": round(1e3 * tb / args.steps, 2),
                "host_issue_ms": hi, "conv_launches_per_step": lc,
                "dtype": "bf16 inputs, fp32 accumulate (convolutions only; BN, loss, optimizer fp32)",
                "note": "explicit opt-in (model.conv_dtype='bf16'): results differ from the fp32 reference beyond 1e-3"}
            log("opt-in bf16 convs: %.1f ms/step" % (1e3 * tb / args.steps))
        if world == 1 and graphed is None and not args.no_f32_line and hasattr(model, "conv_dtype") and conv_dtype != "f32":
            # the same step with every contraction on the exact-fp32 matrix instruction (model.conv_dtype = "f32"): the
            # figure to hold next to the headline, whose contractions run as fp16x2 splits
            prev = model.conv_dtype
            model.conv_dtype = "f32"
            retape()
            for _ in range(2):
                step()
            tf32, loss32 = timed(step, args.steps)
            hi, lc = host_issue()
            model.conv_dtype = prev
            line["exact_f32_convs"] = {
                "value": round(args.batch * args.steps / tf32, 3), "unit": "images/s", "ms_per_step": round(1e3 * tf32 / args.steps, 2),
                "host_issue_ms": hi, "conv_launches_per_step": lc,
                "dtype": "f32 (exact fp32 MFMA, v_mfma_f32_16x16x4_f32)", "final_loss": loss32,
                "note": "model.conv_dtype='f32': every convolution on the exact-fp32 MFMA kernels; never the headline value"}
            if gf is not None and args.size == 620:
                tfl = args.batch * args.steps / tf32 * gf / 1e3
                line["exact_f32_convs"].update(achieved_tflops=round(tfl, 2), peak=FP32_MFMA_PEAK_TFLOPS,
                                               frac=round(tfl / FP32_MFMA_PEAK_TFLOPS, 4))
            log("exact-fp32 convs: %.1f ms/step" % (1e3 * tf32 / args.steps))
        state["taped"] = None                     # the probes below launch kernels directly; release the tape's pool
        torch.cuda.empty_cache()
        if not args.no_probe:
            # the level passes run batched: every conv launch sees batch * L images
            n_pass = len(model.levels) if (hier and not getattr(model, "sequential_passes", False)) else 1
            line["roofline"] = probe_dominant_kernel(device, args.batch * n_pass, args.size, conv_dtype)
            if args.model != "hrnet":
                # the probe times the parallel-branch launch mix of HRNet (the headline's dominant kernel) at this line's
                # batch; the recorded counters belong to the HRNet step and are not quoted on another model's line
                line["roofline"].update(mfma_busy=None, traffic=None, mfma_busy_source="HRNet-step counters: not quoted on this line",
                                        traffic_source="HRNet-step counters: not quoted on this line",
                                        note="kernel probe = HRNet branch mix (the wave-specialised 3x3 kernel this model's convolutions also run), not this model's own launch mix")
            if args.model == "hrnet":
                line["roofline_other"] = probe_secondary_kernels(device, args.batch * n_pass, args.size, conv_dtype)
            log("probe: %s" % json.dumps(line["roofline"]))
        if world == 1 and not args.no_cpu_baseline:
            log("cpu baseline (oracle train steps, batch 4, on %d cores) ..." % host_cores())
            line["cpu_baseline"] = cpu_baseline(args, tree)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
