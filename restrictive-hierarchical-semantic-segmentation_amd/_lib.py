"""ctypes binding of libhrseg_hip.so (the C ABI declared in include/hrseg.h).

There is no fallback: if the shared library is missing the import of any
product module fails with a clear error (build it with
``python -c "import __graft_entry__ as g; g.build()"`` or
``make -C restrictive-hierarchical-semantic-segmentation_amd/csrc``).
"""
from __future__ import annotations

import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libhrseg_hip.so")

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} not found: the MI355X HIP library is not built. "
        "Run `make -C restrictive-hierarchical-semantic-segmentation_amd/csrc` (needs hipcc); "
        "there is no CPU fallback for the product path.")

_lib = C.CDLL(LIB_PATH)

_p = C.c_void_p
_i = C.c_int
_l = C.c_long
_f = C.c_float


class ConvShape(C.Structure):
    """hrseg_conv_shape_t"""
    _fields_ = [(n, _i) for n in ("B", "Hi", "Wi", "Cin", "ldx", "Ho", "Wo", "Cout", "ldy", "ksize", "stride", "precision")] + \
               [("grad_absmax", _p), ("residual", _p), ("ldr", _i), ("relu", _i), ("stat_partial", _p), ("stat_rows", C.POINTER(C.c_int)), ("x_split", _i), ("w_persistent", _i)]


# hrseg_conv_precision (include/hrseg.h): arithmetic of the convolution contractions
CONV_PRECISION = {"f32": 0, "bf16x3": 1, "bf16x2": 2, "bf16": 3, "auto": 4, "fp16x2": 5}


class BnFwd(C.Structure):
    """hrseg_bn_fwd_t"""
    _fields_ = [("y", _p), ("ldy", _i), ("npix", _l), ("C", _i), ("gamma", _p), ("beta", _p), ("running_mean", _p),
                ("running_var", _p), ("num_batches_tracked", _p), ("momentum", _f), ("eps", _f), ("residual", _p),
                ("ldr", _i), ("relu", _i), ("z", _p), ("ldz", _i), ("coef", _p), ("partial", _p), ("nchunks", _i),
                ("stat_div", _i), ("stat_updates", _i), ("relu_mask", _p), ("z_split", _i), ("residual_split", _i), ("stat_ranks", _i)]


class BnBwd(C.Structure):
    """hrseg_bn_bwd_t"""
    _fields_ = [("dz", _p), ("lddz", _i), ("z", _p), ("ldz", _i), ("relu", _i), ("y", _p), ("ldy", _i), ("coef", _p),
                ("dgamma", _p), ("dbeta", _p), ("dy", _p), ("lddy", _i), ("dres", _p), ("lddres", _i),
                ("dres_accumulate", _i), ("npix", _l), ("C", _i), ("partial", _p), ("nchunks", _i), ("dy_absmax", _p), ("nseg", _i), ("relu_mask", _p), ("sum_ranks", _i)]


# name -> argtypes, exactly the prototypes of include/hrseg.h
PROTOTYPES = {
    "hrseg_conv_fwd": [_p, _p, _p, _p, C.POINTER(ConvShape), _p],
    "hrseg_conv_dgrad": [_p, _p, _p, _i, C.POINTER(ConvShape), _p],
    "hrseg_conv_wgrad": [_p, _p, _p, C.POINTER(ConvShape), _p],
    "hrseg_conv_fwd_group": [_i, _p, _p, _p, _p, C.POINTER(ConvShape), _p],
    "hrseg_conv_dgrad_group": [_i, _p, _p, _p, _p, C.POINTER(ConvShape), _p],
    "hrseg_conv_wgrad_group": [_i, _p, _p, _p, C.POINTER(ConvShape), _p],
    "hrseg_conv_wgrad_group_ws": [_i, _p, _p, _p, C.POINTER(ConvShape), _p, C.c_size_t, _p],
    "hrseg_weight_transpose": [_p, _p, _i, _i, _i, _p],
    "hrseg_weight_transpose_all": [_p, _p, _p, _i, _p],
    "hrseg_bn_stats": [_p, _i, _l, _i, _p, _i, _p],
    "hrseg_bn_finalize": [_p, _i, _l, _i, _p, _p, _p, _p, _p, _f, _f, _p, _p],
    "hrseg_bn_eval_coef": [_p, _p, _p, _p, _f, _i, _p, _p],
    "hrseg_bn_fold": [_p, _p, _p, _p, _p, _p, _f, _i, _i, _p, _p, _p],
    "hrseg_bn_apply": [_p, _i, _p, _p, _i, _i, _p, _i, _l, _i, _p],
    "hrseg_bn_bwd_reduce": [_p, _i, _p, _i, _i, _p, _i, _p, _l, _i, _p, _i, _p],
    "hrseg_bn_bwd_apply": [_p, _i, _p, _i, _p, _i, _i, _p, _i, _p, _p, _p, _p, _p, _i, _p, _i, _i, _l, _i, _i, _p],
    "hrseg_bn_fwd_group": [_i, C.POINTER(BnFwd), _i, _p],
    "hrseg_bn_bwd_group": [_i, C.POINTER(BnBwd), _i, _p],
    "hrseg_bn_fwd_group_phases": [_i, C.POINTER(BnFwd), _i, _i, _p],
    "hrseg_bn_bwd_group_phases": [_i, C.POINTER(BnBwd), _i, _i, _p],
    "hrseg_maxpool2_fwd": [_p, _i, _p, _i, _i, _i, _i, _i, _p],
    "hrseg_maxpool2_bwd": [_p, _i, _p, _i, _p, _i, _i, _i, _i, _i, _i, _p],
    "hrseg_bilinear_fwd": [_p, _i, _i, _i, _i, _i, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p],
    "hrseg_bilinear_bwd": [_p, _i, _i, _i, _i, _i, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p],
    "hrseg_absmax": [_p, _i, _l, _i, _p, _p],
    "hrseg_fuse_sum": [_i, _p, _p, _i, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p],
    "hrseg_add": [_p, _i, _p, _i, _p, _i, _i, _l, _i, _p],
    "hrseg_copy": [_p, _i, _p, _i, _i, _l, _i, _p],
    "hrseg_relu_bwd": [_p, _i, _p, _i, _p, _i, _l, _i, _p],
    "hrseg_nchw_to_nhwc": [_p, _p, _i, _i, _i, _i, _i, _p],
    "hrseg_nhwc_to_nchw": [_p, _i, _p, _i, _i, _i, _i, _p],
    "hrseg_gap_nchw": [_p, _p, _p, _i, _l, _p],
    "hrseg_film_linear_fwd": [_p, _p, _p, _p, _i, _i, _i, _p],
    "hrseg_film_linear_bwd": [_p, _p, _p, _p, _p, _p, _i, _i, _i, _f, _p],
    "hrseg_head_fwd": [_p, _i, _p, _p, _p, _p, _i, _i, _l, _i, _i, _p],
    "hrseg_head_bwd": [_p, _i, _p, _p, _p, _i, _p, _i, _i, _p, _p, _p, _i, _l, _i, _i, _p],
    "hrseg_logits_up_fwd": [_p, _i, _i, _i, _i, _i, _p, _i, _i, _i, _p],
    "hrseg_logits_up_bwd": [_p, _i, _i, _i, _i, _p, _i, _i, _i, _i, _p],
    "hrseg_sigmoid_fwd": [_p, _p, _l, _p],
    "hrseg_sigmoid_bwd": [_p, _l, _l, _l, _p, _p, _i, _i, _i, _l, _p],
    "hrseg_compose_fwd": [_p, _p, _p, _i, _i, _i, _l, _i, _p, _p, _p],
    "hrseg_compose_bwd": [_p, _l, _l, _l, _p, _p, _p, _i, _p, _i, _i, _i, _i, _l, _i, _p, _p, _p],
    "hrseg_loss_partials": [_p, _p, _p, _i, _i, _l, _p],
    "hrseg_loss_finalize": [_p, _p, _i, _i, _p, _p, _p],
    "hrseg_loss_bwd": [_p, _p, _p, _p, _p, _i, _i, _i, _l, _p],
    "hrseg_consistency": [_p, _p, _p, _i, _i, _i, _l, _i, _p, _p, _p],
    "hrseg_consistency_bwd": [_p, _p, _p, _f, _p, _p, _i, _i, _i, _l, _i, _p, _p, _p],
    "hrseg_group_kl": [_p, _p, _p, _i, _i, _i, _l, _i, _p, _p, _p],
    "hrseg_group_kl_bwd": [_p, _p, _p, _f, _p, _i, _i, _i, _l, _i, _p, _p, _p],
    "hrseg_predict_metrics": [_p, _p, _p, _p, _i, _i, _l, _i, _i, _p],
    "hrseg_metric_vectors": [_i, _p, _p, _p, _p, _p],
    "hrseg_adamw": [_p, _p, _p, _p, _l, _f, _f, _f, _f, _f, _f, _f, _f, _p],
    "hrseg_adamw_dev": [_p, _p, _p, _p, _l, _p, _p, _p],
    "hrseg_fill": [_p, _f, _l, _p],
    "hrseg_encode_targets": [_p, _p, C.POINTER(C.c_int), _p, _i, _i, _l, _p],
    "hrseg_combine_levels": [_p, _i, _p, _i, _p, _p, _p, _i, _i, _l, _p],
    "hrseg_weight_images_refresh": [_p],
}
# entry points without the trailing stream argument convention of `call`
RAW_PROTOTYPES = {
    "hrseg_comm_unique_id": [_p],
    "hrseg_comm_init": [C.POINTER(_p), _i, _i, _p],
    "hrseg_comm_allreduce_async": [_p, _p, _l, _p],
    "hrseg_comm_wait": [_p, _p],
    "hrseg_comm_destroy": [_p],
    "hrseg_set_scratch": [_p, C.c_size_t],
    "hrseg_set_weight_image_arena": [_p, C.c_size_t, _p, C.c_size_t, _p, _p, _p, _p],
}

_lib.hrseg_last_error_string.restype = C.c_char_p
_lib.hrseg_last_error_string.argtypes = []
_lib.hrseg_abi_version.restype = _i
_lib.hrseg_abi_version.argtypes = []

_lib.hrseg_conv_x_split_ok.restype = _i
_lib.hrseg_conv_x_split_ok.argtypes = [_i, C.POINTER(ConvShape)]


def conv_x_split_ok(shapes, n=None) -> bool:
    """may the producer of these convolutions' inputs store them pre-split (hrseg_conv_x_split_ok)?"""
    return bool(_lib.hrseg_conv_x_split_ok(len(shapes) if n is None else n, shapes))


_lib.hrseg_conv_wgrad_workspace_bytes.restype = C.c_size_t
_lib.hrseg_conv_wgrad_workspace_bytes.argtypes = [_i, C.POINTER(ConvShape)]


def conv_wgrad_workspace_bytes(shapes, n=None):
    """bytes of workspace the nine-tap weight-gradient path wants for these problems (0: not applicable)"""
    return int(_lib.hrseg_conv_wgrad_workspace_bytes(len(shapes) if n is None else n, shapes))


_scratch = None


def ensure_scratch(device):
    """hand the library its convolution scratch buffer (hrseg_set_scratch) the first time a convolution runs;
    HRSEG_SCRATCH_MB=0 leaves it detached (the layers that want it take their other kernels)"""
    global _scratch
    if _scratch is not None:
        return
    import torch
    nbytes = int(os.environ.get("HRSEG_SCRATCH_MB", "256")) << 20
    if nbytes == 0:
        _scratch = False
        return
    _scratch = torch.empty(nbytes, dtype=torch.uint8, device=device)
    call_raw("hrseg_set_scratch", _scratch.data_ptr(), nbytes)


_image_owner = None


def ensure_image_arena(flat):
    """persistent weight images (hrseg_set_weight_image_arena) for the parameters of `flat` (an engine.FlatParams): the arena
    -- 8 bytes per parameter: forward and data-gradient images -- and the entry table are tensors owned by `flat`; (re)attached
    whenever another model's parameters were the registered ones.  HRSEG_WEIGHT_IMAGES=0: off (an image launch per convolution)"""
    global _image_owner
    if _image_owner is not None and _image_owner() is flat:
        return True
    if os.environ.get("HRSEG_WEIGHT_IMAGES", "1") == "0":
        return False
    import torch
    import weakref
    if getattr(flat, "_img_arena", None) is None:
        flat._img_arena = torch.empty(max(1 << 20, 8 * flat.numel + (1 << 20)), dtype=torch.uint8, device=flat.data.device)
        flat._img_table = torch.empty(40 * 4096, dtype=torch.uint8, device=flat.data.device)
    d, dt = flat.data, flat.data_t
    call_raw("hrseg_set_weight_image_arena", flat._img_arena.data_ptr(), flat._img_arena.numel(), flat._img_table.data_ptr(),
             flat._img_table.numel(), d.data_ptr(), d.data_ptr() + 4 * d.numel(), dt.data_ptr(), dt.data_ptr() + 4 * dt.numel())
    _image_owner = weakref.ref(flat)      # (weak: the registration must not keep a dead model's buffers alive; only the
    return True                           #  engine flags weights as persistent, and it re-registers its own model first)


_lib.hrseg_tune.restype = _i
_lib.hrseg_tune.argtypes = [C.c_char_p, _i]
_lib.hrseg_launch_count.restype = C.c_long
_lib.hrseg_launch_count.argtypes = [C.c_char_p, _i]

_deterministic = False


def launch_count(family=None, reset=False) -> int:
    """convolution launches issued so far by kernel family (hrseg_launch_count; None = all families)"""
    return int(_lib.hrseg_launch_count(None if family is None else family.encode(), int(reset)))


def deterministic() -> bool:
    return _deterministic


_tune_generation = 0


def tune_generation() -> int:
    """bumped by every tune() call: recorded launch tapes are keyed by it (the library plans per call, so a replay would
    follow a new plan anyway, but workspace sizes were fixed when the tape was recorded)"""
    return _tune_generation


def tune(**kv):
    """tile-plan overrides of the library (hrseg_tune; 0 = automatic), e.g. tune(igemm_wtm=2, igemm_kc=1)"""
    global _deterministic, _tune_generation
    _tune_generation += 1
    for k, v in kv.items():
        if _lib.hrseg_tune(k.encode(), int(v)) != 0:
            raise RuntimeError(f"hrseg_tune({k}) failed: {last_error()}")
        if k == "deterministic":
            _deterministic = bool(v)


def set_conv_tune(wtm=0, kc=0, db=0, ksplit=0):
    tune(igemm_wtm=wtm, igemm_kc=kc, igemm_db=db, igemm_ksplit=ksplit)


def set_wgrad_tune(pix=0, db=0, target_blocks=0):
    tune(wgrad_pix=pix, wgrad_db=db, wgrad_blocks=target_blocks)


def set_wgrad_group_plan(mult=0, min_blocks=0, max_blocks=0):
    tune(wgrad_group_mult=mult, wgrad_group_min=min_blocks, wgrad_group_max=max_blocks)


_fn = {}
for _name, _args in PROTOTYPES.items():
    f = getattr(_lib, _name)       # AttributeError here = header and library disagree
    f.argtypes = _args
    f.restype = _i
    _fn[_name] = f


ABI_VERSION = 15    # must equal hrseg_abi_version() of the built library (struct layouts above)


raw = {}
for _name, _args in RAW_PROTOTYPES.items():
    f = getattr(_lib, _name)
    f.argtypes = _args
    f.restype = _i
    raw[_name] = f


def call_raw(name, *args):
    rc = raw[name](*args)
    if rc != 0:
        raise RuntimeError(f"{name} failed ({rc}): {last_error()}")


def abi_version() -> int:
    return int(_lib.hrseg_abi_version())


if abi_version() != ABI_VERSION:
    raise ImportError(f"{LIB_PATH} has ABI version {abi_version()}, this package needs {ABI_VERSION}: rebuild it "
                      "(`make -C restrictive-hierarchical-semantic-segmentation_amd/csrc`)")


def last_error() -> str:
    return (_lib.hrseg_last_error_string() or b"").decode()


def ptr(t):
    """device pointer of a tensor (None -> NULL)"""
    return None if t is None else t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream():
    """raw hipStream_t of torch's current stream on the current device (called once per launch: the
    C-level accessor costs ~0.3 us, building a torch.cuda.Stream object ~4 us)"""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


class Tape:
    """A recorded sequence of C-ABI calls (plus the stream waits and host callbacks between them) that can be issued
    again without the Python engine around it: `replay()` is one loop over pre-built ctypes argument tuples.

    What a train step costs on the host is not the ~1,500 launches (3-4 us each inside the library) but the engine that
    derives them every step: tensor allocations, shape structs, closures, autograd (31 ms of a 52 ms step on a fast
    host, more than the step on a slow one).  A step whose shapes do not change issues the SAME calls with the SAME
    pointers as long as its buffers stay where they are, so train.TapedTrainStep records them once -- inside a private
    torch memory pool whose blocks nobody else can take -- and replays them.  Unlike a captured hipGraph the replay issues
    real launches on the real streams (main + weight-gradient side stream + the collective's stream), so the two-stream
    overlap of the eager step is kept, host callbacks (the bucketed gradient all-reduce) run where they ran, and the
    library still plans every launch itself (tile plans, scratch rings) exactly as in the eager step.

    Entries: (0, cfunc, args, raw stream) | (1, waiting torch stream or None = main, signalling stream or None = main, event) |
    (2, callable).  The main stream is whatever stream is current when replay() is called."""

    def __init__(self):
        self.entries = []
        self.keep = []          # tensors the recorded side-stream work reads: held until the streams join (see engine.py)
        self.main = None        # raw handle of the stream that was current while recording
        self.calls = 0

    def __enter__(self):
        global _tape
        if _tape is not None:
            raise RuntimeError("hrseg_amd: a launch tape is already being recorded")
        self.main = stream()
        _tape = self
        return self

    def __exit__(self, *exc):
        global _tape
        _tape = None
        self.keep = []
        return False

    def replay(self):
        cur, rec = stream(), self.main
        for kind, a, b, c in self.entries:
            if kind == 0:
                rc = a(*b, cur if c == rec else c)
                if rc != 0:
                    raise RuntimeError(f"{a.__name__} failed ({rc}) in a launch-tape replay: {last_error()}")
            elif kind == 1:         # (the entry's own event, created once: wait_stream builds and destroys one per call)
                c.record(b if b is not None else torch.cuda.current_stream())
                (a if a is not None else torch.cuda.current_stream()).wait_event(c)
            else:
                a()


_tape = None


def taping() -> bool:
    """True while a launch tape is being recorded (host readbacks and data-dependent routing must stay out of it)"""
    return _tape is not None


def tape_keep(*tensors):
    """recording: hold `tensors` until the next tape_release() (what Tensor.record_stream does in the eager step -- whose
    allocator decides by GPU timing when such a block may be reused, which a replay cannot reproduce)"""
    _tape.keep.extend(tensors)


def tape_release():
    if _tape is not None:
        _tape.keep = []


def stream_wait(waiter, signaller):
    """waiter.wait_stream(signaller) for two torch streams (None = the current stream), recorded when a tape is active"""
    if _tape is not None:
        _tape.entries.append((1, waiter, signaller, torch.cuda.Event()))
    (waiter if waiter is not None else torch.cuda.current_stream()).wait_stream(
        signaller if signaller is not None else torch.cuda.current_stream())


def host_call(fn):
    """run a host callback now; a recording tape runs it again at the same position of every replay (collectives)"""
    if _tape is not None:
        _tape.entries.append((2, fn, None, None))
    fn()


def call(name, *args):
    """Invoke an entry point on torch's current stream; raise on failure."""
    st = stream()
    f = _fn[name]
    if _tape is not None:
        _tape.entries.append((0, f, args, st))
        _tape.calls += 1
    rc = f(*args, st)
    if rc != 0:
        raise RuntimeError(f"{name} failed ({rc}): {last_error()}")


def ptr_array(tensors):
    """host array of device pointers (None -> NULL)"""
    return (C.c_void_p * len(tensors))(*[None if t is None else t.data_ptr() for t in tensors])


def int_array(values):
    return (C.c_int * len(values))(*values)


def require_gpu():
    if not torch.cuda.is_available():
        raise RuntimeError("hrseg_amd: no GPU visible; the product path runs on MI355X only "
                           "(the CPU oracle under oracle/ is test infrastructure, not a fallback)")


def set_deterministic(on=True):
    """bit-reproducible gradients: every float reduction takes its single-adder form (see csrc/common.h)"""
    tune(deterministic=int(bool(on)))


if os.environ.get("HRSEG_DETERMINISTIC", "0") == "1":
    set_deterministic(True)
if "HRSEG_TUNE" in os.environ:               # "key=value,key=value" A/B switch for tuning runs
    tune(**{kv.split("=")[0]: int(kv.split("=")[1]) for kv in os.environ["HRSEG_TUNE"].split(",") if kv})
