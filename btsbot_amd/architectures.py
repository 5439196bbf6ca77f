"""Drop-in counterparts of the model classes in /root/reference/btsbot/architectures.py.

Same class names, constructor argument (the config dict), forward keyword names
(``image_input`` / ``metadata_input`` / ``input_data``), output shape ([B, 1] fp32 logits) and --
the on-disk contract -- the same ``state_dict()`` keys, shapes and order, so
``load_state_dict(torch.load("pytorch_model.bin"))`` (from_HF.py:74-79) works unchanged.

What differs is underneath: there is no timm / cuDNN / MIOpen graph.  Every ``nn.Parameter`` is a
view into one flat fp32 "master arena"; ``forward`` hands the arena and the input tensors to the
hand-written gfx950 kernels in libbtsbot_hip.so through the C ABI of include/btsbot_hip.h.
There is NO CPU fallback: calling a model whose tensors are not on a HIP device raises.

Covered: ``mm_ConvNeXt`` (:125-171), ``ConvNeXt`` (:104-122), ``um_nn`` (:277-293),
``frozen_fusion`` (:296-372) with ConvNeXt + um_nn branches; convnext_pico / convnext_nano
backbones at 63x63; ``MaxViT`` (:25-55) and ``mm_MaxViT`` (:58-101) with the maxvit_tiny_rw_224
backbone in eval mode (inference; or, under ``model.train()``, a frozen branch put in eval mode whose heads and
metadata branch train -- see ``_check_train_supported``).  ``mm_cnn`` / ``um_cnn`` raise NotImplementedError (legacy
VGG-like CNNs, out of scope per SURVEY.md section 2).
"""
from __future__ import annotations

import ctypes as C
import json
import math
import os
import os.path as path
import re
import warnings
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn as nn

from . import _lib

# timm model tables (convnext_pico / convnext_nano: conv_mlp=True, patch stem, LayerNorm2d)
_CONVNEXT_TABLE = {
    "convnext_pico": ((2, 2, 6, 2), (64, 128, 256, 512)),
    "convnext_nano": ((2, 2, 8, 2), (80, 160, 320, 640)),
}
_DEFAULT_KIND = "convnext_nano.d1h_in1k"   # architectures.py:107,128
# alerts per internal workspace chunk: 28 x 256 -- whole rounds of every stage kernel's grid on 256 CUs (stage0b: one
# alert per workgroup, 512 resident; stage1b: two per workgroup, 512 resident; stage2p: seven per workgroup, 256
# resident, which is where a larger chunk pays: its filter stream per workgroup is shared by 7 alerts instead of 4);
# ~0.3 MB of workspace per alert in the 16-bit modes (2 GB of the 288), 0.4 MB in f32
MAX_CHUNK = int(os.environ.get("BTSBOT_AMD_CHUNK", "7168"))
_MAXVIT_TABLE = {"maxvit_tiny_rw_224": ((2, 2, 5, 2), (64, 128, 256, 512))}
_MAXVIT_DEFAULT_KIND = "maxvit_tiny_rw_224.sw_in1k"   # architectures.py:28,61
MAXVIT_MAX_CHUNK = int(os.environ.get("BTSBOT_AMD_MV_CHUNK", "1024"))  # ~22 MB of bf16 activations per alert:
# a 1024-alert chunk is 22 GB of the 288 GB; measured 19.4k / 20.1k / 20.8k alerts/s at 256 / 512 / 1024


def get_model_image_size(model_kind: str) -> int:
    """architectures.py:10-22."""
    if "maxvit" in model_kind.lower():
        match = re.search(r"_(\d+)\.", model_kind)
        if match:
            return int(match.group(1))
    return 224


def _convnext_table(model_kind: str):
    mk = model_kind.lower()
    for name, tab in _CONVNEXT_TABLE.items():
        if name in mk:
            return tab
    raise ValueError(
        f"btsbot_amd: model_kind {model_kind!r} is not a supported ConvNeXt "
        f"(have {sorted(_CONVNEXT_TABLE)})")


def default_precision() -> str:
    return os.environ.get("BTSBOT_AMD_PRECISION", "f32")


class _Node(nn.Module):
    """Bare container used to reproduce the reference's dotted state-dict key hierarchy."""


class _HipModel(nn.Module):
    """Shared machinery: parameter arena, state-dict key layout, C-ABI calls."""

    _wiring: str = ""

    # -- construction ------------------------------------------------------------------
    def _setup(self, *, table, head_norm: bool, n_meta: int, meta_fc: Tuple[int, int],
               comb_fc: Tuple[int, int], dropouts: Tuple[float, float],
               key_map, precision: Optional[str]):
        self._table = table
        self._cfg_args = dict(
            depths=table[0] if table else (1, 1, 1, 1), dims=table[1] if table else (0, 0, 0, 0),
            head_norm=head_norm, n_meta=n_meta, meta_fc1=meta_fc[0], meta_fc2=meta_fc[1],
            comb_fc1=comb_fc[0], comb_fc2=comb_fc[1], meta_dropout=dropouts[0],
            comb_dropout=dropouts[1])
        self._precision = precision or default_precision()
        self._key_map = key_map
        self._handle: Optional[_lib.Handle] = None
        self._handle_device: Optional[torch.device] = None
        self._packed_version = None
        self._reserved = 0
        self._debug = False
        self._new_handle()
        self._build_parameters()
        self.reset_parameters()

    def _new_handle(self):
        if self._handle is not None:
            self._handle.close()
        cfg = _lib.make_config(self._wiring, self._precision, **self._cfg_args)
        self._handle = _lib.Handle(cfg)
        self._handle_device = None
        self._packed_version = None
        self._reserved = 0
        self._reserved_train = 0
        self._reserved_image = False

    def _build_parameters(self):
        h = self._handle
        self._table_rows = h.params()                      # (canonical, off, numel, shape, is_buf)
        arena = torch.zeros(h.param_floats(), dtype=torch.float32)
        object.__setattr__(self, "_arena", arena)          # plain attribute: not in state_dict
        self._slots: List[tuple] = []      # (key, parent module, leaf name, off, numel, shape, is_buf)
        for canon, off, numel, shape, is_buf in self._table_rows:
            key = self._key_map(canon)
            parent, leaf = self._descend(key)
            view = arena[off:off + numel].view(shape)
            if is_buf:
                parent.register_buffer(leaf, view)
            else:
                parent.register_parameter(leaf, nn.Parameter(view))
            self._slots.append((key, parent, leaf, off, numel, shape, is_buf))
            if canon.endswith("running_var"):               # BatchNorm1d bookkeeping buffer
                parent.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))

    def _tensor_list(self):
        """Current tensor object of every arena slot (buffers are replaced by nn.Module._apply)."""
        return [getattr(parent, leaf) for _k, parent, leaf, *_ in self._slots]

    def _descend(self, key: str):
        parts = key.split(".")
        node = self
        for p in parts[:-1]:
            if p not in node._modules:
                node.add_module(p, _Node())
            node = node._modules[p]
        return node, parts[-1]

    @torch.no_grad()
    def reset_parameters(self):
        """timm-style init for the backbone (trunc-normal .02 weights, zero biases, LayerNorm 1/0,
        layer-scale 1e-6), torch defaults for nn.Linear / nn.BatchNorm1d heads."""
        for (canon, *_), t in zip(self._table_rows, self._tensor_list()):
            leaf = canon.rsplit(".", 1)[-1]
            is_head = canon.startswith(("meta.", "comb."))
            if canon.endswith("gamma"):
                t.fill_(1e-6)
            elif leaf == "running_mean":
                t.zero_()
            elif leaf == "running_var":
                t.fill_(1.0)
            elif t.dim() == 1:
                if leaf == "weight":
                    t.fill_(1.0)                            # LayerNorm / BatchNorm scale
                elif is_head and not canon.startswith("meta.0."):
                    fan_in = self._fan_in(canon)
                    bound = 1.0 / math.sqrt(fan_in) if fan_in > 0 else 0.0
                    t.uniform_(-bound, bound)               # nn.Linear bias default
                else:
                    t.zero_()
            elif is_head:
                nn.init.kaiming_uniform_(t, a=math.sqrt(5))  # nn.Linear weight default
            else:
                nn.init.trunc_normal_(t, std=0.02)

    def _fan_in(self, bias_canon: str) -> int:
        wname = bias_canon.rsplit(".", 1)[0] + ".weight"
        for canon, _off, _n, shape, _b in self._table_rows:
            if canon == wname:
                return int(shape[1]) if len(shape) > 1 else 0
        return 0

    # -- device moves keep the single-arena invariant ----------------------------------------
    def _apply(self, fn, recurse=True):
        super()._apply(fn, recurse)
        tensors = self._tensor_list()
        first = tensors[0]
        if first.dtype != torch.float32:
            raise TypeError("btsbot_amd models keep fp32 master parameters; choose the MFMA operand "
                            "type with set_precision('bf16'|'f16'|'f32') instead of .half()/.bfloat16()")
        arena = torch.zeros(self._arena.numel(), dtype=torch.float32, device=first.device)
        with torch.no_grad():
            for (key, parent, leaf, off, numel, shape, is_buf), t in zip(self._slots, tensors):
                view = arena[off:off + numel].view(shape)
                view.copy_(t.detach())
                if is_buf:
                    parent._buffers[leaf] = view
                else:
                    t.data = view
        object.__setattr__(self, "_arena", arena)
        self._packed_version = None
        return self

    # -- options ------------------------------------------------------------------------------
    def set_precision(self, precision: str):
        """'f32' (exact-fp32 MFMA, parity mode), 'bf16' or 'f16' (MFMA operand type)."""
        if precision not in _lib.PRECISION:
            raise ValueError(f"unknown precision {precision!r}")
        if _lib.PRECISION[precision] != _lib.PRECISION[self._precision]:
            self._precision = precision
            self._new_handle()
        return self

    @property
    def precision(self) -> str:
        return self._precision

    def mark_weights_dirty(self):
        """Call after modifying parameters through ``.data`` (in-place ops on the parameters
        themselves, ``load_state_dict`` and optimisers are tracked automatically)."""
        self._packed_version = None

    def set_debug_taps(self, on: bool = True):
        self._debug = bool(on)
        self._reserved = 0
        _lib.check(_lib.lib().btsbot_set_debug(self._handle.ptr, int(on)), "btsbot_set_debug")

    # -- execution ----------------------------------------------------------------------------
    def _version(self):
        return tuple(t._version for t in self._tensor_list())

    def _prepare(self, device: torch.device, batch: int, train_only: bool = False):
        if device.type != "cuda":
            raise RuntimeError(
                "btsbot_amd: this model runs only on an AMD GPU through libbtsbot_hip.so "
                f"(tensors are on {device}); there is no CPU fallback.  Move the model and its "
                "inputs to 'cuda'.")
        if self._arena.device != device:
            raise RuntimeError(f"btsbot_amd: model parameters are on {self._arena.device} but the "
                               f"input is on {device}")
        L = _lib.lib()
        if self._handle_device is not None and self._handle_device != device:
            self._new_handle()
            if self._debug:
                L.btsbot_set_debug(self._handle.ptr, 1)
        self._handle_device = device
        stream = torch.cuda.current_stream(device).cuda_stream
        ver = self._version()
        if ver != self._packed_version or (not train_only and not getattr(self, "_packed_full", True)):
            # a training step that differentiates the image branch only needs the per-op operand images
            pack = L.btsbot_pack_params_train if train_only else L.btsbot_pack_params
            _lib.check(pack(self._handle.ptr, C.c_void_p(self._arena.data_ptr()), C.c_void_p(stream)),
                       "btsbot_pack_params")
            self._packed_version = ver
            self._packed_full = not train_only
        chunk = min(max(batch, 1), getattr(self, "_max_chunk", MAX_CHUNK))
        if chunk > self._reserved:
            _lib.check(L.btsbot_reserve(self._handle.ptr, chunk), "btsbot_reserve")
            self._reserved = chunk
        return L, stream

    def _check_inputs(self, image, meta):
        ref = image if image is not None else meta
        if ref is None:
            raise ValueError("btsbot_amd: no input tensor")
        batch, dev = ref.shape[0], ref.device
        if image is not None:
            if image.dim() != 4 or tuple(image.shape[1:]) != (3, 63, 63):
                raise ValueError(f"image input must be [B,3,63,63], got {tuple(image.shape)}")
            image = image.to(torch.float32).contiguous()
        if meta is not None:
            n_meta = self._cfg_args["n_meta"]
            if meta.dim() != 2 or meta.shape[1] != n_meta or meta.shape[0] != batch:
                raise ValueError(f"metadata input must be [{batch},{n_meta}], got {tuple(meta.shape)}")
            meta = meta.to(torch.float32).contiguous()
        if dev.type != "cuda":
            raise RuntimeError(
                "btsbot_amd: this model runs only on an AMD GPU through libbtsbot_hip.so "
                f"(input is on {dev}); there is no CPU fallback.  Move the model and its inputs "
                "to 'cuda'.")
        return image, meta, batch, dev

    def _run(self, image: Optional[torch.Tensor], meta: Optional[torch.Tensor],
             want_scores: bool = False):
        image, meta, batch, dev = self._check_inputs(image, meta)
        if self.training:
            return self._run_train(image, meta)
        if batch == 0:
            empty = torch.empty(0, 1, dtype=torch.float32, device=dev)
            return (empty, empty.clone()) if want_scores else empty
        with torch.cuda.device(dev):
            L, stream = self._prepare(dev, batch)
            logits = torch.empty(batch, dtype=torch.float32, device=dev)
            scores = torch.empty(batch, dtype=torch.float32, device=dev) if want_scores else None
            _lib.check(L.btsbot_forward(
                self._handle.ptr,
                C.c_void_p(image.data_ptr() if image is not None else 0),
                C.c_void_p(meta.data_ptr() if meta is not None else 0),
                C.c_void_p(logits.data_ptr()),
                C.c_void_p(scores.data_ptr() if scores is not None else 0),
                batch, 0, 0, C.c_void_p(stream)), "btsbot_forward")
        logits = logits.view(batch, 1)
        if want_scores:
            return logits, scores.view(batch, 1)
        return logits

    # -- training mode ------------------------------------------------------------------------
    def _slot_groups(self):
        """(comb, meta, image) lists of (tensor, off, numel, shape) over the arena slots."""
        comb, meta, image = [], [], []
        for (canon, *_), (key, parent, leaf, off, numel, shape, is_buf), t in zip(
                self._table_rows, self._slots, self._tensor_list()):
            if is_buf:
                continue
            dest = comb if canon.startswith("comb.") else meta if canon.startswith("meta.") else image
            dest.append((t, off, numel, shape))
        return comb, meta, image

    def _dropout_masks(self, batch: int, dev, generator=None):
        """uint8 keep-masks [B, meta_fc1] and [B, comb_fc2] drawn from torch's device RNG (or `generator`, or
        the ones a test planted in ``_forced_masks``)."""
        forced = getattr(self, "_forced_masks", None)
        a = self._cfg_args
        layers = (("meta", a["meta_fc1"], a["meta_dropout"]), ("comb", a["comb_fc2"], a["comb_dropout"]))
        drawn = [(n, w, p) for n, w, p in layers if w > 0 and p > 0.0 and not (forced is not None and n in forced)]
        fresh = {}
        if drawn:
            # both masks from ONE draw and ONE compare (two launches instead of six small ones in front of every step):
            # the thresholds of the layers side by side in a cached vector, the bool result viewed as bytes
            key = (batch, str(dev), tuple(drawn))
            cache = getattr(self, "_mask_thr", None)
            if cache is None or cache[0] != key:
                thr = torch.cat([torch.full((batch * w,), float(p), dtype=torch.float32) for _n, w, p in drawn]).to(dev)
                self._mask_thr = cache = (key, thr)
            keep = torch.rand(cache[1].numel(), device=dev, generator=generator) >= cache[1]
            off = 0
            for n, w, _p in drawn:
                fresh[n] = keep[off:off + batch * w].view(batch, w).view(torch.uint8)
                off += batch * w
        out = []
        for name, width, p in layers:
            if width <= 0 or p <= 0.0:
                out.append(None)
            elif forced is not None and name in forced:
                out.append(forced[name].to(device=dev, dtype=torch.uint8).contiguous())
            else:
                out.append(fresh[name])
        return out

    def _run_train(self, image, meta):
        """model.train() forward: BatchNorm batch statistics (+ running-stat update), dropout,
        activations cached for backward.  Differentiable w.r.t. every parameter that has
        requires_grad (fusion head, metadata branch, ConvNeXt image branch)."""
        comb, metag, imageg = self._slot_groups()
        grad_on = torch.is_grad_enabled()
        trainable = [g for g in imageg + metag + comb if g[0].requires_grad] if grad_on else []
        keep_image = grad_on and any(t.requires_grad for t, *_ in imageg)
        ref = image if image is not None else meta
        batch, dev = ref.shape[0], ref.device
        if batch < 1:
            raise ValueError("btsbot_amd: training-mode forward needs a non-empty batch")
        masks = self._dropout_masks(batch, dev)
        if not trainable:
            return self._forward_train_raw(image, meta, masks)
        return _TrainFn.apply(self, image, meta, masks, (trainable, keep_image),
                              *[g[0] for g in trainable])

    def _image_bn_modules(self):
        """Modules of the image branch that own BatchNorm running statistics (MaxViT's BatchNorm2d)."""
        out = []
        for (canon, *_), (key, parent, leaf, off, numel, shape, is_buf) in zip(self._table_rows, self._slots):
            if is_buf and leaf == "running_mean" and not canon.startswith(("comb.", "meta.")):
                out.append(parent)
        return out

    def train(self, mode: bool = True):
        """nn.Module.train, except that a FROZEN image branch with BatchNorm2d (MaxViT with ``requires_grad_(False)`` on
        its parameters: the regime of the published "-metadata" checkpoints, train.py:224-232) keeps its BatchNorm
        containers in eval mode: ``model.train()`` after a validation pass (train.py:332-340, val.py:55) would
        otherwise flip them back and the next step would refuse to run.  A branch with trainable parameters goes to
        train mode like everything else (batch statistics, the reference's semantics)."""
        super().train(mode)
        if mode and getattr(self, "_inference_only", False):
            _comb, _meta, imageg = self._slot_groups()
            if not any(t.requires_grad for t, *_ in imageg):
                for m in self._image_bn_modules():
                    m.eval()
        return self

    def _check_train_supported(self, keep_image: bool):
        """The MaxViT image branch trains as a whole or not at all: with trainable parameters its BatchNorm2d layers
        must be in train mode (batch statistics + the backward of every layer: maxvit_train.hip, what
        ``model.train()`` gives); frozen (``requires_grad_(False)`` on all its parameters) they must be in eval mode
        (the inference kernels with running statistics folded in; heads and the metadata branch train over fixed
        image features).  The two mixed cases are not built and raise."""
        if not getattr(self, "_inference_only", False):
            return
        bn_train = [m.training for m in self._image_bn_modules()]
        if keep_image and not all(bn_train):
            raise NotImplementedError(
                f"btsbot_amd.{type(self).__name__}: a trainable MaxViT image branch with BatchNorm2d layers in eval "
                "mode is not built; call .train() on the branch (batch statistics) or freeze it")
        if not keep_image and any(bn_train):
            raise NotImplementedError(
                f"btsbot_amd.{type(self).__name__}: BatchNorm2d batch statistics of a FROZEN MaxViT image branch are "
                "not built; put the branch in eval mode (e.g. model.train(); model.maxvit_backbone.eval()) or "
                "call .eval() on the whole model")

    def _forward_train_raw(self, image, meta, masks, keep_image: bool = False):
        self._check_train_supported(keep_image)
        ref = image if image is not None else meta
        batch, dev = ref.shape[0], ref.device
        with torch.cuda.device(dev):
            L, stream = self._prepare(dev, batch, train_only=keep_image and self._reserved_image)
            if batch > self._reserved_train or (keep_image and not self._reserved_image):
                _lib.check(L.btsbot_reserve_train(self._handle.ptr, max(batch, self._reserved_train),
                                                  int(keep_image or self._reserved_image)),
                           "btsbot_reserve_train")
                self._reserved_train = max(batch, self._reserved_train)
                if keep_image and not self._reserved_image:
                    self._reserved_image = True
                    # the dgrad operand images are packed from now on
                    _lib.check(L.btsbot_pack_params(self._handle.ptr,
                                                    C.c_void_p(self._arena.data_ptr()),
                                                    C.c_void_p(stream)), "btsbot_pack_params")
            logits = torch.empty(batch, dtype=torch.float32, device=dev)
            self._live_masks = masks          # must outlive btsbot_backward
            _lib.check(L.btsbot_forward_train(
                self._handle.ptr,
                C.c_void_p(image.data_ptr() if image is not None else 0),
                C.c_void_p(meta.data_ptr() if meta is not None else 0),
                C.c_void_p(logits.data_ptr()), C.c_void_p(0), batch,
                C.c_void_p(masks[0].data_ptr() if masks[0] is not None else 0),
                C.c_void_p(masks[1].data_ptr() if masks[1] is not None else 0),
                C.c_void_p(self._arena.data_ptr()), int(keep_image), C.c_void_p(stream)),
                "btsbot_forward_train")
        # the kernel updated running_mean / running_var inside the arena
        for mod in self.modules():   # (eval-mode BatchNorm does not count batches: a frozen MaxViT branch)
            if "num_batches_tracked" in mod._buffers and mod.training:
                mod._buffers["num_batches_tracked"] += 1
        self._packed_version = None
        return logits.view(batch, 1)

    def _backward_raw(self, dlogits: torch.Tensor, need_meta: bool,
                      need_image: bool = False) -> torch.Tensor:
        """d(loss)/d(param) into the flat gradient arena (master-arena layout); returns the arena."""
        dev = self._arena.device
        if getattr(self, "_grad_arena", None) is None or self._grad_arena.device != dev:
            object.__setattr__(self, "_grad_arena", torch.zeros_like(self._arena))
        dl = dlogits.reshape(-1).to(torch.float32).contiguous()
        with torch.cuda.device(dev):
            stream = torch.cuda.current_stream(dev).cuda_stream
            _lib.check(_lib.lib().btsbot_backward(
                self._handle.ptr, C.c_void_p(dl.data_ptr()), C.c_void_p(self._grad_arena.data_ptr()),
                int(need_meta), int(need_image), C.c_void_p(stream)), "btsbot_backward")
        return self._grad_arena

    def _grad_buckets(self):
        """Arena ranges [(lo, hi)] in the order btsbot_backward() completes them (btsbot_grad_buckets)."""
        lo, hi = (C.c_int64 * 8)(), (C.c_int64 * 8)()
        n = _lib.check(_lib.lib().btsbot_grad_buckets(self._handle.ptr, 8, lo, hi), "btsbot_grad_buckets")
        return [(int(lo[i]), int(hi[i])) for i in range(n)]

    def _wait_grad_bucket(self, bucket: int, stream: int):
        """Make the HIP stream `stream` wait for bucket `bucket` of the last backward (no host sync)."""
        _lib.check(_lib.lib().btsbot_wait_grad_bucket(self._handle.ptr, int(bucket), C.c_void_p(stream)),
                   "btsbot_wait_grad_bucket")

    def set_profile(self, on: bool = True):
        """Bracket every kernel launch of forward() with HIP events (bench.py's roofline leg)."""
        _lib.check(_lib.lib().btsbot_set_profile(self._handle.ptr, int(on)), "btsbot_set_profile")

    def collect_profile(self) -> Dict[str, Tuple[float, int]]:
        """{kernel family: (summed device ms, launches)} since the last collect."""
        L = _lib.lib()
        n = L.btsbot_profile_categories()
        ms = (C.c_double * n)()
        cnt = (C.c_int64 * n)()
        _lib.check(L.btsbot_profile_collect(self._handle.ptr, n, ms, cnt), "btsbot_profile_collect")
        return {L.btsbot_profile_category_name(i).decode(): (ms[i], int(cnt[i])) for i in range(n)}

    def read_tap(self, name: str) -> torch.Tensor:
        """fp32 NHWC copy of 'stem' / 'stage0'..'stage3' from the last forward chunk
        (needs set_debug_taps(True) before the forward)."""
        hw = {"stem": 15, "stage0": 15, "stage1": 7, "stage2": 3, "stage3": 1}[name]
        ci = 0 if name == "stem" else int(name[-1])
        c = self._cfg_args["dims"][ci]
        dev = self._arena.device
        buf = torch.empty(self._reserved * hw * hw * c, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            stream = torch.cuda.current_stream(dev).cuda_stream
            n = _lib.check(_lib.lib().btsbot_read_tap(
                self._handle.ptr, name.encode(), C.c_void_p(buf.data_ptr()), buf.numel(),
                C.c_void_p(stream)), "btsbot_read_tap")
        return buf[:n].view(-1, hw * hw, c)


class _TrainFn(torch.autograd.Function):
    """Autograd node around btsbot_forward_train / btsbot_backward (train.py:510,526)."""

    @staticmethod
    def forward(ctx, model, image, meta, masks, plan, *params):
        trainable, keep_image = plan
        ctx.model = model
        ctx.trainable = trainable
        ctx.keep_image = keep_image
        ctx.inputs = (image, meta, masks)   # the C side re-reads them in backward: keep them alive
        return model._forward_train_raw(image, meta, masks, keep_image)

    @staticmethod
    def backward(ctx, dlogits):
        model, trainable = ctx.model, ctx.trainable
        _comb, metag, _img = model._slot_groups()
        meta_ids = {id(t) for t, *_ in metag}
        need_meta = any(id(t) in meta_ids for t, *_ in trainable)
        arena = model._backward_raw(dlogits, need_meta, ctx.keep_image)
        grads = [arena[off:off + numel].view(shape).clone() for _t, off, numel, shape in trainable]
        return (None, None, None, None, None, *grads)


def _backbone_key_map(bprefix: str, head_norm_key: str, meta_prefix: str, comb_keys):
    def f(canon: str) -> str:
        if canon.startswith("head_norm."):
            return head_norm_key + canon[len("head_norm."):]
        if canon.startswith("meta."):
            return meta_prefix + canon[len("meta."):]
        if canon.startswith("comb."):
            idx, leaf = canon[len("comb."):].split(".")
            return f"{comb_keys[int(idx)]}.{leaf}"
        return bprefix + canon
    return f


def _warn_pretrained(config: dict):
    if config.get("pretrained", True):
        warnings.warn(
            "btsbot_amd: config['pretrained'] is true, but pretrained timm/HF-hub backbones are not "
            "fetched here (no timm, no network); parameters are randomly initialised until "
            "load_state_dict() is called.", stacklevel=3)


class mm_ConvNeXt(_HipModel):
    """architectures.py:125-171 -- ConvNeXt image branch + GELU metadata branch + GELU fusion head."""
    _wiring = "mm_ConvNeXt"

    def __init__(self, config, precision: Optional[str] = None):
        super().__init__()
        self._init_config = dict(config)
        _warn_pretrained(config)
        table = _convnext_table(config.get("model_kind", _DEFAULT_KIND))
        self.convnext_feature_dim = table[1][-1]
        ls = "LS" in config["train_data_version"]
        self._setup(
            table=table, head_norm=ls, n_meta=len(config.get("metadata_cols", [])),
            meta_fc=(config["meta_fc1_neurons"], config["meta_fc2_neurons"]),
            comb_fc=(config["comb_fc1_neurons"], config["comb_fc2_neurons"]),
            dropouts=(config["meta_dropout"], config["comb_dropout"]),
            key_map=_backbone_key_map("convnext_backbone.", "convnext_backbone.head.1.",
                                      "metadata_branch.",
                                      ["combined_head.0", "combined_head.2", "combined_head.5"]),
            precision=precision or config.get("precision"))

    def forward(self, image_input: torch.Tensor, metadata_input: torch.Tensor) -> torch.Tensor:
        return self._run(image_input, metadata_input)


class ConvNeXt(_HipModel):
    """architectures.py:104-122 -- image-only ConvNeXt with a pooled, normalised MLP head."""
    _wiring = "ConvNeXt"

    def __init__(self, config, precision: Optional[str] = None):
        super().__init__()
        self._init_config = dict(config)
        _warn_pretrained(config)
        table = _convnext_table(config.get("model_kind", _DEFAULT_KIND))
        self._setup(
            table=table, head_norm=True, n_meta=0, meta_fc=(0, 0),
            comb_fc=(config["fc1_neurons"], config["fc2_neurons"]),
            dropouts=(0.0, config["dropout"]),
            key_map=_backbone_key_map("convnext.", "convnext.head.1.", "",
                                      ["convnext.head.3", "convnext.head.5", "convnext.head.8"]),
            precision=precision or config.get("precision"))

    def forward(self, input_data: torch.Tensor) -> torch.Tensor:
        return self._run(input_data, None)


class um_nn(_HipModel):
    """architectures.py:277-293 -- metadata-only MLP."""
    _wiring = "um_nn"

    def __init__(self, config, precision: Optional[str] = None):
        super().__init__()
        self._init_config = dict(config)
        self._setup(
            table=None, head_norm=False, n_meta=len(config.get("metadata_cols", [])),
            meta_fc=(config["meta_fc1_neurons"], config["meta_fc2_neurons"]), comb_fc=(0, 0),
            dropouts=(config["meta_dropout"], 0.0),
            key_map=_backbone_key_map("", "", "network.", ["network.6"]),
            precision=precision or config.get("precision"))

    def forward(self, input_data: torch.Tensor) -> torch.Tensor:
        return self._run(None, input_data)


class frozen_fusion(_HipModel):
    """architectures.py:296-372 -- two trained uni-modal models with their heads removed
    (ConvNeXt -> pool, LayerNorm2d, flatten; um_nn -> BN, Linear, ReLU, Dropout, Linear) feeding a
    ReLU fusion head.  This is what the published "-metadata" HF checkpoints instantiate
    (to_HF.py:142-162)."""
    _wiring = "frozen_fusion"

    @staticmethod
    def _branch_config(config, which):
        cfg = config.get(f"{which}_model_config", None)
        if cfg is None:                                   # architectures.py:324-326
            with open(path.join(config[f"{which}_model_dir"], "report.json"), "r") as f:
                cfg = json.load(f)["train_config"]
        return cfg

    def __init__(self, config, precision: Optional[str] = None):
        super().__init__()
        self._init_config = dict(config)
        icfg = self._branch_config(config, "image")
        mcfg = self._branch_config(config, "meta")
        if icfg["model_name"] not in ("ConvNeXt", "MaxViT") or mcfg["model_name"] != "um_nn":
            raise NotImplementedError(
                "btsbot_amd.frozen_fusion supports a ConvNeXt or MaxViT image branch and a um_nn "
                f"metadata branch (got {icfg['model_name']} / {mcfg['model_name']})")
        _warn_pretrained(icfg)
        if icfg["model_name"] == "MaxViT":
            # architectures.py:304-308: the MaxViT branch keeps head[0:1] = its global pool (no parameters)
            model_kind = icfg.get("model_kind", _MAXVIT_DEFAULT_KIND)
            self.image_size = get_model_image_size(model_kind)
            self._wiring = "frozen_fusion_MaxViT"
            self._inference_only = True
            self._max_chunk = MAXVIT_MAX_CHUNK
            table, head_norm = _maxvit_table(model_kind), False
            bprefix, hn_key = "image_branch.maxvit.", ""
        else:
            table, head_norm = _convnext_table(icfg.get("model_kind", _DEFAULT_KIND)), True
            bprefix, hn_key = "image_branch.convnext.", "image_branch.convnext.head.1."
        self._setup(
            table=table, head_norm=head_norm, n_meta=len(mcfg.get("metadata_cols", [])),
            meta_fc=(mcfg["meta_fc1_neurons"], mcfg["meta_fc2_neurons"]),
            comb_fc=(config["comb_fc1_neurons"], config["comb_fc2_neurons"]),
            dropouts=(mcfg["meta_dropout"], config["comb_dropout"]),
            key_map=_backbone_key_map(bprefix, hn_key, "meta_branch.network.",
                                      ["combined_head.0", "combined_head.2", "combined_head.5"]),
            precision=precision or config.get("precision"))
        if icfg["model_name"] == "MaxViT":
            with torch.no_grad():
                for (canon, *_), t in zip(self._table_rows, self._tensor_list()):
                    if canon.endswith("relative_position_bias_table"):
                        nn.init.trunc_normal_(t, std=0.02)
        if not config.get("skip_load_state", False):      # architectures.py:334-335
            self._load_branch(path.join(config["image_model_dir"], "best_model.pth"),
                              "image_branch.")
            self._load_branch(path.join(config["meta_model_dir"], "best_model.pth"),
                              "meta_branch.")

    def _load_branch(self, file: str, prefix: str):
        state = torch.load(file, map_location="cpu")
        own = dict(self.state_dict())
        picked = {}
        for k, v in state.items():
            k = k[len("module."):] if k.startswith("module.") else k
            if prefix + k in own:
                picked[prefix + k] = v
        missing = [k for k in own if k.startswith(prefix) and k not in picked]
        if missing:
            raise RuntimeError(f"{file}: missing keys for {prefix}: {missing[:5]} ...")
        self.load_state_dict(picked, strict=False)

    def forward(self, image_input: torch.Tensor, metadata_input: torch.Tensor) -> torch.Tensor:
        return self._run(image_input, metadata_input)


def _not_built(name, why):
    class _Missing(nn.Module):
        def __init__(self, config=None, *a, **k):
            raise NotImplementedError(f"btsbot_amd.{name}: {why}")
    _Missing.__name__ = _Missing.__qualname__ = name
    return _Missing


def _maxvit_table(model_kind: str):
    mk = model_kind.lower()
    for name, tab in _MAXVIT_TABLE.items():
        if name in mk:
            return tab
    raise ValueError(f"btsbot_amd: model_kind {model_kind!r} is not a supported MaxViT "
                     f"(have {sorted(_MAXVIT_TABLE)})")


class _MaxVitModel(_HipModel):
    _inference_only = True
    _max_chunk = MAXVIT_MAX_CHUNK

    @torch.no_grad()
    def reset_parameters(self):
        super().reset_parameters()
        for (canon, *_), t in zip(self._table_rows, self._tensor_list()):
            if canon.endswith("relative_position_bias_table"):
                nn.init.trunc_normal_(t, std=0.02)

    def read_tap(self, name: str) -> torch.Tensor:
        """fp32 NHWC copy of 'stem' (112x112x64) / 'stage0'..'stage3' (56, 28, 14, 7) of the last chunk."""
        hw = {"stem": 112, "stage0": 56, "stage1": 28, "stage2": 14, "stage3": 7}[name]
        c = self._cfg_args["dims"][0 if name == "stem" else int(name[-1])]
        dev = self._arena.device
        buf = torch.empty(self._reserved * hw * hw * c, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            stream = torch.cuda.current_stream(dev).cuda_stream
            n = _lib.check(_lib.lib().btsbot_read_tap(
                self._handle.ptr, name.encode(), C.c_void_p(buf.data_ptr()), buf.numel(),
                C.c_void_p(stream)), "btsbot_read_tap")
        return buf[:n].view(-1, hw * hw, c)


class MaxViT(_MaxVitModel):
    """architectures.py:25-55 -- image-only MaxViT: bilinear resize to 224, maxvit_tiny_rw_224, global
    average pool, Linear-GELU-Linear-GELU-Dropout-Linear (keys maxvit.head.{1,3,6})."""
    _wiring = "MaxViT"

    def __init__(self, config, precision: Optional[str] = None):
        super().__init__()
        self._init_config = dict(config)
        _warn_pretrained(config)
        model_kind = config.get("model_kind", _MAXVIT_DEFAULT_KIND)
        self.image_size = get_model_image_size(model_kind)
        self._setup(
            table=_maxvit_table(model_kind), head_norm=False, n_meta=0, meta_fc=(0, 0),
            comb_fc=(config["fc1_neurons"], config["fc2_neurons"]),
            dropouts=(0.0, config["dropout"]),
            key_map=_backbone_key_map("maxvit.", "", "",
                                      ["maxvit.head.1", "maxvit.head.3", "maxvit.head.6"]),
            precision=precision or config.get("precision"))

    def forward(self, input_data: torch.Tensor) -> torch.Tensor:
        return self._run(input_data, None)


class mm_MaxViT(_MaxVitModel):
    """architectures.py:58-101 -- MaxViT image branch + GELU metadata branch + GELU fusion head."""
    _wiring = "mm_MaxViT"

    def __init__(self, config, precision: Optional[str] = None):
        super().__init__()
        self._init_config = dict(config)
        _warn_pretrained(config)
        model_kind = config.get("model_kind", _MAXVIT_DEFAULT_KIND)
        self.image_size = get_model_image_size(model_kind)
        table = _maxvit_table(model_kind)
        self.maxvit_feature_dim = table[1][-1]
        self._setup(
            table=table, head_norm=False, n_meta=len(config.get("metadata_cols", [])),
            meta_fc=(config["meta_fc1_neurons"], config["meta_fc2_neurons"]),
            comb_fc=(config["comb_fc1_neurons"], config["comb_fc2_neurons"]),
            dropouts=(config["meta_dropout"], config["comb_dropout"]),
            key_map=_backbone_key_map("maxvit_backbone.", "", "metadata_branch.",
                                      ["combined_head.0", "combined_head.2", "combined_head.5"]),
            precision=precision or config.get("precision"))

    def forward(self, image_input: torch.Tensor, metadata_input: torch.Tensor) -> torch.Tensor:
        return self._run(image_input, metadata_input)


mm_cnn = _not_built("mm_cnn", "legacy VGG-like CNN, out of scope (SURVEY.md section 2)")
um_cnn = _not_built("um_cnn", "legacy VGG-like CNN, out of scope (SURVEY.md section 2)")
