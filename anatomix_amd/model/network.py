"""Drop-in mirror of ``anatomix.model.network`` (reference: anatomix/model/network.py) whose
inference forward runs on hand-written gfx950 kernels through the C ABI of
``include/anatomix_amd.h``.

What is kept from the reference surface:
  * ``Unet(dimension, input_nc, output_nc, num_downs, ngf=24, norm='batch', final_act='none',
    activation='relu', pad_type='reflect', doubleconv=True, residual_connection=False,
    pooling='Max', interp='nearest', use_skip_connection=True, norm_eps=1e-5)`` -- same
    positional order and defaults as network.py:262-279;
  * ``.model`` is an ``nn.Sequential`` of real ``nn.Conv3d / nn.BatchNorm3d / nn.ReLU / nn.MaxPool3d /
    nn.Upsample`` children at the same integer indices, so ``state_dict`` keys, ``load_state_dict(
    strict=True)``, ``.apply(init_fn)``, ``for i, layer in enumerate(net.model)`` all behave
    identically (network.py:465);
  * ``encoder_idx / decoder_idx / res_source / res_dest / use_bias / use_skip_connection /
    residual_connection`` attributes and the two constructor prints (network.py:447-448);
  * ``forward(input, layers=[], encode_only=False, verbose=False)`` (network.py:467).

What is different: for CUDA(ROCm) inputs under ``torch.no_grad()`` the forward -- plain, or with ``layers`` /
``encode_only`` feature taps (``amx_unet_forward_taps``) -- is ONE call into libanatomix_amd.so (20 fused conv
launches for the 6M model) instead of 66 module calls; with autograd enabled the whole network is one
``torch.autograd.Function`` whose forward AND backward run on the HIP kernels (``anatomix_amd/model/train.py``:
train-mode BatchNorm / InstanceNorm, pooling and upsampling adjoints, data and weight gradients -- what the contrastive
step and the segmentation finetuning call).  What the HIP path does not cover raises with the reason
(``hip_unsupported_reason``): CPU tensors, ``dimension`` 1 / 2, ``pad_type`` other than reflect,
``residual_connection=True``, activations other than relu / lrelu / none, input_nc > 16, ngf outside {8, 16, 24, 32} --
unless the stock-module path is explicitly enabled with ``model.allow_torch_path = True`` (or env
AMX_ALLOW_TORCH_PATH=1).  There is no silent fallback.
"""
from __future__ import annotations

import ctypes
import os
import warnings
from functools import partial

import torch
import torch.nn as nn

from .. import _lib


def get_norm_layer(ndims, norm="batch", eps=1e-5):
    """Same contract as network.py:127-168: returns a callable ``Norm(num_features)`` or None."""
    table = {
        "batch": lambda: partial(getattr(nn, f"BatchNorm{ndims}d"), eps=eps),
        "instance": lambda: partial(getattr(nn, f"InstanceNorm{ndims}d"), eps=eps),
        "instance_affine": lambda: partial(getattr(nn, f"InstanceNorm{ndims}d"), affine=True, eps=eps),
        "none": lambda: None,
    }
    if norm not in table:
        raise ValueError(f"Currently unsupported normalization: {norm}")
    return table[norm]()


def get_actvn_layer(activation="relu"):
    """Same contract as network.py:171-204 (one module instance, shared by every site)."""
    makers = {
        "relu": lambda: nn.ReLU(inplace=True),
        "lrelu": lambda: nn.LeakyReLU(0.3, inplace=True),
        "elu": nn.ELU,
        "prelu": nn.PReLU,
        "selu": lambda: nn.SELU(inplace=True),
        "tanh": nn.Tanh,
        "none": lambda: None,
    }
    assert activation in makers, "Unsupported activation: {}".format(activation)
    return makers[activation]()


class Unet(nn.Module):
    """MI355X-native U-Net with the constructor / state_dict / forward surface of the reference
    ``anatomix.model.network.Unet`` (network.py:210-548)."""

    def __init__(self, dimension, input_nc, output_nc, num_downs, ngf=24, norm="batch", final_act="none",
                 activation="relu", pad_type="reflect", doubleconv=True, residual_connection=False,
                 pooling="Max", interp="nearest", use_skip_connection=True, norm_eps=1e-5):
        super().__init__()
        assert dimension in [1, 2, 3], "ndims should be 1--3. found: %d" % dimension
        self.use_bias = norm == "instance"          # network.py:292
        self.residual_connection = residual_connection
        self.use_skip_connection = use_skip_connection
        self.res_source, self.res_dest = [], []
        self.encoder_idx, self.decoder_idx = [], []
        self._cfg = dict(dimension=dimension, input_nc=input_nc, output_nc=output_nc, num_downs=num_downs,
                         ngf=ngf, norm=norm, final_act=final_act, activation=activation, pad_type=pad_type,
                         doubleconv=doubleconv, residual_connection=residual_connection, pooling=pooling,
                         interp=interp, use_skip_connection=use_skip_connection, norm_eps=norm_eps)

        Conv = getattr(nn, "Conv%dd" % dimension)
        Pool = getattr(nn, "%sPool%dd" % (pooling, dimension))
        Norm = get_norm_layer(dimension, norm, eps=norm_eps)
        act = get_actvn_layer(activation)          # ONE shared instance, as in the reference
        final = get_actvn_layer(final_act)
        seq = []

        def block(cin, cout):
            """conv -> norm -> act, recording the residual source/destination indices."""
            seq.append(Conv(cin, cout, kernel_size=3, stride=1, bias=self.use_bias, padding="same",
                            padding_mode=pad_type))
            self.res_source.append(len(seq) - 1)
            if Norm is not None:
                seq.append(Norm(cout))
            if act is not None:
                seq.append(act)
            self.res_dest.append(len(seq) - 1)

        block(input_nc, ngf)                                        # stem
        width = ngf
        for level in range(num_downs):                              # encoder
            grow = 1 if level == 0 else 2
            block(width, width * grow)
            if doubleconv:
                block(width * grow, width * grow)
            self.encoder_idx.append(len(seq) - 1)
            seq.append(Pool(2))
            width *= grow
        block(width, width * 2)                                     # bottleneck
        if doubleconv:
            block(width * 2, width * 2)
        mult = 2 ** num_downs
        for _ in range(num_downs):                                  # decoder
            self.decoder_idx.append(len(seq))
            seq.append(nn.Upsample(scale_factor=2, mode=interp))
            cat_mult = mult + mult // 2 if use_skip_connection else mult
            block(ngf * cat_mult, ngf * (mult // 2))
            if doubleconv:
                block(ngf * (mult // 2), ngf * (mult // 2))
            mult //= 2
        print("Encoder skip connect id", self.encoder_idx)
        print("Decoder skip connect id", self.decoder_idx)
        seq.append(Conv(ngf * mult, output_nc, kernel_size=3, stride=1, bias=self.use_bias, padding="same",
                        padding_mode=pad_type))                    # bare output conv
        if final is not None:
            seq.append(final)
        self.model = nn.Sequential(*seq)

        # ---- HIP-path state (not part of the reference surface)
        self._precision = os.environ.get("AMX_PRECISION") or None      # None: chosen per configuration, see `precision`
        self.allow_torch_path = os.environ.get("AMX_ALLOW_TORCH_PATH", "0") == "1"
        self._handle = None
        self._handle_key = None
        self._weights_dirty = True
        self._workspace = None
        self._side_streams, self._side_ws = None, [None, None]
        self.concurrent_chunks = 4       # batches of >= 2 chunks of this many volumes run as chunks on two streams (0: off)
        self._warned = False
        self.register_load_state_dict_post_hook(lambda module, incompatible: module._mark_dirty())

    # ------------------------------------------------------------------------------------------
    # storage precision of the HIP path
    # ------------------------------------------------------------------------------------------
    # Measured rel-L2 of the 16-bit storage modes against the fp32 reference at 128^3: networks whose norm
    # folds into the convolution (eval BatchNorm / none) hold the 1e-3 tolerance in f16; networks that normalise with statistics
    # of their own activations (InstanceNorm: `anatomix-dev`) amplify every operand rounding and do NOT (1.1e-2).
    _F16_ERROR_WITH_LIVE_NORM = 1.1e-2

    @property
    def precision(self):
        """Storage precision of the HIP path: ``"f16"``, ``"bf16"``, ``"f16x2"``, ``"strict"`` (= ``"bf16x2"``) or ``"f16x2mx"``.

        Unless set explicitly (attribute or env ``AMX_PRECISION``) it is the fastest mode that keeps the features within 1e-3
        (relative) of the reference's fp32 inference (convex_adam_utils.py:194-219 runs the network in fp32): ``"f16"`` where the
        norm folds into the convolution weights (the 6 M ``anatomix`` variant); for InstanceNorm networks (``anatomix-dev``,
        load_from_hf.py:18-24), on which single 16-bit storage is 10x outside the tolerance, ``"f16x2mx"`` -- f16 hi + lo pairs whose
        correction products run on the block-scaled fp8 matrix instruction (2.5e-4 rel-L2 / 4.4e-4 max-norm on ``anatomix-dev`` at
        128^3, 1.34x the throughput of ``"strict"``; its f16 range is guarded like ``"f16"``: AmxOverflowError, never silent Inf) --
        or ``"strict"`` (1.65e-4, fp32's exponent range) where that mode is not implemented (input_nc > 1)."""
        if self._precision is not None:
            return self._precision
        if self._cfg["norm"] in ("instance", "instance_affine"):
            return "f16x2mx" if self._cfg["input_nc"] == 1 else "strict"
        return "f16"

    @property
    def train_precision(self):
        """Storage precision of the differentiable HIP path (model/train.py): the explicit setting if there is one, else
        ``"bf16"`` -- the reference trains under bf16 autocast (supcl_model.py:603-661)."""
        return self._precision if self._precision is not None else "bf16"

    @precision.setter
    def precision(self, value):
        if value is not None and value not in _lib.PRECISION:
            raise ValueError(f"precision must be one of {sorted(_lib.PRECISION)} or None (per-configuration default), got {value!r}")
        if value in ("f16", "bf16") and self._cfg["norm"] in ("instance", "instance_affine") and not getattr(self, "_warned_precision", False):
            self._warned_precision = True
            warnings.warn(f"anatomix_amd.Unet: precision={value!r} on an InstanceNorm network is outside the 1e-3 tolerance of the fp32 "
                          f"reference (measured rel-L2 {self._F16_ERROR_WITH_LIVE_NORM:.1e} on anatomix-dev at 128^3); the default for this "
                          "configuration is 'f16x2mx' (or 'strict')")
        self._precision = value

    # ------------------------------------------------------------------------------------------
    # bookkeeping: repack weights whenever parameters may have changed
    # ------------------------------------------------------------------------------------------
    def _mark_dirty(self):
        self._weights_dirty = True
        self.__dict__["_sig_tensors"] = None

    # The native handle and the workspace belong to ONE module object on ONE device.  copy.copy / copy.deepcopy / pickling /
    # nn.DataParallel replicas (which copy __dict__) must not share them: a copy starts without a handle and builds its own on
    # its first HIP forward (sharing the raw pointer would be a use-after-free on the first re-create and a double free in
    # __del__; ctypes pointers do not pickle either).
    _PER_OBJECT_STATE = ("_handle", "_handle_key", "_workspace", "_uploaded_sig", "_sig_tensors", "_side_streams", "_side_ws", "_sw_streams", "_sw_ws")

    def __getstate__(self):
        state = self.__dict__.copy()
        for k in self._PER_OBJECT_STATE:
            state[k] = None
        state["_weights_dirty"] = True
        return state

    def __copy__(self):
        new = self.__class__.__new__(self.__class__)
        new.__dict__.update(self.__getstate__())
        return new

    def _replicate_for_data_parallel(self):
        replica = super()._replicate_for_data_parallel()
        for k in self._PER_OBJECT_STATE:
            replica.__dict__[k] = None
        replica.__dict__["_weights_dirty"] = True
        return replica

    def _param_signature(self):
        """In-place updates (optimizer steps, nn.init.*, net.apply(init_func) as pretraining_networks.py:687-715 does) bump
        every tensor's version counter: comparing them per call catches changes no hook sees.  The list of tensors is cached
        together with WHERE each one lives (owner's ``_parameters`` / ``_buffers`` dict and key) and with the children of
        ``self.model``: the per-forward cost is one identity test per tensor and per child plus ~100 (pointer, version) pairs,
        not a walk of the module tree -- and a Parameter object replaced on a submodule (``m.model[0].weight = nn.Parameter(w)``,
        weight_norm / parametrize / prune) or a swapped child (``m.model[1] = nn.BatchNorm3d(..)``) is still seen: the cache is
        rebuilt, the new tensors give a new signature, and the weights are packed again."""
        cache = self.__dict__.get("_sig_tensors")
        if cache is not None:
            slots, children = cache
            kids = self.model._modules
            if len(kids) != len(children) or any(a is not b for a, b in zip(kids.values(), children)) or \
                    any(d.get(k) is not t for d, k, t in slots):
                cache = None
        if cache is None:
            slots = []
            for mod in self.modules():
                slots += [(mod._parameters, k, t) for k, t in mod._parameters.items() if t is not None]
                slots += [(mod._buffers, k, t) for k, t in mod._buffers.items() if t is not None]
            cache = self.__dict__["_sig_tensors"] = (slots, list(self.model._modules.values()))
        return tuple((t.data_ptr(), t._version) for _, _, t in cache[0])

    def _weights_stale(self):
        return self._weights_dirty or getattr(self, "_uploaded_sig", None) != self._param_signature()

    def refresh_weights(self):
        """Call after mutating parameters in place (e.g. a manual ``param.data.copy_``)."""
        self._mark_dirty()

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        self._mark_dirty()
        self._workspace = None
        return out

    def train(self, mode=True):
        self._mark_dirty()
        return super().train(mode)

    def __del__(self):
        try:
            if getattr(self, "_handle", None) is not None:
                _lib.load().amx_unet_destroy(self._handle)
        except Exception:
            pass

    # ------------------------------------------------------------------------------------------
    # HIP path
    # ------------------------------------------------------------------------------------------
    def hip_unsupported_reason(self, x, layers=()):
        """None if this call can run on the HIP kernels, else a human-readable reason."""
        c = self._cfg
        if not x.is_cuda:
            return "input is not on a GPU device"
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters())):
            return "autograd is enabled on a configuration the differentiable HIP path (anatomix_amd.model.train) does not cover; wrap inference in torch.no_grad()"
        if c["dimension"] != 3 or c["pad_type"] != "reflect" or c["residual_connection"]:
            return "only dimension=3, pad_type='reflect', residual_connection=False are implemented"
        if c["norm"] == "batch" and self.training:
            return "train-mode BatchNorm with encode_only is not implemented in the HIP path; call .eval()"
        if c["norm"] not in _lib.NORM:
            return f"norm='{c['norm']}' is not implemented in the HIP path"
        if c["activation"] not in _lib.ACT or c["final_act"] not in _lib.ACT:
            return "activation not implemented in the HIP path"
        if c["interp"] not in _lib.INTERP or c["pooling"] not in _lib.POOL:
            return f"interp='{c['interp']}' / pooling='{c['pooling']}' is not implemented in the HIP path"
        if not 1 <= c["input_nc"] <= 16 or c["ngf"] not in (8, 16, 24, 32) or c["output_nc"] < 1:
            return ("HIP path needs 1 <= input_nc <= 16 and ngf in {8, 16, 24, 32} (the first layer stores 16 or 32 channels; 8 and 24 -- the "
                    "reference's default width -- run with the ngf-wide tensors padded to 16 / 32)")
        if c["num_downs"] < 1 or c["num_downs"] > 7 or (c["ngf"] << c["num_downs"]) > 2048:
            return "HIP path needs 1 <= num_downs <= 7 and at most 2048 channels at the bottleneck"
        if x.dim() != 5 or x.shape[1] != c["input_nc"]:
            return f"expected input of shape [N, {c['input_nc']}, D, H, W]"
        return None

    def _ensure_handle(self, device):
        lib = _lib.load()
        key = (device.index, self.precision)
        if self._handle is not None and self._handle_key == key:
            return lib
        if self._handle is not None:
            lib.amx_unet_destroy(self._handle)
            self._handle = None
        c = self._cfg
        cfg = _lib.UnetCfg(
            input_nc=c["input_nc"], output_nc=c["output_nc"], num_downs=c["num_downs"], ngf=c["ngf"],
            norm=_lib.NORM[c["norm"]], norm_eps=float(c["norm_eps"]), activation=_lib.ACT[c["activation"]],
            act_slope=0.3, final_act=_lib.ACT[c["final_act"]], pooling=_lib.POOL[c["pooling"]],
            interp=_lib.INTERP[c["interp"]], doubleconv=int(bool(c["doubleconv"])),
            use_skip=int(bool(c["use_skip_connection"])), precision=_lib.PRECISION[self.precision])
        h = ctypes.c_void_p()
        with torch.cuda.device(device):
            _lib.check(lib.amx_unet_create(ctypes.byref(h), ctypes.byref(cfg)))
        self._handle, self._handle_key = h, key
        self._weights_dirty = True
        return lib

    def _upload_weights(self, lib, device):
        stream = ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)
        keep = []   # fp32 staging copies must outlive the enqueued pack kernels (same stream => safe)
        nconv = lib.amx_unet_num_convs(self._handle)
        mi, ci, co, ni = ctypes.c_int(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int()

        def dev32(t):
            if t is None:
                return None
            t = t.detach().to(device=device, dtype=torch.float32).contiguous()
            keep.append(t)
            return t

        for k in range(nconv):
            _lib.check(lib.amx_unet_conv_info(self._handle, k, ctypes.byref(mi), ctypes.byref(ci),
                                              ctypes.byref(co), ctypes.byref(ni)))
            conv = self.model[mi.value]
            w, b = dev32(conv.weight), dev32(conv.bias)
            g = be = mu = var = None
            if ni.value >= 0:
                nm = self.model[ni.value]
                g, be = dev32(getattr(nm, "weight", None)), dev32(getattr(nm, "bias", None))
                mu, var = dev32(getattr(nm, "running_mean", None)), dev32(getattr(nm, "running_var", None))
            _lib.check(lib.amx_unet_load_conv(self._handle, mi.value, _lib.ptr(w), _lib.ptr(b), _lib.ptr(g),
                                              _lib.ptr(be), _lib.ptr(mu), _lib.ptr(var), stream))
        self._weights_dirty = False
        self._uploaded_sig = self._param_signature()

    def _get_workspace(self, lib, n, d, h, w, device):
        need = lib.amx_unet_workspace_bytes(self._handle, n, d, h, w)
        ws = self._workspace
        if ws is None or ws.numel() < need or ws.device != device:
            ws = torch.empty(need, dtype=torch.uint8, device=device)
            self._workspace = ws
        return ws, need

    def forward_hip(self, x):
        """[N,1,D,H,W] -> [N,output_nc,D,H,W] fp32, all work enqueued on the current HIP stream."""
        device = x.device
        lib = self._ensure_handle(device)
        with torch.cuda.device(device):
            if self._weights_stale():
                self._upload_weights(lib, device)
            xin = x.detach()
            if xin.dtype != torch.float32 or not xin.is_contiguous():
                xin = xin.float().contiguous()
            n, _, d, h, w = xin.shape
            y = torch.empty((n, self._cfg["output_nc"], d, h, w), dtype=torch.float32, device=device)
            if self.concurrent_chunks and n >= 2 * self.concurrent_chunks and not torch.cuda.is_current_stream_capturing():
                self._forward_chunks(lib, xin, y, device)
            else:
                ws, need = self._get_workspace(lib, n, d, h, w, device)
                stream = ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)
                _lib.check(lib.amx_unet_forward(self._handle, _lib.ptr(xin), _lib.ptr(y), n, d, h, w, _lib.ptr(ws),
                                                need, stream))
        return y if x.dtype == torch.float32 else y.to(x.dtype)

    def _forward_chunks(self, lib, xin, y, device):
        """Batches of at least two chunks (``concurrent_chunks`` volumes each, default 4) run as chunks on two HIP streams:
        the deep levels of the U-Net launch fewer workgroups than the GPU has compute units, and the bandwidth-bound
        full-resolution layers of the other chunk fill them (2538 -> 2625 volumes/s at batch 8, tools/two_stream.py).  Every
        stream has its own workspace; the packed weights are shared and read-only.  Results do not depend on the split
        (volumes are independent)."""
        n, _, d, h, w = xin.shape
        c = self.concurrent_chunks
        if self._side_streams is None or self._side_streams[0].device != device:
            self._side_streams = [torch.cuda.Stream(device) for _ in range(2)]
            self._side_ws = [None, None]
        need = lib.amx_unet_workspace_bytes(self._handle, c, d, h, w)
        cur = torch.cuda.current_stream(device)
        ready = cur.record_event()
        for k, i0 in enumerate(range(0, n, c)):
            i1 = min(i0 + c, n)
            s = self._side_streams[k & 1]
            if k < 2:
                s.wait_event(ready)
            ws = self._side_ws[k & 1]
            if ws is None or ws.numel() < need:
                ws = self._side_ws[k & 1] = torch.empty(need, dtype=torch.uint8, device=device)
            _lib.check(lib.amx_unet_forward(self._handle, _lib.ptr(xin[i0:]), _lib.ptr(y[i0:]), i1 - i0, d, h, w, _lib.ptr(ws), need,
                                            ctypes.c_void_p(s.cuda_stream)))
        for s in self._side_streams:
            cur.wait_stream(s)

    def check_numerics(self, synchronize=True):
        """Raises ``_lib.AmxOverflowError`` if a forward of this module (f16 / f16x2 storage) produced values outside the f16
        range since the last check; the output of that forward was overwritten with NaN on the device.  With
        ``synchronize`` the current stream is drained first, so every forward enqueued so far is covered.  The same error is
        raised by the next forward if nobody asked in between."""
        if self._handle is None:
            return
        dev = torch.device("cuda", self._handle_key[0]) if self._handle_key[0] is not None else torch.device("cuda")
        with torch.cuda.device(dev):
            stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            _lib.check(_lib.load().amx_unet_numerics_status(self._handle, int(bool(synchronize)), stream))

    def forward_hip_taps(self, x, layers, encode_only=False):
        """Unet.forward(input, layers, encode_only) (network.py:475-529) on the HIP kernels.  Features are fp32 NCDHW
        tensors collected in traversal order; with ``encode_only`` the forward stops after module ``layers[-1]``."""
        device = x.device
        lib = self._ensure_handle(device)
        nmod = len(self.model)
        stop = layers[-1] if (encode_only and 0 <= layers[-1] < nmod) else -1
        want = sorted({int(l) for l in layers if 0 <= int(l) < nmod and (stop < 0 or int(l) <= stop)})
        with torch.cuda.device(device):
            if self._weights_stale():
                self._upload_weights(lib, device)
            xin = x.detach()
            if xin.dtype != torch.float32 or not xin.is_contiguous():
                xin = xin.float().contiguous()
            n, _, d, h, w = xin.shape
            ws, need = self._get_workspace(lib, n, d, h, w, device)
            y = torch.empty((n, self._cfg["output_nc"], d, h, w), dtype=torch.float32, device=device)
            feats = []
            ch, lv = ctypes.c_int(), ctypes.c_int()
            for m in want:
                _lib.check(lib.amx_unet_module_info(self._handle, m, ctypes.byref(ch), ctypes.byref(lv)))
                feats.append(torch.empty((n, ch.value, d >> lv.value, h >> lv.value, w >> lv.value), dtype=torch.float32,
                                         device=device))
            mods = (ctypes.c_int * max(len(want), 1))(*want)
            outs = (ctypes.c_void_p * max(len(want), 1))(*[f.data_ptr() for f in feats])
            stream = ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)
            _lib.check(lib.amx_unet_forward_taps(self._handle, _lib.ptr(xin), _lib.ptr(y), n, d, h, w, _lib.ptr(ws), need,
                                                 mods, len(want), outs, stop, stream))
        if x.dtype != torch.float32:
            y, feats = y.to(x.dtype), [f.to(x.dtype) for f in feats]
        return feats if stop >= 0 else (y, feats)

    def profile_forward(self, x):
        """One forward with a hipEvent around every launch (amx_unet_forward_profiled).  Returns
        (output, [dict(kernel, module_idx, cin, cout, n, d, h, w, ms, flops, bytes), ...])."""
        reason = self.hip_unsupported_reason(x)
        if reason is not None:
            raise RuntimeError("profile_forward needs the HIP path: " + reason)
        device = x.device
        lib = self._ensure_handle(device)
        with torch.cuda.device(device):
            if self._weights_stale():
                self._upload_weights(lib, device)
            xin = x.detach().float().contiguous()
            n, _, d, h, w = xin.shape
            ws, need = self._get_workspace(lib, n, d, h, w, device)
            y = torch.empty((n, self._cfg["output_nc"], d, h, w), dtype=torch.float32, device=device)
            stream = ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)
            recs = (_lib.LaunchRecord * 256)()
            cnt = ctypes.c_int(0)
            _lib.check(lib.amx_unet_forward_profiled(self._handle, _lib.ptr(xin), _lib.ptr(y), n, d, h, w,
                                                     _lib.ptr(ws), need, stream, recs, 256, ctypes.byref(cnt)))
        out = []
        for r in recs[:cnt.value]:
            out.append(dict(kernel=r.kernel.decode(), module_idx=r.module_idx, cin=r.cin, cout=r.cout, n=r.n,
                            d=r.d, h=r.h, w=r.w, ms=r.ms, flops=r.flops, bytes=r.bytes))
        return y, out

    # ------------------------------------------------------------------------------------------
    # stock-module path (explicit opt-in): same traversal as network.py:467-548
    # ------------------------------------------------------------------------------------------
    def _forward_torch(self, x, layers, encode_only, verbose):
        feat, feats, pending_skips = x, [], []
        saved = None
        want = len(layers) > 0
        for idx, layer in enumerate(self.model):
            feat = layer(feat)
            if verbose and want:
                print(idx, layer.__class__.__name__, feat.size())
            if self.residual_connection and idx in self.res_source:
                saved = feat
            if self.residual_connection and idx in self.res_dest:
                assert saved.size() == feat.size()
                feat = feat + 0.1 * saved
            if self.use_skip_connection:
                if idx in self.decoder_idx:
                    feat = torch.cat((pending_skips.pop(), feat), dim=1)
                if idx in self.encoder_idx:
                    pending_skips.append(feat)
            if want:
                if idx in layers:
                    feats.append(feat)
                if encode_only and idx == layers[-1]:
                    return feats
        return (feat, feats) if want else feat

    @torch.compiler.disable      # the reference wraps the module in torch.compile (README.md:64, train.py:223): the HIP
    def forward(self, input, layers=[], encode_only=False, verbose=False):   # path is opaque to Dynamo -> clean graph break
        """Same call contract as network.py:467: tensor without ``layers``; ``(out, feats)`` with
        ``layers``; ``feats`` alone with ``encode_only``."""
        train_reason = None
        wants_grad = torch.is_grad_enabled() and (input.requires_grad or any(p.requires_grad for p in self.parameters()))
        batch_stats = self._cfg["norm"] == "batch" and (self.training or wants_grad)   # eval + autograd: frozen statistics
        instance = self._cfg["norm"] in ("instance", "instance_affine") and wants_grad
        if input.is_cuda and (batch_stats or instance) and not encode_only:
            # train-mode BatchNorm (batch statistics) and/or autograd: the differentiable HIP path (model/train.py)
            from . import train
            train_reason = train.unsupported_reason(self, input, list(layers))
            if train_reason is None:
                return train.forward_train(self, input, list(layers))
        reason = self.hip_unsupported_reason(input, layers)
        if reason is not None and train_reason is not None:
            reason = train_reason
        if reason is None:
            if len(layers) > 0:
                if verbose:
                    print("anatomix_amd.Unet: verbose per-layer shapes are only printed on the stock torch path")
                return self.forward_hip_taps(input, list(layers), encode_only)
            return self.forward_hip(input)
        if not self.allow_torch_path:
            raise RuntimeError(
                "anatomix_amd.Unet: this call cannot run on the HIP kernels (" + reason + "). Set "
                "model.allow_torch_path = True (or AMX_ALLOW_TORCH_PATH=1) to run it on the stock torch modules.")
        if not self._warned:
            warnings.warn("anatomix_amd.Unet: running on stock torch modules: " + reason)
            self._warned = True
        return self._forward_torch(input, layers, encode_only, verbose)
