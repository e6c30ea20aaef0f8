"""Tensor-level wrappers over the C ABI (include/ganslate_hip.h). Every call enqueues on torch's current
stream; tensors are device tensors owned by the caller. No fallback: a missing library raises in lib.load()."""
import ctypes as C
import os

import torch

import functools

from . import lib as L
from ..nn.native.spec import GConv, WGrad
from ..nn.native.twin import Twin, TwinSplit, is_twin


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream():
    """torch's current HIP stream as a raw handle. torch.cuda.current_stream() costs ~8 us of Python per call, which at
    ~1100 launches per step was half of the host's enqueue time; the C accessor is ~0.3 us."""
    if _raw_stream is not None:
        return C.c_void_p(_raw_stream(torch.cuda.current_device()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class HipOps(TwinSplit):
    """The product backend: hand-written gfx950 kernels behind libganslate_hip.so."""
    name = "hip"
    act_dtype = torch.bfloat16   # storage type of activations and weight packs

    def __init__(self, device=None):
        self.lib = L.load()
        if not torch.cuda.is_available():
            raise RuntimeError("HipOps needs an MI355X (torch.cuda.is_available() is False); there is no CPU path")
        self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        L.check(self.lib.gs_init(self.device.index or 0), "gs_init")
        self._desc_cache = {}
        self._timing_filter, self._timing_events, self._timing_images = None, {}, {}
        self.sync_options()

    # ---- kernel-selection switches ------------------------------------------------------------------------
    # The library reads no environment variable (gs_set_option, include/ganslate_hip.h); the GS_* variables of the
    # host side are mapped onto its options here, when the backend is created and whenever a model is built.
    ENV_OPTIONS = {"GS_SPLITK": "splitk", "GS_SPLITK_MAXB": "splitk_max_blocks", "GS_SPLITK_TARGET": "splitk_target",
                   "GS_HCONV": "hconv", "GS_HCONV_WIDE": "hconv_wide", "GS_HCONVW_PERSIST": "hconvw_persist", "GS_HSTRIP_REGS": "hstrip_regs", "GS_GCONV_TWIN": "gconv_twin", "GS_GCONV_SMALLK": "gconv_smallk", "GS_GCONV_PERSIST": "gconv_persist", "GS_HCONVT_PERSIST": "hconvt_persist", "GS_RING_APPLY": "ring_apply", "GS_NORM_XCD": "norm_xcd", "GS_WGRAD_ROWS": "wgrad_rows", "GS_SPLITK_MULTI": "splitk_multi", "GS_SPLITK_RING": "splitk_ring", "GS_GCONV_RING4": "gconv_ring4", "GS_WGRAD_TWIN": "wgrad_twin",
                   "GS_HWGRAD": "hwgrad", "GS_HWGRAD_WIDE": "hwgrad_wide", "GS_HWGRAD_PLANES": "hwgrad_planes",
                   "GS_BWD_PPB": "norm_bwd_ppb", "GS_APPLY_U": "norm_apply_unroll", "GS_GCONV_TILE288": "gconv_tile288", "GS_GCONV_MULTI": "gconv_multi",
                   "GS_HCONVW_RING": "hconvw_ring", "GS_HCONVT": "hconvt", "GS_HSTRIP": "hstrip",
                   "GS_WFOLD_ROWS": "wfold_rows", "GS_HWGRAD_FT": "hwgrad_ft", "GS_GCONV_BIG": "gconv_big", "GS_HCONV_BOX8": "hconv_box8",
                   "GS_HCONV5": "hconv5", "GS_RING_DBG": "ring_dbg", "GS_HWGRAD2": "hwgrad2", "GS_HCONV2": "hconv2", "GS_PWISE": "pwise"}

    def set_option(self, name, value):
        L.check(self.lib.gs_set_option(name.encode(), int(value)), "gs_set_option")
        self._desc_cache = {k: v for k, v in self._desc_cache.items()
                            if not (isinstance(k, tuple) and k[0] in ("splitk", "wgrad_ws", "multi"))}

    def get_option(self, name):
        v = C.c_int(0)
        L.check(self.lib.gs_get_option(name.encode(), C.byref(v)), "gs_get_option")
        return v.value

    def sync_options(self):
        if not hasattr(self, "_option_defaults"):
            self._option_defaults = {}
            for opt in self.ENV_OPTIONS.values():      # (an older build of the library — GANSLATE_HIP_LIB in an A/B — may not know
                try:                                   # the newest switches: those are skipped)
                    self._option_defaults[opt] = self.get_option(opt)
                except L.HipError:
                    pass
        for env, opt in self.ENV_OPTIONS.items():
            if opt not in self._option_defaults:
                continue
            want = int(os.environ[env]) if env in os.environ else self._option_defaults[opt]
            if want != self.get_option(opt):
                self.set_option(opt, want)

    # ---- per-kernel timing with HIP events on the launch stream (used by bench.py) ------------------
    def enable_kernel_timing(self, select):
        """select(kind, spec, flag) -> label or None. kind "gconv": spec = GConv class, flag = fused norm-backward
        epilogue; kind "wgrad": spec = WGrad, flag = merged pair launch. Launches with a label are bracketed by HIP
        events on the stream they are launched on."""
        self._timing_filter, self._timing_events, self._timing_images = select, {}, {}
        return self._timing_events

    def disable_kernel_timing(self):
        self._timing_filter = None

    def _time_begin(self, kind, spec, flag, N=None):
        if self._timing_filter is None:
            return None
        label = self._timing_filter(kind, spec, flag)
        if label is None:
            return None
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        self._timing_events.setdefault(label, []).append((e0, e1))
        self._timing_images.setdefault(label, []).append(N)       # images per launch (a twin launch covers both networks)
        return e1

    def kernel_timing_result(self):
        """{label: (launches, average milliseconds)}"""
        torch.cuda.synchronize()
        return {label: (len(ev), sum(a.elapsed_time(b) for a, b in ev) / max(len(ev), 1))
                for label, ev in self._timing_events.items()}

    def kernel_timing_images(self):
        """{label: average number of images per timed launch}"""
        return {label: sum(v for v in ns if v) / max(sum(1 for v in ns if v), 1) for label, ns in self._timing_images.items()}

    # ---- descriptors ------------------------------------------------------------------------------------
    def _gdesc(self, g: GConv, N, in_cs, in_co, out_cs, out_co, act, slope, stats_slots, stats_slot0, accumulate=False):
        key = (id(g), N, in_cs, in_co, out_cs, out_co, act, slope, stats_slots, stats_slot0, accumulate)
        d = self._desc_cache.get(key)
        if d is None:
            d = L.GConvDesc()
            d.N, d.Hi, d.Wi, d.Ci, d.in_cs, d.in_co = N, g.Hi, g.Wi, g.Ci, in_cs, in_co
            d.Di, d.Do, d.Dc, d.pz = g.Di, g.Do, g.Dc, g.pz
            d.Ho, d.Wo, d.Co, d.out_cs, d.out_co = g.Ho, g.Wo, g.Co, out_cs, out_co
            d.Hc, d.Wc, d.so, d.py, d.px, d.si = g.Hc, g.Wc, g.so, g.py, g.px, g.si
            d.T, d.Kp, d.w_rows = g.T, g.Kp, g.w_rows
            d.border, d.act, d.slope = L.BORDER[g.border], L.ACT[act], slope
            d.stats_slots, d.stats_slot0, d.accumulate = stats_slots, stats_slot0, int(accumulate)
            for i, (a, b, c) in enumerate(zip(g.dh, g.dw, g.dd)):
                d.dh[i], d.dw[i], d.dd[i] = a, b, c
            self._desc_cache[key] = (d, g)   # keep g alive so id() stays unique
            return d
        return d[0]

    def _splitk_floats(self, d) -> int:
        """workspace floats the split-K form of this launch wants (0: no split); cached per descriptor"""
        key = ("splitk", id(d))
        n = self._desc_cache.get(key)
        if n is None:
            n = int(self.lib.gs_gconv_splitk_ws_floats(C.byref(d)))
            self._desc_cache[key] = n
        return n

    def _cout1(self, d) -> bool:
        """the dot-product kernels of the one-output-channel layer take this launch (GS_COUT1=0: A/B switch)"""
        if os.environ.get("GS_COUT1", "1") == "0":
            return False
        key = ("cout1", id(d))
        v = self._desc_cache.get(key)
        if v is None:
            v = bool(self.lib.gs_conv_cout1_eligible(C.byref(d)))
            self._desc_cache[key] = v
        return v

    def tile_m(self, g: GConv, N: int) -> int:
        """pixel-tile height the kernel will pick for this class at batch N (the choice depends on the grid size)"""
        return self.lib.gs_tile_m(C.byref(self._gdesc(g, N, g.Ci, 0, g.Co, 0, "none", 0.0, 0, 0)))

    # ---- twin batches (nn/native/twin.py): which batch a launch sees --------------------------------------------
    # Kernel choice — and with it the number of statistics / partial-sum slots per image — depends on the batch of the
    # LAUNCH. A twin batch of N images is one launch of N where the kernel picks the weight set per image
    # (gs_gconv_twin_native) and two launches of N / 2 otherwise; the planning calls below take `twin` and answer for the
    # launches that will actually run, while their buffers are sized for all N images.
    def twin_native(self, g: GConv, N: int, ring: bool = False, fused: bool = False) -> bool:
        """ring: the fused data gradient on the unpadded domain (hconvw RING); fused: the padded-domain fused launch"""
        if os.environ.get("GS_TWIN_NATIVE", "1") == "0":
            return False
        d = self._gdesc(g, N, g.Ci, 0, g.Co, 0, "none", 0.0, 0, 0)
        if ring:
            return self.lib.gs_gconv_ring_slots(C.byref(d)) > 0
        if not fused and getattr(g, "co_real", 0) == 1 and self._cout1(d):
            return True
        if fused:
            if os.environ.get("GS_TWIN_FUSED", "1") == "0":
                return False
            f = L.GConvFuse()           # (fold 0 on a padded output domain: not the ring form)
            return bool(self.lib.gs_gconv_twin_native(C.byref(d), C.byref(f)))
        return bool(self.lib.gs_gconv_twin_native(C.byref(d), None))

    def fused_norm_plan(self, g: GConv, N: int, C_: int, force: bool = False, twin: bool = False):
        """(slots, scratch) for fusing the reduction pass of the consumer's InstanceNorm backward into the data-gradient
        launch of class g, or None when this backend / layer shape does not fuse (narrow layers run on the halo kernel)"""
        if g.so != 1 or g.si not in (1, 2) or g.Co <= 64 or g.Co != C_ or os.environ.get("GS_FUSE_NORM", "1") == "0":
            return None
        if g.si == 2 and os.environ.get("GS_FUSE_SI2", "1") == "0":     # (A/B switch: data gradients of transposed convs)
            return None
        Nl = N // 2 if (twin and not self.twin_native(g, N, fused=True)) else N        # the batch of the launch(es)
        d = self._gdesc(g, Nl, g.Ci, 0, g.Co, 0, "none", 0.0, 0, 0)
        if self._splitk_floats(d) and not force:   # few output tiles, long K: split-K wins over the fused epilogue
            return None
        tm = self.tile_m(g, Nl)
        slots = (g.pixels + tm - 1) // tm
        return slots, torch.empty(N * (slots + 1) * 3 * C_, dtype=torch.float32, device=self.device)

    @staticmethod
    def _fuse_struct(fuse):
        f = L.GConvFuse()
        f.y, f.mean_rstd, f.partial = fuse["y"].data_ptr(), fuse["mean_rstd"].data_ptr(), fuse["partial"].data_ptr()
        f.g2 = fuse["g2"].data_ptr() if fuse.get("g2") is not None else None
        yd = fuse["y"].shape[1:-1]
        f.Dy, f.Hy, f.Wy = (1,) + tuple(yd) if len(yd) == 2 else tuple(yd)
        f.fold, f.fold_mode, f.act, f.slope = fuse["fold"], L.BORDER[fuse["fold_mode"]], L.ACT[fuse["act"]], \
            float(fuse.get("slope", 0.2))
        return f

    def _multi_descs(self, classes, N, in_cs, out_cs):
        key = ("multi_fused", tuple(id(g) for g in classes), N, in_cs, out_cs)
        ent = self._desc_cache.get(key)
        if ent is None:
            descs = [self._gdesc(g, N, in_cs, 0, out_cs, 0, "none", 0.0, 0, 0) for g in classes]
            arr = (C.POINTER(L.GConvDesc) * len(descs))(*[C.pointer(d) for d in descs])
            ent = (arr, descs)
            self._desc_cache[key] = ent
        return ent

    def fused_multi_plan(self, classes, N: int, C_: int, twin: bool = False):
        """(slots, scratch) when the output-parity classes of a stride-2 conv's data gradient run as ONE halo-resident launch
        that can carry the reduction pass of the consumer's InstanceNorm backward in its epilogue (hconvt.hip), else None"""
        # (GS_FUSE_MULTI=0: A/B switch. Worth 0.3 % once the fused instantiation stopped spilling)
        g = classes[0]
        if len(classes) != 4 or g.Co != C_ or os.environ.get("GS_FUSE_NORM", "1") == "0" or \
                os.environ.get("GS_FUSE_MULTI", "1") == "0":
            return None
        Nl = N // 2 if (twin and not self.multi_twin_native(classes, N)) else N      # the batch of the launch(es)
        arr, _ = self._multi_descs(classes, Nl, g.Ci, g.Co)
        slots = self.lib.gs_gconv_multi_fused_slots(arr, len(classes))
        if slots <= 0:
            return None
        return slots, torch.empty(N * (slots + 1) * 3 * C_, dtype=torch.float32, device=self.device)

    def fused_ring_plan(self, g: GConv, N: int, C_: int, twin: bool = False):
        """(slots, scratch) when the fused data gradient of a reflect-padded 3x3 layer can run on the unpadded domain
        (class g = Lowered.dgrad_ring; the launch folds the ring itself, hconvw.hip RING), else None"""
        if g is None or g.Co != C_ or os.environ.get("GS_FUSE_NORM", "1") == "0":
            return None
        Nl = N // 2 if (twin and not self.twin_native(g, N, ring=True)) else N
        slots = self.lib.gs_gconv_ring_slots(C.byref(self._gdesc(g, Nl, g.Ci, 0, g.Co, 0, "none", 0.0, 0, 0)))
        if slots <= 0:
            return None
        return slots, torch.empty(N * (slots + 1) * 3 * C_, dtype=torch.float32, device=self.device)

    def ring_apply_plan(self, g: GConv, N: int, C_: int, twin: bool = False):
        """the rendezvous buffer (zero int32 words, one per launching stream, left zero by every launch) when the ring-form
        launch of class g can also carry the consumer's whole InstanceNorm backward (gs_gconv_ring_apply: dy and the total
        gradient come out of the data-gradient launch, no gs_inorm_act_backward behind it), else None. Twin batches only
        where the launch is one launch. GS_RING_APPLY=0 switches it off."""
        if g is None or g.Co != C_:
            return None
        if twin and not self.twin_native(g, N, ring=True):
            return None
        words = self.lib.gs_gconv_ring_apply_words(C.byref(self._gdesc(g, N, g.Ci, 0, g.Co, 0, "none", 0.0, 0, 0)))
        if words <= 0:
            return None
        if not hasattr(self, "_ring_sync"):
            self._ring_sync = {}
        key = int(_stream().value or 0)                        # one buffer per launching stream
        buf = self._ring_sync.get(key)
        if buf is None or buf.numel() < words:
            buf = torch.zeros(max(words, 4096), dtype=torch.int32, device=self.device)
            self._ring_sync[key] = buf
        return buf

    def gconv_ring_apply(self, g: GConv, x, wpack, dy, total, fuse, sync):
        """the ring-form fused data gradient with the norm backward applied in the launch: writes dy (and total = gx + g2)"""
        N = x.shape[0]
        d = self._gdesc(g, N, x.shape[-1], 0, dy.shape[-1], 0, "none", 0.0, 0, 0, False)
        f = self._fuse_struct(fuse)
        tw = None
        if isinstance(wpack, Twin):
            tw = L.Twin()
            tw.n_split, tw.w_delta, tw.bias_delta = N // 2, wpack.delta(), 0
            w = C.c_void_p(wpack.a.data_ptr() + 2 * g.pack_offset)
        else:
            w = C.c_void_p(wpack.data_ptr() + 2 * g.pack_offset)
        t_end = self._time_begin("gconv", g, True, N)
        L.check(self.lib.gs_gconv_ring_apply(C.byref(d), _ptr(x), w, C.byref(f), _ptr(dy), _ptr(total), _ptr(sync),
                                             C.byref(tw) if tw is not None else None, _stream()), "gs_gconv_ring_apply")
        if t_end is not None:
            t_end.record()

    def multi_twin_native(self, classes, N: int) -> bool:
        """a twin batch of N images over the output-parity classes of one layer runs as ONE launch (the halo-resident class
        kernel picks the packs per box, gs_gconv_multi_twin_native) — else as two launches of N / 2"""
        if os.environ.get("GS_TWIN_NATIVE", "1") == "0" or os.environ.get("GS_TWIN_MULTI", "1") == "0" or N % 2:
            return False
        g = classes[0]
        key = ("multi_twin", tuple(id(c) for c in classes), N)
        v = self._desc_cache.get(key)
        if v is None:
            arr, _ = self._multi_descs(classes, N, g.Ci, g.Co)
            v = bool(self.lib.gs_gconv_multi_twin_native(arr, len(classes)))
            self._desc_cache[key] = v
        return v

    def stat_slots(self, g: GConv, N: int, twin: bool = False, multi=None) -> int:
        """partial-statistics slots per image the kernel writes for this class at batch N. multi: the classes of the layer
        when g is one of several output-parity classes (gconv_classes): a twin batch of those runs as ONE launch where
        multi_twin_native says so and as two launches of N / 2 otherwise, whatever a single-class launch of g would do"""
        if twin and multi:
            Nl = N if self.multi_twin_native(multi, N) else N // 2
        else:
            Nl = N // 2 if (twin and not self.twin_native(g, N)) else N
        return self.lib.gs_gconv_stat_slots(C.byref(self._gdesc(g, Nl, g.Ci, 0, g.Co, 0, "none", 0.0, 0, 0)))

    # ---- convolution family -------------------------------------------------------------------------------
    def gconv(self, g: GConv, x, wpack, bias, out, *, in_cs=None, in_co=0, out_cs=None, out_co=0, act="none",
              slope=0.2, stats=None, stats_slots=0, stats_slot0=0, accumulate=False, fuse=None):
        N = x.shape[0]
        in_cs = in_cs if in_cs is not None else x.shape[-1]
        out_cs = out_cs if out_cs is not None else out.shape[-1]
        d = self._gdesc(g, N, in_cs, in_co, out_cs, out_co, act, float(slope), stats_slots, stats_slot0, accumulate)
        if getattr(g, "co_real", 0) == 1 and fuse is None and stats is None and self._cout1(d):
            # one output channel (the PatchGAN's last layer): the dot-product kernel on the vector ALUs (csrc/cout1.hip)
            tw = None
            if is_twin(wpack, bias):
                tw = L.Twin()
                tw.n_split, tw.w_delta = N // 2, wpack.delta()
                tw.bias_delta = bias.delta() if isinstance(bias, Twin) else 0
            w0 = wpack.a if isinstance(wpack, Twin) else wpack
            b0 = bias.a if isinstance(bias, Twin) else bias
            L.check(self.lib.gs_conv_cout1_forward(C.byref(d), _ptr(x), C.c_void_p(w0.data_ptr() + 2 * g.pack_offset),
                                                   _ptr(b0), _ptr(out), C.byref(tw) if tw is not None else None, _stream()),
                    "gs_conv_cout1_forward")
            return
        if is_twin(wpack, bias):      # two networks' weights over one batch (nn/native/twin.py)
            f = self._fuse_struct(fuse) if fuse is not None else None
            ring = f is not None and f.fold > 0 and tuple(fuse["y"].shape[-3:-1]) == (g.Ho, g.Wo)
            if isinstance(wpack, Twin) and not accumulate and \
                    self.twin_native(g, N, ring=ring, fused=f is not None and not ring):
                # the kernel picks the weight set per image: one launch over both networks' images
                tw = L.Twin()
                tw.n_split, tw.w_delta = N // 2, wpack.delta()
                tw.bias_delta = bias.delta() if isinstance(bias, Twin) else 0
                b0 = bias.a if isinstance(bias, Twin) else bias
                t_end = self._time_begin("gconv", g, fuse is not None, N)
                L.check(self.lib.gs_gconv_forward_twin(C.byref(d), _ptr(x), C.c_void_p(wpack.a.data_ptr() + 2 * g.pack_offset),
                                                       _ptr(b0), _ptr(out), _ptr(stats),
                                                       C.byref(f) if f is not None else None, C.byref(tw), _stream()),
                        "gs_gconv_forward_twin")
                if t_end is not None:
                    t_end.record()
                return
            return self.twin_gconv(functools.partial(self.gconv, g), x, wpack, bias, out, in_cs=in_cs, in_co=in_co,
                                   out_cs=out_cs, out_co=out_co, act=act, slope=slope, stats=stats,
                                   stats_slots=stats_slots, stats_slot0=stats_slot0, accumulate=accumulate, fuse=fuse,
                                   C_=g.Co)
        w = C.c_void_p(wpack.data_ptr() + 2 * g.pack_offset)
        t_end = self._time_begin("gconv", g, fuse is not None, N)
        if fuse is not None:     # data gradient + first pass of the consumer's InstanceNorm backward (fused_norm_plan)
            f = self._fuse_struct(fuse)
            L.check(self.lib.gs_gconv_forward_fused(C.byref(d), _ptr(x), w, _ptr(bias), _ptr(out), _ptr(stats),
                                                    C.byref(f), _stream()), "gs_gconv_forward_fused")
            if t_end is not None:
                t_end.record()
            return
        nws = self._splitk_floats(d)
        if nws:      # few output tiles, long K: split-K with a per-launch workspace (stream-safe through the allocator)
            ws = torch.empty(nws, dtype=torch.float32, device=self.device)
            L.check(self.lib.gs_gconv_forward_ws(C.byref(d), _ptr(x), w, _ptr(bias), _ptr(out), _ptr(stats), _ptr(ws),
                                                 nws, _stream()), "gs_gconv_forward_ws")
        else:
            L.check(self.lib.gs_gconv_forward(C.byref(d), _ptr(x), w, _ptr(bias), _ptr(out), _ptr(stats), _stream()),
                    "gs_gconv_forward")
        if t_end is not None:
            t_end.record()

    def gconv_classes(self, classes, x, wpack, bias, out, *, in_co=0, out_co=0, act="none", slope=0.2, stats=None,
                      stats_slots=0, stats_slot0s=None, accumulate=False, fuse=None):
        """every output-parity class of one layer (Lowered.fwd / .dgrad). More than one class: gs_gconv_forward_multi, one
        launch when the classes are mergeable (the library decides; it runs them one by one otherwise). Layers so small
        that even the merged grid leaves the chip empty keep the per-class launches, which split K."""
        if is_twin(wpack, bias) and len(classes) == 1 and fuse is None:
            return self.gconv(classes[0], x, wpack, bias, out, in_co=in_co, out_co=out_co, act=act, slope=slope, stats=stats,
                              stats_slots=stats_slots, stats_slot0=(stats_slot0s[0] if stats_slot0s else 0),
                              accumulate=accumulate)
        if is_twin(wpack, bias):
            N2 = x.shape[0]
            if isinstance(wpack, Twin) and not accumulate and in_co == 0 and out_co == 0 and \
                    self.multi_twin_native(classes, N2):
                # the class kernel picks the packs per box: one launch over both networks' images
                descs = [self._gdesc(g, N2, x.shape[-1], 0, out.shape[-1], 0, act, float(slope), stats_slots,
                                     (stats_slot0s[i] if stats_slot0s else 0)) for i, g in enumerate(classes)]
                arr = (C.POINTER(L.GConvDesc) * len(descs))(*[C.pointer(d) for d in descs])
                ws = (C.c_void_p * len(classes))(*[wpack.a.data_ptr() + 2 * g.pack_offset for g in classes])
                tw = L.Twin()
                tw.n_split, tw.w_delta = N2 // 2, wpack.delta()
                tw.bias_delta = bias.delta() if isinstance(bias, Twin) else 0
                b0 = bias.a if isinstance(bias, Twin) else bias
                f = self._fuse_struct(fuse) if fuse is not None else None
                t_end = self._time_begin("gconv_multi", classes, f is not None, N2)
                L.check(self.lib.gs_gconv_forward_multi_twin(arr, len(classes), _ptr(x), ws, _ptr(None if f is not None else b0),
                                                             _ptr(out), _ptr(None if f is not None else stats),
                                                             C.byref(f) if f is not None else None, C.byref(tw), _stream()),
                        "gs_gconv_forward_multi_twin")
                if t_end is not None:
                    t_end.record()
                return
            return self.twin_gconv(functools.partial(self.gconv_classes, classes), x, wpack, bias, out, in_co=in_co,
                                   out_co=out_co, act=act, slope=slope, stats=stats, stats_slots=stats_slots,
                                   stats_slot0s=stats_slot0s, accumulate=accumulate, fuse=fuse, C_=classes[0].Co)
        N = x.shape[0]
        if fuse is not None:     # fused_multi_plan said yes: all classes + the consumer's norm-backward sums in one launch
            arr, _ = self._multi_descs(classes, N, x.shape[-1], out.shape[-1])
            base = wpack.data_ptr()
            ws = (C.c_void_p * len(classes))(*[base + 2 * g.pack_offset for g in classes])
            t_end = self._time_begin("gconv_multi", classes, True, N)
            L.check(self.lib.gs_gconv_forward_multi_fused(arr, len(classes), _ptr(x), ws, _ptr(out),
                                                          C.byref(self._fuse_struct(fuse)), _stream()),
                    "gs_gconv_forward_multi_fused")
            if t_end is not None:
                t_end.record()
            return
        slot0 = lambda i: stats_slot0s[i] if stats_slot0s else 0
        g0 = classes[0]
        merged = len(classes) > 1 and not accumulate
        if merged:
            key = ("multi", tuple(id(g) for g in classes), N, x.shape[-1], in_co, out.shape[-1], out_co, act, float(slope),
                   stats_slots, tuple(stats_slot0s or ()))
            ent = self._desc_cache.get(key)
            if ent is None:
                descs = [self._gdesc(g, N, x.shape[-1], in_co, out.shape[-1], out_co, act, float(slope), stats_slots,
                                     slot0(i)) for i, g in enumerate(classes)]
                tn = 16 if g0.Co <= 16 else (64 if g0.Co <= 64 else 128)
                tm = self.tile_m(g0, N)
                blocks = len(classes) * N * ((g0.pixels + tm - 1) // tm) * ((g0.Co + tn - 1) // tn)
                use = blocks >= 128 or not any(self._splitk_floats(d) for d in descs)
                arr = (C.POINTER(L.GConvDesc) * len(descs))(*[C.pointer(d) for d in descs])
                # a merged grid that still leaves the chip empty: split K over it (one launch + one finalize for all classes)
                nws = 0 if use else int(self.lib.gs_gconv_multi_splitk_ws_floats(arr, len(descs)))
                ent = (arr, descs, use or nws > 0, nws)
                self._desc_cache[key] = ent
            merged = ent[2]
        if not merged:
            for i, g in enumerate(classes):
                self.gconv(g, x, wpack, bias, out, in_co=in_co, out_co=out_co, act=act, slope=slope, stats=stats,
                           stats_slots=stats_slots, stats_slot0=slot0(i), accumulate=accumulate)
            return
        base = wpack.data_ptr()
        ws = (C.c_void_p * len(classes))(*[base + 2 * g.pack_offset for g in classes])
        t_end = self._time_begin("gconv_multi", classes, False, N)
        if ent[3]:
            part = torch.empty(ent[3], dtype=torch.float32, device=self.device)
            L.check(self.lib.gs_gconv_forward_multi_ws(ent[0], len(classes), _ptr(x), ws, _ptr(bias), _ptr(out), _ptr(stats),
                                                       _ptr(part), ent[3], _stream()), "gs_gconv_forward_multi_ws")
        else:
            L.check(self.lib.gs_gconv_forward_multi(ent[0], len(classes), _ptr(x), ws, _ptr(bias), _ptr(out), _ptr(stats),
                                                    _stream()), "gs_gconv_forward_multi")
        if t_end is not None:
            t_end.record()

    @staticmethod
    def can_merge_wgrad(w: WGrad) -> bool:
        """two backward passes of this layer can share one launch (the wide halo kernel's eligibility, hwgrad.hip)"""
        if w.si != 1 or w.P % 64 or w.Q % 64 or os.environ.get("GS_WGRAD_PAIR", "1") == "0":
            return False
        if w.T == 9 and w.Da == 1:
            return True
        # 3x3x3 layers of volumes run as three depth planes of the same kernel (hwgrad.hip)
        return (w.T == 27 and w.Da > 1 and os.environ.get("GS_HWGRAD_PLANES", "1") != "0"
                and all(w.dd[9 * k + t] == w.dd[9 * k] and w.dh[9 * k + t] == w.dh[t] and w.dw[9 * k + t] == w.dw[t]
                        for k in range(3) for t in range(9)))

    def _wdesc(self, w: WGrad, N, a_cs, a_co, g_cs, g_co, fresh=False):
        d = L.WGradDesc()
        d.dw_fresh = int(fresh)
        d.N, d.Ha, d.Wa, d.P = N, w.Ha, w.Wa, w.P
        d.Da, d.Dg = w.Da, w.Dg
        d.a_cs, d.a_co = a_cs, a_co
        d.Hg, d.Wg, d.Q = w.Hg, w.Wg, w.Q
        d.g_cs, d.g_co = g_cs, g_co
        d.si, d.T, d.border, d.dw_ld = w.si, w.T, L.BORDER[w.border], w.T * w.Q
        for i, (p, q, r) in enumerate(zip(w.dh, w.dw, w.dd)):
            d.dh[i], d.dw_[i], d.dd[i] = p, q, r
        return d

    def wgrad_adam(self, w: WGrad, a, g, p, m, v, hyper_dev, packs=None, tr=None) -> bool:
        """weight gradient + Adam of ONE layer in one launch (gs_wgrad_adam): p / m / v are the layer's slices of the flat
        buffers, packs = (inv_f slice, fpack, inv_d slice, dpack) as for adam_step_dev; tr = (base int32[T], kp int32[T], pack):
        the layer's transposed pack, written by the same launch (gs_adam_fuse.tr_*). Returns False (nothing launched) where
        the layer does not run as a one-split im2col launch — the caller then runs wgrad() and the optimiser as usual."""
        key = ("wadam", id(w), a.shape[0], a.shape[-1], g.shape[-1])
        ent = self._desc_cache.get(key)
        if ent is None:
            d = self._wdesc(w, a.shape[0], a.shape[-1], 0, g.shape[-1], 0)
            ok = bool(self.lib.gs_wgrad_adam_eligible(C.byref(d))) and getattr(w, "p_real", 0) != 1
            ent = (d, ok, w)
            self._desc_cache[key] = ent
        if not ent[1]:
            return False
        ad = L.AdamFuse()
        ad.p, ad.m, ad.v, ad.hyper = p.data_ptr(), m.data_ptr(), v.data_ptr(), hyper_dev.data_ptr()
        if packs is not None:
            inv_f, fpack, inv_d, dpack = packs
            if inv_f is not None:
                ad.inv_f, ad.fpack = inv_f.data_ptr(), fpack.data_ptr()
            if inv_d is not None:
                ad.inv_d, ad.dpack = inv_d.data_ptr(), dpack.data_ptr()
        if tr is not None:
            ad.tr_base, ad.tr_kp, ad.tr_pack = tr[0].data_ptr(), tr[1].data_ptr(), tr[2].data_ptr()
        t_end = self._time_begin("wgrad", w, False, a.shape[0])
        L.check(self.lib.gs_wgrad_adam(C.byref(ent[0]), _ptr(a), _ptr(g), C.byref(ad), _stream()), "gs_wgrad_adam")
        if t_end is not None:
            t_end.record()
        return True

    def wgrad(self, w: WGrad, a, g, dw, *, a_cs=None, a_co=0, g_cs=None, g_co=0, pair=None, fresh=False):
        """dw += the weight gradient. fresh: the caller guarantees that dw holds zeros (the layer's first weight gradient since
        the optimiser cleared the buffer, NativeNet.wgrad_fresh) — a hint (gs_wgrad_desc.dw_fresh), never a requirement"""
        twin = is_twin(dw)
        fresh = bool(fresh) and not twin and pair is None and os.environ.get("GS_WGRAD_FRESH", "1") != "0"     # (A/B switch)
        if twin and (os.environ.get("GS_WGRAD_DET", "1") == "0" or os.environ.get("GS_TWIN_NATIVE", "1") == "0"
                     or os.environ.get("GS_TWIN_WGRAD", "1") == "0"):       # (A/B switch)
            return self.twin_wgrad(w, a, g, dw, a_cs=a_cs, a_co=a_co, g_cs=g_cs, g_co=g_co, pair=pair)
        key = ("w", id(w), a.shape[0], a_cs, a_co, g_cs, g_co, fresh)
        ent = self._desc_cache.get(key)
        if ent is None:
            d = L.WGradDesc()
            d.dw_fresh = int(fresh)
            d.N, d.Ha, d.Wa, d.P = a.shape[0], w.Ha, w.Wa, w.P
            d.Da, d.Dg = w.Da, w.Dg
            d.a_cs, d.a_co = (a_cs if a_cs is not None else a.shape[-1]), a_co
            d.Hg, d.Wg, d.Q = w.Hg, w.Wg, w.Q
            d.g_cs, d.g_co = (g_cs if g_cs is not None else g.shape[-1]), g_co
            d.si, d.T, d.border, d.dw_ld = w.si, w.T, L.BORDER[w.border], w.T * w.Q
            for i, (p, q, r) in enumerate(zip(w.dh, w.dw, w.dd)):
                d.dh[i], d.dw_[i], d.dd[i] = p, q, r
            ent = (d, w)
            self._desc_cache[key] = ent
        if getattr(w, "p_real", 0) == 1 and os.environ.get("GS_COUT1", "1") != "0" and \
                os.environ.get("GS_WGRAD_DET", "1") != "0":
            ckey = ("cout1_w", id(ent[0]))
            nws = self._desc_cache.get(ckey)
            if nws is None:
                nws = int(self.lib.gs_wgrad_cout1_ws_floats(C.byref(ent[0])))
                self._desc_cache[ckey] = nws
            if nws > 0 and (not twin or dw.delta() % 16 == 0):
                # one output channel: x[q] times the 4 x 4 patch of dy around q on the vector ALUs (csrc/cout1.hip)
                tw = None
                if twin:
                    tw = L.Twin()
                    tw.n_split, tw.dw_delta = a.shape[0] // 2, dw.delta()
                dw0 = dw.a if twin else dw
                for aa, gg in ((a, g),) + ((tuple(pair),) if pair is not None else ()):
                    ws = torch.empty(nws, dtype=torch.float32, device=self.device)
                    L.check(self.lib.gs_wgrad_cout1_ws(C.byref(ent[0]), _ptr(aa), _ptr(gg), _ptr(dw0), _ptr(ws), nws,
                                                       C.byref(tw) if tw is not None else None, _stream()), "gs_wgrad_cout1_ws")
                return
        if twin:      # both networks' images in one launch where the layer's kernel has the form, else the two halves
            wkey = ("wgrad_ws", id(ent[0]), pair is not None, "twin")
            nws = self._desc_cache.get(wkey)
            if nws is None:
                nws = int(self.lib.gs_wgrad_ws_floats_twin(C.byref(ent[0]), int(pair is not None))) \
                    if self.lib.gs_wgrad_twin_native(C.byref(ent[0]), int(pair is not None)) else -1
                self._desc_cache[wkey] = nws
            if nws <= 0:
                return self.twin_wgrad(w, a, g, dw, a_cs=a_cs, a_co=a_co, g_cs=g_cs, g_co=g_co, pair=pair)
            tw = L.Twin()
            tw.n_split, tw.dw_delta = a.shape[0] // 2, dw.delta()
            ws = torch.empty(nws, dtype=torch.float32, device=self.device)
            a2, g2 = pair if pair is not None else (None, None)
            t_end = self._time_begin("wgrad", w, pair is not None, a.shape[0])
            L.check(self.lib.gs_wgrad_ws_twin(C.byref(ent[0]), _ptr(a), _ptr(g), _ptr(a2), _ptr(g2), _ptr(dw.a), _ptr(ws), nws,
                                              C.byref(tw), _stream()), "gs_wgrad_ws_twin")
            if t_end is not None:
                t_end.record()
            return
        t_end = self._time_begin("wgrad", w, pair is not None, a.shape[0])
        if os.environ.get("GS_WGRAD_DET", "1") != "0":
            # deterministic accumulation (default): partial sums to a per-launch workspace, fixed-order second stage
            wkey = ("wgrad_ws", id(ent[0]), pair is not None)        # (set_option drops these plans)
            nws = self._desc_cache.get(wkey)
            if nws is None:
                nws = int(self.lib.gs_wgrad_ws_floats(C.byref(ent[0]), int(pair is not None)))
                if nws < 0:
                    L.check(2, "gs_wgrad_ws_floats")
                self._desc_cache[wkey] = nws
            # (0 floats: single-contributor layers accumulate straight into dw, no workspace)
            ws = torch.empty(nws, dtype=torch.float32, device=self.device) if nws else None   # stream-safe via the allocator
            a2, g2 = pair if pair is not None else (None, None)
            L.check(self.lib.gs_wgrad_ws(C.byref(ent[0]), _ptr(a), _ptr(g), _ptr(a2), _ptr(g2), _ptr(dw), _ptr(ws), nws,
                                         _stream()), "gs_wgrad_ws")
        elif pair is not None:      # (a2, g2): the same layer's operands from another backward pass, one launch
            L.check(self.lib.gs_wgrad_pair(C.byref(ent[0]), _ptr(a), _ptr(g), _ptr(pair[0]), _ptr(pair[1]), _ptr(dw),
                                           _stream()), "gs_wgrad_pair")
        else:
            L.check(self.lib.gs_wgrad(C.byref(ent[0]), _ptr(a), _ptr(g), _ptr(dw), _stream()), "gs_wgrad")
        if t_end is not None:
            t_end.record()

    def bias_grad(self, dy, C_, db, *, cs=None, co=0):
        if is_twin(db):
            return self.twin_bias_grad(dy, C_, db, cs=cs, co=co)
        pixels = dy.numel() // dy.shape[-1]
        if C_ % 8:                                          # db has exactly C_ floats: whole octets, then a head of < 8
            full = C_ // 8 * 8
            cs_ = cs if cs is not None else dy.shape[-1]
            if full:
                self.bias_grad(dy, full, db[:full], cs=cs_, co=co)
            nws = int(self.lib.gs_bias_grad_ws_floats(pixels, 8))
            ws = torch.empty(nws, dtype=torch.float32, device=self.device)
            L.check(self.lib.gs_bias_grad_head_ws(_ptr(dy), pixels, cs_, co + full, C_ - full, _ptr(db[full:]),
                                                  _ptr(ws), nws, _stream()), "gs_bias_grad_head_ws")
            return
        if os.environ.get("GS_WGRAD_DET", "1") != "0":      # deterministic: partial sums + fixed-order second stage
            nws = int(self.lib.gs_bias_grad_ws_floats(pixels, C_))
            ws = torch.empty(nws, dtype=torch.float32, device=self.device)
            L.check(self.lib.gs_bias_grad_ws(_ptr(dy), pixels, C_, cs if cs is not None else dy.shape[-1], co, _ptr(db),
                                             _ptr(ws), nws, _stream()), "gs_bias_grad_ws")
            return
        L.check(self.lib.gs_bias_grad(_ptr(dy), pixels, C_, cs if cs is not None else dy.shape[-1], co, _ptr(db),
                                      _stream()), "gs_bias_grad")

    # ---- InstanceNorm + activation ----------------------------------------------------------------------
    def inorm_finalize(self, partial, N, slots, Cc, hw, mean_rstd, eps=1e-5):
        L.check(self.lib.gs_inorm_finalize(_ptr(partial), N, slots, Cc, hw, eps, _ptr(mean_rstd), _stream()),
                "gs_inorm_finalize")

    def inorm_act_forward(self, y, mean_rstd, res, x, act="none", slope=0.2):
        N, Cc = y.shape[0], y.shape[-1]
        L.check(self.lib.gs_inorm_act_forward(_ptr(y), _ptr(mean_rstd), _ptr(res), _ptr(x), N,
                                              y.numel() // (N * Cc), Cc,
                                              L.ACT[act], slope, _stream()), "gs_inorm_act_forward")

    def inorm_stats_act_forward(self, y, partial, slots, mean_rstd, res, x, act="none", slope=0.2, eps=1e-5):
        """inorm_finalize + inorm_act_forward as one launch (the apply kernel sums the slots of its own channels)"""
        N, Cc = y.shape[0], y.shape[-1]
        L.check(self.lib.gs_inorm_stats_act_forward(_ptr(y), _ptr(partial), slots, eps, _ptr(mean_rstd), _ptr(res),
                                                    _ptr(x), N, y.numel() // (N * Cc), Cc, L.ACT[act], slope,
                                                    _stream()), "gs_inorm_stats_act_forward")

    def inorm_act_backward(self, g_pad, g2, y, mean_rstd, dy, gsum, fold=0, fold_mode="reflect", act="none",
                           slope=0.2, bias_grad=None, pre=None):
        N, D, H, W, Cc = y.shape if y.dim() == 5 else (y.shape[0], 1) + tuple(y.shape[1:])
        scratch, pre_slots = None, 0
        if pre is not None:          # reduction pass already done by the fused data-gradient launch
            pre_slots, scratch = pre
        elif mean_rstd is not None:
            n = self.lib.gs_inorm_backward_scratch_floats(N, D, H, W, Cc)
            scratch = torch.empty(n, dtype=torch.float32, device=y.device)
        L.check(self.lib.gs_inorm_act_backward(_ptr(g_pad), _ptr(g2), _ptr(y), _ptr(mean_rstd), _ptr(dy),
                                               _ptr(gsum), _ptr(scratch), _ptr(bias_grad), N, D, H, W, Cc, fold,
                                               L.BORDER[fold_mode], L.ACT[act], slope, pre_slots, _stream()),
                "gs_inorm_act_backward")
        if mean_rstd is None:
            return None
        # the per-image totals [N][3][C] sit behind the partial slots: (holder tensor, float offset) for norm_bias_grads
        return scratch, scratch.numel() - N * 3 * Cc

    def norm_bias_grads(self, items):
        """bias gradients of the convs in front of InstanceNorms for many layers in one launch;
        items: (sums holder, float offset, mean_rstd, db, N, C, hw)"""
        if not items:
            return
        arr = (L.NormDbItem * len(items))()
        for a, (holder, off, mean_rstd, db, N, Cc, hw) in zip(arr, items):
            a.sums, a.mean_rstd, a.db = holder.data_ptr() + 4 * off, mean_rstd.data_ptr(), db.data_ptr()
            a.N, a.C, a.inv_hw = N, Cc, 1.0 / hw
        L.check(self.lib.gs_norm_bias_grads(arr, len(items), _stream()), "gs_norm_bias_grads")

    # ---- generalised norm / activation for skip-connection graphs (U-Net) ---------------------------------
    def _norm_ex_desc(self, y, act1, act2, slope, drop_p, seed, seed_dev=None):
        d = L.NormExDesc()
        d.seed_dev = seed_dev.data_ptr() if seed_dev is not None else None      # int32[2] device tensor (lo, hi)
        d.N, d.W, d.C = y.shape[0], y.shape[-2], y.shape[-1]       # geometry-free kernels: a volume is D*H rows
        d.H = y.numel() // (d.N * d.W * d.C)
        d.act1, d.act2, d.slope = L.ACT[act1], L.ACT[act2], slope
        d.drop_p, d.seed_lo, d.seed_hi = drop_p, seed & 0xffffffff, (seed >> 32) & 0xffffffff
        return d

    def norm_act_forward_ex(self, y, mean_rstd, x1, x2=None, act1="none", act2="none", slope=0.2, x1_co=0, x2_co=0,
                            drop_p=0.0, seed=0, seed_dev=None):
        d = self._norm_ex_desc(y, act1, act2, slope, drop_p, seed, seed_dev)
        d.x1_cs, d.x1_co = x1.shape[-1], x1_co
        if x2 is not None:
            d.x2_cs, d.x2_co = x2.shape[-1], x2_co
        L.check(self.lib.gs_norm_act_forward_ex(C.byref(d), _ptr(y), _ptr(mean_rstd), _ptr(x1), _ptr(x2), _stream()),
                "gs_norm_act_forward_ex")

    def norm_act_backward_ex(self, g1, g2, y, mean_rstd, dy, act1="none", act2="none", slope=0.2, g1_co=0, g2_co=0,
                             drop_p=0.0, seed=0, bias_grad=None, seed_dev=None):
        d = self._norm_ex_desc(y, act1, act2, slope, drop_p, seed, seed_dev)
        d.g1_cs, d.g1_co = g1.shape[-1], g1_co
        if g2 is not None:
            d.g2_cs, d.g2_co = g2.shape[-1], g2_co
        scratch = None
        if mean_rstd is not None:
            scratch = torch.empty(self.lib.gs_norm_backward_ex_scratch_floats(C.byref(d)), dtype=torch.float32,
                                  device=y.device)
        L.check(self.lib.gs_norm_act_backward_ex(C.byref(d), _ptr(g1), _ptr(g2), _ptr(y), _ptr(mean_rstd), _ptr(dy),
                                                 _ptr(scratch), _ptr(bias_grad), _stream()),
                "gs_norm_act_backward_ex")

    # ---- V-Net elementwise family (InstanceNorm3d -> [+res] -> PReLU -> [+res]) on channel slices ----------------
    @staticmethod
    def _pdesc(y, C_, y_co, res, res_mode, res_mod, res_co):
        d = L.PNormDesc()
        d.N, d.C = y.shape[0], C_
        d.pixels = y.numel() // (y.shape[0] * y.shape[-1])
        d.y_cs, d.y_co = y.shape[-1], y_co
        d.res_mode, d.res_mod = res_mode, res_mod
        if res is not None:
            d.res_cs, d.res_co = res.shape[-1], res_co
        return d

    def pnorm_forward(self, y, mean_rstd, out, *, C, slope=None, res=None, res_mode=0, res_mod=0, y_co=0, res_co=0,
                      out_co=0):
        if is_twin(slope):
            return self.twin_pnorm_forward(y, mean_rstd, out, C=C, slope=slope, res=res, res_mode=res_mode, res_mod=res_mod,
                                           y_co=y_co, res_co=res_co, out_co=out_co)
        d = self._pdesc(y, C, y_co, res, res_mode, res_mod, res_co)
        d.out_cs, d.out_co = out.shape[-1], out_co
        L.check(self.lib.gs_pnorm_forward(L.C.byref(d), _ptr(y), _ptr(mean_rstd), _ptr(res), _ptr(slope), _ptr(out),
                                          _stream()), "gs_pnorm_forward")

    def pnorm_backward(self, g, y, mean_rstd, dy, *, C, slope=None, dslope=None, g2=None, res=None, res_mode=0,
                       res_mod=0, gres=None, bias_grad=None, g_co=0, g2_co=0, y_co=0, res_co=0, dy_co=0, gres_co=0):
        if is_twin(slope, dslope, bias_grad):
            return self.twin_pnorm_backward(g, y, mean_rstd, dy, C=C, slope=slope, dslope=dslope, g2=g2, res=res,
                                            res_mode=res_mode, res_mod=res_mod, gres=gres, bias_grad=bias_grad, g_co=g_co,
                                            g2_co=g2_co, y_co=y_co, res_co=res_co, dy_co=dy_co, gres_co=gres_co)
        d = self._pdesc(y, C, y_co, res, res_mode, res_mod, res_co)
        d.g_cs, d.g_co = g.shape[-1], g_co
        if g2 is not None:
            d.g2_cs, d.g2_co = g2.shape[-1], g2_co
        d.dy_cs, d.dy_co = dy.shape[-1], dy_co
        if gres is not None:
            d.gres_cs, d.gres_co = gres.shape[-1], gres_co
        scratch = None
        if mean_rstd is not None or (slope is not None and dslope is not None):
            scratch = torch.empty(self.lib.gs_pnorm_backward_scratch_floats(L.C.byref(d)), dtype=torch.float32,
                                  device=y.device)
        L.check(self.lib.gs_pnorm_backward(L.C.byref(d), _ptr(g), _ptr(g2), _ptr(y), _ptr(mean_rstd), _ptr(res),
                                           _ptr(slope), _ptr(dy), _ptr(gres), _ptr(dslope), _ptr(bias_grad),
                                           _ptr(scratch), _stream()), "gs_pnorm_backward")

    def slice_stats(self, x, co, C, mean_rstd, eps=1e-5):
        """mean / rstd [N][2][C] of channels [co, co + C) of x (any NHWC / NDHWC activation tensor): gs_slice_stats +
        gs_inorm_finalize"""
        N = x.shape[0]
        pixels = x.numel() // (N * x.shape[-1])
        slots = int(self.lib.gs_slice_stats_slots(pixels))
        part = torch.empty(N * slots * 2 * C, dtype=torch.float32, device=x.device)
        L.check(self.lib.gs_slice_stats(_ptr(x), N, pixels, x.shape[-1], co, C, _ptr(part), _stream()), "gs_slice_stats")
        self.inorm_finalize(part, N, slots, C, pixels, mean_rstd, eps)

    def add_views(self, dst, src, C, dst_co=0, src_co=0, accumulate=True):
        pixels = dst.numel() // dst.shape[-1]
        L.check(self.lib.gs_add_views(_ptr(dst), dst.shape[-1], dst_co, _ptr(src), src.shape[-1], src_co, pixels, C,
                                      int(accumulate), _stream()), "gs_add_views")

    def repeat_backward(self, g, g_img, C, g_co=0):
        N, Cin = g_img.shape[0], g_img.shape[1]
        L.check(self.lib.gs_repeat_backward(_ptr(g), g.shape[-1], g_co, _ptr(g_img), N, Cin, C,
                                            g_img.numel() // (N * Cin), _stream()), "gs_repeat_backward")

    # ---- network boundary -----------------------------------------------------------------------------------
    @staticmethod
    def _img_dims(img):
        """(N, C, rows, W) of an NCHW image or NCDHW volume: the layout-only kernels see a volume as D*H rows"""
        N, Cc = img.shape[0], img.shape[1]
        return N, Cc, img.numel() // (N * Cc * img.shape[-1]), img.shape[-1]

    def image_to_act(self, img, act_t):
        N, Cc, H, W = self._img_dims(img)
        L.check(self.lib.gs_image_to_act(_ptr(img), _ptr(act_t), N, Cc, H, W, act_t.shape[-1], _stream()),
                "gs_image_to_act")

    def image_pair_to_act(self, a, b, act_t):
        """torch.cat([a, b], dim=1) converted in one pass"""
        N, Ca, H, W = self._img_dims(a)
        L.check(self.lib.gs_image_pair_to_act(_ptr(a), Ca, _ptr(b), b.shape[1], _ptr(act_t), N, H, W, act_t.shape[-1],
                                              _stream()), "gs_image_pair_to_act")

    def image_pair_to_act_backward(self, g, ga, gb, Ca, Cb):
        """gradient of image_pair_to_act into the images' own tensors (None: not needed)"""
        ref = ga if ga is not None else gb
        N, _, H, W = self._img_dims(ref)
        L.check(self.lib.gs_image_pair_to_act_backward(_ptr(g), _ptr(ga), Ca, _ptr(gb), Cb, N, H, W, g.shape[-1], _stream()),
                "gs_image_pair_to_act_backward")

    def act_to_image(self, act_t, img, act="none"):
        N, Cc, H, W = self._img_dims(img)
        L.check(self.lib.gs_act_to_image(_ptr(act_t), _ptr(img), N, Cc, H, W, act_t.shape[-1], L.ACT[act],
                                         _stream()), "gs_act_to_image")

    def act_to_image_backward(self, g_img, out_img, g_act, act="none"):
        N, Cc, H, W = self._img_dims(g_img)
        L.check(self.lib.gs_act_to_image_backward(_ptr(g_img), _ptr(out_img), _ptr(g_act), N, Cc, H, W,
                                                  g_act.shape[-1], L.ACT[act], _stream()),
                "gs_act_to_image_backward")

    def image_to_act_backward(self, g_pad, g_img, fold=0, fold_mode="reflect", accumulate=False):
        N, Cc = g_img.shape[0], g_img.shape[1]
        D, H, W = g_img.shape[2:] if g_img.dim() == 5 else (1,) + tuple(g_img.shape[2:])
        L.check(self.lib.gs_image_to_act_backward(_ptr(g_pad), _ptr(g_img), N, Cc, D, H, W, g_pad.shape[-1], fold,
                                                  L.BORDER[fold_mode], int(accumulate), _stream()),
                "gs_image_to_act_backward")

    # ---- W-fold boundary transforms of the k7 stem / last layer (csrc/wfold.hip) ---------------------------------
    def image_unfold(self, img, act_t, k, p, border):
        N, Cc, rows, W = self._img_dims(img)
        L.check(self.lib.gs_image_unfold(_ptr(img), _ptr(act_t), N, Cc, rows, W, act_t.shape[-1], k, p,
                                         L.BORDER[border], _stream()), "gs_image_unfold")

    def image_unfold_backward(self, g, g_img, k, p, fold, border, accumulate=False):
        N, Cc = g_img.shape[0], g_img.shape[1]
        D, H, W = g_img.shape[2:] if g_img.dim() == 5 else (1,) + tuple(g_img.shape[2:])
        L.check(self.lib.gs_image_unfold_backward(_ptr(g), _ptr(g_img), N, Cc, D, H, W, g.shape[-1], k, p, fold,
                                                  L.BORDER[border], int(accumulate), _stream()),
                "gs_image_unfold_backward")

    def shiftadd_to_image(self, z, bias, img, k, act="none"):
        if is_twin(bias):
            return self.twin_shiftadd_to_image(z, bias, img, k, act=act)
        N, Cc, rows, W = self._img_dims(img)
        L.check(self.lib.gs_shiftadd_to_image(_ptr(z), _ptr(bias), _ptr(img), N, Cc, rows, W, z.shape[-1], k,
                                              L.ACT[act], _stream()), "gs_shiftadd_to_image")

    def shiftadd_to_image_backward(self, g_img, out_img, gz, k, act="none"):
        N, Cc, rows, W = self._img_dims(g_img)
        L.check(self.lib.gs_shiftadd_to_image_backward(_ptr(g_img), _ptr(out_img), _ptr(gz), N, Cc, rows, W,
                                                       gz.shape[-1], k, L.ACT[act], _stream()),
                "gs_shiftadd_to_image_backward")

    # ---- losses ---------------------------------------------------------------------------------------------
    # ---- device-side image preprocessing (csrc/imgproc.hip) ---------------------------------------------
    def u8_resample_h(self, img, out, bounds, kk):
        """Pillow's horizontal 8-bit pass: img (H, W, C) uint8 -> out (H, W', C) uint8; bounds (W', 2), kk (W', ksize) int32"""
        H, W, Cc = img.shape
        L.check(self.lib.gs_u8_resample_h(_ptr(img), _ptr(out), H, W, out.shape[1], Cc, _ptr(bounds), _ptr(kk),
                                          kk.shape[1], _stream()), "gs_u8_resample_h")

    def u8_resample_v(self, tmp, out, bounds, kk):
        """Pillow's vertical 8-bit pass: tmp (H, W', C) uint8 -> out (H', W', C) uint8 (the image between two resizes)"""
        H, W2, Cc = tmp.shape
        L.check(self.lib.gs_u8_resample_v(_ptr(tmp), _ptr(out), H, W2, out.shape[0], Cc, _ptr(bounds), _ptr(kk),
                                          kk.shape[1], _stream()), "gs_u8_resample_v")

    def u8_resample_v_crop_normalize(self, tmp, out, out_h, bounds, kk, top, left, flip):
        """Pillow's vertical pass on the crop window + flip + ToTensor + Normalize(0.5, 0.5): tmp (H, W', C) uint8 ->
        out (C, fh, fw) fp32 (a contiguous slice of the NCHW batch)"""
        H, W2, Cc = tmp.shape
        assert out.is_contiguous() and out.dtype == torch.float32 and out.shape[0] == Cc
        L.check(self.lib.gs_u8_resample_v_crop_normalize(_ptr(tmp), _ptr(out), H, W2, out_h, Cc, _ptr(bounds), _ptr(kk),
                                                         kk.shape[1], int(top), int(left), out.shape[1], out.shape[2],
                                                         int(bool(flip)), _stream()),
                "gs_u8_resample_v_crop_normalize")

    VOL_DTYPES = {torch.float32: 0, torch.int16: 1}

    def patch_zscore(self, volume, start, size, out, scale_to_range=(-1.0, 1.0)):
        """gs_patch_zscore: out (d, h, w) fp32 = z_score_normalize(volume[z:z+d, y:y+h, x:x+w], scale_to_range) for a dense
        (D, H, W) fp32 / int16 volume resident on the device (normalization.py:18-30); scale_to_range None: plain z-score"""
        assert volume.dim() == 3 and volume.is_contiguous() and volume.dtype in self.VOL_DTYPES, (volume.shape, volume.dtype)
        assert out.is_contiguous() and out.dtype == torch.float32 and out.numel() == size[0] * size[1] * size[2]
        nws = getattr(self, "_patch_ws", None) or int(self.lib.gs_patch_zscore_ws_floats())
        self._patch_ws = nws
        ws = torch.empty(nws, dtype=torch.float32, device=self.device)        # per launch: stream-safe through the allocator
        st, sz = (C.c_int32 * 3)(*[int(v) for v in start]), (C.c_int32 * 3)(*[int(v) for v in size])
        lo, hi = scale_to_range if scale_to_range else (0.0, 0.0)
        L.check(self.lib.gs_patch_zscore(_ptr(volume), self.VOL_DTYPES[volume.dtype], *volume.shape, st, sz,
                                         int(bool(scale_to_range)), float(lo), float(hi), _ptr(out), _ptr(ws), _stream()),
                "gs_patch_zscore")

    ADV_MODES = {"lsgan": 0, "vanilla": 1, "wgangp": 2, "nonsaturating": 3}

    def adv_loss(self, x, mode, target_is_real, label, loss=None, grad=None, grad_scale=None):
        """gs_adv_loss: `x` is one discriminator map (B, ...), fp32; nonsaturating reduces per sample (loss: B floats)"""
        L.check(self.lib.gs_adv_loss(_ptr(x), x.numel(), x.shape[0], self.ADV_MODES[mode], int(bool(target_is_real)),
                                     float(label), _ptr(loss), _ptr(grad), _ptr(grad_scale), _stream()), "gs_adv_loss")

    def mse_const(self, x, target, loss=None, grad=None, grad_scale=None):
        L.check(self.lib.gs_mse_const(_ptr(x), x.numel(), float(target), _ptr(loss), _ptr(grad), _ptr(grad_scale),
                                      _stream()), "gs_mse_const")

    def l1(self, a, b, loss=None, grad_a=None, grad_scale=None):
        L.check(self.lib.gs_l1(_ptr(a), _ptr(b), a.numel(), _ptr(loss), _ptr(grad_a), _ptr(grad_scale), _stream()),
                "gs_l1")

    def mean(self, x, out):
        L.check(self.lib.gs_mean(_ptr(x), x.numel(), _ptr(out), _stream()), "gs_mean")

    def scalar_affine(self, xs, rows, consts=None):
        """[c_r + sum_k rows[r][k] * xs[k]] as a fresh fp32 vector [R]; xs = 0-d fp32 device tensors or None (= 0)"""
        K, R = len(xs), len(rows)
        out = torch.empty(R, dtype=torch.float32, device=self.device)
        px = (C.c_void_p * K)(*[(_ptr(x) if x is not None else None) for x in xs])
        m = (C.c_float * (R * K))(*[float(v) for row in rows for v in row])
        c = (C.c_float * R)(*[float(v) for v in consts]) if consts is not None else None
        L.check(self.lib.gs_scalar_affine(px, K, m, c, R, _ptr(out), _stream()), "gs_scalar_affine")
        return out

    # ---- feature taps (CUT) ------------------------------------------------------------------------------------
    def tap_gather(self, src, pid, c):
        """src [n, ..., cs] NHWC activations, pid int64 [P] flat pixel ids -> fp32 [n, P, c]"""
        n, cs = src.shape[0], src.shape[-1]
        pid = pid.contiguous()
        out = torch.empty((n, pid.numel(), c), dtype=torch.float32, device=self.device)
        L.check(self.lib.gs_tap_gather(_ptr(src), n, src.numel() // (n * cs), cs, _ptr(pid), pid.numel(), c, _ptr(out),
                                       _stream()), "gs_tap_gather")
        return out

    def tap_scatter_add(self, dst, pid, g, W, f0=0):
        """dst [n, Hp, Wp, cs] (a gradient on the domain padded by f0) += g [n, P, c] at the unpadded pixels pid"""
        n, cs = dst.shape[0], dst.shape[-1]
        pid, g = pid.contiguous(), g.contiguous().float()
        L.check(self.lib.gs_tap_scatter_add(_ptr(dst), n, dst.numel() // (n * cs), cs, _ptr(pid), pid.numel(), g.shape[-1], W,
                                            dst.shape[-2], f0, _ptr(g), _stream()), "gs_tap_scatter_add")

    def tap_rows_sum(self, g, db):
        g = g.contiguous().float()
        L.check(self.lib.gs_tap_rows_sum(_ptr(g), g.numel() // g.shape[-1], g.shape[-1], _ptr(db), _stream()),
                "gs_tap_rows_sum")

    def zeros_like_act(self, t):
        out = torch.empty_like(t)
        nbytes = out.numel() * out.element_size()
        if nbytes % 16 or out.data_ptr() % 16:
            return out.zero_()
        L.check(self.lib.gs_zero_bytes(_ptr(out), nbytes, _stream()), "gs_zero_bytes")
        return out

    def image_tap_gather(self, x, pid, pad):
        N, Cc, H, W = x.shape
        pid = pid.contiguous()
        out = torch.empty((N, pid.numel(), Cc), dtype=torch.float32, device=self.device)
        L.check(self.lib.gs_image_tap_gather(_ptr(x), N, Cc, H, W, pad, _ptr(pid), pid.numel(), _ptr(out), _stream()),
                "gs_image_tap_gather")
        return out

    def image_tap_scatter(self, g, pid, shape, pad):
        N, Cc, H, W = shape
        pid, g = pid.contiguous(), g.contiguous().float()
        gx = torch.empty(shape, dtype=torch.float32, device=self.device)
        L.check(self.lib.gs_image_tap_scatter(_ptr(g), N, Cc, H, W, pad, _ptr(pid), pid.numel(), _ptr(gx), _stream()),
                "gs_image_tap_scatter")
        return gx

    def flip_w_if(self, x, flag):
        """x.flip(-1) where the int32 device flag is set, x otherwise (a copy either way)"""
        x = x.contiguous().float()
        out = torch.empty_like(x)
        L.check(self.lib.gs_flip_w_if(_ptr(x), _ptr(out), x.numel() // x.shape[-1], x.shape[-1], _ptr(flag), _stream()),
                "gs_flip_w_if")
        return out

    def sum2(self, a, b):
        out = torch.empty_like(a)
        L.check(self.lib.gs_sum2_f32(_ptr(a), _ptr(b), _ptr(out), a.numel(), _stream()), "gs_sum2_f32")
        return out

    def ssim_distance(self, x, y, out):
        NC = x.numel() // (x.shape[-1] * x.shape[-2])
        H, W = x.shape[-2], x.shape[-1]
        scratch = torch.empty(self.lib.gs_ssim_scratch_floats(NC, H, W), dtype=torch.float32, device=x.device)
        L.check(self.lib.gs_ssim_distance(_ptr(x), _ptr(y), NC, H, W, _ptr(out), _ptr(scratch), _stream()),
                "gs_ssim_distance")

    # ---- optimiser -----------------------------------------------------------------------------------------
    def adam_step(self, p, g, m, v, lr, beta1, beta2, eps, step, grad_scale=1.0, zero_grad=True):
        bc1 = 1.0 - beta1 ** step
        bc2_sqrt = (1.0 - beta2 ** step) ** 0.5
        hyper = (C.c_float * 6)(lr, beta1, beta2, eps, bc1, bc2_sqrt)
        L.check(self.lib.gs_adam_step(_ptr(p), _ptr(g), _ptr(m), _ptr(v), p.numel(), hyper, float(grad_scale),
                                      int(zero_grad), _stream()), "gs_adam_step")

    def adam_step_dev(self, p, g, m, v, hyper_dev, grad_scale=1.0, zero_grad=True, packs=None):
        """hyper_dev: device float32[6] = lr, beta1, beta2, eps, 1-beta1^t, sqrt(1-beta2^t) (see NativeAdam.prepare).
        packs = (inv_f, fpack, inv_d, dpack): the update also writes the pack groups that are 8 consecutive master elements
        (NativeNet.fused_pack_targets)"""
        if packs is not None:
            inv_f, fpack, inv_d, dpack = packs
            L.check(self.lib.gs_adam_step_dev_packs(_ptr(p), _ptr(g), _ptr(m), _ptr(v), p.numel(), _ptr(hyper_dev),
                                                    float(grad_scale), int(zero_grad), _ptr(inv_f), _ptr(fpack), _ptr(inv_d),
                                                    _ptr(dpack), _stream()), "gs_adam_step_dev_packs")
            return
        L.check(self.lib.gs_adam_step_dev(_ptr(p), _ptr(g), _ptr(m), _ptr(v), p.numel(), _ptr(hyper_dev),
                                          float(grad_scale), int(zero_grad), _stream()), "gs_adam_step_dev")

    def adam_step_dev_ranges(self, p, g, m, v, ranges_dev, max_len, hyper_dev, grad_scale=1.0, zero_grad=True, packs=None):
        """adam_step_dev over the ranges [start, end) (device int64 [n][2]) of the flat buffers in one launch; packs index the
        whole buffers"""
        inv_f, fpack, inv_d, dpack = packs if packs is not None else (None, None, None, None)
        L.check(self.lib.gs_adam_step_dev_packs_ranges(_ptr(p), _ptr(g), _ptr(m), _ptr(v), _ptr(ranges_dev), ranges_dev.shape[0],
                                                       int(max_len), _ptr(hyper_dev), float(grad_scale), int(zero_grad),
                                                       _ptr(inv_f), _ptr(fpack), _ptr(inv_d), _ptr(dpack), _stream()),
                "gs_adam_step_dev_packs_ranges")

    def pool_query(self, pool, images, out, code_dev):
        B = images.shape[0]
        nbytes = images[0].numel() * images.element_size()
        L.check(self.lib.gs_pool_query(_ptr(pool), _ptr(images), _ptr(out), _ptr(code_dev), B, nbytes, _stream()),
                "gs_pool_query")

    def repack(self, master, index, pack):
        L.check(self.lib.gs_repack_bf16(_ptr(master), _ptr(index), _ptr(pack), pack.numel(), _stream()),
                "gs_repack_bf16")

    def repack_groups(self, master, gindex, pack, index=None):
        """pack[8 g + j] = bf16(master[gindex[g] + j]); gindex[g] = -1: zeros, -2: the group's own entries of `index`"""
        L.check(self.lib.gs_repack_bf16_groups(_ptr(master), _ptr(gindex), _ptr(index) if index is not None else None,
                                               _ptr(pack), gindex.numel(), _stream()), "gs_repack_bf16_groups")

    def repack_tiled_groups(self, master, gindex, pack, seg, tiles):
        """all transposed segments of a pack: seg int64 [nseg, 5] = (pack offset, gindex offset, rows, kp, first tile) on the
        device; pack[off + (8 G + j) * kp + k] = bf16(master[gindex[goff + G * kp + k] + j])"""
        L.check(self.lib.gs_repack_bf16_tiled_groups(_ptr(master), _ptr(gindex), _ptr(pack), _ptr(seg), seg.shape[0],
                                                     int(tiles), _stream()), "gs_repack_bf16_tiled_groups")

    def repack_tiled(self, master, index, pack, rows, kp):
        """one [rows][kp] pack segment whose master indices run along the rows (see NativeNet._get_packs)"""
        L.check(self.lib.gs_repack_bf16_tiled(_ptr(master), _ptr(index), _ptr(pack), rows, kp, _stream()),
                "gs_repack_bf16_tiled")

    def ssim_distance_backward(self, x, y, grad_y, grad_scale=None):
        NC = x.numel() // (x.shape[-1] * x.shape[-2])
        H, W = x.shape[-2], x.shape[-1]
        scratch = torch.empty(self.lib.gs_ssim_backward_scratch_floats(NC, H, W), dtype=torch.float32, device=x.device)
        L.check(self.lib.gs_ssim_distance_backward(_ptr(x), _ptr(y), NC, H, W, _ptr(grad_scale), _ptr(grad_y),
                                                   _ptr(scratch), _stream()), "gs_ssim_distance_backward")

    # ---- SelfAttentionBlock (csrc/attn.hip) ----------------------------------------------------------------------
    ATTN_KEYS = ("gamma", "wq", "bq", "wk", "bk", "wv", "bv")

    def _attn_args(self, x, tensors):
        Cc = int(x.shape[-1])
        d = L.AttnDesc(int(x.shape[0]), int(x.numel() // (x.shape[0] * Cc)), Cc)
        p = L.AttnParams()
        for k in self.ATTN_KEYS:
            t = None if tensors is None else tensors.get(k)
            if t is not None:
                assert t.dtype == torch.float32 and t.is_contiguous() and t.device == x.device, k
            setattr(p, k, t.data_ptr() if t is not None else None)
        return d, p

    def attn_forward(self, x, params, need_backward=None):
        """x: NDHWC activation [B, ..., C]; params: {gamma [1], wq [C/8, C], bq, wk, bk, wv [C, C], bv} fp32 (torch layout)
        -> (out like x, saved state for attn_backward)"""
        assert x.is_contiguous() and x.dtype == self.act_dtype
        d, p = self._attn_args(x, params)
        # (a pass that will not run backward — inference, validation — takes the forward part of the scratch only)
        need_bwd = True if need_backward is None else bool(need_backward)     # (callers inside an autograd Function say so)
        size = self.lib.gs_attn_work_bytes(C.byref(d)) if need_bwd else self.lib.gs_attn_forward_work_bytes(C.byref(d))
        work = torch.empty(int(size), dtype=torch.uint8, device=self.device)
        out = torch.empty_like(x)
        L.check(self.lib.gs_attn_forward(C.byref(d), _ptr(x), C.byref(p), _ptr(out), _ptr(work), _stream()), "gs_attn_forward")
        return out, ((x, work) if need_bwd else None)

    def attn_backward(self, saved, dout, params, grads):
        """-> dx; parameter gradients are ADDED into the tensors of `grads` (same keys as params; None: skipped)"""
        if saved is None:
            raise RuntimeError("attn_backward: this forward pass kept no state (it ran under no_grad / need_backward=False)")
        x, work = saved
        d, p = self._attn_args(x, params)
        _, g = self._attn_args(x, grads)
        dx = torch.empty_like(x)
        L.check(self.lib.gs_attn_backward(C.byref(d), _ptr(x), _ptr(dout.contiguous()), C.byref(p),
                                          C.byref(g) if grads is not None else None, _ptr(work), _ptr(dx), _stream()),
                "gs_attn_backward")
        return dx

    # ---- PatchNCE + patch MLP (csrc/patchnce.hip) ---------------------------------------------------------------
    def _nce_desc(self, channels, batch, patches, nc, nce_T, lambda_nce):
        d = L.PatchNCEDesc()
        d.levels, d.batch, d.patches, d.nc, d.nce_T, d.lambda_nce = len(channels), batch, patches, nc, nce_T, lambda_nce
        for i, c in enumerate(channels):
            d.channels[i] = c
        return d

    def patchnce_forward(self, xq, xk, params, *, batch, nc=256, nce_T=0.07, lambda_nce=1.0):
        """xq / xk: lists of [batch, patches_l, C_l] fp32 (target / source patches per level), params: flat fp32 MLP
        parameters -> (loss per level [L] fp32, saved state for patchnce_backward).
        Levels normally share one patch count and run as ONE launch group. FeaturePatchMLP draws min(num_patches,
        pixels of the level) ids per level (cut.py:262-268), so on small inputs deep levels carry fewer patches: then every
        level runs as its own launch group with its own row count, weighted lambda / (levels * batch * patches_l) like
        the reference's per-level .mean() (cut.py:218-226)."""
        n_lv = len(xq)
        channels = [int(t.shape[-1]) for t in xq]
        for l, (q, k) in enumerate(zip(xq, xk)):
            if q.dim() != 3 or tuple(q.shape) != tuple(k.shape) or int(q.shape[0]) != batch:
                raise ValueError(f"patchnce_forward: level {l}: target {tuple(q.shape)} / source {tuple(k.shape)} patches "
                                 f"must both be [batch={batch}, patches, channels]")
        per_level = [int(t.shape[1]) for t in xq]
        if nc != 256 or max(per_level) > 256 or min(per_level) < 1 or max(channels) > 256:
            raise NotImplementedError(
                f"gs_patchnce_*: the fused PatchNCE kernels take mlp_nc == 256, 1..256 patches per image and level "
                f"and <= 256 channels per level (got mlp_nc={nc}, patches={per_level}, channels={channels}); "
                "num_patches: 0 (every pixel) and wider features have no HIP path")
        xq = [t.contiguous().float() for t in xq]
        xk = [t.contiguous().float() for t in xk]
        loss = torch.empty(n_lv, dtype=torch.float32, device=self.device)
        if len(set(per_level)) == 1:
            groups = [(list(range(n_lv)), 0, lambda_nce)]
        else:      # one group per level; its lambda carries the 1 / levels of the whole loss
            offs, off = [], 0
            for c in channels:
                offs.append(off)
                off += nc * c + nc + nc * nc + nc
            groups = [([l], offs[l], lambda_nce / n_lv) for l in range(n_lv)]
        saved = []
        for lv, poff, lam in groups:
            d = self._nce_desc([channels[l] for l in lv], batch, per_level[lv[0]], nc, nce_T, lam)
            work = torch.empty(int(self.lib.gs_patchnce_work_bytes(C.byref(d))), dtype=torch.uint8, device=self.device)
            gq, gk = [xq[l] for l in lv], [xk[l] for l in lv]
            pq = (C.c_void_p * len(lv))(*[t.data_ptr() for t in gq])
            pk = (C.c_void_p * len(lv))(*[t.data_ptr() for t in gk])
            L.check(self.lib.gs_patchnce_forward(C.byref(d), pq, pk, _ptr(params[poff:]), _ptr(work),
                                                 _ptr(loss[lv[0]:]), _stream()), "gs_patchnce_forward")
            saved.append((d, gq, work, poff))
        return loss, saved

    def patchnce_backward(self, saved, params, grads, grad_scale=None):
        """-> list of d(sum of level losses)/d xq[l] * grad_scale; parameter gradients are added into `grads`"""
        out = []
        for d, xq, work, poff in saved:
            dxq = [torch.empty_like(t) for t in xq]
            pq = (C.c_void_p * len(xq))(*[t.data_ptr() for t in xq])
            pd = (C.c_void_p * len(xq))(*[t.data_ptr() for t in dxq])
            L.check(self.lib.gs_patchnce_backward(C.byref(d), pq, pd, _ptr(params[poff:]), _ptr(grads[poff:]),
                                                  _ptr(work), _ptr(grad_scale), _stream()), "gs_patchnce_backward")
            out += dxq
        return out
