"""Weight bank: the weight preparation of a whole model as TWO launches (``pwg_weight_bank_*``).

Per parameter epoch every convolution needs its weight-norm row scale, its packed forward image and -- when a data
gradient will be taken -- its packed data-gradient image.  Lazily, per layer, that is up to three 3-10 us launches
per layer (HiFi-GAN V1 training step: 551 launches, 4.8 ms of kernel time).  The bank owns persistent buffers for
all images of a module tree and a device table describing them; ``ensure()`` refreshes everything with one row-scale
launch and one packing launch and hands every layer a :class:`functional.PreparedWeights` for the current parameter
values, so the layers' own ``prepared()`` finds its cache warm.  The arithmetic is that of the per-layer kernels,
bit for bit (tests/test_weight_bank_gpu.py).

The image buffers are overwritten by the next ``ensure()`` after a parameter update; a ``PreparedWeights`` handed out
earlier is marked stale then and raises if a (retained) autograd graph still tries to read it.
"""
import ctypes

import torch

from . import _lib, ops
from . import functional as Fn
from .layers.conv import _ConvNd


class WeightBank:
    def __init__(self, module):
        self.module = module
        self._sig = None
        self._layers = []
        self._table = None
        self._info = None
        self._bufs = []       # per layer: (w3, scale, fwd, bwd, desc)
        self._handed = []     # PreparedWeights of the current generation
        self._has_bwd = False

    # -- which layers: every convolution of the tree whose effective weight is a pure function of its parameters
    def _collect(self):
        return [m for m in self.module.modules()
                if isinstance(m, _ConvNd) and not m.has_spectral_norm and m.raw_weight.is_cuda]

    def _signature(self, layers):
        return tuple((id(m), m.raw_weight.data_ptr(), m.weight_g.data_ptr() if m.has_weight_norm else 0,
                      tuple(m.raw_weight.shape)) for m in layers)

    def _build(self, layers):
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("WeightBank: the table must be built before a stream capture (run one eager step first)")
        dev = layers[0].raw_weight.device
        items = (_lib.BankItem * len(layers))()
        bufs = []
        for it, m in zip(items, layers):
            desc = m.make_desc(1, m._probe_len())
            w3 = m._w3(m.raw_weight.detach())
            assert w3.data_ptr() == m.raw_weight.data_ptr()  # a view of the parameter, not a copy
            scale = None
            if m.has_weight_norm:
                g = m.weight_g.detach().reshape(-1)
                assert g.data_ptr() == m.weight_g.data_ptr()
                scale = torch.empty(w3.shape[0], device=dev, dtype=torch.float32)
                it.g, it.scale = g.data_ptr(), scale.data_ptr()
            fwd = bwd = None
            if getattr(m, "bank_images", True):  # (False: layers that normally run inside a fused multi-layer kernel
                # -- the WaveNet block's four convolutions -- get their row scale here and pack lazily if ever needed)
                # zero-filled ONCE: the packing launch writes the real elements only, the padding of an image
                # (channels up to a multiple of 16, rows up to a multiple of 128, unused polyphase taps) stays zero
                fwd = torch.zeros(ops.packed_weight_floats(desc), device=dev, dtype=torch.float32)
                n_bwd = _lib.lib().pwg_conv1d_packed_weight_bwd_floats(ctypes.byref(desc))
                bwd = torch.zeros(n_bwd, device=dev, dtype=torch.float32) if n_bwd else None
            it.w, it.desc = w3.data_ptr(), desc
            it.fwd = None if fwd is None else fwd.data_ptr()
            it.bwd = None if bwd is None else bwd.data_ptr()
            bufs.append((w3, scale, fwd, bwd, desc))
        n_bytes = _lib.lib().pwg_weight_bank_table_bytes(len(layers))
        host = (ctypes.c_char * n_bytes)()
        info = (ctypes.c_int32 * 8)()
        _lib.check(_lib.lib().pwg_weight_bank_build(items, len(layers), host, n_bytes, info), "weight_bank_build")
        table = torch.frombuffer(host, dtype=torch.uint8).clone().to(dev)
        torch.cuda.synchronize(dev)
        self._layers, self._bufs, self._table, self._info = layers, bufs, table, info
        self._sig = self._signature(layers)

    def _current(self, with_bwd):
        if not self._layers or (with_bwd and not self._has_bwd):
            return False
        for m, pw in zip(self._layers, self._handed):
            if m._cache_packed is not pw or m._cache_key != m._params_key():
                return False
        return len(self._handed) == len(self._layers)

    def ensure(self, with_bwd=True):
        """Make every layer's prepared weights current (no launch when they already are)."""
        layers = self._collect()
        if not layers:
            return
        if self._sig != self._signature(layers):
            self._build(layers)
            self._handed = []
        if self._current(with_bwd):
            return
        for pw in self._handed:
            pw._stale = True
        with torch.no_grad():
            _lib.check(_lib.lib().pwg_weight_bank_prepare(ctypes.c_void_p(self._table.data_ptr()), self._info,
                                                          int(bool(with_bwd)), ops._stream()), "weight_bank_prepare")
        self._handed = []
        self._has_bwd = bool(with_bwd)
        for m, (w3, scale, fwd, bwd, desc) in zip(self._layers, self._bufs):
            key = m._params_key()
            pw = Fn.PreparedWeights(key, w3, scale, fwd, desc)
            if with_bwd and bwd is not None:
                pw._bwd = bwd
            m._cache_packed, m._cache_key = pw, key
            self._handed.append(pw)
